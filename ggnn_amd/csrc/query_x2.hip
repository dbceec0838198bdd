// query kernel, paired form: TWO searches per wave64, phase-interleaved.
// Reference: QueryKernel::operator(), src/ggnn/query/query_layer.cu:39-97 (one search each).
//
// Why: a blocking 10 000-query launch is 9.8 one-search waves per SIMD with 7 resident -- 1.4
// rounds, the second one at low occupancy, and a lone wave is bound by the three dependent memory
// round trips of a pop (graph row -> code rows -> float rows), not by its instructions.  Here a wave
// carries two INDEPENDENT searches, each still using all 64 lanes for every step (same SortedList,
// same filter, same pre-screen, same distance engine as query.hip -- the per-search state evolution
// is the one-search kernel's, statement for statement), but their phases alternate:
//
//     A: pop, graph row, filter, request code rows      B: the same
//     A: pre-screen verdicts, request float rows        B: the same
//     A: distances, accept/push replay                  B: the same
//
// so the loads of one search are in flight while the other computes, 10 000 queries are 5 000 waves
// = one resident round at 5 waves per SIMD, and every SIMD has 10 searches to interleave instead
// of 7.  All wave-uniform values stay in SGPRs (two copies); nothing is shared between the two
// searches except the instruction stream.
//
// Selected by launch_query (query.hip) for one-register lists with the pre-screen on float32 rows;
// every other configuration runs the one-search kernel.
#include "query_args.hpp"
#include "traversal.hpp"

namespace ggnn_amd {

// tuning knobs (A/B builds): float rows requested ahead per search, operands kept in LDS
#ifndef GGNN_X2_FSTEPS
#define GGNN_X2_FSTEPS 2
#endif
#ifndef GGNN_X2_Q_LDS
#define GGNN_X2_Q_LDS 1
#endif
#ifndef GGNN_X2_QC_LDS
#define GGNN_X2_QC_LDS 1
#endif

template <typename BaseT, int LPR, int NCH, int MODE, class PSC, int HB>
struct PairedSearch {
  using DE = DistEngine<BaseT, LPR, NCH>;
  using Chunk = typename DE::Chunk;
  static constexpr int PROWS = PSC::ROWS;
  static constexpr int PSTEPS = 24 / PSC::ROWS;  // code rows of one graph row in one round
  static constexpr int FROWS = DE::ROWS;
  static constexpr int FSTEPS = GGNN_X2_FSTEPS;  // float rows requested ahead (the rest: rare)
  // one-register lists: query_sizing gives cache 256 (sorted 64, ring 192: HB 1) or cache 512
  // (ring 448 / 480: HB 2) -- a compile-time constant here, so that the LDS addresses of the two
  // searches differ by an immediate offset instead of living in separate address registers
  static constexpr int kCache = HB == 1 ? 256 : 512;
  // behind the one-search layout (traversal.hpp): the query's chunks and its pre-screen codes, read
  // back where they are used instead of being held in 12 registers per search
  static constexpr int kQInts = LPR * NCH * 4;
  static constexpr int kQcInts = PSC::LPR * PSC::NCH * 4;
  static constexpr int kSetInts = kCache + static_cast<int>(WaveLds::extra_ints) +
                                  HB * 64 * kVisSlots + kVisStash;
  static constexpr int kLdsInts = kSetInts + kQInts + kQcInts;

  SortedList<1, HB> sl;
  DE de;
  PSC ps;
  int* lds_base;
  float xi0;
  uint32_t n;         // query number
  uint32_t ite;
  bool live;
  int stage;          // 0: nothing pending, 1: code rows requested, 2: float rows requested
  int spec_key, spec_row;
  uint32_t cnt_dist, cnt_pop;
  uint2 rows;
  int nsurv, neval;
  float s_thr;

  // rows in flight between the phases: locals of the caller's loop body, NOT members -- a member
  // written on some paths only is carried around the loop and stays live for good
  struct CodeRows {
    uint4 cv[PSTEPS];
  };
  struct FloatRows {
    Chunk fv[FSTEPS][NCH];
  };

  GGNN_DEV WaveLds lds() const { return WaveLds(lds_base, kCache); }
  GGNN_DEV void reload_q()
  {
#if GGNN_X2_Q_LDS
    const Chunk* qb = reinterpret_cast<const Chunk*>(lds_base + kSetInts);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      de.q[c] = qb[c * LPR + de.g];
#endif
  }
  GGNN_DEV void reload_qc()
  {
#if GGNN_X2_QC_LDS
    const uint4* qb = reinterpret_cast<const uint4*>(lds_base + kSetInts + kQInts);
#pragma unroll
    for (int c = 0; c < PSC::NCH; ++c)
      ps.qc[c] = qb[c * PSC::LPR + ps.g];
#endif
  }

  GGNN_DEV void start(const QueryArgs& a, uint32_t n_, int* lds_region)
  {
    const int lane = threadIdx.x;
    n = n_;
    lds_base = lds_region;
    live = n < a.Nq;
    stage = 0;
    ite = 0;
    cnt_dist = cnt_pop = 0;
    rows = make_uint2(0u, 0u);
    spec_key = spec_row = kEmptyKey;
    if (!live)
      return;
    const BaseT* base = static_cast<const BaseT*>(a.base);
    const BaseT* qrow = static_cast<const BaseT*>(a.query) + static_cast<size_t>(n) * a.D;
    // query_layer.cu:48-50 (xi from the MAX nn1 distance, quirk Q4)
    const float nn1 = a.nn1_stats[1];
    xi0 = (MODE == kL2) ? (nn1 * nn1) * a.tau * a.tau : nn1 * a.tau;
    de.template load_query<MODE>(base, a.D, qrow);
    ps.load(a.ps_codes, a.ps_params, a.ps_Dc, reinterpret_cast<const float*>(qrow), a.D);
    // lanes of the first row group publish their chunks (every group holds the same ones)
    {
      Chunk* qb = reinterpret_cast<Chunk*>(lds_base + kSetInts);
      uint4* qcb = reinterpret_cast<uint4*>(lds_base + kSetInts + kQInts);
      if (lane < LPR) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          qb[c * LPR + lane] = de.q[c];
      }
      if (lane < PSC::LPR) {
#pragma unroll
        for (int c = 0; c < PSC::NCH; ++c)
          qcb[c * PSC::LPR + lane] = ps.qc[c];
      }
    }
    const WaveLds l = lds();
    sl.init(a.KQuery, a.sorted, kCache, xi0, l.known, static_cast<int>(a.vis_slots));
    // fetch_unfiltered(d_starting_points, nullptr, S), query_layer.cu:54-55 (not interleaved)
    for (uint32_t i = 0; i < a.num_start; i += kKBlock) {
      const int cand = (lane < (int)kKBlock && i + lane < a.num_start) ? a.start[i + lane]
                                                                        : kEmptyKey;
      cnt_dist += fetch<MODE, false>(sl, de, l, cand, nullptr, ps, rows);
    }
    __syncthreads();
  }

  // pop, graph row, filter, code rows requested (query_layer.cu:57-77, fetch():241-261)
  GGNN_DEV void phase1(const QueryArgs& a, CodeRows& cr)
  {
    const int lane = threadIdx.x;
    stage = 0;
    if (!live)
      return;
    if (ite >= a.max_iters) {
      live = false;
      return;
    }
    ++ite;
    const WaveLds l = lds();
    const float d0 = sl.dist_at(0);
    sl.xi = (MODE == kL2) ? fminf(xi0, d0 * a.tau * a.tau) : fminf(xi0, d0 * a.tau);
    const int anchor = sl.pop(sl.criteria(), l.known);
    if (anchor == kEmptyKey) {
      live = false;
      return;
    }
    ++cnt_pop;
    const bool in_row = lane < (int)kKBlock && lane < (int)a.KBuild;
    int cand;
    if (anchor == spec_key)
      cand = spec_row;
    else
      cand = in_row ? a.graph0[static_cast<size_t>(static_cast<uint32_t>(anchor)) * a.KBuild + lane]
                    : kEmptyKey;
    cand = lower_half_to_both(cand);
    cand = sl.filter(cand, l.known);
    const unsigned long long surv = __ballot(lane < 32 && cand != kEmptyKey);
    nsurv = __popcll(surv);
    // graph row of the current queue head, in case it is still the head at the next pop
    spec_key = sl.key_at(sl.BEST);
    if (spec_key != kEmptyKey)
      spec_row = in_row ? a.graph0[static_cast<size_t>(static_cast<uint32_t>(spec_key)) * a.KBuild +
                                   lane]
                        : kEmptyKey;
    if (nsurv == 0)
      return;
    cnt_dist += nsurv;
    __syncthreads();
    if (lane < 32 && cand != kEmptyKey)
      l.ckeys[__popcll(surv & ((1ull << lane) - 1ull))] = cand;
    __syncthreads();
    s_thr = ps.threshold(sl.criteria());
    if (!(s_thr < inf_f())) {
      // no usable bound (list not full, zero-norm cosine query): every candidate is evaluated
      neval = nsurv;
      stage = 3;
      return;
    }
    rows.y += nsurv;
    const int grp = lane / PSC::LPR;
#pragma unroll
    for (int s = 0; s < PSTEPS; ++s) {
      const int r = s * PROWS + grp;
      // slots past the end read the code row of the first candidate (cached, verdict ignored)
      const int m = l.ckeys[r < nsurv ? r : 0];
      const uint8_t* row = ps.row_ptr(m);
      if (ps.all_chunks || ps.chunk_valid(0))
        cr.cv[s] = ps.load_chunk(row, 0);
      else
        cr.cv[s] = make_uint4(0u, 0u, 0u, 0u);
    }
    stage = 1;
  }

  // pre-screen verdicts (prescreen_pass), float rows of the survivors requested
  GGNN_DEV void phase2(const QueryArgs&, const CodeRows& cr, FloatRows& fr)
  {
    if (stage != 1 && stage != 3)
      return;
    const int lane = threadIdx.x;
    const WaveLds l = lds();
    if (stage == 1) {
      reload_qc();
      const int pgrp = lane / PSC::LPR;
      int kk[PSTEPS];
#pragma unroll
      for (int s = 0; s < PSTEPS; ++s) {
        const int r = s * PROWS + pgrp;
        kk[s] = r < nsurv ? l.ckeys[r] : kEmptyKey;
      }
      __syncthreads();  // all keys of the round are in registers before the in-place compaction
      int npass = 0;
#pragma unroll
      for (int s = 0; s < PSTEPS; ++s) {
        if (s * PROWS >= nsurv)
          break;
        const uint4 v[1] = {cr.cv[s]};
        const float S = group_sum<PSC::LPR>(ps.partial(v));
        const bool pass = (kk[s] != kEmptyKey) && (ps.g == 0) && !(S >= s_thr);
        const unsigned long long pm = __ballot(pass);
        if (pass)
          l.ckeys[npass + __popcll(pm & ((1ull << lane) - 1ull))] = kk[s];
        npass += __popcll(pm);
      }
      neval = npass;
      if (neval == 0) {
        stage = 0;
        return;
      }
      __syncthreads();
    }
    const int grp = lane / LPR;
#pragma unroll
    for (int s = 0; s < FSTEPS; ++s) {
      if (s > 0 && s * FROWS >= neval)
        break;
      const int r = s * FROWS + grp;
      const int m = l.ckeys[r < neval ? r : 0];
      const BaseT* row = de.row_ptr(m);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (de.all_chunks || de.chunk_valid(c))
          fr.fv[s][c] = de.load_chunk(row, c);
        else
          fr.fv[s][c] = ChunkOf<BaseT>::zero();
      }
    }
    stage = 2;
  }

  // distances of the survivors, accept/push replay (simple_knn_cache.cuh:268-286)
  GGNN_DEV void phase3(const QueryArgs&, const FloatRows& fr)
  {
    if (stage != 2)
      return;
    const int lane = threadIdx.x;
    const WaveLds l = lds();
    reload_q();
    const int grp = lane / LPR;
#pragma unroll
    for (int s = 0; s < FSTEPS; ++s) {
      if (s * FROWS >= neval)
        break;
      if (s > 0)
        asm volatile("" ::: "memory");
      float x, y;
      de.template partial<MODE>(fr.fv[s], x, y);
      x = group_sum<LPR>(x);
      if (MODE == kCos)
        y = group_sum<LPR>(y);
      const int r = s * FROWS + grp;
      if (r < neval && de.g == 0)
        l.cd0[r] = (MODE == kCos) ? de.finish_cos(x, y) : x;
    }
    if (neval > FSTEPS * FROWS)  // more survivors than rows requested ahead: the rest right here
      compute_distances<MODE, DE, FSTEPS>(de, l, neval, nullptr, FSTEPS * FROWS);
    rows.x += neval;
    __syncthreads();
    const float cd = lane < neval ? l.cd0[lane] : inf_f();
    const int ck = lane < neval ? l.ckeys[lane] : kEmptyKey;
    unsigned long long m = __ballot(cd < sl.criteria());
    while (m) {
      const int j = __ffsll(static_cast<long long>(m)) - 1;
      m &= m - 1;
      const float d = rdlanef(cd, j);
      const int k = rdlane(ck, j);
      if (d < sl.criteria())
        sl.push(k, d);
    }
    stage = 0;
  }

  // write_best + dists, query_layer.cu:81-90 (EMPTY becomes -1 + offset, as in the reference)
  GGNN_DEV void finish(const QueryArgs& a) const
  {
    if (n >= a.Nq)
      return;
    const uint32_t lane = threadIdx.x;
    const size_t out_row = (static_cast<size_t>(n) * a.shards_per_gpu + a.on_gpu_shard) * a.KQuery;
    const int32_t id_offset = static_cast<int32_t>(a.on_gpu_shard * a.N_base);
    if (lane < a.KQuery) {
      a.ids[out_row + lane] = sl.key[0] + id_offset;
      a.dists[out_row + lane] = sl.dist[0];
    }
    if (lane == 0) {
      if (a.n_dist)
        a.n_dist[n] = cnt_dist;
      if (a.n_pop)
        a.n_pop[n] = cnt_pop;
      if (a.n_rows)
        a.n_rows[n] = rows;
    }
  }
};

#ifndef GGNN_QUERY_X2_WAVES
#define GGNN_QUERY_X2_WAVES 5
#endif

template <typename BaseT, int LPR, int NCH, int MODE, class PSC, int HB>
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(GGNN_QUERY_X2_WAVES)))
query_x2_kernel(const QueryArgs a)
{
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  const uint32_t w = block_linear_index();
  if (2 * w >= a.Nq)
    return;
  using PS = PairedSearch<BaseT, LPR, NCH, MODE, PSC, HB>;
  PS A, B;
  A.start(a, 2 * w, lds_raw);
  B.start(a, 2 * w + 1, lds_raw + PS::kLdsInts);
  while (A.live || B.live) {
    typename PS::CodeRows ca, cb;
    typename PS::FloatRows fa, fb;
    A.phase1(a, ca);
    B.phase1(a, cb);
    A.phase2(a, ca, fa);
    B.phase2(a, cb, fb);
    A.phase3(a, fa);
    B.phase3(a, fb);
  }
  A.finish(a);
  B.finish(a);
}

template <typename BaseT, int LPR, int NCH, int MODE>
static bool launch_x2_cfg(const QueryArgs& args, hipStream_t stream)
{
  using PSC = typename PsFor<LPR, NCH, MODE>::type;
  if (args.KBuild > 24 / PSC::ROWS * PSC::ROWS)
    return false;
  const uint32_t hb = vis_hash_regs(args.cache - args.sorted);
  if (args.cache != (hb == 1 ? 256u : 512u))
    return false;
  const dim3 grid = grid_for((static_cast<uint64_t>(args.Nq) + 1) / 2);
  using P1 = PairedSearch<BaseT, LPR, NCH, MODE, PSC, 1>;
  using P2 = PairedSearch<BaseT, LPR, NCH, MODE, PSC, 2>;
  const size_t lds1 = 2 * sizeof(int) * P1::kLdsInts, lds2 = 2 * sizeof(int) * P2::kLdsInts;
  if (hb == 1)
    hipLaunchKernelGGL((query_x2_kernel<BaseT, LPR, NCH, MODE, PSC, 1>), grid, dim3(kWave), lds1,
                       stream, args);
  else if (hb == 2)
    hipLaunchKernelGGL((query_x2_kernel<BaseT, LPR, NCH, MODE, PSC, 2>), grid, dim3(kWave), lds2,
                       stream, args);
  else
    return false;
  return true;
}

bool launch_query_x2(const QueryArgs& args, ggnn_dtype dtype, ggnn_measure measure, bool use_ps,
                     hipStream_t stream)
{
  if (dtype != GGNN_F32 || !use_ps || args.sorted > 64 || args.KBuild > kKBlock)
    return false;
  const DistConfig dc = pick_dist_config(args.D, dtype);
  const bool l2 = measure == GGNN_EUCLIDEAN;
  if (dc.lpr == 16 && dc.nch == 2)
    return l2 ? launch_x2_cfg<float, 16, 2, kL2>(args, stream)
              : launch_x2_cfg<float, 16, 2, kCos>(args, stream);
  if (dc.lpr == 8 && dc.nch == 2)
    return l2 ? launch_x2_cfg<float, 8, 2, kL2>(args, stream)
              : launch_x2_cfg<float, 8, 2, kCos>(args, stream);
  return false;
}

}  // namespace ggnn_amd
