// query kernel: best-first graph traversal, one wave64 per query.
// Reference: QueryKernel::operator(), src/ggnn/query/query_layer.cu:39-97; host sizing
// QueryKernelsImpl::query, src/ggnn/query/query_kernels.cu:50-186.
#include <algorithm>

#include "traversal.hpp"
#include "query_args.hpp"

namespace ggnn_amd {


template <class PSC, typename BaseT>
GGNN_DEV void load_prescreen(PSC& ps, const QueryArgs& a, const BaseT* qrow)
{
  if constexpr (PSC::enabled)
    ps.load(a.ps_codes, a.ps_params, a.ps_Dc, reinterpret_cast<const float*>(qrow), a.D);
}

// occupancy target of the common instantiations (one register of list per lane, narrow rows):
// a tuning knob, 1 = leave it to the compiler
#ifndef GGNN_QUERY_WAVES
#define GGNN_QUERY_WAVES 7
#endif
// early rows: the requested code rows (15 registers) are live across the pop's bookkeeping and the
// membership test; with the float query row in LDS (DistEngine<.., QL>) the kernels still fit the
// 72 registers of 7 waves.  The global-ring variants spill 9-15 registers there: 80 (6 waves; they
// exist for caches whose LDS ring would allow fewer)
#ifndef GGNN_QUERY_WAVES_GR
#define GGNN_QUERY_WAVES_GR 7
#endif

// EARLY (R = 1, KBuild <= 24; traversal.hpp "Early rows"): the first-read rows of a pop's neighbours
// are requested before the pop's bookkeeping and the membership test instead of after them.
// GR (with EARLY and a hashed set): the visited ring in global memory (SortedList<R, HB, true>).
// (fetch_early<.., COUNT = false> -- the sorted part of the cache tested behind the verdicts, for the
// candidates still in the race -- is used by the merge kernel only: measured here on one box,
// round 6, it is 3 % SLOWER on 10 000-query batches (1.186 -> 1.225 ms headline, 7.36 -> 7.57 ms
// lowrank24 at 1.0 / 750: the per-candidate compare chain sits between the verdicts and the float
// rows of a wave that is bound by its own latency) and even on 100 000-query batches.)
template <typename BaseT, int LPR, int NCH, int R, int MODE, class PSC, int HB = 0, bool EARLY = false,
          bool GR = false>
__global__ void __launch_bounds__(kWave) __attribute__((
    amdgpu_waves_per_eu((R == 1 && NCH <= 3) ? (GR ? GGNN_QUERY_WAVES_GR : GGNN_QUERY_WAVES) : 1)))
query_kernel(const QueryArgs a)
{
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  // tag-set form: the visited ring is not in LDS, the candidate scratch follows the sorted keys
  const WaveLds lds(lds_raw, (is_tag_set(HB) || GR) ? a.sorted : a.cache);
  const int lane = threadIdx.x;
  const uint32_t n = block_linear_index();
  if (n >= a.Nq)
    return;

  const BaseT* base = static_cast<const BaseT*>(a.base);
  const BaseT* query = static_cast<const BaseT*>(a.query);

  // query_layer.cu:48-50 (xi from the MAX nn1 distance, quirk Q4)
  const float nn1 = a.nn1_stats[1];
  const float xi = (MODE == kL2) ? (nn1 * nn1) * a.tau * a.tau : nn1 * a.tau;

  // early rows + pre-screen: the float query row waits in LDS behind the wave's other regions (its
  // registers are needed while the requested code rows are live across the membership test)
  using DE = DistEngine<BaseT, LPR, NCH, EARLY && PSC::enabled>;
  DE de;
  de.template load_query<MODE>(
      base, a.D, query + static_cast<size_t>(n) * a.D,
      lds_raw + (is_tag_set(HB) ? tag_set_lds_ints(a.sorted, static_cast<uint32_t>(-HB))
                                : wave_lds_ints(GR ? a.sorted : a.cache, HB)));
  PSC ps;
  load_prescreen(ps, a, query + static_cast<size_t>(n) * a.D);

  SortedList<R, HB, GR> sl;
  if constexpr (GR && !is_tag_set(HB))
    sl.init_global_ring(a.KQuery, a.sorted, a.cache, xi, lds.known, static_cast<int>(a.vis_slots),
                        a.ring + static_cast<size_t>(n) * (a.cache - a.sorted));
  else if constexpr (is_tag_set(HB))
    sl.init_tagged(a.KQuery, a.sorted, a.cache, xi, lds.known, static_cast<int>(a.vis_slots),
                   a.ring + static_cast<size_t>(n) * (a.cache - a.sorted));
  else
    sl.init(a.KQuery, a.sorted, a.cache, xi, lds.known, static_cast<int>(a.vis_slots));

  uint32_t cnt_dist = 0, cnt_pop = 0;
  uint2 cnt_rows = make_uint2(0u, 0u);

  // fetch_unfiltered(d_starting_points, nullptr, S), query_layer.cu:54-55
  for (uint32_t i = 0; i < a.num_start; i += kKBlock) {
    const int cand = (lane < (int)kKBlock && i + lane < a.num_start) ? a.start[i + lane]
                                                                      : kEmptyKey;
    cnt_dist += fetch<MODE, false>(sl, de, lds, cand, nullptr, ps, cnt_rows);
  }

  // Speculation that hides one of the dependent memory latencies per pop: while the pre-screen
  // and distance phases of this pop run, the graph row of the current queue head is loaded; if
  // that key is still the head at the next pop (no closer candidate was pushed: 73 % of the pops)
  // the row is there.  The load is issued from fetch()'s after-filter hook, i.e. after the wait
  // for this pop's own graph row -- issued before it, the two waits merge into one vmcnt(0).
  int spec_key = kEmptyKey, spec_row = kEmptyKey;
#ifdef GGNN_PHASE_CYCLES
  phase_begin();
#endif
  for (uint32_t ite = 0; ite < a.max_iters; ++ite) {
    // query_layer.cu:58-63
    const float d0 = sl.dist_at(0);
    sl.xi = (MODE == kL2) ? fminf(xi, d0 * a.tau * a.tau) : fminf(xi, d0 * a.tau);
    if constexpr (EARLY) {
      // the same pop, reordered: decide -> graph row (speculated, else loaded now) -> request the
      // neighbours' first-read rows -> bookkeeping of the pop and membership test under that latency
      const int anchor = sl.peek(sl.criteria());
      if (anchor == kEmptyKey)
        break;
      ++cnt_pop;
      const bool in_row = lane < static_cast<int>(a.KBuild);  // KBuild <= 24 (host)
      int cand;
      if (anchor == spec_key)
        cand = in_row ? spec_row : kEmptyKey;
      else
        cand = in_row ? a.graph0[static_cast<size_t>(static_cast<uint32_t>(anchor)) * a.KBuild + lane]
                      : kEmptyKey;
      // The speculative row is loaded UNCONDITIONALLY (an empty queue reads row 0, lanes past the
      // row its last entry; masked where the row is consumed): a load under a branch leaves the
      // two paths with different numbers of loads in flight, and the compiler then waits for the
      // requested code rows with vmcnt(0) -- i.e. also for this load, issued a moment earlier
      // (found in the ISA; the wait is vmcnt(1) now and the row travels during the verdicts).
      auto prefetch_head_row = [&]() {
        spec_key = sl.key_at(sl.BEST);
        spec_row = a.graph0[static_cast<size_t>(static_cast<uint32_t>(max(spec_key, 0))) * a.KBuild +
                            min(lane, static_cast<int>(a.KBuild) - 1)];
        __builtin_amdgcn_s_setprio(1);  // (the membership test is done: see below)
      };
      // Wave priority.  What a wave does WHILE its requested rows travel (bookkeeping of the pop,
      // membership test) is free as long as it finishes before they arrive; everything else --
      // verdicts -> float rows -> distances -> replay -> peek -> graph row -> the next requests --
      // is on the way to the wave's next memory request.  The seven waves of a SIMD compete for
      // its issue slots (VALU issue ~0.7 busy), so the first kind runs at priority 0 and yields
      // to waves of the second kind (priority 1): a pure scheduling hint, results unchanged.
      // Same box, alternating runs: 1M x 128 f32 1.234-1.242 -> 1.209-1.210 ms, uint8 0.883-0.885
      // -> 0.860-0.870, 12.5M x 96 2.014 -> 1.931 ms, 100k-query batches -1.5 ... -2.6 %.  (The
      // inverse assignment: +1 %; only the bookkeeping at low priority: +1 %; a third level for
      // peek -> requests: -0.3 %, inside the noise.)
      if constexpr (PSC::enabled) {
        EarlyRows<PSC> er;
        er.issue(ps, cand);
        __builtin_amdgcn_s_setprio(0);
        sl.pop_commit(anchor, lds.known);
        cnt_dist += fetch_early<MODE>(sl, de, lds, cand, er, ps, cnt_rows, prefetch_head_row);
      }
      else {
        EarlyRows<DE> er;
        er.issue(de, cand);
        __builtin_amdgcn_s_setprio(0);
        sl.pop_commit(anchor, lds.known);
        cnt_dist += fetch_early<MODE>(sl, de, lds, cand, er, ps, cnt_rows, prefetch_head_row);
      }
      continue;
    }
    const int anchor = sl.pop(sl.criteria(), lds.known);
    GGNN_TICK(0);  // pop
    if (anchor == kEmptyKey)
      break;
    ++cnt_pop;
    // query_layer.cu:69-77
    const int32_t* row = a.graph0 + static_cast<size_t>(static_cast<uint32_t>(anchor)) * a.KBuild;
    for (uint32_t i = 0; i < a.KBuild; i += kKBlock) {
      const bool in_row = lane < (int)kKBlock && i + lane < a.KBuild;
      int cand;
      if (i == 0 && anchor == spec_key)
        cand = spec_row;
      else
        cand = in_row ? row[i + lane] : kEmptyKey;
      auto prefetch_head_row = [&]() {
        if (i == 0) {
          spec_key = sl.key_at(sl.BEST);
          if (spec_key != kEmptyKey)
            spec_row = in_row ? a.graph0[static_cast<size_t>(static_cast<uint32_t>(spec_key)) *
                                             a.KBuild + lane]
                              : kEmptyKey;
        }
      };
      cnt_dist += fetch<MODE, true>(sl, de, lds, cand, nullptr, ps, cnt_rows, prefetch_head_row);
    }
  }

#ifdef GGNN_PHASE_CYCLES
  phase_end();
#endif
  // write_best + dists, query_layer.cu:81-90 (EMPTY becomes -1 + offset, as in the reference)
  const size_t out_row = (static_cast<size_t>(n) * a.shards_per_gpu + a.on_gpu_shard) * a.KQuery;
  const int32_t id_offset = static_cast<int32_t>(a.on_gpu_shard * a.N_base);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = r * kWave + lane;
    if (i < a.KQuery) {
      a.ids[out_row + i] = sl.key[r] + id_offset;
      a.dists[out_row + i] = sl.dist[r];
    }
  }
  if (lane == 0) {
    if (a.n_dist)
      a.n_dist[n] = cnt_dist;
    if (a.n_pop)
      a.n_pop[n] = cnt_pop;
    if (a.n_rows)
      a.n_rows[n] = cnt_rows;
  }
}

// Same kernel with the LDS-resident list (SORTED > 512, i.e. KQuery > 495).
template <typename BaseT, int LPR, int NCH, int MODE, class PSC>
__global__ void __launch_bounds__(kWave) query_kernel_lds(const QueryArgs a)
{
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  // [cache keys][sorted dists][ckeys 32 | cd0 32 | cd1 32]
  int* keys = lds_raw;
  float* dists = reinterpret_cast<float*>(lds_raw + a.cache);
  const WaveLds lds(lds_raw + a.cache + a.sorted, 0);
  const int lane = threadIdx.x;
  const uint32_t n = block_linear_index();
  if (n >= a.Nq)
    return;
  const BaseT* base = static_cast<const BaseT*>(a.base);
  const BaseT* query = static_cast<const BaseT*>(a.query);
  const float nn1 = a.nn1_stats[1];
  const float xi = (MODE == kL2) ? (nn1 * nn1) * a.tau * a.tau : nn1 * a.tau;
  DistEngine<BaseT, LPR, NCH> de;
  de.template load_query<MODE>(base, a.D, query + static_cast<size_t>(n) * a.D);
  PSC ps;
  load_prescreen(ps, a, query + static_cast<size_t>(n) * a.D);
  LdsList sl;
  sl.init(a.KQuery, a.sorted, a.cache, xi, keys, dists);
  uint32_t cnt_dist = 0, cnt_pop = 0;
  uint2 cnt_rows = make_uint2(0u, 0u);
  for (uint32_t i = 0; i < a.num_start; i += kKBlock) {
    const int cand = (lane < (int)kKBlock && i + lane < a.num_start) ? a.start[i + lane]
                                                                      : kEmptyKey;
    cnt_dist += fetch<MODE, false>(sl, de, lds, cand, nullptr, ps, cnt_rows);
  }
  for (uint32_t ite = 0; ite < a.max_iters; ++ite) {
    __syncthreads();
    const float d0 = sl.dist_at(0);
    sl.xi = (MODE == kL2) ? fminf(xi, d0 * a.tau * a.tau) : fminf(xi, d0 * a.tau);
    const int anchor = sl.pop(sl.criteria());
    if (anchor == kEmptyKey)
      break;
    ++cnt_pop;
    const int32_t* row = a.graph0 + static_cast<size_t>(static_cast<uint32_t>(anchor)) * a.KBuild;
    for (uint32_t i = 0; i < a.KBuild; i += kKBlock) {
      const int cand = (lane < (int)kKBlock && i + lane < a.KBuild) ? row[i + lane] : kEmptyKey;
      cnt_dist += fetch<MODE, true>(sl, de, lds, cand, nullptr, ps, cnt_rows);
    }
  }
  __syncthreads();
  const size_t out_row = (static_cast<size_t>(n) * a.shards_per_gpu + a.on_gpu_shard) * a.KQuery;
  const int32_t id_offset = static_cast<int32_t>(a.on_gpu_shard * a.N_base);
  for (uint32_t i = lane; i < a.KQuery; i += kWave) {
    a.ids[out_row + i] = keys[i] + id_offset;
    a.dists[out_row + i] = dists[i];
  }
  if (lane == 0) {
    if (a.n_dist)
      a.n_dist[n] = cnt_dist;
    if (a.n_pop)
      a.n_pop[n] = cnt_pop;
    if (a.n_rows)
      a.n_rows[n] = cnt_rows;
  }
}

void query_sizing(uint32_t D, uint32_t k_query, uint32_t max_iterations, uint32_t* cache_size,
                  uint32_t* sorted_size)
{
  // query_kernels.cu:55-110
  GGNN_REQUIRE(k_query >= 1 && k_query <= 6000, GGNN_INVALID_ARGUMENT, "KQuery must be in [1, 6000]");
  GGNN_REQUIRE(max_iterations <= 8192, GGNN_INVALID_ARGUMENT, "max_iterations must be <= 8192");
  GGNN_REQUIRE(D >= 1 && D <= 4096, GGNN_INVALID_ARGUMENT, "D must be in [1, 4096]");
  const uint32_t required_sorted = next_multiple32(k_query + 1 + 16);
  const uint32_t cache =
      std::max(std::max(256u, required_sorted + 32u), bit_ceil_u32(max_iterations));
  GGNN_REQUIRE(cache <= 8192, GGNN_INVALID_ARGUMENT, "cache size exceeds 8192");
  *cache_size = cache;
  *sorted_size = std::max(cache < 512u ? 64u : 32u, required_sorted);
}

template <typename BaseT, int LPR, int NCH, int MODE, class PSC>
static void launch_query_lds(const QueryArgs& args, uint32_t sorted, hipStream_t stream);

// layouts whose first row read is 8 lanes x one 16-byte chunk: Prescreen<8,1> next to any float
// layout, or rows of <= 128 bytes read directly
template <int LPR, int NCH, class PSC>
constexpr bool early_rows_layout()
{
  return PSC::enabled ? (PsLayout<PSC>::lpr == 8 && PsLayout<PSC>::nch == 1) : (LPR == 8 && NCH == 1);
}

template <typename BaseT, int LPR, int NCH, int MODE, class PSC>
static void launch_query_r(const QueryArgs& args, uint32_t sorted, hipStream_t stream)
{
  const size_t lds = wave_lds_bytes(args.cache);
  // one list register per lane: the visited ring is mirrored in a hash set (traversal.hpp) when it
  // is short enough for one or two bucket registers
  // (not for the two-chunk float layouts without pre-screen: four rows of two chunks in flight leave
  // no register for it at 7 waves per SIMD -- measured 2.73 vs 2.54 ms with the spills)
  const bool fits = PSC::enabled || NCH == 1;
  const uint32_t hb = (sorted <= 64 && fits) ? vis_hash_regs(args.cache - sorted) : 0;
  // early rows (traversal.hpp): graph rows of <= 24 neighbours, first row read 8 lanes x 16 bytes
  // (hook QUERY_EARLY = 0: the round-1..4 order, A/B and test hook)
  if constexpr (early_rows_layout<LPR, NCH, PSC>()) {
    // the tag set of long rings (513..2016 iterations): early rows only when the search cannot
    // wrap its ring -- then the set is ring-less too (no store per pop: a store in flight turns
    // every wait for the requested rows into vmcnt(0)); otherwise the round-4 order below
    const bool tagged = hb == 0 && fits && args.ring && (args.tag_bits == 8 || args.tag_bits == 9);
    const bool tagged_ringless = tagged && args.max_iters <= args.cache - sorted &&
                                 hook(kHookQueryGlobalRing) != 0;
    if (args.KBuild <= 8 * kEarlySteps && sorted <= 64 && tagged_ringless &&
        hook(kHookQueryEarly) != 0) {
      const size_t tag_lds = tag_set_lds_bytes(sorted, args.cache - sorted) +
                             DistEngine<BaseT, LPR, NCH, PSC::enabled>::kQueryLdsBytes;
      if (args.tag_bits == 8)
        hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, -8, true, true>),
                           grid_for(args.Nq), dim3(kWave), tag_lds, stream, args);
      else
        hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, -9, true, true>),
                           grid_for(args.Nq), dim3(kWave), tag_lds, stream, args);
      return;
    }
    if (args.KBuild <= 8 * kEarlySteps && sorted <= 64 && !tagged && hook(kHookQueryEarly) != 0) {
      // (hook QUERY_LDS_PAD: extra bytes of LDS per wave -- occupancy experiments without a rebuild)
      const size_t qrow = DistEngine<BaseT, LPR, NCH, PSC::enabled>::kQueryLdsBytes +
                          static_cast<size_t>(std::clamp<int64_t>(hook(kHookQueryLdsPad), 0, 32768));
      // a search that cannot wrap its visited ring needs no ring: buckets + stash ARE the set
      // (SortedList<R, HB, true>; launch_query allocates the overflow lists then)
      const bool ringless = args.ring && args.tag_bits == 0;
      if (hb == 1 && ringless)
        hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 1, true, true>),
                           grid_for(args.Nq), dim3(kWave), wave_lds_bytes(sorted, 1) + qrow, stream,
                           args);
      else if (hb == 1)
        hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 1, true>), grid_for(args.Nq),
                           dim3(kWave), wave_lds_bytes(args.cache, 1) + qrow, stream, args);
      else if (hb == 2 && ringless)
        hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 2, true, true>),
                           grid_for(args.Nq), dim3(kWave), wave_lds_bytes(sorted, 2) + qrow, stream,
                           args);
      else if (hb == 2)
        hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 2, true>), grid_for(args.Nq),
                           dim3(kWave), wave_lds_bytes(args.cache, 2) + qrow, stream, args);
      else
        hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 0, true>), grid_for(args.Nq),
                           dim3(kWave), lds + qrow, stream, args);
      return;
    }
  }
  // long rings (searches of 1000-2000 iterations): tag set + ring in global memory (traversal.hpp)
  if (hb == 0 && sorted <= 64 && fits && args.ring && args.tag_bits == 8)
    hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, -8>), grid_for(args.Nq),
                       dim3(kWave), tag_set_lds_bytes(sorted, args.cache - sorted), stream, args);
  else if (hb == 0 && sorted <= 64 && fits && args.ring && args.tag_bits == 9)
    hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, -9>), grid_for(args.Nq),
                       dim3(kWave), tag_set_lds_bytes(sorted, args.cache - sorted), stream, args);
  else if (hb == 1)
#ifdef GGNN_PHASE_CYCLES
    hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 1>), grid_for(args.Nq), dim3(kWave),
                       16384 + 256, stream, args);
#else
    hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 1>), grid_for(args.Nq), dim3(kWave),
                       wave_lds_bytes(args.cache, 1), stream, args);
#endif
  else if (hb == 2)
    hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 2>), grid_for(args.Nq), dim3(kWave),
                       wave_lds_bytes(args.cache, 2), stream, args);
  else if (sorted <= 64)
    hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 1, MODE, PSC>), grid_for(args.Nq), dim3(kWave), lds,
                       stream, args);
  else if (sorted <= 128)
    hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 2, MODE, PSC>), grid_for(args.Nq), dim3(kWave), lds,
                       stream, args);
  else if (sorted <= 256)
    hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 4, MODE, PSC>), grid_for(args.Nq), dim3(kWave), lds,
                       stream, args);
  else if (sorted <= 512)  // KQuery <= 495: eight list registers per lane
    hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 8, MODE, PSC>), grid_for(args.Nq), dim3(kWave), lds,
                       stream, args);
  else if constexpr (!PSC::enabled) {
    // KQuery <= 1007 / 2031: 16 / 32 list registers per lane.  A push is then 16 / 32 lock-step
    // register steps (~12 instructions each) instead of a walk through LDS with a round trip or
    // two per 64 entries -- with K this large nearly every evaluated candidate is pushed, so
    // the pushes ARE the search (K = 1000 / 4 000 iterations: 24 pushes per pop).  Launched
    // without the pre-screen (launch_query_cfg): a loose criteria rejects nothing.
    if (sorted <= 1024)
      hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 16, MODE, PSC>), grid_for(args.Nq), dim3(kWave),
                         lds, stream, args);
    else if (sorted <= 2048)
      hipLaunchKernelGGL((query_kernel<BaseT, LPR, NCH, 32, MODE, PSC>), grid_for(args.Nq), dim3(kWave),
                         lds, stream, args);
    else
      launch_query_lds<BaseT, LPR, NCH, MODE, PSC>(args, sorted, stream);
  }
  else
    launch_query_lds<BaseT, LPR, NCH, MODE, PSC>(args, sorted, stream);
}

template <typename BaseT, int LPR, int NCH, int MODE, class PSC>
static void launch_query_lds(const QueryArgs& args, uint32_t sorted, hipStream_t stream)
{
  {
    // SORTED > 512: sorted list in LDS (keys [cache] + dists [sorted] + candidate scratch)
    const size_t lds_big = (args.cache + sorted + WaveLds::extra_ints) * sizeof(int);
    GGNN_REQUIRE(lds_big <= 64 * 1024, GGNN_UNSUPPORTED, "cache too large for one workgroup");
    hipLaunchKernelGGL((query_kernel_lds<BaseT, LPR, NCH, MODE, PSC>), grid_for(args.Nq), dim3(kWave),
                       lds_big, stream, args);
  }
}

template <typename BaseT, int LPR, int NCH>
static void launch_query_cfg(const QueryArgs& args, bool use_ps, ggnn_measure measure,
                             hipStream_t stream)
{
  if constexpr (std::is_same<BaseT, float>::value) {
    if (use_ps && args.sorted <= 512) {
      if (measure == GGNN_EUCLIDEAN)
        launch_query_r<BaseT, LPR, NCH, kL2, typename PsFor<LPR, NCH, kL2>::type>(
            args, args.sorted, stream);
      else
        launch_query_r<BaseT, LPR, NCH, kCos, typename PsFor<LPR, NCH, kCos>::type>(
            args, args.sorted, stream);
      return;
    }
  }
  if (measure == GGNN_EUCLIDEAN)
    launch_query_r<BaseT, LPR, NCH, kL2, NoPrescreen>(args, args.sorted, stream);
  else
    launch_query_r<BaseT, LPR, NCH, kCos, NoPrescreen>(args, args.sorted, stream);
}

void launch_query(const QueryLaunch& a, hipStream_t stream)
{
  if (a.Nq == 0)
    return;
  check_vector_layout(a.base, a.D, a.dtype);
  check_vector_layout(a.query, a.D, a.dtype);
  QueryArgs args{};
  args.base = a.base;
  args.query = a.query;
  args.graph0 = a.graph0;
  args.start = a.start;
  args.nn1_stats = a.nn1_stats;
  args.ids = a.ids;
  args.dists = a.dists;
  args.n_dist = a.n_dist;
  args.n_pop = a.n_pop;
  args.n_rows = reinterpret_cast<uint2*>(a.n_rows);
  args.D = a.D;
  args.Nq = a.Nq;
  args.N_base = a.N_base;
  args.KBuild = a.KBuild;
  args.num_start = a.num_start;
  args.KQuery = a.k_query;
  query_sizing(a.D, a.k_query, a.max_iterations, &args.cache, &args.sorted);
  args.max_iters = a.max_iterations;
  args.shards_per_gpu = a.shards_per_gpu;
  args.on_gpu_shard = a.on_gpu_shard;
  args.tau = a.tau_query;
  args.vis_slots = vis_slots_hook();
  // long rings: per-query visited rings as stream-ordered scratch of this launch
  const uint32_t vis = args.cache - args.sorted;
  const bool use_ps = a.ps_codes && a.ps_params && a.dtype == GGNN_F32;
  // what launch_query_cfg / launch_query_r will pick for this shape, decided HERE so that the
  // per-query scratch below is only allocated for kernels that use it (round-5 advisor finding:
  // Nq x ring x 4 bytes -- 77 MB per 100k-query launch -- also went to layouts that keep the ring
  // in LDS): the early-rows layouts (first row read 8 lanes x 16 bytes: Prescreen<8, 1> next to
  // every float layout up to 128 dimensions, or rows of <= 128 bytes read directly) and the
  // layouts that carry a hashed / tag set at all (pre-screened, or one chunk per lane)
  const DistConfig dc = pick_dist_config(a.D, a.dtype);
  const bool ps_kernel = use_ps && args.sorted <= 512;
  const bool early_shape = ps_kernel ? !((dc.lpr == 16 && dc.nch == 4) || dc.lpr == 64)
                                     : (dc.lpr == 8 && dc.nch == 1);
  const bool set_shape = ps_kernel || dc.nch == 1;
  // ring-less hashed set (launch_query_r: early rows) when the search cannot wrap its ring: the
  // overflow lists of the launch (hook QUERY_GLOBAL_RING = 0: ring in LDS, A/B and test hook)
  const bool global_ring = args.sorted <= 64 && vis_hash_regs(vis) != 0 && a.max_iterations <= vis &&
                           a.KBuild <= 8 * kEarlySteps && hook(kHookQueryEarly) != 0 &&
                           hook(kHookQueryGlobalRing) != 0 && early_shape && set_shape;
  if (global_ring || (args.sorted <= 64 && set_shape && tag_set_usable(vis, a.N_base) &&
                      hook(kHookVisTagSet) != 0)) {
    args.tag_bits = global_ring ? 0 : tag_set_bucket_bits(vis);
    try {
      args.ring = static_cast<int32_t*>(
          scratch_alloc(static_cast<size_t>(a.Nq) * vis * sizeof(int32_t), stream));
    }
    catch (const Error& e) {
      // no room for the rings (Nq x ring x 4 bytes): the ring-scan kernel needs none
      if (e.status != GGNN_OUT_OF_MEMORY)
        throw;
      (void)hipGetLastError();
      args.ring = nullptr;
    }
  }
  struct RingGuard {
    void* p;
    hipStream_t s;
    ~RingGuard()
    {
      if (p)
        scratch_free(p, s);
    }
  } ring_guard{args.ring, stream};
  if (use_ps) {
    GGNN_REQUIRE(a.ps_Dc == prescreen_code_dim(a.D), GGNN_INVALID_ARGUMENT,
                 "pre-screen code rows must have prescreen_code_dim(D) bytes");
    GGNN_REQUIRE((reinterpret_cast<uintptr_t>(a.ps_codes) & 15u) == 0 &&
                     (reinterpret_cast<uintptr_t>(a.ps_params) & 15u) == 0,
                 GGNN_INVALID_ARGUMENT, "pre-screen buffers must be 16-byte aligned");
    args.ps_codes = a.ps_codes;
    args.ps_params = a.ps_params;
    args.ps_Dc = a.ps_Dc;
  }

#define GGNN_LAUNCH_QUERY(T, LPR, NCH) launch_query_cfg<T, LPR, NCH>(args, use_ps, a.measure, stream)
  GGNN_DISPATCH_DIST(a.dtype, a.D, GGNN_LAUNCH_QUERY);
#undef GGNN_LAUNCH_QUERY
  GGNN_HIP_CHECK(hipGetLastError());
}

#ifdef GGNN_PHASE_CYCLES
extern "C" int ggnn_debug_phase_cycles(unsigned long long* out16, int reset)
{
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_acc), 16 * sizeof(unsigned long long)) != hipSuccess)
    return 1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_acc), z, sizeof(z)) != hipSuccess)
      return 1;
  }
  return 0;
}
#endif

}  // namespace ggnn_amd
