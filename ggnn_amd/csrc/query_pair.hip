// query kernel, two searches per wave64 (traversal_pair.hpp): searches whose sorted part is 32
// keys (257+ iterations, KQuery <= 15) and that cannot wrap their visited ring.
// Reference: QueryKernel::operator(), src/ggnn/query/query_layer.cu:39-97; host sizing
// QueryKernelsImpl::query, src/ggnn/query/query_kernels.cu:50-186.
#include <algorithm>

#include "traversal_pair.hpp"
#include "query_args.hpp"

namespace ggnn_amd {

// register budget: 5 waves per SIMD = 96 VGPRs (20 waves = 40 searches per CU: a 10 000-query batch
// is resident at once)
#ifndef GGNN_PAIR_WAVES
#define GGNN_PAIR_WAVES 5
#endif

template <typename BaseT, int LPR, int NCH, int MODE, class PSC, int NB, int SLOTS>
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(GGNN_PAIR_WAVES)))
query_pair_kernel(const QueryArgs a)
{
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  using L = PairLayout<NB, SLOTS>;
  using DE = DistEngine<BaseT, LPR, NCH, PSC::enabled>;
  using PL = PairList<NB, SLOTS>;
  const int lane = threadIdx.x;
  const int li = lane & (kHalf - 1);
  const bool upper = lane >= kHalf;
  const uint32_t pair = block_linear_index();
  if (2 * pair >= a.Nq)
    return;
  // the upper half of the last wave of an odd batch idles: no candidates, no pops, no stores (its
  // loads are clamped to the last search's rows)
  const uint32_t n_raw = 2 * pair + (upper ? 1u : 0u);
  const bool live = n_raw < a.Nq;
  const uint32_t n = live ? n_raw : a.Nq - 1;
  constexpr int kHalfInts = static_cast<int>(L::ints(DE::kQueryLdsBytes));
  int* lds = lds_raw + (upper ? kHalfInts : 0);

  const BaseT* base = static_cast<const BaseT*>(a.base);
  const BaseT* qrow = static_cast<const BaseT*>(a.query) + static_cast<size_t>(n) * a.D;

  // query_layer.cu:48-50 (xi from the MAX nn1 distance, quirk Q4)
  const float nn1 = a.nn1_stats[1];
  const float xi0 = (MODE == kL2) ? (nn1 * nn1) * a.tau * a.tau : nn1 * a.tau;

  DE de;
  de.template load_query<MODE>(base, a.D, qrow, lds + L::kQrow);
  PSC ps;
  if constexpr (PSC::enabled)
    ps.load(a.ps_codes, a.ps_params, a.ps_Dc, reinterpret_cast<const float*>(qrow), a.D);

  PL sl;
  sl.init(static_cast<int>(a.KQuery), xi0, lds,
          a.ring + static_cast<size_t>(n) * (a.cache - a.sorted), static_cast<int>(a.vis_slots));

  PairCounters cnt{0u, 0u, 0u, 0u};
  const int kbuild = static_cast<int>(a.KBuild);

  // fetch_unfiltered(d_starting_points, nullptr, S), query_layer.cu:54-55: chunks of 32 per half
  for (uint32_t i = 0; i < a.num_start; i += kKBlock) {
    const int cand = (live && i + li < a.num_start) ? a.start[i + li] : kEmptyKey;
    if constexpr (PSC::enabled) {
      PairRows<PSC, 8> er;
      er.issue(ps, cand);
      fetch_pair<MODE, false, 8>(sl, de, cand, er, ps, cnt, NoHook{});
    }
    else {
      PairRows<DE, 8> er;
      er.issue(de, cand);
      fetch_pair<MODE, false, 8>(sl, de, cand, er, ps, cnt, NoHook{});
    }
  }

  // speculative graph row of the queue head (query.hip): loaded unconditionally
  int spec_key = kEmptyKey, spec_row = kEmptyKey;
  bool finished = !live;
  for (uint32_t ite = 0; ite < a.max_iters; ++ite) {
    // query_layer.cu:58-63
    const float d0 = sl.dist_at(0);
    sl.xi = (MODE == kL2) ? fminf(xi0, d0 * a.tau * a.tau) : fminf(xi0, d0 * a.tau);
    // decide the pop (simple_knn_cache.cuh:218-224); a search that found nothing to pop has ended
    const int k0 = sl.key_at(sl.BEST);
    const float dq = sl.dist_at(sl.BEST);
    const bool go = !finished && k0 != kEmptyKey && dq < sl.criteria();
    finished = !go;
    if (!__any(go))
      break;
    const int anchor = go ? k0 : kEmptyKey;
    cnt.n_pop += go ? 1u : 0u;
    const bool in_row = li < kbuild;  // KBuild <= 24 (host)
    const bool hit = anchor == spec_key;
    const bool need = go && !hit;
    int cand = (go && hit && in_row) ? spec_row : kEmptyKey;
    if (__any(need)) {
      const int loaded =
          a.graph0[static_cast<size_t>(static_cast<uint32_t>(need ? anchor : 0)) * a.KBuild +
                   min(li, kbuild - 1)];
      cand = (need && in_row) ? loaded : cand;
    }
    auto prefetch_head_row = [&]() {
      spec_key = sl.key_at(sl.BEST);
      spec_row = a.graph0[static_cast<size_t>(static_cast<uint32_t>(max(spec_key, 0))) * a.KBuild +
                          min(li, kbuild - 1)];
      __builtin_amdgcn_s_setprio(1);
    };
    // wave priority as in query.hip: bookkeeping and membership test (while the rows travel) at 0
    if constexpr (PSC::enabled) {
      PairRows<PSC, 6> er;
      er.issue(ps, cand);
      __builtin_amdgcn_s_setprio(0);
      sl.pop_commit(anchor, go);
      fetch_pair<MODE, true, 6>(sl, de, cand, er, ps, cnt, prefetch_head_row);
    }
    else {
      PairRows<DE, 6> er;
      er.issue(de, cand);
      __builtin_amdgcn_s_setprio(0);
      sl.pop_commit(anchor, go);
      fetch_pair<MODE, true, 6>(sl, de, cand, er, ps, cnt, prefetch_head_row);
    }
  }

  // write_best + dists, query_layer.cu:81-90 (EMPTY becomes -1 + offset, as in the reference)
  const size_t out_row = (static_cast<size_t>(n) * a.shards_per_gpu + a.on_gpu_shard) * a.KQuery;
  const int32_t id_offset = static_cast<int32_t>(a.on_gpu_shard * a.N_base);
  if (live && li < static_cast<int>(a.KQuery)) {
    a.ids[out_row + li] = sl.key + id_offset;
    a.dists[out_row + li] = sl.dist;
  }
  if (live && li == 0) {
    if (a.n_dist)
      a.n_dist[n] = cnt.n_dist;
    if (a.n_pop)
      a.n_pop[n] = cnt.n_pop;
    if (a.n_rows)
      a.n_rows[n] = make_uint2(cnt.float_rows, cnt.code_rows);
  }
}

// buckets of the pair kernel's tag set: {bucket bits, tags per bucket}, {0, 0} = not served
struct PairSet {
  int nb, slots;
};
static PairSet pick_pair_set(uint32_t vis, uint32_t max_iters, uint32_t n_base)
{
  // every key must fit nb + 16 bits (the tag is exact: traversal.hpp "long rings")
  auto fits = [n_base](int nb) { return static_cast<uint64_t>(n_base) <= (1ull << (nb + 16)); };
  if (vis <= 480 && max_iters <= 320 && fits(8))
    return {8, 4};  // <= 320 keys in 1024 slots
  if (vis <= 992 && fits(8))
    return {8, 8};  // <= 992 keys in 2048 slots
  if (vis <= 2016 && fits(9))
    return {9, 8};
  return {0, 0};
}

template <typename BaseT, int LPR, int NCH, int MODE, class PSC>
static bool launch_pair_set(const QueryArgs& args, const PairSet set, hipStream_t stream)
{
  using DE = DistEngine<BaseT, LPR, NCH, PSC::enabled>;
  const dim3 grid = grid_for((static_cast<uint64_t>(args.Nq) + 1) / 2);
#define GGNN_PAIR_LAUNCH(NB, SL)                                                                   \
  do {                                                                                             \
    using Lay = PairLayout<NB, SL>;                                                                \
    const size_t lds_bytes = 2 * Lay::ints(DE::kQueryLdsBytes) * sizeof(int);                      \
    hipLaunchKernelGGL((query_pair_kernel<BaseT, LPR, NCH, MODE, PSC, NB, SL>), grid, dim3(kWave), \
                       lds_bytes, stream, args);                                                   \
  } while (0)
  if (set.nb == 8 && set.slots == 4)
    GGNN_PAIR_LAUNCH(8, 4);
  else if (set.nb == 8)
    GGNN_PAIR_LAUNCH(8, 8);
  else
    GGNN_PAIR_LAUNCH(9, 8);
#undef GGNN_PAIR_LAUNCH
  return true;
}

template <typename BaseT, int LPR, int NCH>
static bool launch_pair_cfg(const QueryArgs& args, const PairSet set, bool use_ps,
                            ggnn_measure measure, hipStream_t stream)
{
  if constexpr (std::is_same<BaseT, float>::value) {
    if (!use_ps)
      return false;  // float rows are read behind the 128-byte code rows only
    if (measure == GGNN_EUCLIDEAN)
      return launch_pair_set<BaseT, LPR, NCH, kL2, Prescreen<8, 1, kL2>>(args, set, stream);
    return launch_pair_set<BaseT, LPR, NCH, kCos, Prescreen<8, 1, kCos>>(args, set, stream);
  }
  else {
    if (measure == GGNN_EUCLIDEAN)
      return launch_pair_set<BaseT, LPR, NCH, kL2, NoPrescreen>(args, set, stream);
    return launch_pair_set<BaseT, LPR, NCH, kCos, NoPrescreen>(args, set, stream);
  }
}

// true: the launch is a candidate for the pair kernel (launch_query then provides args.ring, the
// overflow lists)
bool query_pair_eligible(const QueryLaunch& a, uint32_t sorted, uint32_t cache)
{
  if (hook(kHookQueryPair) == 0 || hook(kHookQueryEarly) == 0 || hook(kHookQueryGlobalRing) == 0)
    return false;
  const uint32_t vis = cache - sorted;
  if (sorted != static_cast<uint32_t>(kHalf) || a.KBuild > 24 || a.KBuild < 1 || a.max_iterations > vis)
    return false;
  const uint32_t row_bytes = a.D * (a.dtype == GGNN_F32 ? 4u : 1u);
  if (a.dtype == GGNN_F32) {
    // first read: 8 lanes x 16 bytes of pre-screen codes
    if (!(a.ps_codes && a.ps_params) || row_bytes > 512)
      return false;
  }
  else if (row_bytes > 128)
    return false;
  return pick_pair_set(vis, a.max_iterations, a.N_base).nb != 0;
}

bool launch_query_pair(const QueryArgs& args, ggnn_dtype dtype, ggnn_measure measure, bool use_ps,
                       hipStream_t stream)
{
  const PairSet set = pick_pair_set(args.cache - args.sorted, args.max_iters, args.N_base);
  if (set.nb == 0 || !args.ring)
    return false;
  const DistConfig dc = pick_dist_config(args.D, dtype);
  if (dtype == GGNN_F32) {
    if (dc.lpr == 8 && dc.nch == 1)
      return launch_pair_cfg<float, 8, 1>(args, set, use_ps, measure, stream);
    if (dc.lpr == 8 && dc.nch == 2)
      return launch_pair_cfg<float, 8, 2>(args, set, use_ps, measure, stream);
    if (dc.lpr == 8 && dc.nch == 3)
      return launch_pair_cfg<float, 8, 3>(args, set, use_ps, measure, stream);
    if (dc.lpr == 16 && dc.nch == 2)
      return launch_pair_cfg<float, 16, 2>(args, set, use_ps, measure, stream);
    return false;
  }
  if (dc.lpr == 8 && dc.nch == 1)
    return launch_pair_cfg<uint8_t, 8, 1>(args, set, use_ps, measure, stream);
  return false;
}

}  // namespace ggnn_amd
