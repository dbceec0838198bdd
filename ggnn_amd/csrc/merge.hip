// merge kernel: hierarchical search from the top segment down to layer_btm for every point of
// layer_btm; one wave64 per point.
// Reference: MergeKernel::operator() and get_top_seg_offset, src/ggnn/construction/merge_layer.cu
// :40-158; sizing include/ggnn/construction/merge_layer.cuh:40-65.
#include "traversal.hpp"

namespace ggnn_amd {

struct MergeArgs {
  const void* base;
  const int32_t* graph_all;
  const int32_t* translation_all;
  const int32_t* selection_all;
  const float* nn1_stats;
  int32_t* graph_buffer;
  float* nn1_dist_buffer;
  uint32_t* n_dist;
  uint4* n_work;  // optional [N_btm]: distance evaluations, float rows, code rows, pops per point
  uint32_t D, KBuild, S, G, S0, S0_off, layer_top, layer_btm, sorted, N_btm;
  uint32_t Ns_off[kLayers], STs_off[kLayers];
  float tau;
  // optional pre-screen copy of the base coded for this measure (prescreen.hip); float32 only
  const uint8_t* ps_codes;
  const float* ps_params;
  uint32_t ps_Dc;
  uint32_t vis_slots;  // usable keys per bucket of the hashed visited set (kVisSlots; test hook)
  uint32_t xcd_map;    // XCD-aware block -> point mapping (common.hpp)
};

// occupancy target of the common instantiations (as for the query kernel): a tuning knob
#ifndef GGNN_MERGE_WAVES
#define GGNN_MERGE_WAVES 7
#endif

constexpr uint32_t kMergeCache = 256;      // merge_layer.cuh:44
constexpr uint32_t kMergeIterations = 200; // merge_layer.cuh:43

uint32_t merge_sorted_size(uint32_t KBuild)
{
  // merge_layer.cuh:68-69 (CACHE_SIZE = 256 < 512)
  return std::max(64u, next_multiple32(KBuild + 1 + 16));
}

#ifndef GGNN_MERGE_WAVES_EARLY
#define GGNN_MERGE_WAVES_EARLY 6   // see GGNN_QUERY_WAVES_EARLY (query.hip)
#endif

// EARLY (R = 1, KBuild <= 24): the pop order of traversal.hpp "Early rows", as in the query kernel
// COUNT = false (launches without work counters: every production build): the membership test
// against the sorted part runs behind the verdicts, for the candidates that are still in the race
// (fetch_early<.., false>, SortedList::drop_sorted); same graph.  Build 0.448 -> 0.434 s (1M x 128).
template <typename BaseT, int LPR, int NCH, int R, int MODE, class PSC, int HB = 0, bool EARLY = false,
          bool COUNT = true>
__global__ void __launch_bounds__(kWave)
    __attribute__((amdgpu_waves_per_eu((R == 1 && NCH <= 3) ? ((EARLY && PSC::enabled) ? GGNN_MERGE_WAVES_EARLY
                                                                                       : GGNN_MERGE_WAVES)
                                                            : 1)))
    merge_kernel(const MergeArgs a)
{
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  const WaveLds lds(lds_raw, kMergeCache);
  const int lane = threadIdx.x;
  const uint32_t un = xcd_contiguous_index(block_linear_index(), a.N_btm, a.xcd_map != 0);
  if (un >= a.N_btm)
    return;
  const int n = static_cast<int>(un);
  const BaseT* base = static_cast<const BaseT*>(a.base);
  const uint32_t K = a.KBuild;

  // merge_layer.cu:74-76 (xi from the MEAN nn1 distance, quirk Q4)
  const float nn1 = a.nn1_stats[0];
  const float xi = (MODE == kL2) ? (nn1 * nn1) * a.tau * a.tau : nn1 * a.tau;

  const int m = (!a.layer_btm) ? n : a.translation_all[a.STs_off[a.layer_btm] + un];

  // (early rows + pre-screen: the point's float row waits in LDS, see query.hip)
  using DE = DistEngine<BaseT, LPR, NCH, EARLY && PSC::enabled>;
  DE de;
  de.template load_query<MODE>(base, a.D, base + static_cast<size_t>(static_cast<uint32_t>(m)) * a.D,
                               lds_raw + wave_lds_ints(kMergeCache, HB));
  // exact pre-screen (traversal.hpp): the point is coded like any query, which reproduces its
  // stored codes and gives its own coding error
  PSC ps;
  if constexpr (PSC::enabled)
    ps.load(a.ps_codes, a.ps_params, a.ps_Dc,
            reinterpret_cast<const float*>(base + static_cast<size_t>(static_cast<uint32_t>(m)) * a.D),
            a.D);
  uint2 rows_read = make_uint2(0u, 0u);

  SortedList<R, HB> sl;
  sl.init(K + 1, a.sorted, kMergeCache, xi, lds.known, static_cast<int>(a.vis_slots));
  uint32_t cnt_dist = 0, cnt_pop = 0;

  {
    // get_top_seg_offset, merge_layer.cu:40-61
    uint32_t seg_btm = un / a.S;
    if (!a.layer_btm) {
      const uint32_t offset_points = a.S0_off * (a.S0 + 1);
      seg_btm = (un < offset_points) ? un / (a.S0 + 1) : a.S0_off + (un - offset_points) / a.S0;
    }
    uint32_t powG = a.G;
    for (uint32_t i = 1; i < a.layer_top - a.layer_btm; ++i)
      powG *= a.G;
    const uint32_t s_offset = (seg_btm / powG) * a.S;
    // fetch starting points, merge_layer.cu:86-94
    for (uint32_t i = 0; i < a.S; i += kKBlock) {
      const int cand = (lane < (int)kKBlock && i + lane < a.S)
                           ? static_cast<int>(s_offset + i + lane)
                           : kEmptyKey;
      cnt_dist += fetch<MODE, false>(sl, de, lds, cand, a.translation_all + a.STs_off[a.layer_top],
                                     ps, rows_read);
    }
  }

  // hierarchic kNN search, merge_layer.cu:97-119
  for (uint32_t layer = a.layer_top - 1; layer >= a.layer_btm && layer != 0xffffffffu; --layer) {
    sl.transform(a.selection_all + a.STs_off[layer + 1], lds.known,
                 reinterpret_cast<float*>(lds.known + a.sorted));
    const int32_t* tr = (!layer) ? nullptr : a.translation_all + a.STs_off[layer];
    if (layer == a.layer_btm) {
      const int cand = (lane == 0) ? n : kEmptyKey;
      cnt_dist += fetch<MODE, false>(sl, de, lds, cand, tr, ps, rows_read);
    }
    // graph row of the queue head loaded ahead (query.hip: same speculation, same hook)
    int spec_key = kEmptyKey, spec_row = kEmptyKey;
    const int32_t* layer_graph = a.graph_all + static_cast<size_t>(a.Ns_off[layer]) * K;
    for (uint32_t ite = 0; ite < kMergeIterations; ++ite) {
      if constexpr (EARLY) {
        const int anchor = sl.peek(sl.criteria());
        if (anchor == kEmptyKey)
          break;
        ++cnt_pop;
        const bool in_row = lane < static_cast<int>(K);  // K <= 24 (host)
        int cand;
        if (anchor == spec_key)
          cand = in_row ? spec_row : kEmptyKey;
        else
          cand = in_row ? layer_graph[static_cast<size_t>(static_cast<uint32_t>(anchor)) * K + lane]
                        : kEmptyKey;
        // unconditional load, masked at the use (query.hip: a load under a branch turns the wait
        // for the requested code rows into vmcnt(0))
        auto prefetch_head_row = [&]() {
          spec_key = sl.key_at(sl.BEST);
          spec_row = layer_graph[static_cast<size_t>(static_cast<uint32_t>(max(spec_key, 0))) * K +
                                 min(lane, static_cast<int>(K) - 1)];
        };
        if constexpr (PSC::enabled) {
          EarlyRows<PSC> er;
          er.issue(ps, cand, tr);
          sl.pop_commit(anchor, lds.known);
          cnt_dist += fetch_early<MODE, COUNT>(sl, de, lds, cand, er, ps, rows_read, prefetch_head_row, tr);
        }
        else {
          EarlyRows<DE> er;
          er.issue(de, cand, tr);
          sl.pop_commit(anchor, lds.known);
          cnt_dist += fetch_early<MODE, COUNT>(sl, de, lds, cand, er, ps, rows_read, prefetch_head_row, tr);
        }
        continue;
      }
      const int anchor = sl.pop(sl.criteria(), lds.known);
      if (anchor == kEmptyKey)
        break;
      ++cnt_pop;
      const int32_t* row = layer_graph + static_cast<size_t>(static_cast<uint32_t>(anchor)) * K;
      for (uint32_t j = 0; j < K; j += kKBlock) {
        const bool in_row = lane < (int)kKBlock && j + lane < K;
        int cand;
        if (j == 0 && anchor == spec_key)
          cand = spec_row;
        else
          cand = in_row ? row[j + lane] : kEmptyKey;
        auto prefetch_head_row = [&]() {
          if (j == 0) {
            spec_key = sl.key_at(sl.BEST);
            if (spec_key != kEmptyKey)
              spec_row = in_row ? layer_graph[static_cast<size_t>(static_cast<uint32_t>(spec_key)) *
                                                  K + lane]
                                : kEmptyKey;
          }
        };
        cnt_dist += fetch<MODE, true>(sl, de, lds, cand, tr, ps, rows_read, prefetch_head_row);
      }
    }
  }

  // own index among the first K entries (merge_layer.cu:121-134); keys of the best list are unique
  int own = -1;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = r * kWave + lane;
    const unsigned long long mm = __ballot(i < K && sl.key[r] == n);
    if (mm)
      own = r * kWave + __ffsll(static_cast<long long>(mm)) - 1;
  }
  // write K neighbours skipping self (Q3: own == -1 drops entry 0), merge_layer.cu:135-145
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = r * kWave + lane;
    int nk = lane_down1(sl.key[r]);
    if (r + 1 < R) {
      const int bk = rdlane(sl.key[r + 1], 0);
      if (lane == 63)
        nk = bk;
    }
    if (i < (int)K) {
      const int idx = (i >= own) ? nk : sl.key[r];
      a.graph_buffer[static_cast<size_t>(un) * K + i] = (idx != kEmptyKey) ? idx : n;
    }
  }
  // nn1 = first non-zero distance after self, merge_layer.cu:147-157
  if (!a.layer_btm) {
    int i = own + 1;
    float dist;
    do {
      dist = sl.dist_at(i);
      ++i;
    } while (dist == 0.0f && i < sl.BEST);
    if (MODE == kL2)
      dist = sqrtf(dist);
    if (lane == 0)
      a.nn1_dist_buffer[un] = dist;
  }
  if (lane == 0 && a.n_dist)
    a.n_dist[un] = cnt_dist;
  if (lane == 0 && a.n_work)
    a.n_work[un] = make_uint4(cnt_dist, rows_read.x, rows_read.y, cnt_pop);
}

template <typename BaseT, int LPR, int NCH, int MODE, class PSC>
static void launch_merge_r(const MergeArgs& args, hipStream_t stream)
{
  const size_t lds = wave_lds_bytes(kMergeCache);
  // visited ring of 192 entries mirrored in a hash set (traversal.hpp) where the registers allow
  // it at 7 waves per SIMD (see launch_query_r)
  constexpr bool early_layout = PSC::enabled ? (PsLayout<PSC>::lpr == 8 && PsLayout<PSC>::nch == 1)
                                             : (LPR == 8 && NCH == 1);
  if constexpr (early_layout) {
    // hook MERGE_EARLY = 0: the round-1..4 order (A/B and test hook)
    if (args.sorted <= 64 && args.KBuild <= 8 * kEarlySteps && hook(kHookMergeEarly) != 0) {
      constexpr size_t qrow = DistEngine<BaseT, LPR, NCH, PSC::enabled>::kQueryLdsBytes;
      if (args.n_dist || args.n_work || hook(kHookMergeCounting) != 0)
        hipLaunchKernelGGL((merge_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 1, true>),
                           grid_for(xcd_grid_blocks(args.N_btm, args.xcd_map != 0)), dim3(kWave),
                           wave_lds_bytes(kMergeCache, 1) + qrow, stream, args);
      else
        hipLaunchKernelGGL((merge_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 1, true, false>),
                           grid_for(xcd_grid_blocks(args.N_btm, args.xcd_map != 0)), dim3(kWave),
                           wave_lds_bytes(kMergeCache, 1) + qrow, stream, args);
      return;
    }
  }
  if (args.sorted <= 64 && (PSC::enabled || NCH == 1))
    hipLaunchKernelGGL((merge_kernel<BaseT, LPR, NCH, 1, MODE, PSC, 1>), grid_for(xcd_grid_blocks(args.N_btm, args.xcd_map != 0)), dim3(kWave),
                       wave_lds_bytes(kMergeCache, 1), stream, args);
  else if (args.sorted <= 64)
    hipLaunchKernelGGL((merge_kernel<BaseT, LPR, NCH, 1, MODE, PSC>), grid_for(xcd_grid_blocks(args.N_btm, args.xcd_map != 0)), dim3(kWave),
                       lds, stream, args);
  else if (args.sorted <= 128)
    hipLaunchKernelGGL((merge_kernel<BaseT, LPR, NCH, 2, MODE, PSC>), grid_for(xcd_grid_blocks(args.N_btm, args.xcd_map != 0)), dim3(kWave),
                       lds, stream, args);
  else if (args.sorted <= 256)
    hipLaunchKernelGGL((merge_kernel<BaseT, LPR, NCH, 4, MODE, PSC>), grid_for(xcd_grid_blocks(args.N_btm, args.xcd_map != 0)), dim3(kWave),
                       lds, stream, args);
  else
    throw Error(GGNN_UNSUPPORTED, "KBuild too large for the merge cache");
}

template <typename BaseT, int LPR, int NCH>
static void launch_merge_cfg(const MergeArgs& args, bool use_ps, ggnn_measure measure,
                             hipStream_t stream)
{
  if constexpr (std::is_same<BaseT, float>::value) {
    if (use_ps) {
      if (measure == GGNN_EUCLIDEAN)
        launch_merge_r<BaseT, LPR, NCH, kL2, typename PsFor<LPR, NCH, kL2>::type>(args, stream);
      else
        launch_merge_r<BaseT, LPR, NCH, kCos, typename PsFor<LPR, NCH, kCos>::type>(args, stream);
      return;
    }
  }
  if (measure == GGNN_EUCLIDEAN)
    launch_merge_r<BaseT, LPR, NCH, kL2, NoPrescreen>(args, stream);
  else
    launch_merge_r<BaseT, LPR, NCH, kCos, NoPrescreen>(args, stream);
}

void launch_merge(const MergeLaunch& a, hipStream_t stream)
{
  const ggnn_graph_config& c = a.cfg;
  GGNN_REQUIRE(a.layer_top > a.layer_btm && a.layer_top < kLayers, GGNN_INVALID_ARGUMENT,
               "merge needs layer_top > layer_btm");
  check_vector_layout(a.base, c.D, a.dtype);
  MergeArgs args{};
  args.base = a.base;
  args.graph_all = a.graph_all;
  args.translation_all = a.translation_all;
  args.selection_all = a.selection_all;
  args.nn1_stats = a.nn1_stats;
  args.graph_buffer = a.graph_buffer;
  args.nn1_dist_buffer = a.nn1_dist_buffer;
  args.n_dist = a.n_dist;
  args.n_work = reinterpret_cast<uint4*>(a.n_work);
  args.D = c.D;
  args.KBuild = c.KBuild;
  args.S = c.S;
  args.G = c.G;
  args.S0 = c.S0;
  args.S0_off = c.S0_off;
  args.layer_top = a.layer_top;
  args.layer_btm = a.layer_btm;
  args.sorted = merge_sorted_size(c.KBuild);
  args.N_btm = c.Ns[a.layer_btm];
  for (uint32_t l = 0; l < kLayers; ++l) {
    args.Ns_off[l] = c.Ns_offsets[l];
    args.STs_off[l] = c.STs_offsets[l];
  }
  args.tau = a.tau_build;
  args.vis_slots = vis_slots_hook();
  args.xcd_map = (hook(kHookXcdMap) & 1) != 0;
  GGNN_REQUIRE(args.sorted < kMergeCache, GGNN_UNSUPPORTED, "KBuild too large for the merge cache");
  if (!args.N_btm)
    return;

  const bool use_ps = a.ps_codes && a.ps_params && a.dtype == GGNN_F32;
  if (use_ps) {
    GGNN_REQUIRE(a.ps_Dc == prescreen_code_dim(c.D), GGNN_INVALID_ARGUMENT,
                 "pre-screen code rows must be D rounded up to 16");
    args.ps_codes = a.ps_codes;
    args.ps_params = a.ps_params;
    args.ps_Dc = a.ps_Dc;
  }
#define GGNN_LAUNCH_MERGE(T, LPR, NCH) launch_merge_cfg<T, LPR, NCH>(args, use_ps, a.measure, stream)
  GGNN_DISPATCH_DIST(a.dtype, c.D, GGNN_LAUNCH_MERGE);
#undef GGNN_LAUNCH_MERGE
  GGNN_HIP_CHECK(hipGetLastError());
}

}  // namespace ggnn_amd
