// sym kernel: for every local neighbour s of point n, a constrained (<= 20 pops) search from s
// back to n; if n is unreachable, request an inverse link at a near point through an atomic slot
// counter.  One wave64 per point.
// Reference: SymQueryKernel::operator(), src/ggnn/construction/sym_query_layer.cu:39-145;
// SimpleKNNSymCache, include/ggnn/cuda_utils/simple_knn_sym_cache.cuh:34-488.
#include "traversal.hpp"

namespace ggnn_amd {

struct SymArgs {
  const void* base;
  const int32_t* graph;        // layer view [N_layer x K]
  const int32_t* translation;  // layer view or null
  const float* nn1_stats;
  int32_t* sym_buffer;   // [N_layer x KF]
  uint32_t* sym_atomic;  // [N_layer]
  uint4* n_work;  // optional [N_layer]: distance evaluations, float rows, code rows, pops per point
  uint32_t D, KBuild, N_layer, sorted, first_n, count;
  float tau;
  // optional pre-screen copy of the base coded for this measure (prescreen.hip); float32 only
  const uint8_t* ps_codes;
  const float* ps_params;
  uint32_t ps_Dc;
  uint32_t xcd_map;  // XCD-aware block -> point mapping (common.hpp)
};

constexpr uint32_t kSymCache = 128;          // sym_query_layer.cuh:43
constexpr uint32_t kSymPathIterations = 20;  // sym_query_layer.cuh:42

uint32_t sym_sorted_size(uint32_t KBuild)
{
  // sym_query_layer.cuh:63-64
  return std::max(64u, next_multiple32(KBuild / 2 + 16));
}

// query + "half" point distance engine (simple_knn_sym_cache.cuh:143-283)
template <typename BaseT, int LPR, int NCH>
struct SymEngine : DistEngine<BaseT, LPR, NCH> {
  using DEB = DistEngine<BaseT, LPR, NCH>;
  using Chunk = typename DEB::Chunk;
  static constexpr int EPC = DEB::EPC;
  float half[NCH][EPC];
  float half_norm;

  // init_start_point, simple_knn_sym_cache.cuh:159-189: half = q + (0.5-EPS)*(start - q);
  // returns the distances of the start point itself
  template <int MODE>
  GGNN_DEV void set_half(int other_m, float& dq, float& dh)
  {
    const BaseT* row = this->row_ptr(other_m);
    Chunk v[NCH];
    const float w = 0.5f - 0.1f;
    float qn = 0.f, hn = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      v[c] = ChunkOf<BaseT>::zero();
      if (this->chunk_valid(c))
        v[c] = this->load_chunk(row, c);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float qq = ChunkOf<BaseT>::get(this->q[c], e);
        const float o = ChunkOf<BaseT>::get(v[c], e);
        half[c][e] = fmaf(w, o - qq, qq);
        if (MODE == kCos) {
          qn = fmaf(qq, qq, qn);
          hn = fmaf(half[c][e], half[c][e], hn);
        }
      }
    }
    if (MODE == kCos) {
      this->q_norm = group_sum<LPR>(qn);
      half_norm = group_sum<LPR>(hn);
    }
    float a, b, nn;
    partial2<MODE>(v, a, b, nn);
    finish<MODE>(a, b, nn, dq, dh);
  }

  template <int MODE>
  GGNN_DEV void partial2(const Chunk (&v)[NCH], float& a, float& b, float& nn) const
  {
    a = 0.f;
    b = 0.f;
    nn = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float o = ChunkOf<BaseT>::get(v[c], e);
        const float qq = ChunkOf<BaseT>::get(this->q[c], e);
        if (MODE == kL2) {
          const float dq = qq - o;
          a = fmaf(dq, dq, a);
          const float dh = half[c][e] - o;
          b = fmaf(dh, dh, b);
        }
        else {
          a = fmaf(qq, o, a);
          b = fmaf(half[c][e], o, b);
          nn = fmaf(o, o, nn);
        }
      }
    }
  }
  // group reduction + normalisation (simple_knn_sym_cache.cuh:253-273)
  template <int MODE>
  GGNN_DEV void finish(float a, float b, float nn, float& dq, float& dh) const
  {
    a = group_sum<LPR>(a);
    b = group_sum<LPR>(b);
    if (MODE == kCos) {
      nn = group_sum<LPR>(nn);
      const float qn = nn * this->q_norm;
      const float hn = nn * half_norm;
      a = (qn > 0.0f) ? fabsf(1.0f - a / sqrtf(qn)) : 1.0f;
      b = (hn > 0.0f) ? fabsf(1.0f - b / sqrtf(hn)) : 1.0f;
    }
    dq = a;
    dh = b;
  }
};

// fetch of the sym cache, simple_knn_sym_cache.cuh:405-436
template <int MODE, int R, class SE>
GGNN_DEV void sym_distances(SortedList<R>& sl, const SE& se, const WaveLds& lds, const int nsurv,
                            const int32_t* translation, float criteria_half);

// ps: exact pre-screen on the distance to the query point (traversal.hpp): a candidate whose lower
// bound already reaches criteria_sym() = s_dists[0] + xi at the start of the fetch fails the
// first half of the acceptance test (simple_knn_sym_cache.cuh:431) whatever its distance to the
// half point is -- the criteria only tightens during a fetch -- so its float row is not read.
template <int MODE, int R, class SE, class PS>
GGNN_DEV void sym_fetch(SortedList<R>& sl, const SE& se, const WaveLds& lds, int cand,
                        const int32_t* translation, float criteria_half, const PS& ps, uint4& work)
{
  const int lane = threadIdx.x;
  cand = lower_half_to_both(cand);
  cand = sl.filter(cand, lds.known);
  const unsigned long long surv = __ballot(lane < 32 && cand != kEmptyKey);
  const int nsurv = __popcll(surv);
  if (nsurv == 0)
    return;
  __syncthreads();
  if (lane < 32 && cand != kEmptyKey)
    lds.ckeys[__popcll(surv & ((1ull << lane) - 1ull))] = cand;
  __syncthreads();
  int nsurv_eval = nsurv;
  work.x += nsurv;
  if constexpr (PS::enabled) {
    const float s_thr = ps.threshold(sl.dist_at(0) + sl.xi);
    if (s_thr < inf_f()) {
      nsurv_eval = prescreen_pass(ps, lds, nsurv, s_thr, translation);
      work.z += nsurv;
      if (nsurv_eval == 0)
        return;
      __syncthreads();
    }
  }
  work.y += nsurv_eval;
  sym_distances<MODE, R>(sl, se, lds, nsurv_eval, translation, criteria_half);
}

template <int MODE, int R, class SE>
GGNN_DEV void sym_distances(SortedList<R>& sl, const SE& se, const WaveLds& lds, const int nsurv,
                            const int32_t* translation, float criteria_half)
{
  constexpr int STEPS = StepsOf<SE::LPR, SE::NCH>::value;
  constexpr int ROWS = SE::ROWS;
  using Chunk = typename SE::Chunk;
  const int lane = threadIdx.x;
  const int grp = lane / SE::LPR;
  for (int s0 = 0; s0 < nsurv; s0 += ROWS * STEPS) {
    Chunk v[STEPS][SE::NCH];
    int rr[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int r = s0 + s * ROWS + grp;
      const bool valid = r < nsurv;
      rr[s] = valid ? r : -1;
      // slots past the end read the row of the first candidate (cached, result never stored)
      int m = lds.ckeys[valid ? r : s0];
      if (translation)
        m = translation[m];
      const auto* row = se.row_ptr(m);
#pragma unroll
      for (int c = 0; c < SE::NCH; ++c) {
        if (se.all_chunks || se.chunk_valid(c))
          v[s][c] = se.load_chunk(row, c);
        else
          v[s][c] = ChunkOf<typename SE::Base>::zero();
      }
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s0 + s * ROWS >= nsurv)
        break;
      float a, b, nn, dq, dh;
      se.template partial2<MODE>(v[s], a, b, nn);
      se.template finish<MODE>(a, b, nn, dq, dh);
      if (rr[s] >= 0 && se.g == 0) {
        lds.cd0[rr[s]] = dq;
        lds.cd1[rr[s]] = dh;
      }
    }
  }
  __syncthreads();
  const float cd = lane < nsurv ? lds.cd0[lane] : inf_f();
  const float ch = lane < nsurv ? lds.cd1[lane] : inf_f();
  const int ck = lane < nsurv ? lds.ckeys[lane] : kEmptyKey;
  // criteria_sym() = s_dists[0] + xi never increases during a fetch
  unsigned long long m = __ballot(cd < sl.dist_at(0) + sl.xi && ch < criteria_half);
  while (m) {
    const int j = __ffsll(static_cast<long long>(m)) - 1;
    m &= m - 1;
    const float d = rdlanef(cd, j);
    const int k = rdlane(ck, j);
    if (d < sl.dist_at(0) + sl.xi)
      sl.push(k, d);
  }
}

template <typename BaseT, int LPR, int NCH, int R, int MODE, class PSC>
// one-chunk layouts fit 7 waves per SIMD without spilling (79 -> 71 registers for uint8 rows)
__global__ void __launch_bounds__(kWave)
    __attribute__((amdgpu_waves_per_eu((R == 1 && NCH == 1) ? 7 : 1))) sym_kernel(const SymArgs a)
{
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  const WaveLds lds(lds_raw, kSymCache);
  const int lane = threadIdx.x;
  const uint32_t bi = xcd_contiguous_index(block_linear_index(), a.count, a.xcd_map != 0);
  if (bi >= a.count)
    return;
  const uint32_t un = a.first_n + bi;
  if (un >= a.N_layer)
    return;
  const int n = static_cast<int>(un);
  const BaseT* base = static_cast<const BaseT*>(a.base);
  const uint32_t K = a.KBuild;
  const uint32_t KF = K / 2;
  const uint32_t KL = K - KF;

  const float nn1 = a.nn1_stats[0];
  const float xi = (MODE == kL2) ? (nn1 * nn1) * a.tau * a.tau : nn1 * a.tau;

  const int m = a.translation ? a.translation[un] : n;
  SymEngine<BaseT, LPR, NCH> se;
  se.template load_query<MODE>(base, a.D, base + static_cast<size_t>(static_cast<uint32_t>(m)) * a.D);
  // the point is coded like a query (as in the merge kernel)
  PSC ps;
  if constexpr (PSC::enabled)
    ps.load(a.ps_codes, a.ps_params, a.ps_Dc,
            reinterpret_cast<const float*>(base + static_cast<size_t>(static_cast<uint32_t>(m)) * a.D),
            a.D);

  SortedList<R> sl;
  sl.init(KF, a.sorted, kSymCache, xi, lds.known);
  uint4 work = make_uint4(0u, 0u, 0u, 0u);

  for (uint32_t i = 0; i < KL; i += kKBlock) {
    // s_sym_ids, sym_query_layer.cu:67-75
    const int my_sym = (lane < (int)kKBlock && i + lane < KL)
                           ? a.graph[static_cast<size_t>(un) * K + i + lane]
                           : kEmptyKey;
    for (uint32_t k = 0; i + k < KL && k < kKBlock; ++k) {
      const int other_n = rdlane(my_sym, k);
      // init_start_point, simple_knn_sym_cache.cuh:159-201
      const int other_m = a.translation ? a.translation[other_n] : other_n;
      float dq, dh;
      se.template set_half<MODE>(other_m, dq, dh);
      ++work.x;  // the start point's own row
      ++work.y;
      const float criteria_half = dh + xi;
      sl.reset(lds.known);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int li = r * kWave + lane;
        if (li == 0 || li == sl.BEST) {
          sl.key[r] = other_n;
          sl.dist[r] = dq;
        }
      }

      bool found = false;
      for (uint32_t ite = 0; ite < kSymPathIterations && !found; ++ite) {
        const int anchor = sl.pop(sl.dist_at(0) + sl.xi, lds.known);
        if (anchor == kEmptyKey)
          break;
        ++work.w;
        // neighbours at the anchor + its pending inverse links, sym_query_layer.cu:96-119
        for (uint32_t i2 = 0; i2 < K; i2 += kKBlock) {
          const uint32_t k2 = i2 + lane;
          int other_id = kEmptyKey;
          if (lane < (int)kKBlock && k2 < K) {
            other_id = (k2 < KL)
                           ? a.graph[static_cast<size_t>(static_cast<uint32_t>(anchor)) * K + k2]
                           : a.sym_buffer[static_cast<size_t>(static_cast<uint32_t>(anchor)) * KF +
                                          (k2 - KL)];
          }
          if (__any(lane < (int)kKBlock && k2 < K && other_id == n)) {
            found = true;
            break;
          }
          sym_fetch<MODE>(sl, se, lds, other_id, a.translation, criteria_half, ps, work);
        }
      }

      if (!found) {
        // request an inverse link at the nearest point with a free slot, sym_query_layer.cu:121-141
        for (uint32_t i3 = 0; i3 < KF; ++i3) {
          const int other = sl.key_at(i3);
          if (other == kEmptyKey)
            break;
          uint32_t pos = 0;
          if (lane == 0)
            pos = atomicAdd(&a.sym_atomic[other], 1u);
          pos = static_cast<uint32_t>(uni(static_cast<int>(pos)));
          if (pos < KF) {
            if (lane == 0)
              a.sym_buffer[static_cast<size_t>(static_cast<uint32_t>(other)) * KF + pos] = n;
            break;
          }
        }
      }
    }
  }
  if (lane == 0 && a.n_work)
    a.n_work[un] = work;
}

template <typename BaseT, int LPR, int NCH, int MODE, class PSC>
static void launch_sym_r(const SymArgs& args, hipStream_t stream)
{
  const size_t lds = wave_lds_bytes(kSymCache);
  if (args.sorted <= 64)
    hipLaunchKernelGGL((sym_kernel<BaseT, LPR, NCH, 1, MODE, PSC>), grid_for(xcd_grid_blocks(args.count, args.xcd_map != 0)), dim3(kWave),
                       lds, stream, args);
  else if (args.sorted <= 128)
    hipLaunchKernelGGL((sym_kernel<BaseT, LPR, NCH, 2, MODE, PSC>), grid_for(xcd_grid_blocks(args.count, args.xcd_map != 0)), dim3(kWave),
                       lds, stream, args);
  else
    throw Error(GGNN_UNSUPPORTED, "KBuild too large for the sym cache");
}

template <typename BaseT, int LPR, int NCH>
static void launch_sym_cfg(const SymArgs& args, bool use_ps, ggnn_measure measure,
                           hipStream_t stream)
{
  if constexpr (std::is_same<BaseT, float>::value) {
    if (use_ps) {
      if (measure == GGNN_EUCLIDEAN)
        launch_sym_r<BaseT, LPR, NCH, kL2, typename PsFor<LPR, NCH, kL2>::type>(args, stream);
      else
        launch_sym_r<BaseT, LPR, NCH, kCos, typename PsFor<LPR, NCH, kCos>::type>(args, stream);
      return;
    }
  }
  if (measure == GGNN_EUCLIDEAN)
    launch_sym_r<BaseT, LPR, NCH, kL2, NoPrescreen>(args, stream);
  else
    launch_sym_r<BaseT, LPR, NCH, kCos, NoPrescreen>(args, stream);
}

void launch_sym(const SymLaunch& a, hipStream_t stream)
{
  check_vector_layout(a.base, a.D, a.dtype);
  SymArgs args{};
  args.base = a.base;
  args.graph = a.graph_layer;
  args.translation = a.translation;
  args.nn1_stats = a.nn1_stats;
  args.sym_buffer = a.sym_buffer;
  args.sym_atomic = a.sym_atomic;
  args.n_work = reinterpret_cast<uint4*>(a.n_work);
  args.D = a.D;
  args.KBuild = a.KBuild;
  args.N_layer = a.N_layer;
  args.sorted = sym_sorted_size(a.KBuild);
  args.first_n = a.first_n;
  args.count = std::min(a.count, a.N_layer > a.first_n ? a.N_layer - a.first_n : 0u);
  args.tau = a.tau_build;
  args.xcd_map = (hook(kHookXcdMap) & 2) != 0;
  GGNN_REQUIRE(args.sorted < kSymCache, GGNN_UNSUPPORTED, "KBuild too large for the sym cache");
  if (!args.count)
    return;

  const bool use_ps = a.ps_codes && a.ps_params && a.dtype == GGNN_F32;
  if (use_ps) {
    GGNN_REQUIRE(a.ps_Dc == prescreen_code_dim(a.D), GGNN_INVALID_ARGUMENT,
                 "pre-screen code rows must be D rounded up to 16");
    args.ps_codes = a.ps_codes;
    args.ps_params = a.ps_params;
    args.ps_Dc = a.ps_Dc;
  }
#define GGNN_LAUNCH_SYM(T, LPR, NCH) launch_sym_cfg<T, LPR, NCH>(args, use_ps, a.measure, stream)
  GGNN_DISPATCH_DIST(a.dtype, a.D, GGNN_LAUNCH_SYM);
#undef GGNN_LAUNCH_SYM
  GGNN_HIP_CHECK(hipGetLastError());
}

}  // namespace ggnn_amd
