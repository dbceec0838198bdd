// bf_query: exhaustive scan, the exact ground-truth path.
// Reference: BruteForceQueryKernel::operator(), src/ggnn/query/bf_query_layer.cu:39-65 (one
// block per query, N sequential block-reductions) and KBestList (k_best_list.cuh:29-142).
//
// This kernel keeps the reference's arithmetic (direct difference form, Q2 tie rule: equal
// distances keep the lower base index first) but restructures the scan for wave64: one wave
// per (query, base slice), 64/LPR base rows per coalesced 16 B/lane load instruction, K-best
// list in registers.  Slices are merged by a second tiny kernel.  (An MFMA Q x B^T tile path
// for large query batches is planned on top of this parity anchor, see DESIGN.md.)
#include <cstdlib>

#include "hooks.hpp"
#include "traversal.hpp"

namespace ggnn_amd {

struct BfArgs {
  const void* base;
  const void* query;
  int32_t* ids;    // [slices x Nq x K] partial results (or final when slices == 1)
  float* dists;
  uint32_t D, Nq, N_base, K, slices, rows_per_slice;
  // optional subset: only the queries qlist[0, *qcount) are scanned (device memory; the grid is
  // sized for all Nq queries and the surplus blocks leave at once)
  const uint32_t* qlist;
  const uint32_t* qcount;
};

template <typename BaseT, int LPR, int NCH, int R, int MODE>
__global__ void __launch_bounds__(kWave) bf_query_kernel(const BfArgs a)
{
  constexpr int ROWS = kWave / LPR;
  constexpr int STEPS = StepsOf<LPR, NCH>::value;
  using DE = DistEngine<BaseT, LPR, NCH>;
  using Chunk = typename DE::Chunk;
  __shared__ float s_d[ROWS * STEPS];

  const int lane = threadIdx.x;
  const uint32_t bid = block_linear_index();
  if (bid >= a.Nq * a.slices)
    return;
  uint32_t n = bid / a.slices;
  const uint32_t slice = bid % a.slices;
  if (a.qlist) {
    if (n >= *a.qcount)
      return;
    n = a.qlist[n];
  }
  const BaseT* base = static_cast<const BaseT*>(a.base);
  const BaseT* query = static_cast<const BaseT*>(a.query);

  DE de;
  de.template load_query<MODE>(base, a.D, query + static_cast<size_t>(n) * a.D);
  const int grp = lane / LPR;

  SortedList<R> best;  // only key/dist/BEST are used
  best.BEST = a.K;
  best.SORTED = a.K;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    best.key[r] = kEmptyKey;
    best.dist[r] = inf_f();
  }

  const uint32_t begin = slice * a.rows_per_slice;
  const uint32_t end = min(a.N_base, begin + a.rows_per_slice);
  for (uint32_t i0 = begin; i0 < end; i0 += ROWS * STEPS) {
    Chunk v[STEPS][NCH];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const uint32_t row = i0 + s * ROWS + grp;
      const bool valid = row < end;
      const BaseT* rp = de.row_ptr(valid ? row : begin);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        v[s][c] = ChunkOf<BaseT>::zero();
        if (valid && de.chunk_valid(c))
          v[s][c] = de.load_chunk(rp, c);
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      float x, y;
      de.template partial<MODE>(v[s], x, y);
      x = group_sum<LPR>(x);
      if (MODE == kCos)
        y = group_sum<LPR>(y);
      if (de.g == 0)
        s_d[s * ROWS + grp] = (MODE == kCos) ? de.finish_cos(x, y) : x;
    }
    __syncthreads();
    // visit the batch in base order (bf_query_layer.cu:52-57)
    const uint32_t cnt = min((uint32_t)(ROWS * STEPS), end - i0);
    const float cd = lane < (int)cnt ? s_d[lane] : inf_f();
    unsigned long long m = __ballot(cd < best.dist_at(a.K - 1));
    while (m) {
      const int j = __ffsll(static_cast<long long>(m)) - 1;
      m &= m - 1;
      const float d = rdlanef(cd, j);
      if (d < best.dist_at(a.K - 1))
        best.push_best_stable(static_cast<int>(i0 + j), d);
    }
  }

  const size_t out = (static_cast<size_t>(slice) * a.Nq + n) * a.K;  // [slices, Nq, K]
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = r * kWave + lane;
    if (i < a.K) {
      a.ids[out + i] = best.key[r];
      a.dists[out + i] = best.dist[r];
    }
  }
}

// k > 256: the K-best list lives in LDS (dists [K] | ids [K]); stable insertion by a wave-wide
// shift, rare after the first few thousand rows.
template <typename BaseT, int LPR, int NCH, int MODE>
__global__ void __launch_bounds__(kWave) bf_query_lds_kernel(const BfArgs a)
{
  constexpr int ROWS = kWave / LPR;
  constexpr int STEPS = StepsOf<LPR, NCH>::value;
  using DE = DistEngine<BaseT, LPR, NCH>;
  using Chunk = typename DE::Chunk;
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  float* best_d = reinterpret_cast<float*>(lds_raw);
  int* best_i = lds_raw + a.K;
  float* s_d = reinterpret_cast<float*>(lds_raw + 2 * a.K);

  const int lane = threadIdx.x;
  const uint32_t bid = block_linear_index();
  if (bid >= a.Nq * a.slices)
    return;
  uint32_t n = bid / a.slices;
  const uint32_t slice = bid % a.slices;
  if (a.qlist) {
    if (n >= *a.qcount)
      return;
    n = a.qlist[n];
  }
  const BaseT* base = static_cast<const BaseT*>(a.base);
  DE de;
  de.template load_query<MODE>(base, a.D, static_cast<const BaseT*>(a.query) + static_cast<size_t>(n) * a.D);
  const int grp = lane / LPR;
  for (uint32_t i = lane; i < a.K; i += kWave) {
    best_d[i] = inf_f();
    best_i[i] = kEmptyKey;
  }
  __syncthreads();

  const uint32_t begin = slice * a.rows_per_slice;
  const uint32_t end = min(a.N_base, begin + a.rows_per_slice);
  for (uint32_t i0 = begin; i0 < end; i0 += ROWS * STEPS) {
    Chunk v[STEPS][NCH];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const uint32_t row = i0 + s * ROWS + grp;
      const bool valid = row < end;
      const BaseT* rp = de.row_ptr(valid ? row : begin);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        v[s][c] = ChunkOf<BaseT>::zero();
        if (valid && de.chunk_valid(c))
          v[s][c] = de.load_chunk(rp, c);
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      float x, y;
      de.template partial<MODE>(v[s], x, y);
      x = group_sum<LPR>(x);
      if (MODE == kCos)
        y = group_sum<LPR>(y);
      if (de.g == 0)
        s_d[s * ROWS + grp] = (MODE == kCos) ? de.finish_cos(x, y) : x;
    }
    __syncthreads();
    const uint32_t cnt = min((uint32_t)(ROWS * STEPS), end - i0);
    const float cd = lane < (int)cnt ? s_d[lane] : inf_f();
    unsigned long long m = __ballot(cd < best_d[a.K - 1]);
    while (m) {
      const int j = __ffsll(static_cast<long long>(m)) - 1;
      m &= m - 1;
      const float d = rdlanef(cd, j);
      if (!(d < best_d[a.K - 1]))
        continue;
      // stable insert (k_best_list.cuh:77-109): chunks from the right so that every entry is
      // read before it is overwritten
      const int id = static_cast<int>(i0 + j);
      for (int c0 = (static_cast<int>(a.K) - 1) / kWave * kWave; c0 >= 0; c0 -= kWave) {
        const int k = c0 + lane;
        const bool own = k < static_cast<int>(a.K);
        const float cur = own ? best_d[k] : inf_f();
        const float prev = (own && k > 0) ? best_d[k - 1] : -inf_f();
        const int previ = (own && k > 0) ? best_i[k - 1] : kEmptyKey;
        __syncthreads();
        if (own && d < cur) {
          const bool first = !(d < prev);
          best_d[k] = first ? d : prev;
          best_i[k] = first ? id : previ;
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  const size_t out = (static_cast<size_t>(slice) * a.Nq + n) * a.K;
  for (uint32_t i = lane; i < a.K; i += kWave) {
    a.ids[out + i] = best_i[i];
    a.dists[out + i] = best_d[i];
  }
}

template <typename BaseT, int LPR, int NCH, int MODE>
static void launch_bf_r(const BfArgs& args, hipStream_t stream)
{
  const dim3 grid = grid_for(static_cast<uint64_t>(args.Nq) * args.slices);
  if (args.K <= 64)
    hipLaunchKernelGGL((bf_query_kernel<BaseT, LPR, NCH, 1, MODE>), grid, dim3(kWave), 0, stream,
                       args);
  else if (args.K <= 128)
    hipLaunchKernelGGL((bf_query_kernel<BaseT, LPR, NCH, 2, MODE>), grid, dim3(kWave), 0, stream,
                       args);
  else if (args.K <= 256)
    hipLaunchKernelGGL((bf_query_kernel<BaseT, LPR, NCH, 4, MODE>), grid, dim3(kWave), 0, stream,
                       args);
  else
    hipLaunchKernelGGL((bf_query_lds_kernel<BaseT, LPR, NCH, MODE>), grid, dim3(kWave),
                       (2 * args.K + 64) * sizeof(int), stream, args);
}

bool bf_mfma_supported(const BfLaunch& a);
void launch_bf_query_mfma(const BfLaunch& a, hipStream_t stream);

// scan of all queries (qlist == nullptr) or of the subset qlist[0, *qcount); slices > 1 needs
// tmp_ids / tmp_dists of [slices x Nq x K] entries
static void launch_bf_scan(const BfLaunch& a, uint32_t slices, uint32_t rows_per_slice,
                           const uint32_t* qlist, const uint32_t* qcount, int32_t* tmp_ids,
                           float* tmp_dists, hipStream_t stream)
{
  BfArgs args{};
  args.base = a.base;
  args.query = a.query;
  args.D = a.D;
  args.Nq = a.Nq;
  args.N_base = a.N_base;
  args.K = a.k_query;
  args.slices = slices;
  args.rows_per_slice = rows_per_slice;
  args.qlist = qlist;
  args.qcount = qcount;
  args.ids = slices > 1 ? tmp_ids : a.ids;
  args.dists = slices > 1 ? tmp_dists : a.dists;

#define GGNN_LAUNCH_BF(T, LPR, NCH)                         \
  do {                                                      \
    if (a.measure == GGNN_EUCLIDEAN)                        \
      launch_bf_r<T, LPR, NCH, kL2>(args, stream);          \
    else                                                    \
      launch_bf_r<T, LPR, NCH, kCos>(args, stream);         \
  } while (0)
  GGNN_DISPATCH_DIST(a.dtype, a.D, GGNN_LAUNCH_BF);
#undef GGNN_LAUNCH_BF
  GGNN_HIP_CHECK(hipGetLastError());
  if (slices > 1)
    // slices are in ascending base order, so "lower part first" on ties keeps Q2
    launch_merge_results_subset(a.Nq, a.k_query, slices, a.k_query, 0, tmp_ids, tmp_dists, a.ids,
                                a.dists, qlist, qcount, stream);
}

static void slice_rows(uint32_t N_base, uint32_t& slices, uint32_t& rows_per_slice)
{
  const uint32_t row_quant = 64;  // multiple of ROWS*STEPS for every configuration
  slices = std::max(1u, std::min(slices, 64u));
  rows_per_slice = (N_base + slices - 1) / slices;
  rows_per_slice = (rows_per_slice + row_quant - 1) / row_quant * row_quant;
  slices = std::max(1u, (N_base + rows_per_slice - 1) / rows_per_slice);
}

// exact answers for the queries the MFMA path could not certify (bf_mfma.hip): the scan kernel
// over qlist[0, *qcount), results written to the rows of those queries in a.ids / a.dists
size_t bf_rescan_tmp_entries(const BfLaunch& a, uint32_t* slices_out)
{
  // few queries are expected: split the base so that even a handful of them keep the chip busy,
  // within a bounded scratch size (slices x Nq x K entries)
  uint32_t slices = std::min(32u, std::max(1u, 32768u / std::max(1u, a.Nq)));
  slices = std::min(slices, std::max(1u, a.N_base / 4096u));
  uint32_t rows = 0;
  slice_rows(a.N_base, slices, rows);
  *slices_out = slices;
  return slices > 1 ? static_cast<size_t>(slices) * a.Nq * a.k_query : 0;
}
void launch_bf_rescan(const BfLaunch& a, const uint32_t* qlist, const uint32_t* qcount,
                      int32_t* tmp_ids, float* tmp_dists, hipStream_t stream)
{
  uint32_t slices = 0, rows = 0;
  (void)bf_rescan_tmp_entries(a, &slices);
  slice_rows(a.N_base, slices, rows);
  launch_bf_scan(a, slices, rows, qlist, qcount, tmp_ids, tmp_dists, stream);
}

void launch_bf_query(const BfLaunch& a, hipStream_t stream)
{
  if (a.n_rescanned)
    GGNN_HIP_CHECK(hipMemsetAsync(a.n_rescanned, 0, sizeof(uint32_t), stream));
  if (a.Nq == 0)
    return;
  // large batches: Q x B^T on the matrix cores (bf_mfma.hip); hook BF_SCAN = 1 forces the scan
  const bool force_scan = hook(kHookBfScan) == 1;
  if (!force_scan && bf_mfma_supported(a)) {
    launch_bf_query_mfma(a, stream);
    return;
  }
  check_vector_layout(a.base, a.D, a.dtype);
  check_vector_layout(a.query, a.D, a.dtype);
  GGNN_REQUIRE(a.k_query >= 1 && a.k_query <= 6000, GGNN_INVALID_ARGUMENT,
               "KQuery must be in [1, 6000]");

  // enough waves to fill the chip: split the base into slices when there are few queries
  uint32_t slices = 1;
  const uint32_t target_waves = 256 * 16;
  if (a.Nq < target_waves)
    slices = std::min((target_waves + a.Nq - 1) / a.Nq, std::max(1u, a.N_base / 4096u));
  uint32_t rows_per_slice = 0;
  slice_rows(a.N_base, slices, rows_per_slice);

  int32_t* tmp_ids = nullptr;
  float* tmp_dists = nullptr;
  if (slices > 1) {
    const size_t n = static_cast<size_t>(a.Nq) * slices * a.k_query;
    GGNN_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&tmp_ids), n * sizeof(int32_t), stream));
    GGNN_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&tmp_dists), n * sizeof(float), stream));
  }
  launch_bf_scan(a, slices, rows_per_slice, nullptr, nullptr, tmp_ids, tmp_dists, stream);
  if (slices > 1) {
    GGNN_HIP_CHECK(hipFreeAsync(tmp_ids, stream));
    GGNN_HIP_CHECK(hipFreeAsync(tmp_dists, stream));
  }
}

}  // namespace ggnn_amd
