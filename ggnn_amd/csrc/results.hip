// Per-query post-processing of shard results on the device.
//  * sort_shard_results: GPUInstance::sortQueryResults, src/ggnn/base/gpu_instance.cu:745-790
//    (cub::DeviceSegmentedRadixSort of [Nq] segments of shards*K pairs) restated as a stable
//    rank sort -- segments are tiny (shards*K), one wave per query.
//  * merge_results: ResultMerger::merge, src/ggnn/base/result_merger.cpp:51-149 (CPU heap merge
//    after D2H) done on the device straight from the all-gather buffer.
#include "common.hpp"

namespace ggnn_amd {

// order of floats as a radix sort sees them (-0.0 folded onto +0.0 as in CUB 2.x)
__device__ __forceinline__ uint32_t radix_key(float f)
{
  if (f == 0.0f)
    f = 0.0f;
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// one wave per row; rank_i = #{j : key_j < key_i or (key_j == key_i and j < i)}
__global__ void __launch_bounds__(kWave) sort_rows_kernel(uint32_t Nq, uint32_t row_len,
                                                          int32_t* ids, float* dists)
{
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  uint32_t* s_key = reinterpret_cast<uint32_t*>(lds_raw);
  int32_t* s_id = lds_raw + row_len;
  float* s_dist = reinterpret_cast<float*>(lds_raw + 2 * row_len);
  const uint32_t n = block_linear_index();
  if (n >= Nq)
    return;
  int32_t* ri = ids + static_cast<size_t>(n) * row_len;
  float* rd = dists + static_cast<size_t>(n) * row_len;
  for (uint32_t i = threadIdx.x; i < row_len; i += kWave) {
    const float d = rd[i];
    s_key[i] = radix_key(d);
    s_id[i] = ri[i];
    s_dist[i] = d;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < row_len; i += kWave) {
    const uint32_t k = s_key[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < row_len; ++j) {
      const uint32_t kj = s_key[j];
      rank += (kj < k) || (kj == k && j < i);
    }
    ri[rank] = s_id[i];
    rd[rank] = s_dist[i];
  }
}

void launch_sort_shard_results(uint32_t Nq, uint32_t row_len, int32_t* ids, float* dists,
                               hipStream_t stream)
{
  if (!Nq || row_len <= 1)
    return;
  GGNN_REQUIRE(row_len <= 12000, GGNN_UNSUPPORTED, "result rows longer than 12000 entries");
  const size_t lds = 3 * static_cast<size_t>(row_len) * sizeof(int);
  // more than the default 64 KB of dynamic LDS (KQuery x shards_per_gpu > 5461) has to be asked for
  if (lds > 64 * 1024)
    GGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
  hipLaunchKernelGGL(sort_rows_kernel, grid_for(Nq), dim3(kWave), lds, stream, Nq, row_len, ids,
                     dists);
  GGNN_HIP_CHECK(hipGetLastError());
}

// one thread per query: k-way merge of num_parts sorted rows (num_parts is small)
__global__ void merge_results_kernel(uint32_t Nq, uint32_t k, uint32_t num_parts, uint32_t stride,
                                     uint32_t id_offset_per_part, const int32_t* parts_ids,
                                     const float* parts_dists, int32_t* ids_out, float* dists_out,
                                     const uint32_t* qlist, const uint32_t* qcount,
                                     uint32_t first, uint32_t count, size_t part_elems)
{
  // queries [first, first + count) of the Nq rows per part (or the listed ones); the outputs are
  // indexed by the query number as well
  uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= count)
    return;
  n += first;
  if (qlist) {
    if (n >= *qcount)
      return;
    n = qlist[n];
  }
  constexpr uint32_t kMaxParts = 64;
  uint32_t pos[kMaxParts];
  float head[kMaxParts];
  for (uint32_t p = 0; p < num_parts; ++p) {
    pos[p] = 0;
    head[p] = parts_dists[p * part_elems + static_cast<size_t>(n) * stride];
  }
  for (uint32_t j = 0; j < k; ++j) {
    uint32_t bp = 0;
    bool have = false;
    float bd = 0.f;
    for (uint32_t p = 0; p < num_parts; ++p) {
      if (pos[p] >= stride)
        continue;
      // ascending; NaN-free inputs; ties: lower part first
      if (!have || head[p] < bd) {
        have = true;
        bd = head[p];
        bp = p;
      }
    }
    const size_t src = bp * part_elems + static_cast<size_t>(n) * stride + pos[bp];
    ids_out[static_cast<size_t>(n) * k + j] =
        parts_ids[src] + static_cast<int32_t>(bp * id_offset_per_part);
    dists_out[static_cast<size_t>(n) * k + j] = bd;
    ++pos[bp];
    if (pos[bp] < stride)
      head[bp] = parts_dists[src + 1];
  }
}

void launch_merge_results(uint32_t Nq, uint32_t k, uint32_t num_parts, uint32_t stride,
                          uint32_t id_offset_per_part, const int32_t* parts_ids,
                          const float* parts_dists, int32_t* ids_out, float* dists_out,
                          hipStream_t stream)
{
  launch_merge_results_subset(Nq, k, num_parts, stride, id_offset_per_part, parts_ids, parts_dists,
                              ids_out, dists_out, nullptr, nullptr, stream);
}

void launch_merge_results_subset(uint32_t Nq, uint32_t k, uint32_t num_parts, uint32_t stride,
                                 uint32_t id_offset_per_part, const int32_t* parts_ids,
                                 const float* parts_dists, int32_t* ids_out, float* dists_out,
                                 const uint32_t* qlist, const uint32_t* qcount,
                                 hipStream_t stream)
{
  launch_merge_results_range(Nq, k, num_parts, stride, id_offset_per_part, parts_ids, parts_dists,
                             ids_out, dists_out, qlist, qcount, 0, Nq, stream);
}

void launch_merge_results_range(uint32_t Nq, uint32_t k, uint32_t num_parts, uint32_t stride,
                                uint32_t id_offset_per_part, const int32_t* parts_ids,
                                const float* parts_dists, int32_t* ids_out, float* dists_out,
                                const uint32_t* qlist, const uint32_t* qcount, uint32_t first,
                                uint32_t count, hipStream_t stream, size_t part_elems)
{
  if (!Nq || !count)
    return;
  GGNN_REQUIRE(num_parts >= 1 && num_parts <= 64, GGNN_INVALID_ARGUMENT,
               "number of parts must be in [1, 64]");
  GGNN_REQUIRE(static_cast<uint64_t>(num_parts) * stride >= k, GGNN_INVALID_ARGUMENT,
               "not enough candidates to merge");
  const uint32_t block = 128;
  hipLaunchKernelGGL(merge_results_kernel, dim3((count + block - 1) / block), dim3(block), 0,
                     stream, Nq, k, num_parts, stride, id_offset_per_part, parts_ids, parts_dists,
                     ids_out, dists_out, qlist, qcount, first, count,
                     part_elems ? part_elems : static_cast<size_t>(Nq) * stride);
  GGNN_HIP_CHECK(hipGetLastError());
}

}  // namespace ggnn_amd
