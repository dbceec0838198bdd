// TEST INFRASTRUCTURE ONLY.  Harness around the REFERENCE's own device code:
// /root/reference/include/ggnn/cuda_utils/k_best_list.cuh is #included from where it lies (it
// needs nothing but <cstdint>/<limits>, so hipcc compiles it unchanged for gfx950); nothing of it
// is copied into this repository.  Built by oracle/Makefile into oracle/_ref/ (git-ignored, ships
// to the GPU box), used by tests/test_gpu_ref_kbest.py to pin the oracle's KBestList emulation
// -- and through it bf_query / top -- against the reference implementation executed on MI355X.
#include <hip/hip_runtime.h>

#include <cstdint>

#include <ggnn/cuda_utils/k_best_list.cuh>

template <uint32_t BLOCK>
__global__ void __launch_bounds__(BLOCK) kbest_script_kernel(uint32_t best_size, const float* dists,
                                                             const int32_t* ids, uint32_t n,
                                                             int check_worst, float* out_d,
                                                             int32_t* out_i)
{
  ggnn::KBestList<int32_t, float, BLOCK> best(best_size);
  __syncthreads();
  for (uint32_t i = 0; i < n; ++i) {
    // bf_query_layer.cu:52-57 guards with worst(); top_merge_layer.cu:68 does not
    if (!check_worst || dists[i] < best.worst())
      best.add_unique(dists[i], ids[i]);
    __syncthreads();
  }
  for (uint32_t k = threadIdx.x; k < best_size; k += BLOCK) {
    out_d[k] = best.s_dists[k];
    out_i[k] = best.s_ids[k];
  }
}

extern "C" int ref_kbest_script(uint32_t block, uint32_t best_size, const float* d_dists,
                                const int32_t* d_ids, uint32_t n, int check_worst, float* d_out_d,
                                int32_t* d_out_i)
{
  const size_t smem = best_size * (sizeof(float) + sizeof(int32_t));
#define LAUNCH(B)                                                                             \
  hipLaunchKernelGGL((kbest_script_kernel<B>), dim3(1), dim3(B), smem, 0, best_size, d_dists, \
                     d_ids, n, check_worst, d_out_d, d_out_i)
  switch (block) {
    case 32: LAUNCH(32); break;
    case 64: LAUNCH(64); break;
    case 128: LAUNCH(128); break;
    case 256: LAUNCH(256); break;
    default: return -1;
  }
#undef LAUNCH
  if (hipGetLastError() != hipSuccess)
    return -2;
  return hipDeviceSynchronize() == hipSuccess ? 0 : -3;
}
