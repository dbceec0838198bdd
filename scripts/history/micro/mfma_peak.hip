// Practical peak of v_mfma_f32_32x32x2_f32 on this part: register-only chains, no memory.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
// Prints TFLOP/s for 1..4 waves per SIMD and 1 / 2 / 4 independent accumulator chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0)
{
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int i = 0; i < 16; ++i)
      acc[c][i] = 0.f;
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64 / CHAINS; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int i = 0; i < 16; ++i)
      s += acc[c][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
void run(int waves_per_simd)
{
  const int blocks = 256 * waves_per_simd, iters = 4000;
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(blocks) * 4 * iters * 64 * 4096.0;
  printf("chains %d waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", CHAINS, waves_per_simd, ms, flops / ms / 1e9);
  hipFree(out);
}

int main()
{
  for (int w = 1; w <= 4; ++w) {
    run<1>(w);
    run<2>(w);
    run<4>(w);
  }
  return 0;
}
