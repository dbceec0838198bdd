// Where a tiled f32 MFMA contraction loses the matrix pipe: the chain of the bf kernel with its
// surroundings added one at a time (register-only -> B operand from LDS -> barrier per tile ->
// tile staging global->reg->LDS -> threshold test).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_steps.hip -o scripts/micro/mfma_steps
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int DP = 132, ROWS = 32;

__device__ float operand(unsigned i, int full)
{
  // integer-valued (0..255: few mantissa bits set) or full-mantissa values of the same magnitude
  unsigned x = i * 2654435761u;
  x ^= x >> 15;
  const float v = static_cast<float>(x & 255u);
  return full ? (v - 127.37f) * 1.0001234f + static_cast<float>((x >> 8) & 4095u) * 1.1e-4f : v;
}

template <int V>
__global__ void __launch_bounds__(256) k(const float* base, float* out, int tiles, float a0,
                                         int full = 0)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  float aq[64];
  for (int i = 0; i < 64; ++i)
    aq[i] = operand(i * 256 + tid, full);
  for (int i = tid; i < 2 * ROWS * DP; i += 256)
    lds[i] = operand(77777 + i, full);
  __syncthreads();
  float4 r[4];
  float thr = 1e30f;
  unsigned long long hits = 0;
  const float* src = base + (size_t)blockIdx.x * 4096 * 37;
  f32x16 total = {0};
  for (int tt = 0; tt < tiles; ++tt) {
    if (V >= 3) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = tid + 256 * e;
        r[e] = *reinterpret_cast<const float4*>(src + (size_t)tt * 4096 + idx * 4);
      }
    }
    f32x16 acc = {0};
    const float* bt = lds + (tt & 1) * ROWS * DP + j * DP + h * 64;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float4 bv;
      if (V >= 1)
        bv = *reinterpret_cast<const float4*>(bt + 4 * u);
      else
        bv = make_float4(aq[u], aq[u + 1], aq[u + 2], aq[u + 3]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 0], bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 1], bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 2], bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 3], bv.w, acc, 0, 0, 0);
    }
    if (V >= 4) {
      unsigned long long any = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        any |= __ballot(fmaf(-2.f, acc[i], 3.f + i) < thr);
      if (any) {
        hits += any;
        thr = -1e30f;
      }
    }
    else
      total += acc;
    if (V >= 3) {
      float* dst = lds + ((tt + 1) & 1) * ROWS * DP;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = tid + 256 * e;
        *reinterpret_cast<float4*>(dst + (idx >> 5) * DP + 4 * (idx & 31)) = r[e];
      }
    }
    if (V >= 2)
      __syncthreads();
  }
  float s = hits;
  for (int i = 0; i < 16; ++i)
    s += total[i];
  out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(int waves_per_simd, const float* base, float* out, int full = 0)
{
  const int blocks = 256 * waves_per_simd, tiles = 4000;
  const size_t lds = 2 * ROWS * DP * 4 + (waves_per_simd == 2 ? 40000 : 16000);  // pins blocks/CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, base, out, 50, 1.f, full);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, base, out, tiles, 1.f, full);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(blocks) * 4 * tiles * 64 * 4096.0;
  static const char* names[] = {"registers only", "+ B operand from LDS", "+ barrier per tile",
                                "+ tile staging", "+ threshold test"};
  printf("waves/SIMD %d  %-22s %-14s %.3f ms  %.1f TFLOP/s\n", waves_per_simd, names[V],
         full ? "full mantissa" : "integer-valued", ms, flops / ms / 1e9);
}

int main()
{
  float *base, *out;
  const size_t n = (size_t)768 * 4096 * 37 + 4001 * 4096;
  hipMalloc(&base, n * 4);
  hipMemset(base, 0, n * 4);
  hipMalloc(&out, 768 * 256 * 4);
  for (int w = 2; w <= 3; ++w) {
    run<0>(w, base, out);
    run<1>(w, base, out);
    run<2>(w, base, out);
    run<3>(w, base, out);
    run<4>(w, base, out);
  }
  // operand values: the same kernels on full-mantissa operands (the staged tile of variant 3 / 4
  // comes from `base`, which main() fills accordingly)
  for (int rep = 0; rep < 2; ++rep) {
    run<1>(3, base, out, 0);
    run<1>(3, base, out, 1);
    run<2>(3, base, out, 1);
  }
  return 0;
}
