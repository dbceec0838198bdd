// 8-bit pre-screen copy of a float32 base and its validation probe.
//
// Not part of the reference: an exact pruning aid of the query kernel (traversal.hpp, "Exact
// pre-screen").  Every row x is stored a second time as codes c in [0,255]^Dc with
//     x^_d = o_d + s * c_d,   o_d = min over rows of x_d,   255 s >= max range over dimensions
// together with e_max >= max over rows ||x - x^||_2 (computed in double, rounded up).  The scale
// is first tried as a power of two: integer (or dyadic) data with a range <= 255 s per dimension
// (SIFT) is then coded without loss (e_max = 0).  Otherwise the rows are coded again with the
// tightest scale, max range / 255.
// For the cosine measure the rows are coded after normalisation to unit length (a zero row stays
// zero); e_max then bounds the error against the exactly normalised row.
#include "traversal.hpp"

namespace ggnn_amd {

namespace {

constexpr uint32_t kPsBlocks = 1024;
constexpr uint32_t kPsThreads = 256;

// partial[b][0][d] = min, partial[b][1][d] = max over the rows b, b+B, ... ; flags[0] != 0 when
// a non-finite value was seen
__global__ void __launch_bounds__(kPsThreads)
    ps_minmax_kernel(const float* base, uint32_t N, uint32_t D, uint32_t B, const float* inv_norm,
                     float* partial, uint32_t* flags)
{
  const uint32_t b = blockIdx.x;
  bool bad = false;
  for (uint32_t d = threadIdx.x; d < D; d += kPsThreads) {
    float mn = inf_f(), mx = -inf_f();
    for (uint64_t r = b; r < N; r += B) {
      float v = base[r * D + d];
      bad |= !(fabsf(v) < inf_f());
      if (inv_norm)
        v *= inv_norm[r];
      bad |= !(fabsf(v) < inf_f());
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
    partial[(static_cast<size_t>(b) * 2 + 0) * D + d] = mn;
    partial[(static_cast<size_t>(b) * 2 + 1) * D + d] = mx;
  }
  if (bad)
    atomicOr(flags, 1u);
}

// one block: per-dimension offsets, the common scale, header of params
__global__ void __launch_bounds__(kPsThreads)
    ps_finalize_kernel(const float* partial, const uint32_t* flags, uint32_t D, uint32_t Dc,
                       uint32_t B, int measure, float* params)
{
  __shared__ float s_range[kPsThreads];
  __shared__ float s_osq[kPsThreads];
  float* offs = params + kPsHeader;
  float range = 0.f, osq = 0.f;
  for (uint32_t d = threadIdx.x; d < Dc; d += kPsThreads) {
    float o = 0.f;
    if (d < D) {
      float mn = inf_f(), mx = -inf_f();
      for (uint32_t b = 0; b < B; ++b) {
        mn = fminf(mn, partial[(static_cast<size_t>(b) * 2 + 0) * D + d]);
        mx = fmaxf(mx, partial[(static_cast<size_t>(b) * 2 + 1) * D + d]);
      }
      o = mn;
      range = fmaxf(range, mx - mn);
    }
    offs[d] = o;
    osq = fmaf(o, o, osq);
  }
  s_range[threadIdx.x] = range;
  s_osq[threadIdx.x] = osq;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t t = 1; t < kPsThreads; ++t) {
      range = fmaxf(range, s_range[t]);
      osq += s_osq[t];
    }
    float s = 1.f;
    if (range > 0.f && range < inf_f()) {
      s = exp2f(ceilf(log2f(range / 255.f)));
      while (255.f * s < range)
        s *= 2.f;
    }
    const bool ok = flags[0] == 0 && range < inf_f() && s >= 0x1p-60f && s <= 0x1p60f &&
                    osq < inf_f();
    params[0] = s;
    params[1] = 1.f / s;  // exact: s is a power of two
    params[2] = 0.f;      // e_max, raised by ps_encode_kernel
    params[3] = sqrtf(osq) * (1.f + 1e-6f);
    params[4] = ok ? 1.f : 0.f;
    params[5] = range;  // for ps_retry_kernel
    params[6] = 1.f;    // coding pass wanted
    params[7] = static_cast<float>(measure);
  }
}

// after the first coding pass: lossy with the power-of-two scale -> ask for a second pass with
// the tightest scale
__global__ void ps_retry_kernel(float* params)
{
  const float range = params[5];
  if (params[4] != 0.f && params[2] > 0.f && range > 0.f) {
    const float s = range / 255.f * (1.f + 0x1p-20f);
    params[0] = s;
    params[1] = 1.f / s;
    params[2] = 0.f;
    params[6] = 1.f;
  }
  else
    params[6] = 0.f;
}

// sum of squares of one row in double, lpr lanes per row; every lane of the row gets the total
__device__ __forceinline__ double ps_row_norm2(const float* base, uint64_t row, uint32_t N,
                                               uint32_t D, uint32_t lpr, uint32_t g)
{
  double acc = 0.0;
  if (row < N) {
    const float* x = base + row * D;
    for (uint32_t d = g; d < D; d += lpr)
      acc += static_cast<double>(x[d]) * static_cast<double>(x[d]);
  }
  for (uint32_t off = lpr / 2; off; off >>= 1)
    acc += __shfl_xor(acc, off);
  return acc;
}

// cosine: 1 / |x| per row in float (0 for a zero row), used by the range pass and the coder
__global__ void __launch_bounds__(kPsThreads)
    ps_inv_norm_kernel(const float* base, uint32_t N, uint32_t D, uint32_t lpr, float* inv_norm)
{
  const uint32_t rows_per_block = kPsThreads / lpr;
  const uint64_t row = static_cast<uint64_t>(block_linear_index()) * rows_per_block +
                       threadIdx.x / lpr;
  const uint32_t g = threadIdx.x % lpr;
  const double n2 = ps_row_norm2(base, row, N, D, lpr, g);
  if (row < N && g == 0)
    inv_norm[row] = n2 > 0.0 ? static_cast<float>(1.0 / sqrt(n2)) : 0.f;
}

// lpr lanes per row (power of two >= Dc/16 capped at 64); every lane codes 16 dimensions at a time
__global__ void __launch_bounds__(kPsThreads)
    ps_encode_kernel(const float* base, uint32_t N, uint32_t D, uint32_t Dc, uint32_t lpr,
                     const float* inv_norm, uint8_t* codes, float* params)
{
  const uint32_t rows_per_block = kPsThreads / lpr;
  const uint64_t row = static_cast<uint64_t>(block_linear_index()) * rows_per_block +
                       threadIdx.x / lpr;
  const uint32_t g = threadIdx.x % lpr;
  if (params[6] == 0.f)
    return;  // the first pass was lossless
  const float s = params[0], inv_s = params[1];
  const float* offs = params + kPsHeader;
  // cosine: the exactly normalised row (in double) is what the codes are measured against
  double exact_scale = 1.0;
  float code_scale = 1.f;
  if (inv_norm) {
    const double n2 = ps_row_norm2(base, row, N, D, lpr, g);
    exact_scale = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
    code_scale = row < N ? inv_norm[row] : 0.f;
  }
  double err = 0.0;
  if (row < N) {
    const float* x = base + row * D;
    uint8_t* out = codes + row * Dc;
    for (uint32_t d0 = g * 16; d0 < Dc; d0 += lpr * 16) {
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w[j] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t d = d0 + 4 * j + e;
          if (d < D) {
            const float v = x[d];
            const float o = offs[d];
            const float c = fminf(fmaxf(rintf((v * code_scale - o) * inv_s), 0.f), 255.f);
            w[j] |= static_cast<uint32_t>(c) << (8 * e);
            const double res = static_cast<double>(v) * exact_scale -
                               (static_cast<double>(o) + static_cast<double>(s) * c);
            err += res * res;
          }
        }
      }
      *reinterpret_cast<uint4*>(out + d0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  for (uint32_t off = lpr / 2; off; off >>= 1)
    err += __shfl_xor(err, off);
  if (row < N && g == 0 && err > 0.0) {
    // round up generously; the additive term covers the rounding of o + s*c in double
    const float e = static_cast<float>(sqrt(err)) * (1.f + 1e-6f) +
                    1e-12f * (params[3] + 255.f * s * sqrtf(static_cast<float>(Dc)));
    atomicMax(reinterpret_cast<uint32_t*>(params + 2), __float_as_uint(e));
  }
}

// validation probe: evaluates the pre-screen exactly as fetch() does for explicit
// (query, candidate, criteria) triples.  One wave per query; candidates in rounds of 32.
template <int LPR, int NCH, int MODE>
__global__ void __launch_bounds__(kWave)
    ps_probe_kernel(const uint8_t* codes, const float* params, uint32_t D, uint32_t Dc,
                    const float* query, uint32_t Nq, const int32_t* cand, uint32_t M,
                    const float* crit, int32_t* reject, float* s_out)
{
  using PS = Prescreen<LPR, NCH, MODE>;
  const uint32_t n = block_linear_index();
  if (n >= Nq)
    return;
  PS ps;
  ps.load(codes, params, Dc, query + static_cast<size_t>(n) * D, D);
  const int grp = threadIdx.x / LPR;
  for (uint32_t j0 = 0; j0 < M; j0 += PS::ROWS) {
    const uint32_t j = j0 + grp;
    const bool valid = j < M;
    const int k = valid ? cand[static_cast<size_t>(n) * M + j] : 0;
    const uint8_t* row = ps.row_ptr(k);
    uint4 v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      v[c] = make_uint4(0u, 0u, 0u, 0u);
      if (valid && ps.chunk_valid(c))
        v[c] = ps.load_chunk(row, c);
    }
    const float S = group_sum<LPR>(ps.partial(v));
    if (valid && ps.g == 0) {
      const float thr = ps.threshold(crit[static_cast<size_t>(n) * M + j]);
      reject[static_cast<size_t>(n) * M + j] = (S >= thr) ? 1 : 0;
      if (s_out)
        s_out[static_cast<size_t>(n) * M + j] = S;
    }
  }
}

}  // namespace

size_t prescreen_param_floats(uint32_t D)
{
  return kPsHeader + prescreen_code_dim(D);
}
size_t prescreen_scratch_floats(uint32_t N, uint32_t D, ggnn_measure measure)
{
  return static_cast<size_t>(kPsBlocks) * 2 * D + 4 + (measure == GGNN_COSINE ? N : 0);
}

void launch_prescreen_encode(const float* base, uint32_t N, uint32_t D, ggnn_measure measure,
                             uint8_t* codes, float* params, float* scratch, hipStream_t stream)
{
  GGNN_REQUIRE(N >= 1, GGNN_INVALID_ARGUMENT, "empty base");
  GGNN_REQUIRE(D >= 1 && D <= 4096 && D % 4 == 0, GGNN_INVALID_ARGUMENT,
               "D must be a multiple of 4 in [4, 4096]");
  GGNN_REQUIRE(((reinterpret_cast<uintptr_t>(codes) | reinterpret_cast<uintptr_t>(params)) & 15u) ==
                   0,
               GGNN_INVALID_ARGUMENT, "pre-screen buffers must be 16-byte aligned");
  const uint32_t Dc = prescreen_code_dim(D);
  const uint32_t B = std::min(kPsBlocks, N);
  uint32_t* flags = reinterpret_cast<uint32_t*>(scratch + static_cast<size_t>(kPsBlocks) * 2 * D);
  float* inv_norm = measure == GGNN_COSINE ? scratch + static_cast<size_t>(kPsBlocks) * 2 * D + 4
                                           : nullptr;
  uint32_t lpr = 1;
  while (lpr < 64 && lpr * 16 < Dc)
    lpr *= 2;
  const uint32_t rows_per_block = kPsThreads / lpr;
  const dim3 grid = grid_for((static_cast<uint64_t>(N) + rows_per_block - 1) / rows_per_block);
  GGNN_HIP_CHECK(hipMemsetAsync(flags, 0, 4 * sizeof(uint32_t), stream));
  if (inv_norm)
    hipLaunchKernelGGL(ps_inv_norm_kernel, grid, dim3(kPsThreads), 0, stream, base, N, D, lpr,
                       inv_norm);
  hipLaunchKernelGGL(ps_minmax_kernel, dim3(B), dim3(kPsThreads), 0, stream, base, N, D, B,
                     inv_norm, scratch, flags);
  hipLaunchKernelGGL(ps_finalize_kernel, dim3(1), dim3(kPsThreads), 0, stream, scratch, flags, D,
                     Dc, B, static_cast<int>(measure), params);
  hipLaunchKernelGGL(ps_encode_kernel, grid, dim3(kPsThreads), 0, stream, base, N, D, Dc, lpr,
                     inv_norm, codes, params);
  hipLaunchKernelGGL(ps_retry_kernel, dim3(1), dim3(1), 0, stream, params);
  hipLaunchKernelGGL(ps_encode_kernel, grid, dim3(kPsThreads), 0, stream, base, N, D, Dc, lpr,
                     inv_norm, codes, params);
  GGNN_HIP_CHECK(hipGetLastError());
}

void launch_prescreen_probe(const uint8_t* codes, const float* params, uint32_t D,
                            ggnn_measure measure, const float* query, uint32_t Nq,
                            const int32_t* cand, uint32_t M, const float* crit, int32_t* reject,
                            float* s_out, hipStream_t stream)
{
  if (!Nq || !M)
    return;
  GGNN_REQUIRE(D >= 1 && D <= 4096 && D % 4 == 0, GGNN_INVALID_ARGUMENT,
               "D must be a multiple of 4 in [4, 4096]");
  const uint32_t Dc = prescreen_code_dim(D);
  const uint32_t chunks = Dc / 16;
#define GGNN_PROBE(LPR, NCH)                                                                      \
  do {                                                                                            \
    if (measure == GGNN_EUCLIDEAN)                                                                \
      hipLaunchKernelGGL((ps_probe_kernel<LPR, NCH, kL2>), grid_for(Nq), dim3(kWave), 0, stream,  \
                         codes, params, D, Dc, query, Nq, cand, M, crit, reject, s_out);         \
    else                                                                                          \
      hipLaunchKernelGGL((ps_probe_kernel<LPR, NCH, kCos>), grid_for(Nq), dim3(kWave), 0, stream, \
                         codes, params, D, Dc, query, Nq, cand, M, crit, reject, s_out);         \
  } while (0)
  // the same layouts launch_query pairs with the float-row layouts
  if (chunks <= 8)
    GGNN_PROBE(8, 1);
  else if (chunks <= 16)
    GGNN_PROBE(16, 1);
  else if (chunks <= 64)
    GGNN_PROBE(32, 2);
  else
    GGNN_PROBE(64, 4);
#undef GGNN_PROBE
  GGNN_HIP_CHECK(hipGetLastError());
}

}  // namespace ggnn_amd
