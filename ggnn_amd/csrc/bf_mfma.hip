// bf_query, MFMA path: the exhaustive scan is a true Q x B^T contraction, so it runs on the matrix
// cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf chain).
// Reference being replaced: BruteForceQueryKernel, src/ggnn/query/bf_query_layer.cu:39-65 (N
// sequential block reductions per query, no reuse of base rows across queries).
//
// Structure
//   1. row_norms_kernel: |b|^2 for every base row, |q|^2 for every query.
//   2. bf_mfma_kernel: a block = 4 waves x 32 queries; base tiles of 32 rows are staged through
//      LDS once per block (double buffered) and contracted against the wave's query tile that
//      lives in registers.  dist = |q|^2 + |b|^2 - 2 q.b (or the cosine form).  Each query keeps
//      a sorted candidate list of KP = K + margin entries in LDS; a distance enters only if it
//      beats the list's worst entry (rare after the first tiles), insertion is stable so equal
//      distances keep the lower base index first (KBestList rule, k_best_list.cuh:92-103).
//   3. bf_rerank_kernel: the candidates of all base slices are re-evaluated with the reference's
//      direct formula (DistEngine, distance.cuh:119-163) and ranked by (distance, index), so the
//      returned distances are the direct-form values; the expanded form only pre-selects.
//   4. Exactness certificate (per query, in the re-rank kernel) and exact fallback: see
//      "Why the pre-selection cannot lose a neighbour" below.  Queries that cannot be certified
//      are answered by the scan kernel (bf_query.hip), so the result is ALWAYS the top K by
//      (direct-form distance, index) -- the reference's result (bf_query_layer.cu:52-57).
//
// Why the pre-selection cannot lose a neighbour
//   Float32 squared L2 works on a = fl(q - mu), b = fl(x - mu) with mu = a column mean of the
//   base (any vector works: L2 is shift invariant; centring makes the norms comparable to the
//   distances, so the expanded form does not cancel).  With u = 2^-24, for ANY summation order
//     |d_e - ||a-b||^2|  <= (2D+4) u (|a|^2+|b|^2)      (two norm chains, dot chain, 2 roundings)
//     | ||a-b||^2 - ||q-x||^2 | <= 4u (|a|^2+|b|^2)     (rounding of the centred coordinates)
//     d_dir >= ||q-x||^2 (1 - (D+3)u)                   (direct form: diff, fma chain, tree)
//   A slice list keeps the KP smallest (d_e, index); a row left out of the list of slice s has
//   d_e >= w_s (the list's worst entry, +inf while the list is not full).  Hence with
//     E = 1.01 (2D+8) u (|a|^2 + max_rows |b|^2) + 4u w      and      w = min_s w_s
//   every left-out row has d_dir >= (w - E)(1 - 1.01 (D+3) u).  If that exceeds the K-th re-ranked
//   distance the query is certified; otherwise it is re-scanned.  Cosine: both forms are within
//   (2D+8)u of the true |1 - cos| in absolute terms (|cos| <= 1), so E = 2.02 (2D+8) u.
//   uint8 rows on the i8 path: every quantity is an exact integer < 2^24, a left-out row is
//   preceded by KP > K rows of its own slice in exact (distance, index) order: nothing to check.
#include <cstdlib>

#include "bf_common.hpp"
#include "hooks.hpp"

namespace ggnn_amd {

// Stats build (-DGGNN_BF_PHASE, scripts/bf_phase_cycles.py): shader cycles per phase of a tile of the
// single-chunk f32 kernel, per wave (ticks are s_memtime reads fenced against the scheduler)
#ifdef GGNN_BF_PHASE
static __device__ unsigned long long g_bf_phase[8];
static __device__ const float* g_bf_dbg_bound;  // experiment: per-query bound the lists start from
#define GGNN_BF_TICK(i)                                                  \
  do {                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                   \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();       \
    bf_ph[i] += t_now - bf_t;                                            \
    bf_t = t_now;                                                        \
    __builtin_amdgcn_sched_barrier(0);                                   \
  } while (0)
#else
#define GGNN_BF_TICK(i)
#endif

// ---- 1. squared norms ---------------------------------------------------------------------------
// SHIFT (uint8 only): norms of x - 128, the values the i8 matrix path works on; squared L2
// distances do not change under a common shift
// mean (optional): per-column shift, the same fl(x - mean) the tile kernel forms
// max_out (optional): running maximum of the norms as float bits (norms are >= 0, so the
// unsigned order of the bit patterns is the numeric order)
template <typename BaseT, bool SHIFT = false>
__global__ void __launch_bounds__(256) row_norms_kernel(const BaseT* data, uint32_t N, uint32_t D,
                                                       const float* mean, float* out,
                                                       uint32_t* max_out)
{
  constexpr int EPC = ChunkOf<BaseT>::EPC;
  using Chunk = typename ChunkOf<BaseT>::type;
  const uint32_t g = threadIdx.x & 15;
  float vmax = 0.f;
  // grid-stride over groups of 16 rows (one row per 16 lanes, 16 bytes per lane and step), four
  // groups per iteration: with one row in flight per lane the kernel ran at the latency of its
  // loads (0.38 ms for 1M x 128 bytes)
  constexpr int U = 4;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 16;
  for (uint64_t row0 = static_cast<uint64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4); row0 < N;
       row0 += U * stride) {
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      acc[u] = 0.f;
    for (uint32_t e0 = g * EPC; e0 < D; e0 += 16 * EPC) {
      Chunk v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t row = row0 + u * stride;
        v[u] = *reinterpret_cast<const Chunk*>(data + (row < N ? row : row0) * D + e0);
      }
      float m[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e)
        m[e] = SHIFT ? 128.f : 0.f;
      if constexpr (!SHIFT && EPC == 4) {
        if (mean) {
          const float4 mv = *reinterpret_cast<const float4*>(mean + e0);
          m[0] = mv.x, m[1] = mv.y, m[2] = mv.z, m[3] = mv.w;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float x = ChunkOf<BaseT>::get(v[u], e) - m[e];
          acc[u] = fmaf(x, x, acc[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t row = row0 + u * stride;
      const float a = group_sum<16>(acc[u]);
      if (row < N) {
        if (g == 0)
          out[row] = a;
        vmax = fmaxf(vmax, a);
      }
    }
  }
  if (max_out) {
    // one atomic per wave (a million atomics on one address take milliseconds)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
      vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((threadIdx.x & 63) == 0)
      atomicMax(max_out, __float_as_uint(vmax));
  }
}

// out = fl(in - mean), row by row: the query set is shifted once, up front (it is small), so that
// the tile kernels load their A operand without touching the mean
__global__ void __launch_bounds__(256) shift_rows_kernel(const float* in, uint64_t n4, uint32_t D,
                                                        const float* mean, float* out)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4;
       i += static_cast<uint64_t>(gridDim.x) * 256) {
    const uint32_t col = static_cast<uint32_t>((i * 4) % D);
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    const float4 m = *reinterpret_cast<const float4*>(mean + col);
    reinterpret_cast<float4*>(out)[i] = make_float4(v.x - m.x, v.y - m.y, v.z - m.z, v.w - m.w);
  }
}

inline uint32_t norm_grid(uint32_t rows)
{
  // (2048 workgroups: enough to fill the chip with four rows in flight per lane group, and a
  // quarter of the per-wave atomics on the one norm-maximum word, which is what a launch of 8192
  // spent most of its 0.37 ms on)
  return std::max(1u, std::min((rows + 15u) / 16u, 2048u));
}

// Column means of (a sample of) the base: partial[p][c] = sum over rows p, p+P, ... of the
// sampled rows; finalised in a fixed order, so the shift -- and with it every intermediate of the
// tile kernel -- is the same from run to run.  The shift only has to be close to the data: any
// vector gives exact results (see the certificate above).
constexpr uint32_t kBfMeanBlocks = 256;
__global__ void __launch_bounds__(256) col_mean_partial_kernel(const float* data, uint32_t N,
                                                              uint32_t D, uint32_t stride,
                                                              uint32_t rows, float* partial)
{
  for (uint32_t c = threadIdx.x; c < D; c += 256) {
    // four independent chains: the loop is a sequence of dependent-latency loads otherwise
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t r = blockIdx.x;
    for (; r + 3 * kBfMeanBlocks < rows; r += 4 * kBfMeanBlocks) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] += data[static_cast<size_t>(r + j * kBfMeanBlocks) * stride * D + c];
    }
    for (; r < rows; r += kBfMeanBlocks)
      acc[0] += data[static_cast<size_t>(r) * stride * D + c];
    partial[blockIdx.x * D + c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
}
__global__ void __launch_bounds__(256) col_mean_final_kernel(const float* partial, uint32_t D,
                                                            uint32_t rows, float* mean)
{
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D)
    return;
  float acc = 0.f;
  for (uint32_t p = 0; p < kBfMeanBlocks; ++p)
    acc += partial[p * D + c];
  const float m = acc / static_cast<float>(rows);
  // data the mean of which is not finite is left unshifted (the certificate then decides)
  mean[c] = (fabsf(m) < inf_f()) ? m : 0.f;
}

// ---- 2. tile kernel -------------------------------------------------------------------------------
// The K dimension is processed in chunks of CW = 2*Dh columns (Dh = 64 when D > 128, otherwise
// ceil(D/2) rounded up to the vector width).  Lane (j, h) of a wave owns chunk columns
// [h*Dh, (h+1)*Dh) of base row j (B operand) and of query row j (A operand), so MFMA step kk
// contracts global dimension col0 + h*Dh + kk -- any pairing works as long as A and B agree.
template <typename BaseT>
struct TileStage;

// f32: CW/4 float4 per row, 32 rows -> at most 1024 float4 = 4 per thread
template <>
struct TileStage<float> {
  float4 r[4];
  float bn;          // squared norm of tile row threadIdx.x & 31
  uint32_t rows;     // rows of the staged tile that exist (the others get bn_pad)
  float bn_pad;
  // Every lane loads, unconditionally and without initialising its registers first: rows past
  // `end` re-read the last row of the range and columns past D the row's last four -- finite
  // values that never matter (such a row gets the +inf / invalid norm at the store, and the query
  // operand is zero in such a column).  A predicated load needs `r = 0` in front of it, and a
  // VALU write to a register that a load of the previous tile targeted makes the compiler wait for
  // vmcnt(0) at the top of the tile loop -- i.e. for every query piece the chunked kernel had
  // just prefetched (222 -> 165 ms at D = 960 when this wait was found).
  GGNN_DEV void load(const float* base, uint32_t D, uint32_t row0, uint32_t end, uint32_t col0,
                     uint32_t CW, const float* bnorm, float bn_pad_)
  {
    const uint32_t last = end - 1;  // (end > 0; row0 >= end: a group's padding tile, no valid row)
    rows = end > row0 ? end - row0 : 0u;
    bn_pad = bn_pad_;
    bn = bnorm[min(row0 + (threadIdx.x & 31u), last)];
    const uint32_t cpr = CW / 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t idx = threadIdx.x + 256 * e;
      const uint32_t row = idx / cpr, col = col0 + 4 * (idx % cpr);
      r[e] = *reinterpret_cast<const float4*>(base + static_cast<size_t>(min(row0 + row, last)) * D +
                                              min(col, D - 4));
    }
  }
  GGNN_DEV float norm() const { return threadIdx.x < rows ? bn : bn_pad; }  // threads 0..31
  // the shift of columns [col0, col0 + CW) is read from the shift vector in LDS, which is zero
  // past column D (so that the clamped columns stay finite)
  GGNN_DEV void store_shifted(float* tile, uint32_t DP, uint32_t CW, const float* mean_lds,
                              uint32_t col0) const
  {
    if (threadIdx.x < (uint32_t)kBfTileRows)
      tile[threadIdx.x * DP + CW] = threadIdx.x < rows ? bn : bn_pad;
    const uint32_t cpr = CW / 4;
    // the four shift pieces are requested together, ahead of the wait for the staged rows.  The
    // opaque zero keeps the compiler from hoisting them out of the tile loop, where they would
    // cost 16 registers (and with them the third wave per SIMD of the single-chunk kernel).
    uint32_t opaque = 0;
    asm volatile("" : "+v"(opaque));
    float4 m[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      m[e] = *reinterpret_cast<const float4*>(mean_lds + opaque + col0 +
                                              4 * ((threadIdx.x + 256 * e) % cpr));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t idx = threadIdx.x + 256 * e;
      const uint32_t row = idx / cpr, c4 = idx % cpr;
      if (row < (uint32_t)kBfTileRows)
        *reinterpret_cast<float4*>(tile + row * DP + 4 * c4) =
            make_float4(r[e].x - m[e].x, r[e].y - m[e].y, r[e].z - m[e].z, r[e].w - m[e].w);
    }
  }
};

// u8: CW/16 chunks of 16 bytes per row, 32 rows -> at most 256 chunks, one per thread
// (unconditional clamped loads as above)
template <>
struct TileStage<uint8_t> {
  uint4 r;
  float bn;
  uint32_t rows;
  float bn_pad;
  GGNN_DEV void load(const uint8_t* base, uint32_t D, uint32_t row0, uint32_t end, uint32_t col0,
                     uint32_t CW, const float* bnorm, float bn_pad_)
  {
    const uint32_t last = end - 1;
    rows = end > row0 ? end - row0 : 0u;
    bn_pad = bn_pad_;
    bn = bnorm[min(row0 + (threadIdx.x & 31u), last)];
    const uint32_t cpr = CW / 16;
    const uint32_t row = threadIdx.x / cpr, col = col0 + 16 * (threadIdx.x % cpr);
    r = *reinterpret_cast<const uint4*>(base + static_cast<size_t>(min(row0 + row, last)) * D +
                                        min(col, D - 16));
  }
  GGNN_DEV float norm() const { return threadIdx.x < rows ? bn : bn_pad; }  // threads 0..31
  GGNN_DEV void store_shifted(float* tile, uint32_t DP, uint32_t CW, const float*, uint32_t) const
  {
    if (threadIdx.x < (uint32_t)kBfTileRows)
      tile[threadIdx.x * DP + CW] = threadIdx.x < rows ? bn : bn_pad;
    const uint32_t cpr = CW / 16;
    const uint32_t row = threadIdx.x / cpr, c = threadIdx.x % cpr;
    if (row < (uint32_t)kBfTileRows) {
      float* dst = tile + row * DP + 16 * c;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float4 f;
        f.x = ChunkOf<uint8_t>::get(r, 4 * w + 0);
        f.y = ChunkOf<uint8_t>::get(r, 4 * w + 1);
        f.z = ChunkOf<uint8_t>::get(r, 4 * w + 2);
        f.w = ChunkOf<uint8_t>::get(r, 4 * w + 3);
        *reinterpret_cast<float4*>(dst + 4 * w) = f;
      }
    }
  }
};

// A operand of one chunk: aq[kk] = q[col0 + h*Dh + kk] (0 outside the row / the query set)
// (float32 squared L2: `query` is the shifted copy made by shift_rows_kernel)
// (predicated on purpose: with unconditional loads the scheduler hoists all sixteen pieces of the
// next chunk to the top of the chain and spills 55 registers at two waves per SIMD)
GGNN_DEV float4 load_query_piece(const float* qrow, bool qvalid, uint32_t D, uint32_t Dh,
                                 uint32_t col_h, int t)
{
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (qvalid && static_cast<uint32_t>(4 * t) < Dh && col_h + 4 * t < D)
    v = *reinterpret_cast<const float4*>(qrow + col_h + 4 * t);
  return v;
}
GGNN_DEV void load_query_chunk(float (&aq)[64], const float* qrow, bool qvalid, uint32_t D,
                               uint32_t Dh, uint32_t col_h)
{
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 v = load_query_piece(qrow, qvalid, D, Dh, col_h, t);
    aq[4 * t + 0] = v.x;
    aq[4 * t + 1] = v.y;
    aq[4 * t + 2] = v.z;
    aq[4 * t + 3] = v.w;
  }
}
// piece t (4 operands) of a uint8 query chunk: bytes [4t, 4t+4) of the lane's half row
GGNN_DEV float4 load_query_piece(const uint8_t* qrow, bool qvalid, uint32_t D, uint32_t Dh,
                                 uint32_t col_h, int t)
{
  const uint32_t col = col_h + 4 * t;
  const bool ok = qvalid && static_cast<uint32_t>(4 * t) < Dh && col < D;
  uint32_t w = *reinterpret_cast<const uint32_t*>(qrow + min(col, D - 4));
  w = ok ? w : 0u;
  return make_float4(static_cast<float>(w & 0xffu), static_cast<float>((w >> 8) & 0xffu),
                     static_cast<float>((w >> 16) & 0xffu), static_cast<float>(w >> 24));
}
GGNN_DEV void load_query_chunk(float (&aq)[64], const uint8_t* qrow, bool qvalid, uint32_t D,
                               uint32_t Dh, uint32_t col_h)
{
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (qvalid && static_cast<uint32_t>(16 * t) < Dh && col_h + 16 * t < D)
      v = *reinterpret_cast<const uint4*>(qrow + col_h + 16 * t);
#pragma unroll
    for (int e = 0; e < 16; ++e)
      aq[16 * t + e] = ChunkOf<uint8_t>::get(v, e);
  }
}

// Rare path of the epilogue: stable insertion of the tile's distances that beat a list's worst
// entry, candidates in ascending lane order (= ascending base index within each query row).
// dd[r] / thr[r]: distance and threshold of register row r (query row (r&3) + 8*(r>>2) + 4*h).
// thr[r] of EVERY lane equals the worst entry of its query's list at all times (it is set after
// each insertion below), so neither the test of a candidate nor the new threshold needs an LDS
// round trip of its own: one hit = one batch of reads, one batch of writes.
// NC: 64-entry columns of a list (KP <= 64 * NC).
// thr_w (optional): the wave's 32 thresholds in LDS, kept equal to the lists' worst entries for
// kernels that do not hold thr[] in registers between tiles.
template <int NC>
GGNN_DEV void bf_insert_hits_n(const float (&dd)[16], float (&thr)[16], uint32_t row0,
                               float* wave_d, int* wave_id, uint32_t KP, int h, float* thr_w)
{
  const int lane = threadIdx.x & 63;
  const int last = static_cast<int>(KP) - 1;  // entry KP-1: lane last & 63 of column last >> 6
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    unsigned long long m = __ballot(dd[r] < thr[r]);
    while (m) {
      const int l = __ffsll(static_cast<long long>(m)) - 1;
      m &= m - 1;
      const float dl = rdlanef(dd[r], l);
      // (an earlier candidate of this register row may have lowered the threshold since the ballot)
      if (!(dl < rdlanef(thr[r], l)))
        continue;
      const int hh = l >> 5;
      const int qi = (r & 3) + 8 * (r >> 2) + 4 * hh;
      float* Ld = wave_d + qi * KP;
      int* Li = wave_id + qi * KP;
      const int id = static_cast<int>(row0 + (l & 31));
      // lane owns entries k = c*64 + lane; stable insert: everything <= dl stays, the first larger
      // entry becomes dl, the rest shift by one
      // (reads are unconditional -- lanes past the list's end, and lane 0's "previous" entry, read
      // neighbouring LDS words that are never used: a predicated read is an exec-mask region of
      // its own, and this path is a serial chain whose length is what an insertion costs)
      float cur[NC], prev[NC];
      int previ[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int k = c * kWave + lane;
        cur[c] = Ld[k];
        prev[c] = Ld[k - 1];
        previ[c] = Li[k - 1];
      }
      prev[0] = lane == 0 ? -inf_f() : prev[0];  // entry 0 has no predecessor: dl goes there
      // one wave: the loads above are issued for all lanes before the stores below (LDS
      // operations of a wave execute in order); only the compiler must not reorder them
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int k = c * kWave + lane;
        if (k < (int)KP && dl < cur[c]) {
          const bool first = !(dl < prev[c]);  // previous entry stays: insert here
          Ld[k] = first ? dl : prev[c];
          Li[k] = first ? id : previ[c];
        }
      }
      __builtin_amdgcn_wave_barrier();
      // dl went in ahead of the old worst entry, which dropped out: the list now ends in the
      // larger of dl and the old entry KP-2 (the `prev` of the lane that owns entry KP-1)
      float p_last = rdlanef(prev[0], last & 63);
      if constexpr (NC > 1) {
        const float p1 = rdlanef(prev[NC - 1], last & 63);
        p_last = last >= kWave ? p1 : p_last;
      }
#ifdef GGNN_BF_PHASE
      const float worst = fminf(fmaxf(dl, p_last), rdlanef(thr[r], l));
#else
      const float worst = fmaxf(dl, p_last);
#endif
      if (h == hh)
        thr[r] = worst;
    }
  }
  // the LDS copy of the thresholds: lanes 0 and 32 hold the 16 of their half-wave's query rows
  if (thr_w && (lane & 31) == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(thr_w + 8 * g + 4 * h) =
          make_float4(thr[4 * g + 0], thr[4 * g + 1], thr[4 * g + 2], thr[4 * g + 3]);
  }
}
GGNN_DEV void bf_insert_hits(const float (&dd)[16], float (&thr)[16], uint32_t row0, float* wave_d,
                             int* wave_id, uint32_t KP, int h, float* thr_w = nullptr)
{
  static_assert(kBfMaxKP <= 2 * kWave, "lists have at most two 64-entry columns");
  if (KP <= static_cast<uint32_t>(kWave))
    bf_insert_hits_n<1>(dd, thr, row0, wave_d, wave_id, KP, h, thr_w);
  else
    bf_insert_hits_n<2>(dd, thr, row0, wave_d, wave_id, KP, h, thr_w);
}

// insertions of a tile for a kernel whose thresholds live in LDS (thr_w: the wave's 32)
GGNN_DEV void bf_test_and_insert(const float (&dd)[16], float* thr_w, uint32_t row0, float* wave_d,
                                 int* wave_id, uint32_t KP, int h)
{
  float thr[16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 t = *reinterpret_cast<const float4*>(thr_w + 8 * g + 4 * h);
    thr[4 * g + 0] = t.x;
    thr[4 * g + 1] = t.y;
    thr[4 * g + 2] = t.z;
    thr[4 * g + 3] = t.w;
  }
  bf_insert_hits(dd, thr, row0, wave_d, wave_id, KP, h, thr_w);
}

// Query operands of the chunked kernel come from a PACKED copy of the query set
// (pack_query_kernel): [query block][wave][chunk][piece t][lane] float4, i.e. the sixteen bytes lane
// (j, h) needs for piece t of a chunk -- columns 128 c + 64 h + 4 t .. + 3 of query row j -- sit at
// lane * 16 of a contiguous 1 KB block, so one load instruction of a wave is one contiguous
// kilobyte.  (Read from the row-major query set, the same instruction touched 64 different 128-byte
// lines for 16 bytes each -- eight times the bytes through the L1 / L2 path, and the operand
// stream is twice the size of the base tiles to begin with.)  Padding queries and columns past D
// are zero in the copy.  Buffer loads: the piece index goes into the scalar offset, the lane into
// the vector offset, no address arithmetic per load.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct QueryWindow {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t lane_bytes;
  GGNN_DEV void open(const float* packed, uint32_t nch, uint32_t qblock, uint32_t wave,
                     uint32_t lane)
  {
    const uint32_t wave_bytes = nch * 16u * 64u * 16u;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(packed) + (static_cast<size_t>(qblock) * 4 + wave) * (wave_bytes / 4), 0,
        wave_bytes, 0x00020000);
    lane_bytes = lane * 16u;
  }
  // four operands of piece t of the chunk that starts at column col_h - 64 h (a multiple of 128)
  GGNN_DEV float4 piece(uint32_t col_h, int t) const
  {
    const uint32_t c = col_h >> 7;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_bytes,
                                                          (c * 16u + static_cast<uint32_t>(t)) * 1024u, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                       __uint_as_float(v.w));
  }
};

// the packed copy (see QueryWindow); `query` is what the tile kernel works on (the shifted copy
// for float32 squared L2), out has query blocks * 4 * nch * 16 * 64 float4
template <typename BaseT>
__global__ void __launch_bounds__(256) pack_query_kernel(const BaseT* query, uint32_t Nq, uint32_t D,
                                                        uint32_t nch, uint64_t n4, float4* out)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4;
       i += static_cast<uint64_t>(gridDim.x) * 256) {
    const uint32_t lane = static_cast<uint32_t>(i & 63);
    uint64_t r = i >> 6;
    const uint32_t t = static_cast<uint32_t>(r & 15);
    r >>= 4;
    const uint32_t c = static_cast<uint32_t>(r % nch);
    r /= nch;
    const uint32_t w = static_cast<uint32_t>(r & 3);
    const uint64_t q = (r >> 2) * kBfQueriesPerBlock + w * 32 + (lane & 31);
    const uint32_t col = c * 128 + (lane >> 5) * 64 + 4 * t;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < Nq && col < D) {  // (D is a multiple of 4 / 16: all four columns exist)
      if constexpr (std::is_same<BaseT, float>::value)
        v = *reinterpret_cast<const float4*>(query + q * D + col);
      else {
        const uint32_t b = *reinterpret_cast<const uint32_t*>(query + q * D + col);
        v = make_float4(static_cast<float>(b & 0xffu), static_cast<float>((b >> 8) & 0xffu),
                        static_cast<float>((b >> 16) & 0xffu), static_cast<float>(b >> 24));
      }
    }
    out[i] = v;
  }
}

// 4*NU MFMA steps of one (tile, chunk) pair: acc += aq x B with B read from the LDS tile row bt.
// PREFETCH: every group of four query operands is reloaded for the next chunk (columns from
// next_col on) right after its MFMAs have been issued.  The scheduling recipe at the end pins the
// software pipeline -- B operand of step u+1 from LDS, the four MFMAs of step u, the query piece u
// of the next chunk -- so that the prefetched values reuse the operand registers they replace
// (hoisting all sixteen loads needs 64 more registers: 55 spills at two waves per SIMD).
template <int NU, bool PREFETCH>
GGNN_DEV void mfma_chain(f32x16& acc, float (&aq)[64], const float* bt, const QueryWindow& qw,
                         uint32_t next_col)
{
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const float4 bv = *reinterpret_cast<const float4*>(bt + 4 * u);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 0], bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 1], bv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 2], bv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 3], bv.w, acc, 0, 0, 0);
    if (PREFETCH) {
      const float4 nq = qw.piece(next_col, u);
      aq[4 * u + 0] = nq.x;
      aq[4 * u + 1] = nq.y;
      aq[4 * u + 2] = nq.z;
      aq[4 * u + 3] = nq.w;
    }
  }
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read (B operand of step 0)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (u + 1 < NU)
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read of step u+1
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    // the four MFMAs of step u
    if (PREFETCH)
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // query piece u of the next chunk
  }
}

// distance of one accumulator entry (expanded form; invalid base rows give +inf)
template <int MODE>
GGNN_DEV float bf_expand(float dot, float qn, float bn, bool jvalid)
{
  if (MODE == kL2)
    return fmaf(-2.0f, dot, qn + bn);  // bn = +inf for rows past the end
  const float norm_sqr = qn * bn;
  const float d = (norm_sqr > 0.0f) ? fabsf(1.0f - dot / sqrtf(norm_sqr)) : 1.0f;
  return jvalid ? d : inf_f();
}

// T = base tiles per group whose accumulators stay in registers while the K chunks stream by
// (T = 1 for D <= 128: a single chunk, the query tile is loaded once per workgroup).
// NU (T = 1 only): float4 steps per half row, Dh = 4*NU -- a compile-time trip count keeps the
// MFMA chain of a tile in ONE basic block so that the LDS reads of the B operand are scheduled
// ahead of the MFMAs that consume them.
// KPC: the list length KP as a compile-time constant (0: a.KP at run time) -- every list address
// of the epilogue and of the insertions is then a constant offset from one base.
template <typename BaseT, int MODE, int T, int NU, int KPC = 0>
__global__ void __launch_bounds__(256)
    __attribute__((amdgpu_waves_per_eu(T == 1 ? 3 : T <= 3 ? 2 : 1))) bf_mfma_kernel(const BfMfmaArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  // the two tile buffers are addressed as lds_f + offset (never through a pointer array: a select
  // between pointers decays to generic addressing, i.e. flat loads that wait on vmcnt(0) and with
  // it on the prefetch of the next tile)
  // chunk geometry from the template parameter: the staging index arithmetic (idx / cpr, idx % cpr)
  // is then shifts and masks instead of run-time integer division
  constexpr uint32_t Dh = 4 * NU, CW = 2 * Dh, DP = CW + ((CW % 8 == 0) ? 4 : 8);
  const uint32_t tile_floats = kBfTileRows * DP;
  float* list_d = lds_f + 2 * tile_floats;
  int* list_id = reinterpret_cast<int*>(list_d + kBfQueriesPerBlock * a.KP);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: keep it in an SGPR
  const int j = lane & 31, h = lane >> 5;
  const BaseT* base = static_cast<const BaseT*>(a.base);
  const BaseT* query = static_cast<const BaseT*>(a.query);
  const uint32_t KP = KPC ? static_cast<uint32_t>(KPC) : a.KP;
  const uint32_t nch = (a.D + CW - 1) / CW;
  float aq[64];
  // rows of the accumulator registers: i(r) = (r&3) + 8*(r>>2) + 4*h
  // squared norm standing in for rows past the end of the slice: +inf distance for L2
  const float bn_pad = (MODE == kL2) ? inf_f() : 0.f;
  TileStage<BaseT> stage;
  // behind the candidate lists: the shift vector (a.DM floats), and for the single-chunk kernel
  // the 128 query norms and the 128 thresholds = worst list entries (nothing beats -inf:
  // padding queries).  In LDS, not in registers: 48 registers less is what lets a THIRD
  // workgroup share the CU, and three waves per SIMD cover each other's insertions and barriers.
  float* mean_lds = reinterpret_cast<float*>(list_id + kBfQueriesPerBlock * a.KP);
  float* qn_lds = mean_lds + a.DM;
  float* thr_lds = qn_lds + kBfQueriesPerBlock;
  // chunked kernel: row norms of the tiles of the current and the next group (the next group's
  // first tile is staged before the current group's epilogue), [group parity][tile][row]; the
  // epilogue reads them from here -- a global load there is a round trip of its own per group
  float* bn_grp = thr_lds;  // (the single-chunk kernel keeps its thresholds in these words)
  for (uint32_t i = tid; i < a.DM; i += 256)
    mean_lds[i] = (a.mean && i < a.D) ? a.mean[i] : 0.f;

  // Work of this workgroup.  a.equal_ranges: the (query block, unit) pairs -- a unit = T tiles, one
  // accumulator group -- form ONE sequence of Nq/128 * N/(32 T) units that is cut into gridDim.x
  // equal ranges (a.tiles_per_block), so every resident workgroup gets the same amount of work
  // whatever Nq is; a range that crosses a query-block boundary is processed as two (or more)
  // segments, each with fresh lists written to its own part.  Otherwise (the i8 kernels' launch geometry):
  // query block blockIdx.x, base slice blockIdx.y.
  constexpr uint32_t kUnitRows = kBfTileRows * T;
  uint64_t work = 0, work_end = 0;
  if (a.equal_ranges) {
    work = static_cast<uint64_t>(blockIdx.x) * a.tiles_per_block;
    work_end = min(a.total_tiles, work + a.tiles_per_block);
  }
  do {
  uint32_t qblock = blockIdx.x, begin = 0, end = 0, part = blockIdx.y;
  if (a.equal_ranges) {
    if (work >= work_end)
      break;
    qblock = static_cast<uint32_t>(work / a.tiles_per_q);
    const uint64_t q_first = static_cast<uint64_t>(qblock) * a.tiles_per_q;
    const uint32_t t0 = static_cast<uint32_t>(work - q_first);
    const uint32_t cnt =
        static_cast<uint32_t>(min(static_cast<uint64_t>(a.tiles_per_q - t0), work_end - work));
    begin = t0 * kUnitRows;
    end = min(a.N_base, (t0 + cnt) * kUnitRows);
    // parts of a query block are numbered from the first workgroup that touches it
    part = blockIdx.x - static_cast<uint32_t>(q_first / a.tiles_per_block);
    work += cnt;
  }
  else {
    begin = blockIdx.y * a.rows_per_slice;
    end = min(a.N_base, begin + a.rows_per_slice);
  }
  const uint32_t qbase = (qblock * 4 + wave) * 32;

  // candidate lists of this wave's 32 queries; padding queries get -inf so that nothing ever
  // beats their threshold (their lists are never written out)
  for (uint32_t i = lane; i < 32 * KP; i += 64) {
    list_d[wave * 32 * KP + i] = (qbase + i / KP < a.Nq) ? inf_f() : -inf_f();
    list_id[wave * 32 * KP + i] = kEmptyKey;
  }
  const bool qvalid = qbase + j < a.Nq;
  const BaseT* qrow = query + static_cast<size_t>(qvalid ? qbase + j : 0) * a.D;
  {
    if (lane < 32) {  // this wave's entries only: nothing to order against the other waves
      const uint32_t qi = qbase + lane;
      qn_lds[wave * 32 + lane] = qi < a.Nq ? a.qnorm[qi] : 0.f;
      // Thresholds in LDS exist in the single-chunk kernel only.  In the chunked kernel these words
      // are bn_grp (row norms every wave reads in its epilogue), and a wave that enters the next
      // segment of its range early would overwrite them under a slower wave still in the previous
      // segment's epilogue -- no barrier lies between the two (the chunked kernel takes its
      // thresholds from the lists).
      if constexpr (T == 1) {
        float t0v = qi < a.Nq ? inf_f() : -inf_f();
#ifdef GGNN_BF_PHASE
        if (g_bf_dbg_bound && qi < a.Nq)
          t0v = g_bf_dbg_bound[qi];
#endif
        thr_lds[wave * 32 + lane] = t0v;
      }
    }
  }
  __syncthreads();
  const uint32_t ntiles = (end > begin) ? (end - begin + kBfTileRows - 1) / kBfTileRows : 0;
  if (ntiles) {
    if constexpr (T > 1) {
      stage.load(base, a.D, begin, end, 0, CW, a.bnorm, bn_pad);
      stage.store_shifted(lds_f, DP, CW, mean_lds, 0);
      if (tid < kBfTileRows)
        bn_grp[tid] = stage.norm();
    }
    else {
      stage.load(base, a.D, begin, end, 0, CW, a.bnorm, bn_pad);
      stage.store_shifted(lds_f, DP, CW, mean_lds, 0);
    }
  }
  __syncthreads();

  if constexpr (T == 1) {
    // One chunk per row (D <= 128).  The test of a tile follows its own MFMA chain (no software
    // pipeline over tiles: a second accumulator set would cost the third wave per SIMD, and it
    // is the other two waves of the SIMD that keep the matrix pipe busy meanwhile).
    load_query_chunk(aq, qrow, qvalid, a.D, Dh, h * Dh);
    float* wave_d = list_d + wave * 32 * KP;
    int* wave_id = list_id + wave * 32 * KP;
    const float* qn_w = qn_lds + wave * 32;
    float* thr_w = thr_lds + wave * 32;
    // Everything loaded so far (the query chunk) must have arrived BEFORE the loop: a first use
    // inside the loop makes the compiler wait for vmcnt(0) in every iteration, which also waits
    // for the prefetch of the next tile issued just before (s_waitcnt vmcnt(0)).
    __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef GGNN_BF_PHASE
    unsigned long long bf_ph[5] = {0, 0, 0, 0, 0};
    unsigned long long bf_t = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t tt = 0; tt < ntiles; ++tt) {
      const uint32_t row0 = begin + tt * kBfTileRows;
      const bool jvalid = row0 + j < end;
      // norm of this lane's tile row, staged next to the tile (an LDS read: a global load here
      // would make the compiler wait for vmcnt(0), i.e. for the prefetch of the next tile)
      const float bn = lds_f[(tt & 1) * tile_floats + j * DP + CW];
      const bool has_next = tt + 1 < ntiles;
      if (has_next)
        stage.load(base, a.D, row0 + kBfTileRows, end, 0, CW, a.bnorm, bn_pad);
      f32x16 acc = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* bt = lds_f + (tt & 1) * tile_floats + j * DP + h * Dh;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const float4 bv = *reinterpret_cast<const float4*>(bt + 4 * u);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 0], bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 1], bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 2], bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[4 * u + 3], bv.w, acc, 0, 0, 0);
      }
      float dd[16];
      unsigned long long any = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 qn4 = *reinterpret_cast<const float4*>(qn_w + 8 * g + 4 * h);
        const float4 th4 = *reinterpret_cast<const float4*>(thr_w + 8 * g + 4 * h);
        dd[4 * g + 0] = bf_expand<MODE>(acc[4 * g + 0], qn4.x, bn, jvalid);
        dd[4 * g + 1] = bf_expand<MODE>(acc[4 * g + 1], qn4.y, bn, jvalid);
        dd[4 * g + 2] = bf_expand<MODE>(acc[4 * g + 2], qn4.z, bn, jvalid);
        dd[4 * g + 3] = bf_expand<MODE>(acc[4 * g + 3], qn4.w, bn, jvalid);
        any |= __ballot(dd[4 * g + 0] < th4.x) | __ballot(dd[4 * g + 1] < th4.y) |
               __ballot(dd[4 * g + 2] < th4.z) | __ballot(dd[4 * g + 3] < th4.w);
      }
      GGNN_BF_TICK(0);
      if (any)
        bf_test_and_insert(dd, thr_w, row0, wave_d, wave_id, KP, h);
      GGNN_BF_TICK(1);
      if (has_next)
        stage.store_shifted(lds_f + ((tt + 1) & 1) * tile_floats, DP, CW, mean_lds,
                                            0);
      GGNN_BF_TICK(2);
      __syncthreads();
      GGNN_BF_TICK(3);
    }
#ifdef GGNN_BF_PHASE
    if (lane == 0) {
      for (int i = 0; i < 4; ++i)
        atomicAdd(&g_bf_phase[i], bf_ph[i]);
      atomicAdd(&g_bf_phase[4], static_cast<unsigned long long>(ntiles));
    }
#endif
  }
  else {
  // this wave's block of the packed query copy (a.query_packed)
  QueryWindow qw;
  qw.open(a.query_packed, nch, qblock, wave, lane);
  uint32_t p = 0;  // (tile, chunk) pairs processed: buffer p&1 holds the current pair
#ifdef GGNN_BF_PHASE
  unsigned long long bf_ph[5] = {0, 0, 0, 0, 0};
  unsigned long long bf_t = __builtin_amdgcn_s_memtime();
#endif
  // The first query chunk is loaded here, every later one during the last tile of the chunk
  // before it (mfma_chain<PREFETCH>).  NOT inside the loops under `if (first)`: the operand
  // registers would then be a phi of two load sites, which the compiler resolves with 64 copies and
  // a wait for vmcnt(0) at the top of every (group, chunk) iteration -- right behind the sixteen
  // loads the prefetch had just issued.
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 v = qw.piece(h * Dh, t);
    aq[4 * t + 0] = v.x;
    aq[4 * t + 1] = v.y;
    aq[4 * t + 2] = v.z;
    aq[4 * t + 3] = v.w;
  }
  for (uint32_t g0 = 0; g0 < ntiles; g0 += T) {
    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
      acc[t] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (uint32_t c = 0; c < nch; ++c) {
      const uint32_t next_col = ((c + 1 < nch) ? (c + 1) * CW : 0u) + h * Dh;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const uint32_t tt = g0 + t;
        // A group always runs its T tiles: one that lies past the end of the range stages
        // clamped rows with the padding norm (nothing is inserted from it, the epilogue skips it).
        // That makes "last tile of the chunk" a compile-time property (t == T - 1), so there is ONE
        // prefetch site writing the operand registers -- two sites in different copies of the
        // chain are a phi again (64 copies behind a wait for vmcnt(0) per iteration).
        // the pair after this one: next tile of the group, else next chunk, else next group
        bool has_next = true;
        const bool last_of_chunk = t + 1 == T;
        uint32_t n_tile = tt + 1, n_chunk = c;
        if (last_of_chunk) {
          if (c + 1 < nch) {
            n_tile = g0;
            n_chunk = c + 1;
          }
          else if (g0 + T < ntiles) {
            n_tile = g0 + T;
            n_chunk = 0;
          }
          else
            has_next = false;
        }
        if (has_next)
          stage.load(base, a.D, begin + n_tile * kBfTileRows, end, n_chunk * CW, CW, a.bnorm,
                     bn_pad);

        // S += Q_chunk x B_chunk^T for this wave's 32 queries against the 32 tile rows.  During
        // the last tile of a chunk every group of four query operands is reloaded for the NEXT
        // chunk as soon as its MFMAs have been issued: the loads travel while the rest of the
        // chain runs, instead of stalling the first MFMA of the next chunk.
        const float* bt = lds_f + (p & 1) * tile_floats + j * DP + h * Dh;
        // (each copy of the chain is ONE basic block, so the LDS reads of the B operand are
        // scheduled ahead of the MFMAs that consume them)
        if (t + 1 == T)  // constant after unrolling
          mfma_chain<NU, true>(acc[t], aq, bt, qw, next_col);
        else
          mfma_chain<NU, false>(acc[t], aq, bt, qw, next_col);
        GGNN_BF_TICK(0);
        if (has_next) {
          stage.store_shifted(lds_f + ((p + 1) & 1) * tile_floats, DP, CW, mean_lds,
                              n_chunk * CW);
          if (n_chunk == 0 && tid < kBfTileRows)
            bn_grp[((((n_tile / T) & 1) * T) + n_tile % T) * kBfTileRows + tid] = stage.norm();
        }
        GGNN_BF_TICK(2);
        __syncthreads();
        GGNN_BF_TICK(3);
        ++p;
      }
    }

    // epilogue: distances of column j (base row row0+j) to the lane's 16 query rows.  The
    // thresholds (worst list entry of each of the lane's query rows) live in registers and change
    // only after an insertion, so the common case is 16 x (add, fma, compare) and one branch.
    // (thresholds and query norms are fetched here -- from LDS -- not held across the MFMA loop:
    // 32 registers less, which is what lets a second workgroup share the CU)
    float* wave_d2 = list_d + wave * 32 * KP;
    float thr2[16], qn2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t ql = (r & 3) + 8 * (r >> 2) + 4 * h;
      thr2[r] = wave_d2[ql * KP + KP - 1];   // -inf for padding queries (list initialisation)
      qn2[r] = qn_lds[wave * 32 + ql];
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const uint32_t tt = g0 + t;
      if (tt >= ntiles)
        break;
      const uint32_t row0 = begin + tt * kBfTileRows;
      const bool jvalid = row0 + j < end;
      // rows past the end: +inf norm -> +inf distance (L2); handled explicitly for cosine
      const float bn = bn_grp[(((g0 / T) & 1) * T + t) * kBfTileRows + j];
      float dd[16];
      unsigned long long any = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dd[r] = bf_expand<MODE>(acc[t][r], qn2[r], bn, jvalid);
        any |= __ballot(dd[r] < thr2[r]);
      }
      if (!any)
        continue;
      bf_insert_hits(dd, thr2, row0, wave_d2, list_id + wave * 32 * KP, KP, h);
    }
    GGNN_BF_TICK(1);
  }
#ifdef GGNN_BF_PHASE
  if (lane == 0) {
    for (int i = 0; i < 4; ++i)
      atomicAdd(&g_bf_phase[i], bf_ph[i]);
    atomicAdd(&g_bf_phase[4], static_cast<unsigned long long>(p));
  }
#endif
  }  // T > 1

  // partial results of this base slice / segment
  for (uint32_t i = lane; i < 32 * KP; i += 64) {
    const uint32_t qi = qbase + i / KP;
    if (qi < a.Nq) {
      const size_t o = (static_cast<size_t>(part) * a.Nq + qi) * KP + i % KP;
      a.part_ids[o] = list_id[wave * 32 * KP + i];
      a.part_dists[o] = list_d[wave * 32 * KP + i];
    }
  }
  } while (a.equal_ranges);
}

// ---- 2b. uint8 rows, squared L2: the contraction on v_mfma_i32_32x32x32_i8 ------------------------
// Bytes are shifted to signed (x ^ 0x80 = x - 128), which leaves (q - b)^2 unchanged, so
// d = |q'|^2 + |b'|^2 - 2 q'.b' with the integer dot product of the shifted bytes: exact (all terms
// < 2^24 for D <= 128).  Lane (j, h) owns bytes [32m + 16h, 32m + 16h + 16) of row j for MFMA m
// (any assignment of k to lanes works as long as A and B agree).  A tile of 32 rows is 32 x 128
// bytes (row stride 144: conflict-free ds_read_b128, the spare bytes hold the row's norm); four
// tiles are staged per barrier.  With 4 MFMAs per tile the kernel is bound by staging and the
// candidate test, not by the matrix pipe.

constexpr int kBfI8Tiles = 4;  // 32-row tiles per staged block (one barrier per 128 base rows)

template <int NM>  // MFMAs per tile = ceil(D / 32), D <= 128
__global__ void __launch_bounds__(256) bf_mfma_i8_kernel(const BfMfmaArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  uint8_t* lds_b = reinterpret_cast<uint8_t*>(lds_f);
  constexpr int TT = kBfI8Tiles;
  constexpr uint32_t stage_rows = TT * kBfTileRows;
  constexpr uint32_t stage_bytes = stage_rows * kBfI8RowStride;
  float* list_d = lds_f + 2 * stage_bytes / 4;
  int* list_id = reinterpret_cast<int*>(list_d + kBfQueriesPerBlock * a.KP);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const uint8_t* base = static_cast<const uint8_t*>(a.base);
  const uint8_t* query = static_cast<const uint8_t*>(a.query);
  const uint32_t qbase = (blockIdx.x * 4 + wave) * 32;
  const uint32_t begin = blockIdx.y * a.rows_per_slice;
  const uint32_t end = min(a.N_base, begin + a.rows_per_slice);
  const uint32_t KP = a.KP;

  for (uint32_t i = lane; i < 32 * KP; i += 64) {
    list_d[wave * 32 * KP + i] = (qbase + i / KP < a.Nq) ? inf_f() : -inf_f();
    list_id[wave * 32 * KP + i] = kEmptyKey;
  }

  // A operand: the lane's 16-byte pieces of query row j, shifted to signed
  const bool qvalid = qbase + j < a.Nq;
  const uint8_t* qrow = query + static_cast<size_t>(qvalid ? qbase + j : 0) * a.D;
  i32x4 aq[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const uint32_t col = 32 * m + 16 * h;
    aq[m] = i32x4{0, 0, 0, 0};
    if (qvalid && col < a.D) {
      const uint4 v = *reinterpret_cast<const uint4*>(qrow + col);
      aq[m] = i32x4{static_cast<int>(v.x ^ 0x80808080u), static_cast<int>(v.y ^ 0x80808080u),
                    static_cast<int>(v.z ^ 0x80808080u), static_cast<int>(v.w ^ 0x80808080u)};
    }
  }
  float qn[16], thr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t qi = qbase + (r & 3) + 8 * (r >> 2) + 4 * h;
    qn[r] = qi < a.Nq ? a.qnorm[qi] : 0.f;
    thr[r] = qi < a.Nq ? inf_f() : -inf_f();
  }

  // staging: a block of 128 rows is 1024 pieces of 16 bytes, 4 per thread; threads 0..127 carry
  // the row norms (+inf for rows past the end: +inf distance)
  uint4 sv[TT];
  float sbn;
  auto stage_load = [&](uint32_t row0) {
#pragma unroll
    for (int e = 0; e < TT; ++e) {
      const uint32_t p = tid + 256 * e, srow = p >> 3, scol = 16 * (p & 7);
      sv[e] = make_uint4(0u, 0u, 0u, 0u);
      if (row0 + srow < end && scol < a.D) {
        const uint4 v = *reinterpret_cast<const uint4*>(
            base + static_cast<size_t>(row0 + srow) * a.D + scol);
        sv[e] = make_uint4(v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u,
                           v.w ^ 0x80808080u);
      }
    }
    sbn = inf_f();
    if (tid < (int)stage_rows && row0 + tid < end)
      sbn = a.bnorm[row0 + tid];
  };
  auto stage_store = [&](uint32_t buf) {
    uint8_t* t = lds_b + buf * stage_bytes;
#pragma unroll
    for (int e = 0; e < TT; ++e) {
      const uint32_t p = tid + 256 * e, srow = p >> 3, scol = 16 * (p & 7);
      *reinterpret_cast<uint4*>(t + srow * kBfI8RowStride + scol) = sv[e];
    }
    if (tid < (int)stage_rows)
      *reinterpret_cast<float*>(t + tid * kBfI8RowStride + 128) = sbn;
  };

  const uint32_t nstages = (end > begin) ? (end - begin + stage_rows - 1) / stage_rows : 0;
  if (nstages) {
    stage_load(begin);
    stage_store(0);
  }
  __syncthreads();

  float* wave_d = list_d + wave * 32 * KP;
  int* wave_id = list_id + wave * 32 * KP;
  __builtin_amdgcn_s_waitcnt(0x0F70);  // see bf_mfma_kernel: no first use of a load inside the loop
  for (uint32_t st = 0; st < nstages; ++st) {
    const uint32_t row0 = begin + st * stage_rows;
    const uint8_t* blk = lds_b + (st & 1) * stage_bytes;
    const bool has_next = st + 1 < nstages;
    if (has_next)
      stage_load(row0 + stage_rows);
    i32x16 acc[TT];
    float bn[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const uint8_t* trow = blk + (t * kBfTileRows + j) * kBfI8RowStride;
      bn[t] = *reinterpret_cast<const float*>(trow + 128);
      acc[t] = i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const i32x4 b = *reinterpret_cast<const i32x4*>(trow + 32 * m + 16 * h);
        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[m], b, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (row0 + t * kBfTileRows >= end)
        break;  // uniform
      float dd[16];
      unsigned long long any = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dd[r] = bf_expand<kL2>(static_cast<float>(acc[t][r]), qn[r], bn[t], true);
        any |= __ballot(dd[r] < thr[r]);
      }
      if (any)
        bf_insert_hits(dd, thr, row0 + t * kBfTileRows, wave_d, wave_id, KP, h);
    }
    if (has_next)
      stage_store((st + 1) & 1);
    __syncthreads();
  }

  for (uint32_t i = lane; i < 32 * KP; i += 64) {
    const uint32_t qi = qbase + i / KP;
    if (qi < a.Nq) {
      const size_t o = (static_cast<size_t>(blockIdx.y) * a.Nq + qi) * KP + i % KP;
      a.part_ids[o] = list_id[wave * 32 * KP + i];
      a.part_dists[o] = list_d[wave * 32 * KP + i];
    }
  }
}

// ---- 3. exact re-rank ----------------------------------------------------------------------------
struct BfRerankArgs {
  const void* base;
  const void* query;
  const int32_t* part_ids;
  const float* part_dists;   // expanded-form distances of the candidates (certificate)
  const float* qnorm;        // squared norms the tile kernel used (shifted rows)
  const uint32_t* bnorm_max; // float bits of the largest base-row norm
  uint32_t* rescan_count;    // queries that could not be certified ...
  uint32_t* rescan_list;     // ... and their indices (answered by the scan kernel afterwards)
  int32_t* ids;
  float* dists;
  uint32_t D, Nq, K, KP, slices, cap;
  int exact_arith;           // i8 path: integer arithmetic, nothing to certify
};

template <typename BaseT, int LPR, int NCH, int MODE>
__global__ void __launch_bounds__(kWave) bf_rerank_kernel(const BfRerankArgs a)
{
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  const WaveLds lds(lds_raw, 2 * a.cap);
  float* all_d = reinterpret_cast<float*>(lds.known);
  int* all_id = lds.known + a.cap;
  const int lane = threadIdx.x;
  const uint32_t n = block_linear_index();
  if (n >= a.Nq)
    return;
  const BaseT* base = static_cast<const BaseT*>(a.base);
  DistEngine<BaseT, LPR, NCH> de;
  de.template load_query<MODE>(base, a.D, static_cast<const BaseT*>(a.query) + static_cast<size_t>(n) * a.D);

  const uint32_t total = a.slices * a.KP;
  uint32_t count = 0;
  for (uint32_t c0 = 0; c0 < total; c0 += kKBlock) {
    const uint32_t c = c0 + lane;
    int cand = kEmptyKey;
    if (lane < (int)kKBlock && c < total)
      cand = a.part_ids[(static_cast<size_t>(c / a.KP) * a.Nq + n) * a.KP + c % a.KP];
    const unsigned long long surv = __ballot(cand != kEmptyKey);
    const int nsurv = __popcll(surv);
    if (!nsurv)
      continue;
    __syncthreads();
    if (cand != kEmptyKey)
      lds.ckeys[__popcll(surv & ((1ull << lane) - 1ull))] = cand;
    __syncthreads();
    compute_distances<MODE>(de, lds, nsurv, nullptr);
    __syncthreads();
    if (lane < nsurv) {
      all_d[count + lane] = lds.cd0[lane];
      all_id[count + lane] = lds.ckeys[lane];
    }
    count += nsurv;
  }
  __syncthreads();
  int32_t* out_i = a.ids + static_cast<size_t>(n) * a.K;
  float* out_d = a.dists + static_cast<size_t>(n) * a.K;
  for (uint32_t i = lane; i < count; i += kWave) {
    const float d = all_d[i];
    const int id = all_id[i];
    uint32_t rank = 0;
    for (uint32_t jj = 0; jj < count; ++jj) {
      const float dj = all_d[jj];
      rank += (dj < d) || (dj == d && all_id[jj] < id);
    }
    if (rank < a.K) {
      out_i[rank] = id;
      out_d[rank] = d;
    }
    if (rank + 1 == a.K)
      lds.cd0[0] = d;  // K-th distance, for the certificate below
  }
  for (uint32_t k = count + lane; k < a.K; k += kWave) {
    out_i[k] = kEmptyKey;
    out_d[k] = inf_f();
  }

  // ---- certificate (see the top of this file) -------------------------------------------------
  if (a.exact_arith || count < a.K)
    return;  // exact integers, or every row of the base is a candidate
  // w: smallest "worst kept" expanded-form distance over the slices (+inf: list not full)
  float w = inf_f();
  for (uint32_t s = lane; s < a.slices; s += kWave)
    w = fminf(w, a.part_dists[(static_cast<size_t>(s) * a.Nq + n) * a.KP + a.KP - 1]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1)
    w = fminf(w, __shfl_xor(w, o));
  __syncthreads();  // the K-th distance was left in LDS by the lane that ranked it
  const float d_k = lds.cd0[0];
  constexpr float u = 5.9604645e-8f;  // 2^-24
  const float Df = static_cast<float>(a.D);
  bool ok;
  if (MODE == kL2) {
    const float norms = a.qnorm[n] + __uint_as_float(*a.bnorm_max);
    // (+ an absolute term: products and partial sums in the denormal range -- coordinates around
    // 1e-19 and below -- are rounded to a multiple of 2^-149, or flushed, an error the relative
    // model does not see; (2D+8) 2^-125 covers D such roundings of either kind many times over)
    const float E =
        1.01f * (2.f * Df + 8.f) * u * norms + 4.f * u * w + (2.f * Df + 8.f) * 2.3509887e-38f;
    ok = (w - E) * (1.f - 1.01f * (Df + 3.f) * u) > d_k;
  }
  else {
    const float E = 2.02f * (2.f * Df + 8.f) * u;
    ok = w - E > d_k;
  }
  if (!(w < inf_f()))
    ok = true;  // no list is full: nothing was left out
  if (!ok && lane == 0)
    a.rescan_list[atomicAdd(a.rescan_count, 1u)] = n;
}

// ---- host ---------------------------------------------------------------------------------------
bool bf_mfma_supported(const BfLaunch& a)
{
  const uint32_t epc = a.dtype == GGNN_F32 ? 4 : 16;
  if (!(a.D % epc == 0 && a.k_query + 8 <= kBfMaxKP && a.Nq >= 256 && a.N_base >= 4096))
    return false;
  // tiles + candidate lists (+ the shift vector of the chunked kernel) must fit into 160 KB of LDS
  const size_t lists = 2ull * kBfQueriesPerBlock * (a.k_query + 8);
  const size_t tiles = 2ull * kBfTileRows * 132;
  const size_t shift = a.D > 128 ? (a.D + 127) / 128 * 128 + 3 * kBfQueriesPerBlock
                                 : 128 + 2 * kBfQueriesPerBlock;
  return (lists + tiles + shift) * sizeof(float) <= 160 * 1024;
}

size_t bf_rescan_tmp_entries(const BfLaunch& a, uint32_t* slices_out);
void launch_bf_rescan(const BfLaunch& a, const uint32_t* qlist, const uint32_t* qcount,
                      int32_t* tmp_ids, float* tmp_dists, hipStream_t stream);

void launch_bf_query_mfma(const BfLaunch& a, hipStream_t stream)
{
  check_vector_layout(a.base, a.D, a.dtype);
  check_vector_layout(a.query, a.D, a.dtype);
  // a few more candidates than K per slice: makes the certificate (top of this file) succeed for
  // all but near-degenerate queries; correctness does not depend on the value
  // uint8 + squared L2 with rows of up to 128 bytes: integer contraction on the i8 matrix path;
  // lists of up to 24 entries live in registers (bf_i8v2_kernel), longer ones in LDS
  // (bf_mfma_i8_kernel; hook BF_I8_V1 = 1 forces that one: A/B hook)
  const bool use_i8 = a.dtype == GGNN_U8 && a.measure == GGNN_EUCLIDEAN && a.D <= 128 &&
                      hook(kHookBfNoI8) == 0;
  const bool use_i8v2 = use_i8 && a.k_query <= 16 && hook(kHookBfI8V1) == 0;
  // (integer arithmetic needs no certificate margin: the register sets hold exactly K rounded up)
  const uint32_t KP = use_i8v2 ? (a.k_query <= 4 ? 4u : a.k_query <= 10 ? 10u : 16u)
                               : a.k_query + 8;
  // one chunk of 2*Dh columns when the row fits (D <= 128), otherwise chunks of 128 columns
  // half-row width: 64 when D > 128 (K streams in chunks), otherwise 32 / 48 / 64 so that the
  // single chunk covers the row (columns past D are zero in the tile and in the query operand)
  const uint32_t Dh = a.D > 128 ? 64 : (a.D <= 64 ? 32 : a.D <= 96 ? 48 : 64);
  const uint32_t DP = (2 * Dh) + (((2 * Dh) % 8 == 0) ? 4 : 8);  // odd number of 16-B slots/row
  const uint32_t queries_per_block = use_i8v2 ? 256u : static_cast<uint32_t>(kBfQueriesPerBlock);
  const uint32_t qblocks = (a.Nq + queries_per_block - 1) / queries_per_block;
  // one round of resident workgroups (2 per CU x 256 CUs at this register budget): a partial
  // second round costs more than the lower parallelism (measured: 790 blocks 39.8 ms, 474 blocks
  // 31.5 ms for 10k x 1M x 128)
  // (the single-chunk float kernel runs three workgroups per CU when their LDS fits)
  const bool single_chunk = !use_i8 && a.D <= 128;
  uint32_t resident = 512;
  if (single_chunk) {
    const size_t lds1 = (2 * kBfTileRows * DP + 2 * kBfQueriesPerBlock * KP + 2 * Dh +
                         2 * kBfQueriesPerBlock) * sizeof(float);
    resident = 256u * static_cast<uint32_t>(std::clamp<size_t>(160 * 1024 / lds1, 1, 3));
  }
  uint32_t slices = std::max(1u, std::min(32u, resident / std::max(1u, qblocks)));
  if (const int64_t hs = hook(kHookBfSlices); hs > 0)  // tuning hook
    slices = static_cast<uint32_t>(std::clamp<int64_t>(hs, 1, 64));
  uint32_t rows_per_slice = (a.N_base + slices - 1) / slices;
  rows_per_slice = (rows_per_slice + kBfTileRows - 1) / kBfTileRows * kBfTileRows;
  slices = (a.N_base + rows_per_slice - 1) / rows_per_slice;
  // float tile kernels: Nq/128 * N/(32 T) (query block, unit) pairs in one sequence, cut into
  // `resident` equal ranges -- every workgroup slot of the chip gets the same work whatever Nq is
  // (79 query blocks x 9 slices left 7 % of the slots empty).  A query block then has up to
  // ceil(units per query block / range) + 1 parts; ranges are at least 1/32 of a query block (the
  // lists of every part start empty, and what they accept before their thresholds settle is the
  // cost).  Hook BF_SLICES: ranges of 1/BF_SLICES of a query block.
  const bool equal_ranges = !use_i8;
  const int64_t tiles_hook = hook(kHookBfTiles);
  const uint32_t unit_tiles =
      a.D > 128 ? (tiles_hook == 4 ? 4u : tiles_hook == 2 ? 2u : 3u) : 1u;
  const uint32_t unit_rows = unit_tiles * kBfTileRows;
  const uint32_t tiles_per_q = (a.N_base + unit_rows - 1) / unit_rows;
  const uint64_t total_tiles = static_cast<uint64_t>(qblocks) * tiles_per_q;
  uint32_t tiles_per_block = 0, nblocks = 0;
  if (equal_ranges) {
    uint64_t per = (total_tiles + resident - 1) / resident;
    per = std::max<uint64_t>(per, (tiles_per_q + 31) / 32);
    if (const int64_t hs = hook(kHookBfSlices); hs > 0)
      per = (tiles_per_q + slices - 1) / slices;
    per = std::clamp<uint64_t>(per, 1, 0xffffffffu);
    tiles_per_block = static_cast<uint32_t>(per);
    nblocks = static_cast<uint32_t>((total_tiles + per - 1) / per);
    slices = (tiles_per_q + tiles_per_block - 1) / tiles_per_block + 1;  // parts per query
  }

  // float32 squared L2: rows are shifted by a column mean of the base (hook BF_NO_CENTER = 1: test
  // hook that leaves them unshifted so that offset data exercises the re-scan)
  const bool center = a.dtype == GGNN_F32 && a.measure == GGNN_EUCLIDEAN &&
                      hook(kHookBfNoCenter) == 0;

  // one scratch block: [norms N+Nq][mean D][mean partials 256 D][bn_max, rescan_count]
  // [rescan_list Nq][part ids][part dists][re-scan slices ids][re-scan slices dists]
  const size_t parts = static_cast<size_t>(slices) * a.Nq * KP;
  uint32_t rescan_slices = 0;
  const size_t rescan_entries = use_i8 ? 0 : bf_rescan_tmp_entries(a, &rescan_slices);
  // every block starts on a 16-byte boundary (float4 / int4 accesses)
  auto pad4 = [](size_t words) { return (words + 3) / 4 * 4; };
  const size_t n_norms = pad4(static_cast<size_t>(a.N_base) + a.Nq);
  const size_t n_mean = pad4(a.D);
  const size_t n_partial = center ? static_cast<size_t>(kBfMeanBlocks) * n_mean : 0;
  const size_t n_list = pad4(a.Nq);
  const size_t n_parts = pad4(parts);
  const size_t n_rescan = pad4(rescan_entries);
  const size_t n_qshift = center ? static_cast<size_t>(a.Nq) * a.D : 0;  // shifted query copy
  // exchange area of the i8 register-set kernel (bf_i8.hip "bound exchange")
  const size_t n_gthr = use_i8v2 ? pad4(bf_i8v2_exchange_ints(a.Nq, slices)) + pad4(a.Nq) : 0;
  // chunked float kernel: the query set once more, in operand order (QueryWindow)
  const uint32_t pack_chunks = (!use_i8 && a.D > 128) ? (a.D + 127) / 128 : 0;
  const size_t n_qpack = static_cast<size_t>(qblocks) * kBfQueriesPerBlock * pack_chunks * 128;
  const size_t words = n_norms + n_mean + n_partial + 4 + n_list + 2 * n_parts + 2 * n_rescan +
                       pad4(n_qshift) + n_gthr + n_qpack;
  float* scratch = static_cast<float*>(scratch_alloc(words * 4, stream));
  float* bnorm = scratch;
  float* qnorm = bnorm + a.N_base;
  float* mean = scratch + n_norms;
  float* partial = mean + n_mean;
  uint32_t* flags = reinterpret_cast<uint32_t*>(partial + n_partial);  // [0] bn_max [1] count
  uint32_t* rescan_list = flags + 4;
  int32_t* part_ids = reinterpret_cast<int32_t*>(rescan_list + n_list);
  float* part_dists = reinterpret_cast<float*>(part_ids + n_parts);
  int32_t* rescan_ids = reinterpret_cast<int32_t*>(part_dists + n_parts);
  float* rescan_dists = reinterpret_cast<float*>(rescan_ids + n_rescan);
  float* q_shifted = rescan_dists + n_rescan;
  uint32_t* gthr = reinterpret_cast<uint32_t*>(q_shifted + pad4(n_qshift));
  float* q_packed = reinterpret_cast<float*>(gthr + n_gthr);
  GGNN_HIP_CHECK(hipMemsetAsync(flags, 0, 4 * sizeof(uint32_t), stream));
  if (use_i8v2)  // "nothing published"
    GGNN_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(gthr), 0x7fffffff, n_gthr,
                                     stream));
  if (equal_ranges) {
    // not every query block has all `slices` parts: the others read as empty lists
    GGNN_HIP_CHECK(hipMemsetAsync(part_ids, 0xff, parts * sizeof(int32_t), stream));
    GGNN_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(part_dists), 0x7f800000,
                                     parts, stream));
  }

  if (center) {
    // at most ~32k evenly spaced rows: plenty for a shift, negligible next to the scan itself
    const uint32_t stride = std::max(1u, a.N_base / 16384u);
    const uint32_t rows = (a.N_base + stride - 1) / stride;
    hipLaunchKernelGGL(col_mean_partial_kernel, dim3(kBfMeanBlocks), dim3(256), 0, stream,
                       static_cast<const float*>(a.base), a.N_base, a.D, stride, rows, partial);
    hipLaunchKernelGGL(col_mean_final_kernel, dim3((a.D + 255) / 256), dim3(256), 0, stream, partial,
                       a.D, rows, mean);
    const uint64_t n4 = static_cast<uint64_t>(a.Nq) * a.D / 4;
    hipLaunchKernelGGL(shift_rows_kernel,
                       dim3(static_cast<uint32_t>(std::min<uint64_t>((n4 + 255) / 256, 4096))),
                       dim3(256), 0, stream, static_cast<const float*>(a.query), n4, a.D, mean,
                       q_shifted);
  }
  const float* d_mean = center ? mean : nullptr;
  // what the tile and norm kernels see as the query set (the re-rank works on the original rows)
  const void* tile_query = center ? static_cast<const void*>(q_shifted) : a.query;

  if (pack_chunks) {
    const uint64_t n4 = n_qpack / 4;
    const dim3 grid(static_cast<uint32_t>(std::min<uint64_t>((n4 + 255) / 256, 8192)));
    if (a.dtype == GGNN_F32)
      hipLaunchKernelGGL((pack_query_kernel<float>), grid, dim3(256), 0, stream,
                         static_cast<const float*>(tile_query), a.Nq, a.D, pack_chunks, n4,
                         reinterpret_cast<float4*>(q_packed));
    else
      hipLaunchKernelGGL((pack_query_kernel<uint8_t>), grid, dim3(256), 0, stream,
                         static_cast<const uint8_t*>(tile_query), a.Nq, a.D, pack_chunks, n4,
                         reinterpret_cast<float4*>(q_packed));
  }

  BfMfmaArgs m{};
  m.base = a.base;
  m.query = tile_query;
  m.query_packed = q_packed;
  m.mean = d_mean;
  m.bnorm = bnorm;
  m.qnorm = qnorm;
  m.part_ids = part_ids;
  m.part_dists = part_dists;
  m.D = a.D;
  m.Dh = Dh;
  m.DP = DP;
  m.Nq = a.Nq;
  m.N_base = a.N_base;
  m.KP = KP;
  m.slices = slices;
  m.rows_per_slice = rows_per_slice;
  m.equal_ranges = equal_ranges ? 1 : 0;
  m.tiles_per_q = tiles_per_q;
  m.tiles_per_block = tiles_per_block;
  m.total_tiles = total_tiles;
  // the shift vector sits in LDS (single chunk: its 2*Dh columns); 128 query norms and 128
  // thresholds (single chunk only) follow
  // (zero past column D up to the end of the last chunk: see TileStage<float>::load)
  m.DM = a.D > 128 ? (a.D + 127) / 128 * 128 : 2 * Dh;
  // (chunked kernel: [2][T <= 4][32] row norms in place of the thresholds)
  const size_t lds = (2 * kBfTileRows * DP + 2 * kBfQueriesPerBlock * KP + m.DM +
                      (a.D > 128 ? 3 : 2) * kBfQueriesPerBlock) * sizeof(float);
  GGNN_REQUIRE(lds <= 160 * 1024, GGNN_UNSUPPORTED, "k too large for the MFMA brute-force path");
  // rows longer than one chunk: base tiles per accumulator group (hook BF_TILES = 2 | 4)
  const int tiles_per_group = static_cast<int>(unit_tiles);

  BfRerankArgs rr{};
  rr.base = a.base;
  rr.query = a.query;
  rr.part_ids = part_ids;
  rr.part_dists = part_dists;
  rr.qnorm = qnorm;
  rr.bnorm_max = flags;
  rr.rescan_count = flags + 1;
  rr.rescan_list = rescan_list;
  rr.ids = a.ids;
  rr.dists = a.dists;
  rr.D = a.D;
  rr.Nq = a.Nq;
  rr.K = a.k_query;
  rr.KP = KP;
  rr.slices = slices;
  rr.cap = (slices * KP + 3) / 4 * 4;
  rr.exact_arith = use_i8 ? 1 : 0;
  const size_t rr_lds = wave_lds_bytes(2 * rr.cap);

#define GGNN_BF_MFMA(T, MODE_)                                                                    \
  do {                                                                                            \
    hipLaunchKernelGGL((row_norms_kernel<T>), dim3(norm_grid(a.N_base)), dim3(256), 0, stream,   \
                       static_cast<const T*>(a.base), a.N_base, a.D, d_mean, bnorm, flags);       \
    hipLaunchKernelGGL((row_norms_kernel<T>), dim3(norm_grid(a.Nq)), dim3(256), 0, stream,       \
                       static_cast<const T*>(tile_query), a.Nq, a.D,                              \
                       static_cast<const float*>(nullptr), qnorm, static_cast<uint32_t*>(nullptr)); \
    const void* kern = (a.D > 128) ? (tiles_per_group == 2                                             \
                           ? reinterpret_cast<const void*>(&bf_mfma_kernel<T, MODE_, 2, 16>)      \
                           : tiles_per_group == 3                                                 \
                           ? reinterpret_cast<const void*>(&bf_mfma_kernel<T, MODE_, 3, 16>)      \
                           : reinterpret_cast<const void*>(&bf_mfma_kernel<T, MODE_, 4, 16>))     \
                       : (KP == 18) /* k = 10: the list length as a constant (see the kernel) */   \
                           ? ((Dh == 32) ? reinterpret_cast<const void*>(&bf_mfma_kernel<T, MODE_, 1, 8, 18>)  \
                              : (Dh == 48) ? reinterpret_cast<const void*>(&bf_mfma_kernel<T, MODE_, 1, 12, 18>) \
                                           : reinterpret_cast<const void*>(&bf_mfma_kernel<T, MODE_, 1, 16, 18>))\
                       : (Dh == 32) ? reinterpret_cast<const void*>(&bf_mfma_kernel<T, MODE_, 1, 8>)  \
                       : (Dh == 48) ? reinterpret_cast<const void*>(&bf_mfma_kernel<T, MODE_, 1, 12>) \
                                    : reinterpret_cast<const void*>(&bf_mfma_kernel<T, MODE_, 1, 16>);\
    if (lds > 64 * 1024)                                                                          \
      GGNN_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                         static_cast<int>(lds)));                                 \
    void* kargs[] = {&m};                                                                         \
    const dim3 grid = equal_ranges ? dim3(nblocks) : dim3(qblocks, slices);                       \
    GGNN_HIP_CHECK(hipLaunchKernel(kern, grid, dim3(256), kargs, lds, stream));                   \
  } while (0)
  if (use_i8) {
    hipLaunchKernelGGL((row_norms_kernel<uint8_t, true>),
                       dim3(norm_grid(a.N_base)), dim3(256), 0, stream,
                       static_cast<const uint8_t*>(a.base), a.N_base, a.D,
                       static_cast<const float*>(nullptr), bnorm,
                       static_cast<uint32_t*>(nullptr));  // (exact integers: no certificate, no maximum)
    hipLaunchKernelGGL((row_norms_kernel<uint8_t, true>),
                       dim3(norm_grid(a.Nq)), dim3(256), 0, stream,
                       static_cast<const uint8_t*>(a.query), a.Nq, a.D,
                       static_cast<const float*>(nullptr), qnorm, static_cast<uint32_t*>(nullptr));
  }
  if (use_i8v2) {
    m.gthr = hook(kHookBfI8NoShare) ? nullptr : gthr;  // (A/B hook)
    // hook BF_I8_RANKS: bit mask of the published set positions the slices use (31 = all five;
    // the highest bit of a set size alone = the single shared bound of rounds 3-4; negative = auto)
    // default: ONE position -- an exchange costs ~0.2 ms per position and launch (measured: all
    // five 3.08 ms, the last entry alone 3.01, position 1 alone 2.28 for 10k x 1M x 128, k = 10) --
    // the lowest one that at most half of the slices have to reach (12 slices, sets of 10: position
    // 1, the 5th smallest second-best entry; 32 slices: position 0; 4 slices: position 4)
    const int64_t hr = hook(kHookBfI8Ranks);
    if (hr >= 0)
      m.rank_mask = static_cast<uint32_t>(hr) & 31u;
    else
      m.rank_mask = bf_i8v2_default_rank_mask(KP, slices);
    if (slices <= 1)
      m.rank_mask = 0;  // nothing to exchange
    m.refresh_every = static_cast<uint32_t>(std::clamp<int64_t>(hook(kHookBfI8Refresh), 1, 1 << 20));
    // seeds behind the exchange area (both initialised to "nothing" above); hook BF_I8_SEED: rows
    // of the seeding launch (0 = none)
    m.seed = reinterpret_cast<int32_t*>(gthr) + pad4(bf_i8v2_exchange_ints(a.Nq, slices));
    const uint32_t seed_rows = static_cast<uint32_t>(
        std::clamp<int64_t>(hook(kHookBfI8Seed), 0, static_cast<int64_t>(a.N_base)));
    launch_bf_i8v2(m, qblocks, slices, seed_rows, stream);
  }
  else if (use_i8) {
    // (+ 128 words: bf_insert_hits reads up to 126 words past the last list unconditionally)
    const size_t lds8 = 2 * kBfI8Tiles * kBfTileRows * kBfI8RowStride +
                        2 * kBfQueriesPerBlock * KP * sizeof(float) + 128 * sizeof(float);
    const uint32_t nm = (a.D + 31) / 32;
    const void* kern = nm == 1   ? reinterpret_cast<const void*>(&bf_mfma_i8_kernel<1>)
                       : nm == 2 ? reinterpret_cast<const void*>(&bf_mfma_i8_kernel<2>)
                       : nm == 3 ? reinterpret_cast<const void*>(&bf_mfma_i8_kernel<3>)
                                 : reinterpret_cast<const void*>(&bf_mfma_i8_kernel<4>);
    if (lds8 > 64 * 1024)
      GGNN_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds8)));
    void* kargs[] = {&m};
    GGNN_HIP_CHECK(hipLaunchKernel(kern, dim3(qblocks, slices), dim3(256), kargs, lds8, stream));
  }
  else if (a.dtype == GGNN_F32) {
    if (a.measure == GGNN_EUCLIDEAN)
      GGNN_BF_MFMA(float, kL2);
    else
      GGNN_BF_MFMA(float, kCos);
  }
  else {
    if (a.measure == GGNN_EUCLIDEAN)
      GGNN_BF_MFMA(uint8_t, kL2);
    else
      GGNN_BF_MFMA(uint8_t, kCos);
  }
#undef GGNN_BF_MFMA
  GGNN_HIP_CHECK(hipGetLastError());
#define GGNN_BF_RERANK(T, LPR, NCH)                                                                \
  do {                                                                                             \
    if (a.measure == GGNN_EUCLIDEAN)                                                               \
      hipLaunchKernelGGL((bf_rerank_kernel<T, LPR, NCH, kL2>), grid_for(a.Nq), dim3(kWave), rr_lds, \
                         stream, rr);                                                              \
    else                                                                                           \
      hipLaunchKernelGGL((bf_rerank_kernel<T, LPR, NCH, kCos>), grid_for(a.Nq), dim3(kWave),       \
                         rr_lds, stream, rr);                                                      \
  } while (0)
  GGNN_DISPATCH_DIST(a.dtype, a.D, GGNN_BF_RERANK);
#undef GGNN_BF_RERANK
  GGNN_HIP_CHECK(hipGetLastError());
  // queries without a certificate: exact scan (usually none; the surplus blocks leave at once)
  if (!use_i8)
    launch_bf_rescan(a, rescan_list, flags + 1, rescan_ids, rescan_dists, stream);
  if (a.n_rescanned)
    GGNN_HIP_CHECK(hipMemcpyAsync(a.n_rescanned, flags + 1, sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, stream));
  scratch_free(scratch, stream);
}

}  // namespace ggnn_amd

#ifdef GGNN_BF_PHASE
extern "C" int ggnn_debug_bf_bound(const float* bound)
{
  using namespace ggnn_amd;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_bf_dbg_bound), &bound, sizeof(bound)) != hipSuccess;
}
extern "C" int ggnn_debug_bf_phase(unsigned long long* out8, int reset)
{
  using namespace ggnn_amd;
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bf_phase), 8 * sizeof(unsigned long long)) != hipSuccess)
    return 1;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bf_phase), z, sizeof(z)) != hipSuccess)
      return 1;
  }
  return 0;
}
#endif
