// bf_query, uint8 rows with the squared L2 measure and K <= 16: integer contraction on
// v_mfma_i32_32x32x32_i8 with per-query K-best sets in registers.  Separate translation unit:
// compiled with -mllvm -amdgpu-mfma-vgpr-form (Makefile) so that the accumulators and the offset
// block that is their C operand live in ordinary VGPRs -- with the accumulators in AGPRs every
// tile paid 32 v_accvgpr_write (offsets in) + 32 v_accvgpr_read (results out), 4x the 16 VALU
// instructions the tile test itself needs.
// Reference being replaced: BruteForceQueryKernel, src/ggnn/query/bf_query_layer.cu:39-65.
#include <cstdlib>
#include <utility>

#include "bf_common.hpp"

namespace ggnn_amd {

// ---- 2c. uint8 rows, squared L2, short lists (KP <= 24): second-generation i8 kernel ---------------
// What limited the kernel above (profiles/r02_pmc_bf_u8.json: 30 VALU + 21 SALU per MFMA, matrix
// pipe 7 % busy) is not the contraction but everything around it: a tile of 32 rows x 32 queries
// is only 4 MFMAs (128 matrix cycles), while its epilogue converted, expanded and compared all 16
// accumulator registers in float (64+ VALU = 256+ cycles) and every accepted candidate -- about
// one per tile, since each slice list tightens only as KP/rows-seen -- was inserted into a sorted
// LDS list by the whole wave (a chain of dependent LDS round trips).  This kernel changes all three:
//
//  * Threshold folded into the accumulator.  With integer norms, d = |q'|^2 + |b'|^2 - 2 q'.b' < T
//    <=> q'.b' - h > floor(|b'|^2 / 2) for h = floor((|q'|^2 - T) / 2) (conservatively: the test
//    may pass a few non-hits, never drops a hit).  -h is the C operand of the tile's first MFMA
//    (a VGPR block that is only rewritten when thresholds move), so the fast path of a tile is
//    8 v_max3_i32 over the 16 accumulators and ONE compare against the row constant: 9 VALU.
//  * Two query sets (64 queries) per wave share every B operand read from LDS.
//  * Lane = query bookkeeping.  A tile with hits is transposed through LDS (only by the lanes that
//    hit); lane q of the wave then looks at ITS query's values in the hit columns, recomputes
//    the exact integer distance and appends survivors to a small pending list of its own.  The
//    K-best set of a query lives in that lane's REGISTERS (unsorted, replace-the-maximum by
//    (distance, index)); pending candidates are applied for all 64 queries at once in lockstep,
//    so the O(KP) register update is paid per batch, not per candidate.
//  * Thresholds only ever tighten, so stale ones are safe: accumulator offsets are refreshed
//    every few stages.  All slices of a query share a bound through gthr[] (atomicMin of each
//    slice's own KP-th best, read back at refresh points; rows equal to a foreign bound are
//    kept): a row above some slice's KP-th best has KP >= K+8 better rows in that slice, whose
//    final set still contains KP rows at or below the bound -- it cannot be in the top K.  A short
//    first launch over the head of the base seeds gthr[] so that no slice starts unbounded.
// Exact like the kernel above: integer arithmetic throughout, sets are the KP smallest by
// (distance, index) of the rows that passed a bound no top-K row can fail; the re-rank kernel
// orders the union.
constexpr int kI8v2StageRows = 128;  // four 32-row tiles per barrier
#ifndef GGNN_I8_PEND
#define GGNN_I8_PEND 8
#endif
constexpr int kI8v2Pend = GGNN_I8_PEND;         // pending candidates per query before a batch update
// (stages between bound exchanges once past the doubling phase: BfMfmaArgs::refresh_every, hook
//  BF_I8_REFRESH, default 64.  With the single shared bound of rounds 3-4: every 2 stages 3.82 ms,
//  4: 3.54, 8-16: 3.42, 64: 3.50.)
constexpr int kTeInf = 1 << 30;

#ifdef GGNN_I8_STATS
// debug build (make TARGET=libggnn_dbg.so OBJDIR=build_dbg EXTRA=-DGGNN_I8_STATS): event counts of
// the kernel below, read back with ggnn_debug_i8_stats()
__device__ unsigned long long g_i8_stats[16];
#define I8_STAT(i, n) do { if (lane == 0) atomicAdd(&g_i8_stats[i], (unsigned long long)(n)); } while (0)
#define I8_T0() const long long _t0 = clock64()
#define I8_T1(i) I8_STAT(i, clock64() - _t0)
#else
#define I8_STAT(i, n) do { } while (0)
#define I8_T0() do { } while (0)
#define I8_T1(i) do { } while (0)
#endif

// ---- bound exchange between the slices of a query (round 5) ---------------------------------------
// Round 3-4: every slice published the last entry of its K-best set and took the minimum over the
// slices as an extra bound.  That bound is the best "K-th of ONE slice", i.e. about the 65th best
// row seen by all 12 slices together, when the 10th would do -- and each slice starts cold.  Now a
// slice publishes several POSITIONS p of its sorted set, and every slice derives, per position,
//     B_p = the m-th smallest of the published entries [p] over the slices,  m = ceil(KPT / (p+1)):
// m slices each hold p+1 rows at or below B_p, slices hold disjoint rows, so at least KPT >= K rows
// lie at or below B_p and no row above it can be among the K best (rows EQUAL to it are kept).
// Every position is valid on its own with one value per slice, read whenever (entries only ever
// decrease; a stale value is a looser bound), so there is nothing to tear.  For 12 slices and
// KPT = 10: p = 1 (the 5th smallest second-best) sits near the 17th best row overall, p = 0 near
// the 21st, p = 9 (the old bound) near the 65th.  Exchanges follow a doubling schedule (stages 1,
// 2, 4, ... 64, then every 64): the lists move fastest at the start.
// Layout: gl[query][rank][slices padded to 16] ints, 0x7fffffff = nothing published.
constexpr int kI8MaxRanks = 5;
template <int KPT>
__host__ __device__ constexpr int i8_rank_pos(int i)
{
  return KPT == 4    ? (i == 0 ? 0 : i == 1 ? 1 : 3)
         : KPT == 10 ? (i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 2 : i == 3 ? 4 : 9)
                     : (i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 3 : i == 3 ? 7 : 15);
}
template <int KPT>
GGNN_DEV constexpr int i8_ranks()
{
  return KPT == 4 ? 3 : 5;
}
// the M-th smallest of `n4` x 4 published values (unused slots hold 0x7fffffff)
template <int M>
GGNN_DEV int i8_mth_smallest(const int* col, int n4)
{
  int t[M];
#pragma unroll
  for (int k = 0; k < M; ++k)
    t[k] = 0x7fffffff;
  // up to 32 slices in two rounds of 16 (16 registers, not 32: the kernel has none to spare).
  // The four 16-byte loads of a round are issued back to back and waited for ONCE, device-
  // coherent (sc1: the values come from other XCDs).  As sixteen relaxed agent-scope atomic loads
  // -- the first version -- the compiler waited for each one separately: ~17 us per exchange, 12 %
  // of the kernel (stats build), where one round trip is ~1-2 us.  Slots past the slices were
  // initialised to 0x7fffffff by the host (the area is padded to 16 per position).
  for (int b0 = 0; b0 < n4; b0 += 4) {
    i32x4 w0, w1, w2, w3;
    const int* p = col + b0 * 4;
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
        "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
        "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
        : "v"(p)
        : "memory");
    const int v[16] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3],
                       w2[0], w2[1], w2[2], w2[3], w3[0], w3[1], w3[2], w3[3]};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int x = v[e];
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const int lo = min(t[k], x);
        x = max(t[k], x);
        t[k] = lo;
      }
    }
  }
  return t[M - 1];
}

// stages at which a slice exchanges: 1, 2, 4, ... while the lists still move fast, then every
// kI8v2Refresh-th (an exchange is five dependent-free batches of device-coherent loads, ~1.5 stages)
GGNN_DEV bool i8_exchange_stage(uint32_t st, uint32_t every)
{
  if (st == 0)
    return false;
  return st < every ? (st & (st - 1)) == 0 : st % every == 0;
}

GGNN_DEV int i8v2_qrow(int r, int h)
{
  return (r & 3) + 8 * (r >> 2) + 4 * h;  // query row of accumulator register r in half-wave h
}

// LEAN: a register diet for the 16-entry sets, which otherwise need 269 registers = ONE wave per
// SIMD (5.96 ms for k = 16): the accumulator offsets are read from LDS into the accumulators for
// every tile instead of living in 32 VGPRs, one staging register set instead of two, no second
// B-operand set -- 222 registers, two waves per SIMD, 3.92 ms.  The 4- and 10-entry kernels fit two
// waves anyway and are slower on the diet (k = 10: 3.24 vs 3.06 ms, also when forced to three
// waves per SIMD, which costs 14 spilled registers): they keep the offsets in registers.
template <int NM, int KPT, bool LEAN>
__global__ void __launch_bounds__(256)
    __attribute__((amdgpu_waves_per_eu(2))) bf_i8v2_kernel(const BfMfmaArgs a)
{
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  uint8_t* lds_b = reinterpret_cast<uint8_t*>(lds_f);
  constexpr uint32_t SR = kI8v2StageRows;
  constexpr uint32_t stage_bytes = SR * kBfI8RowStride;
  int* bns = reinterpret_cast<int*>(lds_b + 2 * stage_bytes);  // [2][SR] |b'|^2 (rows past the end: 0x3fffffff)
  int* hq_l = bns + 2 * SR;                                     // [4 waves][64] accumulator offsets
  int* qn_l = hq_l + 4 * 64;                                    // [4 waves][64] |q'|^2
  int* te_l = qn_l + 4 * 64;                                    // [4 waves][64] thresholds (d < te)
  int* pc_l = te_l + 4 * 64;                                    // [4 waves][64] pending counts
  int* pd_l = pc_l + 4 * 64;                                    // [4 waves][kPend][64] pending dist
  int* pi_l = pd_l + 4 * kI8v2Pend * 64;                        // [4 waves][kPend][64] pending id

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;     // matrix layout: base row j of the tile, half h
  // bookkeeping layout: lane = query (lane & 31) of set (lane >> 5)
  const uint8_t* base = static_cast<const uint8_t*>(a.base);
  const uint8_t* query = static_cast<const uint8_t*>(a.query);
  const uint32_t qw = (blockIdx.x * 4 + wave) * 64;
  const uint32_t begin = blockIdx.y * a.rows_per_slice;
  const uint32_t end = min(a.N_base, begin + a.rows_per_slice);
  int* hq_w = hq_l + wave * 64;   // LEAN: in matrix order [set][half][register], negated
  int* qn_w = qn_l + wave * 64;
  int* te_w = te_l + wave * 64;
  int* pc_w = pc_l + wave * 64;
  int* pd_w = pd_l + wave * kI8v2Pend * 64;
  int* pi_w = pi_l + wave * kI8v2Pend * 64;

  // ---- lane = query state ----
  const uint32_t my_q = qw + lane;
  const bool my_valid = my_q < a.Nq;
  const int qn_q = my_valid ? static_cast<int>(a.qnorm[my_q]) : 0;
  int Te = my_valid ? kTeInf : -kTeInf;  // candidates need d < Te
  // this query's block of the exchange area: [rank][slices padded to 4]
  const int sl4 = static_cast<int>((a.slices + 15) / 16 * 4);  // int4 groups per position (padded to 16 slices)
  int* gl_q = a.gthr ? reinterpret_cast<int*>(a.gthr) +
                           static_cast<size_t>(my_valid ? my_q : 0) * kI8MaxRanks * 4 * sl4
                     : nullptr;
  const bool gl_q_any = a.gthr != nullptr && a.rank_mask != 0;  // uniform: anything to exchange
  // seeded start: every slice begins with the K-th best distance over the head of the base (one
  // short launch, all queries) instead of "everything passes" -- the cold start of a slice is
  // K (1 + ln(128 / K)) ~ 35 insertions per query in its first stage alone, three quarters of all
  // hits once the slices exchange bounds
  if (my_valid && a.seed && !a.seeding) {
    const int g = a.seed[my_q];
    if (g < 0x7fffffff)
      Te = min(Te, g + 1);
  }
  // the query's K-best set: SORTED ascending by (distance, index), unused slots hold "infinity"
  // (so the threshold is simply the last entry and filling needs no special case)
  int sd[KPT], si[KPT];
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    sd[k] = 0x7fffffff;
    si[k] = 0x7fffffff;
  }
  int hq_used = 0;         // offset the accumulators of this query currently carry
  int Te_seen = 0x7fffffff;
  bool offsets_stale = false;  // wave-uniform
  qn_w[lane] = qn_q;
  te_w[lane] = Te;
  pc_w[lane] = 0;

  // ---- matrix layout state ----
  i32x4 aq[2][NM];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const uint32_t qi = qw + s2 * 32 + j;
    const bool qv = qi < a.Nq;
    const uint8_t* qrow = query + static_cast<size_t>(qv ? qi : 0) * a.D;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const uint32_t col = 32 * m + 16 * h;
      aq[s2][m] = i32x4{0, 0, 0, 0};
      if (qv && col < a.D) {
        const uint4 v = *reinterpret_cast<const uint4*>(qrow + col);
        aq[s2][m] = i32x4{static_cast<int>(v.x ^ 0x80808080u), static_cast<int>(v.y ^ 0x80808080u),
                          static_cast<int>(v.z ^ 0x80808080u), static_cast<int>(v.w ^ 0x80808080u)};
      }
    }
  }
  i32x16 cinit[2];  // (LEAN: unused, the offsets stay in LDS)
  // query (lane & 31) of a set sits in accumulator register (q & 3) + 4 (q >> 3) of half (q >> 2) & 1
  const int my_slot = (lane & 32) + 16 * ((lane >> 2) & 1) + (lane & 3) + 4 * ((lane & 31) >> 3);

  auto refresh_offsets = [&]() {
    I8_T0();
    const int c = qn_q - Te;
    I8_STAT(6, 1);
    hq_used = c >> 1;
    Te_seen = Te;
    offsets_stale = false;
    if constexpr (LEAN) {
      hq_w[my_slot] = -hq_used;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    else {
      hq_w[lane] = hq_used;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          cinit[s2][r] = -hq_w[s2 * 32 + i8v2_qrow(r, h)];
      __builtin_amdgcn_wave_barrier();
    }
    I8_T1(9);
  };
  // offsets of set s2 as this lane's accumulators want them
  auto load_offsets = [&](int s2) __attribute__((always_inline)) -> i32x16 {
    const i32x4* p = reinterpret_cast<const i32x4*>(hq_w + s2 * 32 + 16 * h);
    const i32x4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
    return i32x16{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3],
                  a2[0], a2[1], a2[2], a2[3], a3[0], a3[1], a3[2], a3[3]};
  };

  // applies the pending candidates of all 64 queries in lockstep: one sorted insertion per
  // query and iteration, all in registers (two 64-bit compares and four selects per slot)
  auto flush = [&]() {
    I8_STAT(3, 1);
    I8_T0();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int cnt = min(pc_w[lane], kI8v2Pend);
    pc_w[lane] = 0;
    for (int i = 0; __any(i < cnt); ++i) {
      I8_STAT(4, 1);
      const bool have = i < cnt;
      // (an absent record is "infinity": it changes nothing)
      const int rd = have ? qn_q - pd_w[i * 64 + lane] : 0x7fffffff;
      const int rid = have ? pi_w[i * 64 + lane] : 0x7fffffff;
      const unsigned long long rec =
          (static_cast<unsigned long long>(static_cast<unsigned>(rd)) << 32) | static_cast<unsigned>(rid);
      bool lt_prev = false;  // rec < entry k-1
      int pd_ = 0, pi_ = 0;  // entry k-1 before this insertion
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        const unsigned long long ek =
            (static_cast<unsigned long long>(static_cast<unsigned>(sd[k])) << 32) | static_cast<unsigned>(si[k]);
        const bool lt = rec < ek;
        const int od = sd[k], oi = si[k];
        sd[k] = lt_prev ? pd_ : (lt ? rd : od);
        si[k] = lt_prev ? pi_ : (lt ? rid : oi);
        lt_prev = lt;
        pd_ = od;
        pi_ = oi;
      }
    }
    Te = min(Te, sd[KPT - 1]);  // "infinity" while the set is not full
    te_w[lane] = Te;
    // Thresholds that moved are folded into the accumulator offsets before the next tile (not
    // now: the accumulators being looked at still carry the old offsets).  ANY change counts:
    // squared distances of a high-dimensional base are concentrated, a threshold a few per cent
    // above the current one already passes several times the rows.
    if (__any(Te != Te_seen))
      offsets_stale = true;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    I8_T1(8);
  };

  // threshold exchange with the other slices + accumulator offsets, every few stages
  auto refresh = [&]() {
    if (__any(pc_w[lane] > 0))
      flush();
    if (gl_q && my_valid) {
      // publish this slice's entries (single writer per word), then derive the bounds
      constexpr int NR = i8_ranks<KPT>();
      [&]<int... I>(std::integer_sequence<int, I...>) {
        ((((a.rank_mask >> I) & 1u)
              ? __hip_atomic_store(gl_q + I * 4 * sl4 + blockIdx.y, sd[i8_rank_pos<KPT>(I)],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
              : (void)0),
         ...);
        int bound = 0x7fffffff;
        ((bound = ((a.rank_mask >> I) & 1u)
                      ? min(bound, i8_mth_smallest<(KPT + i8_rank_pos<KPT>(I)) / (i8_rank_pos<KPT>(I) + 1)>(
                                       gl_q + I * 4 * sl4, sl4))
                      : bound),
         ...);
        if (bound < 0x7fffffff)
          Te = min(Te, bound + 1);
      }(std::make_integer_sequence<int, NR>{});
      te_w[lane] = Te;
    }
    if (__any(Te != Te_seen))
      refresh_offsets();
  };

  // Staging: a stage of 128 rows is 1024 pieces of 16 bytes, 4 per thread; threads 0..127 carry
  // the row constants.  Loads run TWO stages ahead of the stage being contracted, in two register
  // sets that alternate (one 8 KB stage per workgroup in flight hid nothing: with ~2 us of memory
  // latency and two workgroups per CU the kernel ran at the speed of its dependent loads); the
  // bytes are shifted to signed on their way into LDS.
  constexpr int PPT = SR / 32;  // pieces per thread
  uint4 sva[PPT], svb[PPT];  // (LEAN: only the first set)
  int bna, bnb = 0;
  // Every load is UNCONDITIONAL, with a clamped address: a load under `if (row < end)` into a
  // register that the other path fills with a constant makes the compiler wait for the load
  // right there (vmcnt(0): the constant may not overtake it), five dependent memory round trips per
  // stage instead of a prefetch (pipeline-only build: 49 % of the wave cycles parked).  Rows past
  // the end of the slice repeat its last row and get the norm sentinel when the stage is stored;
  // columns past D repeat the last 16 and meet zeros in the query operand.
  auto stage_load = [&](uint32_t row0, uint4 (&v)[PPT], int& bn) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < PPT; ++e) {
      const uint32_t p = tid + 256 * e, srow = p >> 3, scol = 16 * (p & 7);
      const uint32_t r = min(row0 + srow, end - 1), c = min(scol, a.D - 16u);
      v[e] = *reinterpret_cast<const uint4*>(base + static_cast<size_t>(r) * a.D + c);
    }
    bn = static_cast<int>(a.bnorm[min(row0 + (tid & (SR - 1)), end - 1)]);
  };
  // row0: first row of the stage being stored
  auto stage_store = [&](uint32_t buf, uint32_t row0, const uint4 (&v)[PPT], int bn)
                         __attribute__((always_inline)) {
    uint8_t* t = lds_b + buf * stage_bytes;
#pragma unroll
    for (int e = 0; e < PPT; ++e) {
      const uint32_t p = tid + 256 * e, srow = p >> 3, scol = 16 * (p & 7);
      *reinterpret_cast<uint4*>(t + srow * kBfI8RowStride + scol) =
          make_uint4(v[e].x ^ 0x80808080u, v[e].y ^ 0x80808080u, v[e].z ^ 0x80808080u,
                     v[e].w ^ 0x80808080u);
    }
    if (tid < (int)SR)  // rows past the end never pass the accumulator test
      bns[buf * SR + tid] = (row0 + tid < end) ? bn : 0x3fffffff;
  };

#ifdef GGNN_I8_STATS
  const long long t_kernel0 = clock64();
#endif
  const uint32_t nstages = (end > begin) ? (end - begin + SR - 1) / SR : 0;
  if (nstages) {
    stage_load(begin, sva, bna);
    stage_store(0, begin, sva, bna);
  }
  refresh_offsets();
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // stage 1 goes into flight now (set b); stage st + 2 is requested at the top of stage st
  if constexpr (!LEAN) {
    if (nstages)
      stage_load(begin + SR, svb, bnb);
  }

  // Hits of one tile-set, in the matrix layout: a lane whose accumulator r passed the test
  // appends (2 q'.b' - |b'|^2, row index) to the pending list of query (r, h), the slot taken with
  // an LDS atomic; the query's bookkeeping lane turns that into the distance (it knows |q'|^2)
  // and decides in the batch update.  Measured on the 1M x 128 base: 16 % of the tile-sets have hits, 6.7
  // on average (a row that is close to one query of the batch is close to many), 87 % of them
  // survive.  A lane whose list is full remembers the register in `redo`; the caller applies the
  // pending candidates and runs the pass again for those.
  // (Tried and dropped: issuing all atomics of a tile-set before the first slot is used -- one LDS
  // round trip instead of one per hit, but the slot array costs more VALU than the latency it
  // hides: 4.8 instead of 3.6 ms.)
  auto scan_pass = [&](const i32x16& acc, const i32x16& off, const int s2, int bnv, int b0,
                       int id, unsigned& redo, const bool first) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool hit = first ? (acc[r] > b0) : ((redo >> r) & 1u);
      if (!__any(hit))
        continue;
      redo &= ~(1u << r);
      if (hit) {
        I8_STAT(1, 1);
        const int q = s2 * 32 + i8v2_qrow(r, h);
        {
          // (no fresh-threshold test here: it costs an LDS round trip in front of the atomic, and
          // the offsets are at most a few tiles stale -- the batch update decides)
          const int slot = atomicAdd(pc_w + q, 1);
          if (slot < kI8v2Pend) {
            pd_w[slot * 64 + q] = 2 * (acc[r] - off[r]) - bnv;  // (off: the tile's offsets)
            pi_w[slot * 64 + q] = id;
            I8_STAT(2, 1);
          }
          else
            redo |= 1u << r;
        }
      }
    }
  };
  auto tile_set = [&](const i32x16& acc, const i32x16& off, const int s2, int bnv, int b0, int id)
                      __attribute__((always_inline)) {
    I8_T0();
    unsigned redo = 0;
    scan_pass(acc, off, s2, bnv, b0, id, redo, true);
    while (__any(redo != 0u)) {
      flush();  // (resets the pending counts: the lists have room again)
      scan_pass(acc, off, s2, bnv, b0, id, redo, false);
    }
    if (__any(pc_w[lane] >= kI8v2Pend / 2))
      flush();
    I8_T1(10);
  };

  auto stage_body = [&](const uint32_t st, uint4 (&v_cur)[PPT], int& bn_cur,
                        const uint4 (&v_next)[PPT], const int& bn_next)
                        __attribute__((always_inline)) {
    const uint32_t row0 = begin + st * SR;
    const uint32_t buf = st & 1;
    const uint8_t* blk = lds_b + buf * stage_bytes;
    // (the exchange first: it waits for its own atomic load with vmcnt(0), which must not have
    // this stage's freshly issued staging loads in front of it)
#if !defined(GGNN_I8_EXP) || GGNN_I8_EXP != 2   // (2: timing experiment without the exchange)
    if (gl_q_any && i8_exchange_stage(st, a.refresh_every)) {
      I8_T0();
      refresh();
      I8_T1(12);
    }
#endif
    // set `cur` held stage st (in LDS since the end of the previous stage): reuse it for st + 2
    stage_load(row0 + 2 * SR, v_cur, bn_cur);
    // the B operands of tile t + 1 are requested while the MFMAs of tile t run (the hit handling
    // in between touches LDS, so the compiler keeps the reads behind it on its own)
    i32x4 bq[2][NM];
#pragma unroll
    for (int m = 0; m < NM; ++m)
      bq[0][m] = *reinterpret_cast<const i32x4*>(blk + j * kBfI8RowStride + 32 * m + 16 * h);
#pragma unroll
    for (int t = 0; t < (int)(SR / kBfTileRows); ++t) {
      if (row0 + t * kBfTileRows >= end)
        break;  // uniform
      if (offsets_stale)
        refresh_offsets();
#ifdef GGNN_I8_STATS
      const long long t_tile0 = clock64();
#endif
      i32x16 acc[2];
      acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[0][0], bq[t & 1][0], cinit[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[1][0], bq[t & 1][0], cinit[1], 0, 0, 0);
#pragma unroll
      for (int m = 1; m < NM; ++m) {
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[0][m], bq[t & 1][m], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[1][m], bq[t & 1][m], acc[1], 0, 0, 0);
      }
      if (t + 1 < (int)(SR / kBfTileRows)) {
        const uint8_t* nrow = blk + ((t + 1) * kBfTileRows + j) * kBfI8RowStride;
#pragma unroll
        for (int m = 0; m < NM; ++m)
          bq[(t + 1) & 1][m] = *reinterpret_cast<const i32x4*>(nrow + 32 * m + 16 * h);
      }
      const int bnv = bns[buf * SR + t * kBfTileRows + j];
      const int b0 = (bnv == 0x3fffffff) ? 0x7fffffff : (bnv >> 1);
      int mx0 = acc[0][0], mx1 = acc[1][0];
#pragma unroll
      for (int r = 1; r < 16; r += 2) {
        mx0 = max(max(mx0, acc[0][r]), acc[0][(r + 1) & 15]);
        mx1 = max(max(mx1, acc[1][r]), acc[1][(r + 1) & 15]);
      }
      I8_STAT(5, 2);
#ifdef GGNN_I8_STATS
      {
        I8_STAT(13, clock64() - t_tile0);
      }
#endif
#if defined(GGNN_I8_EXP) && GGNN_I8_EXP == 1   // timing experiment: no hit handling at all
      if (mx0 == 0x7ffffff0 || mx1 == 0x7ffffff0)
        te_w[lane] = b0;
      continue;
#endif
      if (__any(mx0 > b0)) {
        I8_STAT(0, 1);
        tile_set(acc[0], cinit[0], 0, bnv, b0, static_cast<int>(row0 + t * kBfTileRows) + j);
      }
      if (__any(mx1 > b0)) {
        I8_STAT(0, 1);
        tile_set(acc[1], cinit[1], 1, bnv, b0, static_cast<int>(row0 + t * kBfTileRows) + j);
      }
    }
#ifdef GGNN_I8_STATS
    const long long t_end0 = clock64();
#endif
    if (st + 1 < nstages)
      stage_store((st + 1) & 1, row0 + SR, v_next, bn_next);
#ifdef GGNN_I8_STATS
    const long long t_end1 = clock64();
#endif
    __syncthreads();
#ifdef GGNN_I8_STATS
    I8_STAT(14, t_end1 - t_end0);
    I8_STAT(15, clock64() - t_end1);
#endif
  };
  if constexpr (!LEAN) {
    for (uint32_t st = 0; st < nstages; st += 2) {
      stage_body(st, sva, bna, svb, bnb);
      if (st + 1 < nstages)
        stage_body(st + 1, svb, bnb, sva, bna);
    }
  }
  else {
    for (uint32_t st = 0; st < nstages; ++st) {
      const uint32_t row0 = begin + st * SR;
      const uint32_t buf = st & 1;
      const uint8_t* blk = lds_b + buf * stage_bytes;
      if (gl_q_any && i8_exchange_stage(st, a.refresh_every))
        refresh();
      if (st + 1 < nstages)
        stage_load(row0 + SR, sva, bna);  // written to the other buffer at the end of this stage
#pragma unroll
      for (int t = 0; t < (int)(SR / kBfTileRows); ++t) {
        if (row0 + t * kBfTileRows >= end)
          break;  // uniform
        if (offsets_stale)
          refresh_offsets();
        const uint8_t* trow = blk + (t * kBfTileRows + j) * kBfI8RowStride;
        i32x4 b[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m)
          b[m] = *reinterpret_cast<const i32x4*>(trow + 32 * m + 16 * h);
        i32x16 acc[2];
        acc[0] = load_offsets(0);
        acc[1] = load_offsets(1);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[0][m], b[m], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[1][m], b[m], acc[1], 0, 0, 0);
        }
        const int bnv = bns[buf * SR + t * kBfTileRows + j];
        const int b0 = (bnv == 0x3fffffff) ? 0x7fffffff : (bnv >> 1);
        int mx0 = acc[0][0], mx1 = acc[1][0];
#pragma unroll
        for (int r = 1; r < 16; r += 2) {
          mx0 = max(max(mx0, acc[0][r]), acc[0][(r + 1) & 15]);
          mx1 = max(max(mx1, acc[1][r]), acc[1][(r + 1) & 15]);
        }
        I8_STAT(5, 2);
        if (__any(mx0 > b0)) {
          I8_STAT(0, 1);
          const i32x16 off = load_offsets(0);
          tile_set(acc[0], off, 0, bnv, b0, static_cast<int>(row0 + t * kBfTileRows) + j);
        }
        if (__any(mx1 > b0)) {
          I8_STAT(0, 1);
          const i32x16 off = load_offsets(1);
          tile_set(acc[1], off, 1, bnv, b0, static_cast<int>(row0 + t * kBfTileRows) + j);
        }
      }
      if (st + 1 < nstages)
        stage_store((st + 1) & 1, row0 + SR, sva, bna);
      __syncthreads();
    }
  }

#ifdef GGNN_I8_STATS
  I8_STAT(11, clock64() - t_kernel0);
#endif
  flush();
  if (a.seeding && a.seed && my_valid)
    a.seed[my_q] = sd[KPT - 1];
  if (a.part_ids && my_valid) {
    const size_t o = (static_cast<size_t>(blockIdx.y) * a.Nq + my_q) * KPT;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const bool used = sd[k] != 0x7fffffff;
      a.part_ids[o + k] = used ? si[k] : kEmptyKey;
      a.part_dists[o + k] = used ? static_cast<float>(sd[k]) : inf_f();
    }
  }
}

#ifdef GGNN_I8_STATS
extern "C" int ggnn_debug_i8_stats(unsigned long long* out, int reset)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_i8_stats), sizeof(g_i8_stats)) != hipSuccess)
    return 1;
  if (reset) {
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_i8_stats), z, sizeof(z));
  }
  return 0;
}
#endif

size_t bf_i8v2_lds_bytes()
{
  return 2 * kI8v2StageRows * kBfI8RowStride +
         (2 * kI8v2StageRows + 4 * 4 * 64 + 2 * 4 * kI8v2Pend * 64) * sizeof(int);
}

uint32_t bf_i8v2_default_rank_mask(uint32_t KP, uint32_t slices)
{
  const int n = KP == 4 ? 3 : 5;
  for (int i = 0; i < n; ++i) {
    const uint32_t p = KP == 4    ? static_cast<uint32_t>(i8_rank_pos<4>(i))
                       : KP == 10 ? static_cast<uint32_t>(i8_rank_pos<10>(i))
                                  : static_cast<uint32_t>(i8_rank_pos<16>(i));
    const uint32_t need = (KP + p) / (p + 1);
    if (2 * need <= slices || i == n - 1)
      return 1u << i;
  }
  return 0;
}

size_t bf_i8v2_exchange_ints(uint32_t Nq, uint32_t slices)
{
  return static_cast<size_t>(Nq) * kI8MaxRanks * 16 * ((slices + 15) / 16);
}

void launch_bf_i8v2(const BfMfmaArgs& m, uint32_t qblocks, uint32_t slices, uint32_t seed_rows,
                    hipStream_t stream)
{
  const size_t lds2 = bf_i8v2_lds_bytes();
  const uint32_t nm = (m.D + 31) / 32;
  const uint32_t KP = m.KP;
  GGNN_REQUIRE(KP == 4 || KP == 10 || KP == 16, GGNN_INVALID_ARGUMENT, "unsupported list length");
#define GGNN_I8V2(NM_)                                                                        \
  (KP == 4    ? reinterpret_cast<const void*>(&bf_i8v2_kernel<NM_, 4, false>)                 \
   : KP == 10 ? reinterpret_cast<const void*>(&bf_i8v2_kernel<NM_, 10, false>)                \
              : reinterpret_cast<const void*>(&bf_i8v2_kernel<NM_, 16, true>))
  const void* kern =
      nm == 1 ? GGNN_I8V2(1) : nm == 2 ? GGNN_I8V2(2) : nm == 3 ? GGNN_I8V2(3) : GGNN_I8V2(4);
#undef GGNN_I8V2
  GGNN_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds2)));
  // seeding launch: the first seed_rows rows, all queries, one workgroup per query block, no lists
  // written -- its K-th best distance is every slice's first bound.  (Rounds 3-4 tried this with
  // tens of thousands of rows: 1.1 ms of a 40-workgroup launch, more than it saved.  A few hundred
  // rows are enough -- the first exchange between the slices takes over after 128 rows -- and
  // cost ~20 us.)
  const uint32_t seed = std::min(m.N_base, seed_rows / kI8v2StageRows * kI8v2StageRows);
  if (m.seed && seed >= static_cast<uint32_t>(kI8v2StageRows) && slices > 1) {
    BfMfmaArgs w = m;
    w.part_ids = nullptr;
    w.part_dists = nullptr;
    w.N_base = seed;
    w.rows_per_slice = seed;
    w.slices = 1;
    w.gthr = nullptr;
    w.rank_mask = 0;
    w.seeding = 1;
    void* wargs[] = {&w};
    GGNN_HIP_CHECK(hipLaunchKernel(kern, dim3(qblocks, 1), dim3(256), wargs, lds2, stream));
  }
  BfMfmaArgs mm = m;
  mm.seeding = 0;
  if (!(m.seed && seed >= static_cast<uint32_t>(kI8v2StageRows) && slices > 1))
    mm.seed = nullptr;
  void* kargs[] = {&mm};
  GGNN_HIP_CHECK(hipLaunchKernel(kern, dim3(qblocks, slices), dim3(256), kargs, lds2, stream));
}

}  // namespace ggnn_amd
