// Wave64 device primitives of the GGNN traversal (query / merge / sym kernels), gfx950.
//
// Design: ONE WAVE PER SEARCH.  The reference keeps a per-block cache in shared
// memory and pays >= 3 block barriers plus a SORTED-wide LDS scan per accepted candidate
// (include/ggnn/cuda_utils/simple_knn_cache.cuh:126-213).  Here the sorted part of that cache
// (best list + priority-queue ring) lives in registers, one entry per lane in LOGICAL order, so
// push/pop are a ballot + one cross-lane shift; only the visited ring stays in LDS.  The state
// evolution is identical to the reference's (including the ring-wrap quirk Q1 and the tie
// rules Q2); this is validated on the CPU by oracle/wave_model.hpp against the literal
// emulation, and on the GPU against the oracle.
//
// Distances are computed for all surviving candidates of a fetch at once (LPR lanes per base
// row, 64/LPR rows per wave-wide 16-byte load instruction, i.e. fully coalesced 16 B/lane
// gathers), then the accept/push sequence is replayed in candidate order, which is what the
// reference does one candidate at a time (simple_knn_cache.cuh:268-286).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "hooks.hpp"

namespace ggnn_amd {

#define GGNN_DEV __device__ __forceinline__

// Stats build (-DGGNN_PHASE_CYCLES, scripts/phase_cycles.py): shader cycles per phase of a pop,
// accumulated by lane 0 in LDS behind the first 16 KB of the workgroup's allocation and added to
// g_phase_acc when the wave ends.  Every tick drains the memory counters first, so phases are
// separated (and the kernel is slower than the product build).
#ifdef GGNN_PHASE_CYCLES
static __device__ unsigned long long g_phase_acc[16];
GGNN_DEV void phase_tick(int i)
{
  extern __shared__ __attribute__((aligned(16))) int ph_lds[];
  unsigned long long* p = reinterpret_cast<unsigned long long*>(ph_lds + 4096);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    if (i >= 0)
      p[1 + i] += t - p[0];
    p[0] = t;
  }
}
GGNN_DEV void phase_begin()
{
  extern __shared__ __attribute__((aligned(16))) int ph_lds[];
  unsigned long long* p = reinterpret_cast<unsigned long long*>(ph_lds + 4096);
  if (threadIdx.x < 17)
    p[threadIdx.x] = 0;
  phase_tick(-1);
}
GGNN_DEV void phase_end()
{
  extern __shared__ __attribute__((aligned(16))) int ph_lds[];
  unsigned long long* p = reinterpret_cast<unsigned long long*>(ph_lds + 4096);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (threadIdx.x < 16)
    atomicAdd(&g_phase_acc[threadIdx.x], p[1 + threadIdx.x]);
}
#define GGNN_TICK(i) phase_tick(i)
#else
#define GGNN_TICK(i) ((void)0)
#endif

GGNN_DEV int rdlane(int v, int l)
{
  return __builtin_amdgcn_readlane(v, l);
}
GGNN_DEV float rdlanef(float v, int l)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
GGNN_DEV int uni(int v)
{
  return __builtin_amdgcn_readfirstlane(v);
}
GGNN_DEV float inf_f()
{
  return __builtin_huge_valf();
}

template <int CTRL>
GGNN_DEV float dpp_f(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// whole-wave shifts by one lane as single DPP moves (gfx9 wave_shr / wave_shl); the boundary
// lane keeps its own value, like __shfl_up / __shfl_down
GGNN_DEV int lane_up1(int v)    // lane i <- lane i-1
{
  return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
}
GGNN_DEV int lane_down1(int v)  // lane i <- lane i+1
{
  return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);
}
GGNN_DEV float lane_up1(float v)
{
  return __int_as_float(lane_up1(__float_as_int(v)));
}
GGNN_DEV float lane_down1(float v)
{
  return __int_as_float(lane_down1(__float_as_int(v)));
}

// Cross-half exchange as ONE VALU instruction (gfx950 v_permlane32_swap) instead of an LDS
// crossbar round trip (ds_bpermute): lower[i] / upper[i] = the value lanes i and i+32 hold, both
// delivered to lanes i and i+32.
GGNN_DEV void halves(int v, int& lower, int& upper)
{
  const auto r = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(v),
                                                  static_cast<unsigned>(v), false, false);
  lower = static_cast<int>(r[0]);
  upper = static_cast<int>(r[1]);
}
// lanes j and j+32 <- the value of lane j (j < 32)
GGNN_DEV int lower_half_to_both(int v)
{
  int lo, up;
  halves(v, lo, up);
  return lo;
}
// min over the two half-waves, lane by lane
GGNN_DEV unsigned min_over_halves(unsigned v)
{
  int lo, up;
  halves(static_cast<int>(v), lo, up);
  return min(static_cast<unsigned>(lo), static_cast<unsigned>(up));
}

// sum over groups of LPR consecutive lanes; every lane of a group receives the group total
template <int LPR>
GGNN_DEV float group_sum(float v)
{
  v += dpp_f<0xB1>(v);  // quad_perm:[1,0,3,2]
  v += dpp_f<0x4E>(v);  // quad_perm:[2,3,0,1]
  if (LPR >= 8)
    v += dpp_f<0x141>(v);  // row_half_mirror
  if (LPR >= 16)
    v += dpp_f<0x140>(v);  // row_mirror
  // rows of 16 / half-waves exchanged by v_permlane16_swap / v_permlane32_swap: one VALU
  // instruction each instead of a ds_bpermute round trip (a + b in either order: same bits)
  if (LPR >= 32) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false,
                                                    false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if (LPR >= 64) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false,
                                                    false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}

// LDS layout of one wave (ints): known[CACHE] | ckeys[32] | cd0[32] | cd1[32] | [hashed visited set]
//   known[0,SORTED)        copy of the sorted keys, refreshed at every filtered fetch
//   known[SORTED,CACHE)    visited ring (reference: s_cache[SORTED_SIZE..CACHE_SIZE))
//   buckets[64*HB][8] | stash[32]   only for SortedList<R, HB> with HB > 0 (see there)
struct WaveLds {
  int* known;
  int* ckeys;
  float* cd0;
  float* cd1;
  GGNN_DEV WaveLds(int* base, int cache)
      : known(base), ckeys(base + cache), cd0(reinterpret_cast<float*>(base + cache + 32)),
        cd1(reinterpret_cast<float*>(base + cache + 64))
  {
  }
  static constexpr size_t extra_ints = 96;
};
constexpr int kVisSlots = 8;    // keys per bucket of the hashed visited set
constexpr int kVisStash = 32;   // overflow entries before the filter falls back to the ring scan
// ints of one wave's regions (device and host); hash_regs < 0 (tag set) is sized separately
__host__ __device__ inline size_t wave_lds_ints(uint32_t cache, int hash_regs = 0)
{
  const size_t hash_ints = hash_regs > 0 ? hash_regs * 64 * kVisSlots + kVisStash : 0;
  return cache + WaveLds::extra_ints + hash_ints;
}
inline size_t wave_lds_bytes(uint32_t cache, uint32_t hash_regs = 0)
{
  return wave_lds_ints(cache, static_cast<int>(hash_regs)) * sizeof(int);
}
// bucket registers of the hashed visited set for a visited ring of `vis` entries: ~3 keys per
// bucket on average when the ring is full; 0 = rings too long for it (see the tag set below)
inline uint32_t vis_hash_regs(uint32_t vis)
{
  return vis <= 192 ? 1u : vis <= 480 ? 2u : 0u;
}

// ---- long rings (992 / 2016 keys: searches of 1000-2000 iterations) -----------------------------
// SortedList<1, -nb_bits>.  A scan of the ring costs ~0.75 VALU instructions per key and candidate
// group (768 per fetch on a full 992-key ring, twice the rest of a pop) and 32-bit buckets for such
// a ring cost 8-25 KB of LDS per wave, i.e. the occupancy a search this long lives on.  Here
//   * the ring itself is in GLOBAL memory (one store per pop, never read unless the ring wraps or
//     the set overflows -- both rare), and
//   * the set keeps 16-bit TAGS: with M = nb_bits + 16 >= bits(N_base - 1), k -> (k * C) mod 2^M
//     (C odd) is a bijection on the keys, so (bucket = its top nb_bits, tag = its low 16 bits)
//     identifies the key EXACTLY -- no false positives, 2 bytes per key: 256 buckets x 16 bytes =
//     4 KB for a 992-key ring.  A bucket is SEVEN tags + the number of tags in use in its eighth
//     halfword (round 6; rounds 4-5 kept eight tags and a count byte elsewhere): one 16-byte read
//     brings both, buckets fill from slot 0 (tags are never removed), the probe masks the others.
//     With the candidate scratch cut to what the query kernels use this is 5.0 KB of LDS per wave
//     incl. the float query row -- 5.4 KB were 26 waves per CU, one less than the registers allow.
// A full bucket sends the key to the stash as before; a stash overflow, or the first wrap of the
// ring, switches to the scan of the global ring for the rest of the search (exact in every case).
// HB = -nb_bits selects the tag set with 2^nb_bits buckets (a compile-time constant: the table
// offsets and the hash mask fold into immediates instead of living in scalar registers)
constexpr bool is_tag_set(int hb)
{
  return hb < 0;
}
constexpr uint32_t kTagMul = 0x9E3779B1u;
constexpr int kTagSlots = 7;          // tags per 16-byte bucket (the eighth halfword counts them)
constexpr int kTagScratchInts = 64;   // candidate scratch of the tag-set kernels: ckeys[32] | cd0[32]
// buckets (log2) for a ring of `vis` keys: load factor <= 0.57 with 7 tags per bucket (256 buckets
// for 992 keys, 512 for 2016)
inline uint32_t tag_set_bucket_bits(uint32_t vis)
{
  uint32_t b = 5;
  while ((static_cast<uint32_t>(kTagSlots) << b) * 4 < 7 * vis)
    ++b;
  return b;
}
// usable when every key of the shard fits into nb_bits + 16 bits
inline bool tag_set_usable(uint32_t vis, uint32_t n_base)
{
  if (vis < 481 || vis > 2016)
    return false;
  const uint32_t m = tag_set_bucket_bits(vis) + 16;
  return m >= 32 || (static_cast<uint64_t>(n_base) <= (1ull << m));
}
// LDS of one wave: known[sorted] | ckeys[32] | cd0[32] | buckets (16 bytes each) | stash
__host__ __device__ inline size_t tag_set_lds_ints(uint32_t sorted, uint32_t nb_bits)
{
  const size_t nb = size_t{1} << nb_bits;
  return sorted + kTagScratchInts + kVisStash + nb * 4;
}
inline size_t tag_set_lds_bytes(uint32_t sorted, uint32_t vis)
{
  return tag_set_lds_ints(sorted, tag_set_bucket_bits(vis)) * sizeof(int);
}

// ---------------------------------------------------------------------------------------------
// Sorted part of the cache: logical entry i = r*64 + lane.
//   [0,BEST) best list, [BEST,SORTED) priority queue in logical order (head first).
// Reference: SimpleKNNCache, simple_knn_cache.cuh:58-352.  push() follows the cache's tie rule
// (Q2: a new equal distance goes first), push_best_stable() the KBestList's (k_best_list.cuh:
// 92-103: equal distances keep insertion order).
// ---------------------------------------------------------------------------------------------
// HB > 0: the visited ring is mirrored in an exact hash set so that the membership test of a fetch
// (simple_knn_cache.cuh:246-261 scans the whole cache for every candidate) is one bucket read
// per candidate instead of a scan of up to CACHE entries:
//   * 64*HB buckets of kVisSlots keys in LDS; lane b of hcnt[b >> 6] counts bucket b, so an
//     insert (one per pop, the key is wave-uniform) needs no LDS read;
//   * a key whose bucket is full goes to a small stash that every probe scans (normally empty);
//   * if the stash overflows too, scan_mode switches the filter back to the ring scan for the rest
//     of the search -- the ring is always maintained, so the answer is exact in every case;
//   * when the ring wraps (more pops than ring entries) the overwritten key is removed again.
// The set of keys reported as known is exactly the reference's: sorted part + visited ring.
// GR (with HB > 0): NO visited ring at all -- for searches that cannot wrap it (max_iterations <=
// ring length, the launcher's condition) the ring's only job is to say which keys are known, and
// the buckets + stash hold exactly those keys (nothing is ever removed without a wrap).  A wave's
// LDS then holds only the sorted keys, the scratch and the buckets: for kernels whose occupancy
// the 2 KB ring of a 512-key cache limits.  A key that fits neither its bucket nor the stash goes
// to an overflow list in global memory (`ring_g`, normally empty; every probe scans it when it is
// not).  (First version: the ring itself in global memory, one store per pop -- slower than the
// LDS ring at higher occupancy, 3.69 vs 3.38 ms on the 12.5M x 96 shard: with a store in flight
// the compiler can no longer wait for "all but the newest n" loads -- loads and stores retire out
// of order with respect to each other -- so every wait for the requested code rows became
// vmcnt(0) and also waited for the speculative graph row issued after them.)
template <int R, int HB = 0, bool GR = false>
struct SortedList {
  static_assert(!GR || HB != 0, "ring-less: only with a hashed set or a tag set");
  static constexpr int kHashRegs = HB;
  // (the tag set of long rings keeps its ring in global memory unless the search cannot wrap it:
  // then it is ring-less as well, GR, with the same overflow list)
  static constexpr bool kGlobalRing = is_tag_set(HB) && !GR;
  int ovf_n;                  // GR: keys in the overflow list (global memory)
  static constexpr int NB = 64 * HB;
  int key[R];
  float dist[R];
  int BEST, SORTED, P, VIS;
  int head_in;    // r_prioQ_head - BEST
  int vis_head;   // r0_visited_head - SORTED
  int vis_count;  // valid entries of the visited ring
  float xi;
  int hcnt[HB > 0 ? HB : 1];  // lane b: number of keys in bucket b (+64 per register)
  int stash_n;
  int scan_mode;
  int slots;                  // usable keys per bucket (kVisSlots; tests shrink it)
  int* hbuckets;
  int* hstash;
  // tag set (HB < 0) only
  static constexpr bool kTag = is_tag_set(HB);
  static constexpr int nb_bits = kTag ? -HB : 0;
  int* ring_g;                // [VIS] visited ring of this search in global memory

  GGNN_DEV void init(int best, int sorted, int cache, float xi_, int* known,
                     int usable_slots = kVisSlots)
  {
    BEST = best;
    SORTED = sorted;
    P = sorted - best;
    VIS = cache - sorted;
    xi = xi_;
    slots = usable_slots;
    hbuckets = known + cache + static_cast<int>(WaveLds::extra_ints);
    hstash = hbuckets + NB * kVisSlots;
    reset(known);
  }
  // ring-less form of the hashed set: only known[0, sorted) lives in LDS (WaveLds(base, sorted));
  // ring: this search's overflow list, (cache - sorted) ints of global memory
  GGNN_DEV void init_global_ring(int best, int sorted, int cache, float xi_, int* known,
                                 int usable_slots, int* ring)
  {
    BEST = best;
    SORTED = sorted;
    P = sorted - best;
    VIS = cache - sorted;
    xi = xi_;
    slots = usable_slots;
    ring_g = ring;
    hbuckets = known + sorted + static_cast<int>(WaveLds::extra_ints);
    hstash = hbuckets + NB * kVisSlots;
    reset(known);
  }
  // tag-set form: only known[0, sorted) lives in LDS (WaveLds(base, sorted)); the ring is ring
  GGNN_DEV void init_tagged(int best, int sorted, int cache, float xi_, int* known, int usable_slots,
                            int* ring)
  {
    BEST = best;
    SORTED = sorted;
    P = sorted - best;
    VIS = cache - sorted;
    xi = xi_;
    slots = usable_slots < kTagSlots ? usable_slots : kTagSlots;
    ring_g = ring;
    hbuckets = known + sorted + kTagScratchInts;  // 4 ints per bucket: 7 tags + their number
    hstash = hbuckets + (4 << nb_bits);
    reset(known);
  }
  // visited ring (and its hashed mirror) empty, simple_knn_cache.cuh:73-87
  GGNN_DEV void clear_visited(int* known)
  {
    vis_head = 0;
    vis_count = 0;
    ovf_n = 0;
    if constexpr (kTag) {
      // counts (the upper half of a bucket's last word) to zero; tags need no clearing (masked
      // by the counts), the global ring is only ever read below vis_count
      for (int i = threadIdx.x; i < (1 << nb_bits); i += kWave)
        hbuckets[i * 4 + 3] = 0;
      stash_n = 0;
      scan_mode = 0;
      return;
    }
    if constexpr (!GR) {
      for (int i = SORTED + threadIdx.x; i < SORTED + VIS; i += kWave)
        known[i] = kEmptyKey;
    }
    if constexpr (HB > 0) {
      int4* hb = reinterpret_cast<int4*>(hbuckets);
      for (int i = threadIdx.x; i < NB * kVisSlots / 4; i += kWave)
        hb[i] = make_int4(kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey);
#pragma unroll
      for (int r = 0; r < (HB > 0 ? HB : 1); ++r)
        hcnt[r] = 0;
      stash_n = 0;
      scan_mode = 0;
    }
  }
  GGNN_DEV void reset(int* known)
  {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      key[r] = kEmptyKey;
      dist[r] = inf_f();
    }
    head_in = 0;
    clear_visited(known);
  }

  // bucket of a key: 24-bit multiplicative hash (v_mul_u32_u24 is full rate), top bits
  static GGNN_DEV uint32_t vis_hash(uint32_t k)
  {
    k ^= k >> 20;
    const uint32_t h = __umul24(k, 0x9E3779u);
    return HB == 2 ? (h >> 25) : (h >> 26);
  }
  GGNN_DEV int bucket_count(uint32_t b) const
  {
    if constexpr (HB == 2)
      return (b >> 6) ? rdlane(hcnt[1], b & 63) : rdlane(hcnt[0], b & 63);
    return rdlane(hcnt[0], b & 63);
  }
  GGNN_DEV void bucket_add(uint32_t b, int delta)
  {
    const int lane = threadIdx.x;
#pragma unroll
    for (int r = 0; r < (HB > 0 ? HB : 1); ++r)
      if (lane + 64 * r == static_cast<int>(b))
        hcnt[r] += delta;
  }
  // k: wave-uniform key that has just entered the visited ring
  GGNN_DEV void vis_insert(int k)
  {
    const uint32_t b = vis_hash(static_cast<uint32_t>(k));
    const int c = bucket_count(b);
    if (c < slots) {
      if (threadIdx.x == 0)
        hbuckets[b * kVisSlots + c] = k;
      bucket_add(b, 1);
    }
    else if (stash_n < kVisStash) {
      if (threadIdx.x == 0)
        hstash[stash_n] = k;
      ++stash_n;
    }
    else if constexpr (GR) {
      if (threadIdx.x == 0)
        ring_g[ovf_n] = k;  // (at most one entry per pop, pops <= ring length)
      ++ovf_n;
    }
    else
      scan_mode = 1;  // the ring scan takes over; the set is no longer maintained
  }
  // k: wave-uniform key that the ring is about to overwrite (it is in the set exactly once)
  GGNN_DEV void vis_remove(int k)
  {
    const int lane = threadIdx.x;
    const uint32_t b = vis_hash(static_cast<uint32_t>(k));
    const int c = bucket_count(b);
    __syncthreads();
    const int v = (lane < c) ? hbuckets[b * kVisSlots + lane] : -2;
    const unsigned long long hit = __ballot(v == k);
    if (hit) {
      const int p = __ffsll(static_cast<long long>(hit)) - 1;
      const int last = hbuckets[b * kVisSlots + c - 1];
      __syncthreads();
      if (lane == 0) {
        hbuckets[b * kVisSlots + p] = last;
        hbuckets[b * kVisSlots + c - 1] = kEmptyKey;
      }
      bucket_add(b, -1);
    }
    else {
      const int sv = (lane < stash_n) ? hstash[lane] : -2;
      const unsigned long long shit = __ballot(sv == k);
      if (shit) {
        const int p = __ffsll(static_cast<long long>(shit)) - 1;
        const int last = hstash[stash_n - 1];
        __syncthreads();
        if (lane == 0)
          hstash[p] = last;
        --stash_n;
      }
    }
    __syncthreads();
  }

  // tag set: bucket / tag of a key (see "long rings" above)
  GGNN_DEV uint32_t tag_hash(uint32_t k) const
  {
    const uint32_t h = k * kTagMul;
    return nb_bits + 16 >= 32 ? h : (h & ((1u << (nb_bits + 16)) - 1u));
  }
  // k: wave-uniform key that has just entered the visited ring
  GGNN_DEV void tag_insert(int k)
  {
    const uint32_t h = tag_hash(static_cast<uint32_t>(k));
    const uint32_t b = h >> 16;
    // lane l reads halfword (l & 7) of the bucket: lanes 0-6 a tag, lane 7 the number of tags
    const int lane = threadIdx.x;
    unsigned short* bucket = reinterpret_cast<unsigned short*>(hbuckets) + b * 8;
    const int mine = bucket[lane & 7];
    const int c = rdlane(mine, 7);
    // A key can be popped more than once (quirk Q1 duplicates a queue entry when the ring of the
    // priority queue has wrapped): the set keeps it once, or its copies would fill their bucket
    if (__any(lane < c && mine == static_cast<int>(h & 0xffffu)))
      return;
    if (c < slots) {
      if (threadIdx.x == 0) {
        bucket[c] = static_cast<unsigned short>(h & 0xffffu);
        bucket[7] = static_cast<unsigned short>(c + 1);
      }
    }
    else if (stash_n < kVisStash) {
      if (threadIdx.x == 0)
        hstash[stash_n] = k;
      ++stash_n;
    }
    else if constexpr (GR) {
      if (threadIdx.x == 0)
        ring_g[ovf_n] = k;  // overflow list (at most one entry per pop, pops <= ring length)
      ++ovf_n;
    }
    else
      scan_mode = 1;
  }

  GGNN_DEV float dist_at(int i) const
  {
    if constexpr (R == 1)
      return rdlanef(dist[0], i);
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((i >> 6) == r)
        v = rdlanef(dist[r], i & 63);
    return v;
  }
  GGNN_DEV int key_at(int i) const
  {
    if constexpr (R == 1)
      return rdlane(key[0], i);
    int v = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((i >> 6) == r)
        v = rdlane(key[r], i & 63);
    return v;
  }
  // simple_knn_cache.cuh:121-124
  GGNN_DEV float criteria() const
  {
    return dist_at(BEST - 1) + xi;
  }

  // simple_knn_cache.cuh:126-213 in lane form (oracle/wave_model.hpp::push)
  GGNN_DEV void push(int k, float d)
  {
    bool dup = false;
#pragma unroll
    for (int r = 0; r < R; ++r)
      dup |= (key[r] == k);
    if (__any(dup))
      return;
    // logical index of the entry in physical slot BEST; Q1: nothing shifts into it
    const int qlane = head_in ? BEST + (P - head_in) : -1;
    if constexpr (R >= 16) {
      // long lists: the registers are walked in groups of four with two wave-uniform skips --
      // groups beyond SORTED hold nothing (R comes in steps of 16), and once the lowest
      // register of a best-list group had nothing to shift, nothing below shifts either (the
      // list is sorted).  One branch per FOUR registers: a branch per register serialises the
      // DPP / readlane chains of neighbouring registers and costs more than it skips
      // (K = 1000: 118 ms against 108 without any skip), a `break` sends the arrays to scratch.
      bool settled = false;
#pragma unroll
      for (int g = R / 4 - 1; g >= 0; --g) {
        if (g * 4 * kWave >= SORTED || settled)
          continue;
        bool low_active = false;
#pragma unroll
        for (int r = g * 4 + 3; r >= g * 4; --r)
          low_active = push_step(r, k, d, qlane);
        if ((g * 4 + 1) * kWave <= BEST && !__any(low_active))
          settled = true;
      }
    }
    else {
#pragma unroll
      for (int r = R - 1; r >= 0; --r)
        push_step(r, k, d, qlane);
    }
  }

  // one register of push(): entry i takes (k, d) at the insertion point, its lower neighbour above
  GGNN_DEV bool push_step(const int r, const int k, const float d, const int qlane)
  {
    const int lane = threadIdx.x;
    const int i = r * kWave + lane;
    int pk = lane_up1(key[r]);
    float pd = lane_up1(dist[r]);
    if (r > 0) {
      const int bk = rdlane(key[r - 1], 63);
      const float bd = rdlanef(dist[r - 1], 63);
      if (lane == 0) {
        pk = bk;
        pd = bd;
      }
    }
    const bool first = (i == 0) || (i == BEST);
    const bool active = (dist[r] >= d) && (i < SORTED);
    const bool prev_active = !first && (pd >= d);
    if (active) {
      if (first || !prev_active) {
        key[r] = k;
        dist[r] = d;
      }
      else if (i != qlane && pk != kEmptyKey) {
        key[r] = pk;
        dist[r] = pd;
      }
    }
    return active;
  }

  // KBestList::add_unique, k_best_list.cuh:77-109 (stable insert after equal distances, no
  // duplicate check); only the best list is used (BEST == SORTED).
  GGNN_DEV void push_best_stable(int k, float d)
  {
    const int lane = threadIdx.x;
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
      const int i = r * kWave + lane;
      int pk = lane_up1(key[r]);
      float pd = lane_up1(dist[r]);
      if (r > 0) {
        const int bk = rdlane(key[r - 1], 63);
        const float bd = rdlanef(dist[r - 1], 63);
        if (lane == 0) {
          pk = bk;
          pd = bd;
        }
      }
      const bool active = (d < dist[r]) && (i < BEST);
      const bool prev_active = (i != 0) && (d < pd);
      if (active) {
        if (!prev_active) {
          key[r] = k;
          dist[r] = d;
        }
        else {
          key[r] = pk;
          dist[r] = pd;
        }
      }
    }
  }

  // simple_knn_cache.cuh:215-239 ; crit is criteria() (or criteria_sym() for the sym cache)
  GGNN_DEV int pop(float crit, int* known)
  {
    const int k0 = peek(crit);
    if (k0 != kEmptyKey)
      pop_commit(k0, known);
    return k0;
  }
  // the two halves of pop(): the decision (which key would be popped, simple_knn_cache.cuh:218-224)
  // and the bookkeeping (:225-238).  Kernels that request the rows of the popped key's neighbours
  // between the two hide the bookkeeping under that memory latency.
  GGNN_DEV int peek(float crit) const
  {
    const int k0 = key_at(BEST);
    const float d0 = dist_at(BEST);
    return (k0 == kEmptyKey || d0 >= crit) ? kEmptyKey : k0;
  }
  GGNN_DEV void pop_commit(const int k0, int* known)
  {
    if constexpr (HB > 0 && !GR) {
      if (!scan_mode) {
        if (vis_count == VIS)  // the ring wraps: its oldest key is forgotten
          vis_remove(uni(known[SORTED + vis_head]));
        vis_insert(k0);
      }
    }
    if constexpr (GR && !kTag)
      vis_insert(k0);  // (no wrap: pops <= max_iterations <= ring length)
    if constexpr (GR && kTag)
      tag_insert(k0);
    if constexpr (kTag && !GR) {
      if (vis_count == VIS)
        scan_mode = 1;  // the ring wraps (tags are never removed): the ring scan takes over
      if (!scan_mode)
        tag_insert(k0);
    }
    if constexpr (kGlobalRing) {
      if (threadIdx.x == 0)
        ring_g[vis_head] = k0;
    }
    else if constexpr (!GR) {
      if (threadIdx.x == 0)
        known[SORTED + vis_head] = k0;
    }
    vis_head = (vis_head + 1 >= VIS) ? 0 : vis_head + 1;
    vis_count = (vis_count + 1 > VIS) ? VIS : vis_count + 1;
    const int lane = threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = r * kWave + lane;
      int nk = lane_down1(key[r]);
      float nd = lane_down1(dist[r]);
      if (r + 1 < R) {
        const int bk = rdlane(key[r + 1], 0);
        const float bd = rdlanef(dist[r + 1], 0);
        if (lane == 63) {
          nk = bk;
          nd = bd;
        }
      }
      if (i >= BEST && i < SORTED) {
        const bool last = (i == SORTED - 1);
        key[r] = last ? kEmptyKey : nk;
        dist[r] = last ? inf_f() : nd;
      }
    }
    head_in = (head_in + 1 >= P) ? 0 : head_in + 1;
  }

  // Second half of the membership test of simple_knn_cache.cuh:246-261 for kernels that skipped
  // the scan of the sorted part (filter<.., false>): clears the bits of `m` whose candidate (lane
  // j holds its key in k_of) sits in the best list or the priority queue.  Must run before the
  // first push of the fetch -- the reference tests every candidate against the cache as it is when
  // the fetch starts -- and costs one compare per candidate that is still in the race (the ~5
  // that passed the pre-screen, or the ~4 whose distance beats the criteria) instead of a scan
  // of 32 / 64 keys for all 24.
  GGNN_DEV unsigned long long drop_sorted(unsigned long long m, const int k_of) const
  {
    unsigned long long out = m;
    while (m) {
      const int j = __ffsll(static_cast<long long>(m)) - 1;
      m &= m - 1;
      const int k = rdlane(k_of, j);
      bool hit = false;
#pragma unroll
      for (int r = 0; r < R; ++r)
        hit |= (key[r] == k);
      if (__any(hit))
        out &= ~(1ull << j);
    }
    return out;
  }

  // simple_knn_cache.cuh:297-333 ; goes through LDS (known[0,SORTED)), called once per layer
  GGNN_DEV void transform(const int32_t* selection, int* known, float* scratch_d)
  {
    const int lane = threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = r * kWave + lane;
      if (i < BEST) {
        int k = key[r];
        if (k != kEmptyKey)
          k = selection[k];
        key[r] = k;
        known[i] = k;
        scratch_d[i] = dist[r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = r * kWave + lane;
      if (i >= BEST) {
        const bool copy = (i < 2 * BEST) && (i < SORTED);
        key[r] = copy ? known[i - BEST] : kEmptyKey;
        dist[r] = copy ? scratch_d[i - BEST] : inf_f();
      }
    }
    __syncthreads();
    head_in = 0;
    clear_visited(known);
  }

  // filter part of fetch(): simple_knn_cache.cuh:246-261 / simple_knn_sym_cache.cuh:408-419.
  // cand: lanes j and j+32 hold candidate j (or EMPTY).  Returns cand with known keys blanked.
  // rare: the ring in global memory is scanned, 16 bytes per lane and step, both half-waves.
  // The keys were stored by lane 0 of this wave and another lane's store does not update the
  // vector L1: it is invalidated first (agent-scope acquire)
  // count: valid entries at the head of ring_g
  GGNN_DEV unsigned scan_global_ring(unsigned acc, const unsigned c, const int h, const int count) const
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int4* rp = reinterpret_cast<const int4*>(ring_g);
    const int steps = (count + 3) >> 2;  // the ring length is a multiple of four
    // (one step in flight: more would cost the whole kernel registers for a path that runs
    // for the last few pops of a search, if at all)
    for (int t = h; t < steps; t += 2) {
      int4 e = rp[t];
      if (t * 4 + 1 >= count)
        e.y = kEmptyKey;
      if (t * 4 + 2 >= count)
        e.z = kEmptyKey;
      if (t * 4 + 3 >= count)
        e.w = kEmptyKey;
      acc = min(min(acc, static_cast<unsigned>(e.x) ^ c), static_cast<unsigned>(e.y) ^ c);
      acc = min(min(acc, static_cast<unsigned>(e.z) ^ c), static_cast<unsigned>(e.w) ^ c);
    }
    return acc;
  }

  // AHEAD: the reads of the next pair of 16-byte groups are issued before the current pair is
  // folded (a lone wave otherwise pays one LDS latency per pair).  The early-rows order runs the
  // test under the latency of the candidates' row loads and prefers the 8 registers.
  // SCAN_SORTED = false (kernels that do not count evaluations, fetch_early<.., COUNT = false>):
  // only the VISITED keys are tested here -- the hashed / tag set, stash and overflow list -- and
  // the sorted part is tested later, for the few candidates that would go on to the float rows
  // (drop_sorted below).  Only while the set is in use: once a search has fallen back to the ring
  // scan (scan_mode) the full test runs as before.
  template <bool AHEAD = true, bool SCAN_SORTED = true>
  GGNN_DEV int filter(int cand, int* known) const
  {
    const int lane = threadIdx.x;
    // with the hashed set only the sorted part is scanned; the visited ring is one bucket read
    const bool hashed = (HB != 0) && !scan_mode;
    const bool scan = SCAN_SORTED || !hashed;
    if (scan) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i = r * kWave + lane;
        if (i < SORTED)
          known[i] = key[r];
      }
      __syncthreads();
    }
    const int E = (hashed || kGlobalRing || GR) ? SORTED : SORTED + vis_count;
    const int h = lane >> 5;
    const int4* kp = reinterpret_cast<const int4*>(known);
    // min over (entry XOR cand) is 0 iff some entry equals cand.  Pure VALU on purpose: the
    // v_cmp + s_or form needs fewer VALU instructions but measured slower (scalar dependency
    // chain per entry).
    const unsigned c = static_cast<unsigned>(cand);
    auto fold = [c](unsigned acc, const int4& e) {
      acc = min(min(acc, static_cast<unsigned>(e.x) ^ c), static_cast<unsigned>(e.y) ^ c);
      return min(min(acc, static_cast<unsigned>(e.z) ^ c), static_cast<unsigned>(e.w) ^ c);
    };
    unsigned acc0 = 0xffffffffu, acc1 = 0xffffffffu;
    if constexpr (kTag) {
      if (hashed) {
        // lower half-wave: candidate j probes its bucket (one 16-byte read: 7 tags and, in the
        // last halfword, how many of them are in use); upper half-wave: the stash
        const uint32_t hh = tag_hash(c);
        const uint32_t b = hh >> 16;
        const uint32_t tt = (hh & 0xffffu) * 0x10001u;
        if (h == 0) {
          const int4 w = *reinterpret_cast<const int4*>(hbuckets + b * 4);
          const int v = static_cast<int>(static_cast<unsigned>(w.w) >> 16);
          auto pair = [tt, v](unsigned a, int wv, int slot) {
            const unsigned x = static_cast<unsigned>(wv) ^ tt;
            const unsigned lo = slot < v ? (x & 0xffffu) : 1u;
            const unsigned hi = slot + 1 < v ? (x >> 16) : 1u;
            return min(a, min(lo, hi));
          };
          acc1 = pair(pair(pair(acc1, w.x, 0), w.y, 2), w.z, 4);
          acc1 = min(acc1, 6 < v ? ((static_cast<unsigned>(w.w) ^ tt) & 0xffffu) : 1u);
        }
        else {
          for (int t = 0; t < stash_n; ++t)
            acc1 = min(acc1, static_cast<unsigned>(hstash[t]) ^ c);
        }
      }
      else if constexpr (!GR)
        acc1 = scan_global_ring(acc1, c, h, vis_count);
    }
    if constexpr (GR) {
      // overflow list (normally empty): one key at a time -- this rare path must not add to the
      // register peak of the test, which the requested code rows of the early-rows order share.
      // The keys were stored by lane 0: the vector L1 is invalidated first.
      if (ovf_n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int t = 0; t < ovf_n; ++t)
          acc1 = min(acc1, static_cast<unsigned>(ring_g[t]) ^ c);
      }
    }
    if constexpr (HB > 0) {
      if (hashed) {
        // lanes j and j+32 hold candidate j: each reads one half of its bucket
        const uint32_t b = vis_hash(c);
        acc1 = fold(acc1, *reinterpret_cast<const int4*>(hbuckets + b * kVisSlots + 4 * h));
        for (int t = 0; t < stash_n; ++t)
          acc1 = min(acc1, static_cast<unsigned>(hstash[t]) ^ c);
      }
    }
    const int4* p = kp + h;
    const int T = (E + 7) >> 3;  // >= 4: SORTED >= 32
    if (!scan) {
      // (visited keys only: nothing of known[] is read)
    }
    // the reads of the next pair are issued before the current pair is folded: a lone wave
    // (small batches, tail of a launch) otherwise pays one LDS latency per pair
    else if constexpr (AHEAD) {
      int4 e0 = p[0], e1 = p[2];
      int t = 0;
      for (; t + 4 <= T; t += 2) {
        const int4 n0 = p[2 * t + 4], n1 = p[2 * t + 6];
        acc0 = fold(acc0, e0);
        acc1 = fold(acc1, e1);
        e0 = n0;
        e1 = n1;
      }
      acc0 = fold(acc0, e0);
      acc1 = fold(acc1, e1);
      t += 2;
      if (t < T)
        acc0 = fold(acc0, p[2 * t]);
    }
    else if (R == 1 && E == 32) {
      // one list register per lane: the sorted part has 32 or 64 keys, i.e. 4 or 8 groups of
      // eight -- straight-line code instead of a loop with a run-time trip count (the loop's
      // unrolled body, its remainder loops and their scalar bookkeeping cost more than the scan)
#pragma unroll
      for (int t = 0; t < 4; t += 2) {
        acc0 = fold(acc0, p[2 * t]);
        acc1 = fold(acc1, p[2 * t + 2]);
      }
    }
    else if (R == 1 && E == 64) {
#pragma unroll
      for (int t = 0; t < 8; t += 2) {
        acc0 = fold(acc0, p[2 * t]);
        acc1 = fold(acc1, p[2 * t + 2]);
      }
    }
    else {
      int t = 0;
      for (; t + 2 <= T; t += 2) {
        const int4 e0 = p[2 * t], e1 = p[2 * t + 2];
        acc0 = fold(acc0, e0);
        acc1 = fold(acc1, e1);
      }
      if (t < T)
        acc0 = fold(acc0, p[2 * t]);
    }
    return (min_over_halves(min(acc0, acc1)) == 0u) ? kEmptyKey : cand;
  }
};

// ---------------------------------------------------------------------------------------------
// LDS-resident form of the same cache for SORTED > 256 (KQuery up to 6000): a wave64 port of the
// reference's physical layout and chunked shift-insert (simple_knn_cache.cuh:58-352, emulated
// literally by oracle/ggnn_oracle.cpp::Cache with BLOCK = 64; the result does not depend on the
// chunk width).  Slow path: every push walks the sorted region through LDS.
// ---------------------------------------------------------------------------------------------
struct LdsList {
  int* key;     // [CACHE] physical layout: best | priority-queue ring | visited ring
  float* dist;  // [SORTED]
  int BEST, SORTED, CACHE;
  int pq_head, vis_head, vis_count;
  float xi;

  GGNN_DEV void init(int best, int sorted, int cache, float xi_, int* key_, float* dist_)
  {
    key = key_;
    dist = dist_;
    BEST = best;
    SORTED = sorted;
    CACHE = cache;
    xi = xi_;
    for (int i = threadIdx.x; i < CACHE; i += kWave) {
      key[i] = kEmptyKey;
      if (i < SORTED)
        dist[i] = inf_f();
    }
    pq_head = BEST;
    vis_head = SORTED;
    vis_count = 0;
    __syncthreads();
  }
  GGNN_DEV float dist_at(int i) const { return dist[i]; }
  GGNN_DEV int key_at(int i) const { return key[i]; }
  GGNN_DEV float criteria() const { return dist[BEST - 1] + xi; }

  // simple_knn_cache.cuh:126-213, one chunk of 64 logical entries at a time from the top.  The
  // reference separates "shift", "read the left neighbour", "insert" by barriers (four per chunk);
  // here a chunk is ONE read phase (entry, and the left neighbour's distance) and ONE write phase:
  //  * the left neighbour's distance is read before anything in or below this chunk is written,
  //    and that is the value the reference's later read decides on as well: a slot left of an
  //    insertion point is not active, so nothing shifts into it;
  //  * an inserting lane and the lane left of it never write the same slot for the same reason;
  //  * writes of a chunk touch the chunk itself and the first slot of the chunk above, which was
  //    read an iteration earlier.
  GGNN_DEV void push(int k, float d)
  {
    const int lane = threadIdx.x;
    __syncthreads();
    bool dup = false;
    for (int i = lane; i < SORTED; i += kWave)
      dup |= (key[i] == k);
    if (__any(dup))
      return;
    const int head = pq_head;
    const int head_in = head - BEST;
    for (int block_start = (SORTED + kWave - 1) / kWave * kWave - kWave; block_start >= 0;
         block_start -= kWave) {
      const int li = block_start + lane;
      bool active = li < SORTED;
      int idx = 0, r_key = kEmptyKey;
      float r_dist = 0.f, p_dist = -inf_f();
      if (active) {
        idx = li;
        if (li >= BEST)
          idx = (li + head_in < SORTED) ? li + head_in : li + head_in - SORTED + BEST;
        r_key = key[idx];
        r_dist = dist[idx];
        const bool has_prev = idx != 0 && idx != head;
        const int idx_prev = idx != BEST ? idx - 1 : SORTED - 1;
        if (has_prev)
          p_dist = dist[idx_prev];
        active = r_dist >= d;  // Q2
      }
      __syncthreads();  // every read of the chunk before its writes
      if (active) {
        if (p_dist < d) {
          key[idx] = k;
          dist[idx] = d;
        }
        if (r_key != kEmptyKey) {
          const int idx_next = (idx + 1 == SORTED) ? BEST : idx + 1;
          if (idx_next != BEST && idx_next != head) {  // Q1
            key[idx_next] = r_key;
            dist[idx_next] = r_dist;
          }
        }
      }
      // a chunk of the best list in which nothing shifts: neither does anything below it (the
      // list is sorted) -- on average this halves the walk
      if (block_start + kWave <= BEST && !__any(active))
        break;
      __syncthreads();
    }
    __syncthreads();
  }

  GGNN_DEV int pop(float crit)
  {
    __syncthreads();
    const int k0 = key[pq_head];
    const float d0 = dist[pq_head];
    if (k0 == kEmptyKey || d0 >= crit)
      return kEmptyKey;
    __syncthreads();
    if (threadIdx.x == 0) {
      key[vis_head] = k0;
      key[pq_head] = kEmptyKey;
      dist[pq_head] = inf_f();
    }
    vis_head = (vis_head + 1 >= CACHE) ? SORTED : vis_head + 1;
    vis_count = (vis_count + 1 > CACHE - SORTED) ? CACHE - SORTED : vis_count + 1;
    pq_head = (pq_head + 1 >= SORTED) ? BEST : pq_head + 1;
    __syncthreads();
    return k0;
  }

  GGNN_DEV int filter(int cand, int* /*unused*/) const
  {
    const int lane = threadIdx.x;
    __syncthreads();
    const int E = SORTED + vis_count;  // the visited ring fills contiguously until it wraps
    const int h = lane >> 5;
    const int4* kp = reinterpret_cast<const int4*>(key);
    unsigned acc = 0xffffffffu;
    const unsigned c = static_cast<unsigned>(cand);
    for (int t = 0; t * 8 < E; ++t) {
      const int4 e = kp[t * 2 + h];
      const unsigned a = min(static_cast<unsigned>(e.x) ^ c, static_cast<unsigned>(e.y) ^ c);
      const unsigned b = min(static_cast<unsigned>(e.z) ^ c, static_cast<unsigned>(e.w) ^ c);
      acc = min(acc, min(a, b));
    }
    return (min_over_halves(acc) == 0u) ? kEmptyKey : cand;
  }
};

// ---------------------------------------------------------------------------------------------
// Distance engine.  Reference: Distance, include/ggnn/cuda_utils/distance.cuh:34-164 (squared
// L2 / |1-cos|) and the two-point variant of simple_knn_sym_cache.cuh:143-283.
// LPR lanes cooperate on one base row, each lane owns NCH chunks of 16 bytes.
// ---------------------------------------------------------------------------------------------
template <typename BaseT>
struct ChunkOf;
template <>
struct ChunkOf<float> {
  using type = float4;
  static constexpr int EPC = 4;
  static GGNN_DEV float get(const float4& v, int e)
  {
    return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
  }
  static GGNN_DEV float4 zero()
  {
    return make_float4(0.f, 0.f, 0.f, 0.f);
  }
};
template <>
struct ChunkOf<uint8_t> {
  using type = uint4;
  static constexpr int EPC = 16;
  static GGNN_DEV float get(const uint4& v, int e)
  {
    const uint32_t w = (e >> 2) == 0 ? v.x : (e >> 2) == 1 ? v.y : (e >> 2) == 2 ? v.z : v.w;
    return static_cast<float>((w >> (8 * (e & 3))) & 0xffu);
  }
  static GGNN_DEV uint4 zero()
  {
    return make_uint4(0u, 0u, 0u, 0u);
  }
};

enum DistMode { kL2 = 0, kCos = 1 };

// QL: the query chunks live in LDS (q_lds[c * LPR + g], LPR * NCH chunks = one padded row per
// wave) instead of NCH x 4 registers per lane, and are read where a distance is summed.  For
// kernels whose register peak lies elsewhere (early rows: the requested code rows are live across
// the membership test, the float rows are only needed for the ~4 candidates that pass).
template <typename BaseT, int LPR_, int NCH_, bool QL = false>
struct DistEngine {
  using Base = BaseT;
  using Chunk = typename ChunkOf<BaseT>::type;
  static constexpr int EPC = ChunkOf<BaseT>::EPC;
  static constexpr int LPR = LPR_;
  static constexpr int NCH = NCH_;
  static constexpr int ROWS = kWave / LPR;
  static constexpr bool kQueryInLds = QL;
  static constexpr size_t kQueryLdsBytes = QL ? sizeof(Chunk) * LPR_ * NCH_ : 0;

  const BaseT* base;
  uint32_t D;
  int g;  // lane within the row group
  bool all_chunks;  // wave-uniform: the row fills every chunk of every lane (e.g. D = 128 f32)
  Chunk q[QL ? 1 : NCH];
  Chunk* q_lds;    // QL only
  float q_norm;    // cosine: |q|^2
  uint32_t qq_u8;  // uint8 rows: sum of squares of this lane's query elements

  GGNN_DEV Chunk qchunk(int c) const
  {
    if constexpr (QL)
      return q_lds[c * LPR + g];
    else
      return q[c];
  }

  GGNN_DEV bool chunk_valid(int c) const
  {
    return static_cast<uint32_t>((c * LPR + g) * EPC) < D;
  }
  GGNN_DEV Chunk load_chunk(const BaseT* row, int c) const
  {
    return *reinterpret_cast<const Chunk*>(row + (c * LPR + g) * EPC);
  }
  GGNN_DEV const BaseT* row_ptr(int m) const
  {
    return base + static_cast<size_t>(static_cast<uint32_t>(m)) * D;  // 64-bit addressing
  }

  // distance.cuh:104-117
  // q_lds_: the wave's LPR * NCH chunks of LDS (QL only)
  template <int MODE>
  GGNN_DEV void load_query(const BaseT* base_, uint32_t D_, const BaseT* qrow, void* q_lds_ = nullptr)
  {
    base = base_;
    D = D_;
    g = threadIdx.x % LPR;
    all_chunks = D_ == static_cast<uint32_t>(LPR * NCH * EPC);
    q_lds = static_cast<Chunk*>(q_lds_);
    float nrm = 0.f;
    qq_u8 = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const Chunk qc = chunk_valid(c) ? load_chunk(qrow, c) : ChunkOf<BaseT>::zero();
      if constexpr (QL)
        q_lds[c * LPR + g] = qc;  // (every row group writes the same values)
      else
        q[c] = qc;
      if (MODE == kCos) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float v = ChunkOf<BaseT>::get(qc, e);
          nrm = fmaf(v, v, nrm);
        }
      }
      if constexpr (std::is_same<BaseT, uint8_t>::value) {
        qq_u8 = __builtin_amdgcn_udot4(qc.x, qc.x, qq_u8, false);
        qq_u8 = __builtin_amdgcn_udot4(qc.y, qc.y, qq_u8, false);
        qq_u8 = __builtin_amdgcn_udot4(qc.z, qc.z, qq_u8, false);
        qq_u8 = __builtin_amdgcn_udot4(qc.w, qc.w, qq_u8, false);
      }
    }
    q_norm = 0.f;
    if (MODE == kCos)
      q_norm = group_sum<LPR>(nrm);
    if constexpr (QL)
      __syncthreads();
  }

  // per-lane partial sums over the lane's chunks of one row
  template <int MODE>
  GGNN_DEV void partial(const Chunk (&v)[NCH], float& a, float& b) const
  {
    if constexpr (std::is_same<BaseT, uint8_t>::value) {
      // packed integer arithmetic (v_dot4_u32_u8), exact: sum (o-q)^2 = sum o^2 + sum q^2 - 2 sum oq.
      // Equals the reference's float accumulation whenever that is exact (D <= 258).
      uint32_t ab = 0, bb = 0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const Chunk qc = qchunk(c);
        ab = __builtin_amdgcn_udot4(v[c].x, qc.x, ab, false);
        ab = __builtin_amdgcn_udot4(v[c].y, qc.y, ab, false);
        ab = __builtin_amdgcn_udot4(v[c].z, qc.z, ab, false);
        ab = __builtin_amdgcn_udot4(v[c].w, qc.w, ab, false);
        bb = __builtin_amdgcn_udot4(v[c].x, v[c].x, bb, false);
        bb = __builtin_amdgcn_udot4(v[c].y, v[c].y, bb, false);
        bb = __builtin_amdgcn_udot4(v[c].z, v[c].z, bb, false);
        bb = __builtin_amdgcn_udot4(v[c].w, v[c].w, bb, false);
      }
      if (MODE == kL2) {
        a = static_cast<float>((qq_u8 + bb) - 2u * ab);
        b = 0.f;
      }
      else {
        a = static_cast<float>(ab);
        b = static_cast<float>(bb);
      }
      return;
    }
    a = 0.f;
    b = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const Chunk qc = qchunk(c);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float o = ChunkOf<BaseT>::get(v[c], e);
        const float qq = ChunkOf<BaseT>::get(qc, e);
        if (MODE == kL2) {
          const float diff = o - qq;
          a = fmaf(diff, diff, a);
        }
        else {
          a = fmaf(o, qq, a);
          b = fmaf(o, o, b);
        }
      }
    }
  }
  // distance.cuh:153-158
  GGNN_DEV float finish_cos(float dot, float nrm) const
  {
    const float norm_sqr = q_norm * nrm;
    return (norm_sqr > 0.0f) ? fabsf(1.0f - dot / sqrtf(norm_sqr)) : 1.0f;
  }
};

// rows in flight per distance pass (register budget: STEPS*NCH chunks of 4 VGPRs)
template <int LPR, int NCH>
struct StepsOf {
  static constexpr int value = (NCH <= 2) ? 4 : (NCH <= 4) ? 2 : 1;
};

// Computes the distances of the nsurv compacted candidates in lds.ckeys[0,nsurv) and leaves
// them in lds.cd0[0,nsurv).  Out-of-range chunks are neither loaded nor accumulated.
template <int MODE, class DE, int STEPS = StepsOf<DE::LPR, DE::NCH>::value>
GGNN_DEV void compute_distances(const DE& de, const WaveLds& lds, int nsurv,
                                const int32_t* translation)
{
  constexpr int ROWS = DE::ROWS;
  using Chunk = typename DE::Chunk;
  const int lane = threadIdx.x;
  const int grp = lane / DE::LPR;
  for (int s0 = 0; s0 < nsurv; s0 += ROWS * STEPS) {
    Chunk v[STEPS][DE::NCH];
    int rr[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s > 0 && s0 + s * ROWS >= nsurv)
        break;  // wave-uniform: no rows left for this and the following steps
      const int r = s0 + s * ROWS + grp;
      const bool valid = r < nsurv;
      rr[s] = valid ? r : -1;
      // slots past the end read the row of the first candidate (cached, result never stored):
      // no branch and no zero-fill around the loads
      int m = lds.ckeys[valid ? r : s0];
      if (translation)
        m = translation[m];
      const auto* row = de.row_ptr(m);
#pragma unroll
      for (int c = 0; c < DE::NCH; ++c) {
        if (de.all_chunks || de.chunk_valid(c))
          v[s][c] = de.load_chunk(row, c);
        else
          v[s][c] = ChunkOf<typename DE::Base>::zero();
      }
    }
    GGNN_TICK(6);  // float rows arrived (issue + wait)
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s0 + s * ROWS >= nsurv)
        break;  // wave-uniform: no rows left for this and the following steps
      if (s > 0)
        asm volatile("" ::: "memory");  // keep the branch: the later steps are usually empty
      float a, b;
      de.template partial<MODE>(v[s], a, b);
      a = group_sum<DE::LPR>(a);
      if (MODE == kCos)
        b = group_sum<DE::LPR>(b);
      if (rr[s] >= 0 && de.g == 0)
        lds.cd0[rr[s]] = (MODE == kCos) ? de.finish_cos(a, b) : a;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Exact pre-screen of candidates on a compact copy of the rows (float32 rows).
//
// ~85 % of the distance evaluations of a traversal end in "d >= criteria(): not pushed"
// (simple_knn_cache.cuh:283-285) -- for those the value of d is irrelevant.  Every row x is
// therefore also kept as 8-bit codes c with x^ = o + s*c (per-dimension offset o, one
// scale s, prescreen.hip) and   e_max >= max_rows ||x - x^||.   By the triangle
// inequality  ||q - x|| >= ||q - x^|| - e_max,  so a candidate whose bound already reaches the
// criteria at the start of the fetch can be dropped WITHOUT reading its float row: the reference
// would evaluate it and drop it as well (criteria() only tightens during a fetch).  Survivors of
// the test are evaluated exactly as before, so ids, distances and counters are unchanged; only
// HBM traffic shrinks (D bytes instead of 4D for a rejected candidate).
//
// The query is coded the same way (q^ = o + s*cq, cq clamped to [0,255]) and pays for it with its
// own coding error e_q = ||q - q^||:   ||q - x|| >= ||q^ - x^|| - e_q - e_max,   where
// ||q^ - x^||^2 = s^2 * sum (cq_d - c_d)^2 is exact integer arithmetic on v_dot4_u32_u8.
//
// Rounding: with u = 2^-24, q' = fl(fl(q-o) * fl(1/s)) has ||s q' - (q-o)|| <= 3u(||q|| + ||o||),
// e_q is a float sum of non-negative terms (relative
// error <= (Dc+8)u), and the float distance of the exact phase is >= exact*(1 - (D+8)u).  All of
// these are covered by the relative margin m = 4(Dc+32)u and the absolute slack
// e_q(1+m) + 8u(||q|| + ||o||) + e_max used in threshold().
// ---------------------------------------------------------------------------------------------
constexpr int kPsHeader = 8;  // params: [0] s  [1] 1/s  [2] e_max  [3] ||o||  [4] valid  [8..] o_d
struct NoPrescreen {
  static constexpr bool enabled = false;
};

// Cosine (MODE = kCos): rows and queries are coded after normalisation to unit length; for unit
// vectors |1 - cos| = ||q^ - x^||^2 / 2, so the same bound applies to 2 * criteria.  The float
// evaluation |1 - dot / sqrt(|q|^2 |x|^2)| (distance.cuh:153-158) is within (D+8)u of that in
// ABSOLUTE terms, covered by adding m to the criteria.  A query of zero norm (distance 1 to
// everything in the reference) switches the pre-screen off.
template <int LPR_, int NCH_, int MODE_ = kL2>
struct Prescreen {
  static constexpr bool enabled = true;
  static constexpr int LPR = LPR_;
  static constexpr int NCH = NCH_;
  static constexpr int ROWS = kWave / LPR;
  using Chunk = uint4;

  const uint8_t* codes;
  uint32_t Dc;  // code row length (multiple of 16)
  int g;
  uint4 qc[NCH];  // codes of the query for this lane's dimensions
  uint32_t qq;    // sum of their squares
  float inv_s, slack, m;
  bool usable;
  bool all_chunks;  // wave-uniform: the code row fills every chunk of every lane

  GGNN_DEV bool chunk_valid(int c) const
  {
    return static_cast<uint32_t>((c * LPR + g) * 16) < Dc;
  }
  GGNN_DEV const uint8_t* row_ptr(int k) const
  {
    return codes + static_cast<size_t>(static_cast<uint32_t>(k)) * Dc;
  }
  GGNN_DEV uint4 load_chunk(const uint8_t* row, int c) const
  {
    return *reinterpret_cast<const uint4*>(row + (c * LPR + g) * 16);
  }

  // D: float row length of the query (multiple of 4, <= Dc)
  GGNN_DEV void load(const uint8_t* codes_, const float* params, uint32_t Dc_, const float* qrow,
                     uint32_t D)
  {
    codes = codes_;
    Dc = Dc_;
    g = threadIdx.x % LPR;
    all_chunks = Dc_ == static_cast<uint32_t>(LPR * NCH * 16);
    inv_s = params[1];
    const float* offs = params + kPsHeader;
    constexpr float u = 5.9604645e-8f;  // 2^-24
    m = 4.f * static_cast<float>(Dc + 32) * u;

    // cosine: the norm is needed before coding; the row is read twice (it is in cache) instead
    // of being held in registers
    float qs = 0.f;
    if (MODE_ == kCos) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const uint32_t d0 = static_cast<uint32_t>((c * LPR + g) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (d0 + 4 * j < D) {
            const float4 v = *reinterpret_cast<const float4*>(qrow + d0 + 4 * j);
            qs = fmaf(v.x, v.x, qs);
            qs = fmaf(v.y, v.y, qs);
            qs = fmaf(v.z, v.z, qs);
            qs = fmaf(v.w, v.w, qs);
          }
        }
      }
    }
    float q_scale = 1.f;
    usable = true;
    if (MODE_ == kCos) {
      const float q_norm = sqrtf(group_sum<LPR>(qs));
      usable = q_norm > 0.f && q_norm < inf_f();
      q_scale = usable ? 1.f / q_norm : 0.f;
      qs = 0.f;
    }

    float eq = 0.f;
    qq = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const uint32_t d0 = static_cast<uint32_t>((c * LPR + g) * 16);
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), ov = qv;
        if (d0 + 4 * j < D) {
          qv = *reinterpret_cast<const float4*>(qrow + d0 + 4 * j);
          ov = *reinterpret_cast<const float4*>(offs + d0 + 4 * j);
        }
        const float qe[4] = {qv.x, qv.y, qv.z, qv.w};
        const float oe[4] = {ov.x, ov.y, ov.z, ov.w};
        w[j] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = (qe[e] * q_scale - oe[e]) * inv_s;
          const float code = fminf(fmaxf(rintf(t), 0.f), 255.f);
          const float diff = t - code;
          eq = fmaf(diff, diff, eq);
          qs = fmaf(qe[e], qe[e], qs);
          w[j] |= static_cast<uint32_t>(code) << (8 * e);
        }
        qq = __builtin_amdgcn_udot4(w[j], w[j], qq, false);
      }
      qc[c] = make_uint4(w[0], w[1], w[2], w[3]);
      // one chunk at a time: without this barrier the loads of all chunks are hoisted and the
      // kernel's register allocation is set by this prologue (184 instead of ~110 VGPRs at D=960)
      if (NCH > 2)
        asm volatile("" ::: "memory");
    }
    // norm of the vector that was coded (unit length for the cosine measure)
    const float q_norm = (MODE_ == kCos) ? 1.f : sqrtf(group_sum<LPR>(qs));
    const float e_q = params[0] * sqrtf(group_sum<LPR>(eq)) * (1.f + m);
    slack = e_q + 8.f * u * (q_norm + params[3]) + params[2];
  }

  // sum (cq - c)^2 over this lane's chunks of one code row (exact)
  GGNN_DEV float partial(const uint4 (&v)[NCH]) const
  {
    uint32_t ab = 0, bb = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      ab = __builtin_amdgcn_udot4(v[c].x, qc[c].x, ab, false);
      ab = __builtin_amdgcn_udot4(v[c].y, qc[c].y, ab, false);
      ab = __builtin_amdgcn_udot4(v[c].z, qc[c].z, ab, false);
      ab = __builtin_amdgcn_udot4(v[c].w, qc[c].w, ab, false);
      bb = __builtin_amdgcn_udot4(v[c].x, v[c].x, bb, false);
      bb = __builtin_amdgcn_udot4(v[c].y, v[c].y, bb, false);
      bb = __builtin_amdgcn_udot4(v[c].z, v[c].z, bb, false);
      bb = __builtin_amdgcn_udot4(v[c].w, v[c].w, bb, false);
    }
    return static_cast<float>((qq + bb) - 2u * ab);
  }

  // a candidate with group-summed S >= threshold(crit) has a float distance >= crit
  GGNN_DEV float threshold(float crit) const
  {
    if (!(crit < inf_f()) || !usable)
      return inf_f();
    if (MODE_ == kCos)
      crit = 2.f * (crit + m) * (1.f + m);
    // v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence: m >= 32 * 4 * 2^-24 covers it
    float t = __builtin_amdgcn_sqrtf(crit) * (1.f + m) + slack;
    t = t * inv_s * (1.f + m);
    return t * t * (1.f + m);
  }
};

// code-row layout used next to a float-row layout <LPR, NCH> (a code row has a quarter of the
// 16-byte chunks of the float row)
template <int LPR, int NCH, int MODE>
struct PsFor {
  using type = Prescreen<8, 1, MODE>;
};
template <int MODE>
struct PsFor<16, 4, MODE> {
  using type = Prescreen<16, 1, MODE>;
};
template <int MODE>
struct PsFor<64, 4, MODE> {
  using type = Prescreen<32, 2, MODE>;  // fewer registers than <16, 4> (2 instead of 4 chunks/lane)
};
template <int MODE>
struct PsFor<64, 16, MODE> {
  using type = Prescreen<64, 4, MODE>;
};

// lanes per row / chunks per lane of a pre-screen type (0 for NoPrescreen)
template <class PS>
struct PsLayout {
  static constexpr int lpr = PS::LPR, nch = PS::NCH;
};
template <>
struct PsLayout<NoPrescreen> {
  static constexpr int lpr = 0, nch = 0;
};

// Drops the candidates of lds.ckeys[0,nsurv) whose lower bound reaches the criteria; the others
// are compacted in place (order kept).  Returns their number.  translation: optional map from
// candidate keys to base rows (upper graph layers).
template <class PS>
GGNN_DEV int prescreen_pass(const PS& ps, const WaveLds& lds, int nsurv, float s_thr,
                            const int32_t* translation)
{
  // one-chunk layouts keep the usual KBuild = 24 neighbours of a graph row in flight in one round
  constexpr int STEPS = (PS::NCH == 1 && PS::ROWS <= 8) ? 24 / PS::ROWS
                                                        : StepsOf<PS::LPR, PS::NCH>::value;
  constexpr int ROWS = PS::ROWS;
  const int lane = threadIdx.x;
  const int grp = lane / PS::LPR;
  int npass = 0;
  for (int s0 = 0; s0 < nsurv; s0 += ROWS * STEPS) {
    uint4 v[STEPS][PS::NCH];
    int kk[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int r = s0 + s * ROWS + grp;
      const bool valid = r < nsurv;
      // slots past the end read the code row of the first candidate (cached, verdict ignored)
      int m = lds.ckeys[valid ? r : s0];
      kk[s] = valid ? m : kEmptyKey;
      if (translation)
        m = translation[m];
      const uint8_t* row = ps.row_ptr(m);
#pragma unroll
      for (int c = 0; c < PS::NCH; ++c) {
        if (ps.all_chunks || ps.chunk_valid(c))
          v[s][c] = ps.load_chunk(row, c);
        else
          v[s][c] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    __syncthreads();  // all keys of this round are in registers before the in-place compaction
    GGNN_TICK(4);     // code rows arrived (issue + wait)
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s0 + s * ROWS >= nsurv)
        break;
      const float S = group_sum<PS::LPR>(ps.partial(v[s]));
      const bool pass = (kk[s] != kEmptyKey) && (ps.g == 0) && !(S >= s_thr);
      const unsigned long long pm = __ballot(pass);
      if (pass)
        lds.ckeys[npass + __popcll(pm & ((1ull << lane) - 1ull))] = kk[s];
      npass += __popcll(pm);
    }
  }
  return npass;
}

// fetch(): simple_knn_cache.cuh:241-289.  cand: lane j (<32) holds candidate key j or EMPTY.
// Returns the number of distance evaluations (of the reference: pre-screened candidates count);
// rows.x / rows.y are advanced by the numbers of float / code rows actually read.
// after_filter(): hook called once the candidate keys have been consumed by the filter -- the
// place for a caller's own prefetch: a load issued BEFORE fetch() is waited for together with the
// candidates' graph row (the compiler merges the waits to vmcnt(0)), i.e. not overlapped at all.
struct NoHook {
  GGNN_DEV void operator()() const {}
};
template <int MODE, bool FILTER, class SL, class DE, class PS, class HOOK = NoHook>
GGNN_DEV int fetch(SL& sl, const DE& de, const WaveLds& lds, int cand,
                   const int32_t* translation, const PS& ps, uint2& rows,
                   HOOK&& after_filter = NoHook{})
{
  const int lane = threadIdx.x;
  cand = lower_half_to_both(cand);
  GGNN_TICK(1);  // graph row arrived
  if (FILTER)
    cand = sl.filter(cand, lds.known);
  const unsigned long long surv = __ballot(lane < 32 && cand != kEmptyKey);
  const int nsurv = __popcll(surv);
  GGNN_TICK(2);  // filter
  after_filter();
  if (nsurv == 0)
    return 0;
  __syncthreads();
  if (lane < 32 && cand != kEmptyKey)
    lds.ckeys[__popcll(surv & ((1ull << lane) - 1ull))] = cand;
  __syncthreads();
  int neval = nsurv;
  GGNN_TICK(3);  // compaction
  if constexpr (PS::enabled) {
    const float s_thr = ps.threshold(sl.criteria());
    if (s_thr < inf_f()) {
      neval = prescreen_pass(ps, lds, nsurv, s_thr, translation);
      rows.y += nsurv;
      GGNN_TICK(5);  // pre-screen verdicts + compaction
      if (neval == 0)
        return nsurv;
      __syncthreads();
    }
  }
  // after the pre-screen only a handful of candidates are left: fewer rows in flight, fewer VGPRs
  // (three chunks per lane, 8 rows per load instruction: one step already has 8 rows in flight)
  constexpr int kSteps = StepsOf<DE::LPR, DE::NCH>::value;
  constexpr int kExactSteps = !PS::enabled ? kSteps : (DE::NCH == 3 && DE::ROWS >= 8) ? 1
                              : (kSteps > 2)                                         ? 2
                                                                                     : kSteps;
  compute_distances<MODE, DE, kExactSteps>(de, lds, neval, translation);
  rows.x += neval;
  __syncthreads();
  GGNN_TICK(7);  // exact distances
  const float cd = lane < neval ? lds.cd0[lane] : inf_f();
  const int ck = lane < neval ? lds.ckeys[lane] : kEmptyKey;
  // criteria() never increases during a fetch, so candidates failing it now fail it later
  unsigned long long m = __ballot(cd < sl.criteria());
  while (m) {
    const int j = __ffsll(static_cast<long long>(m)) - 1;
    m &= m - 1;
    const float d = rdlanef(cd, j);
    const int k = rdlane(ck, j);
    if (d < sl.criteria())
      sl.push(k, d);
  }
  GGNN_TICK(8);  // accept / push replay
  return nsurv;
}
template <int MODE, bool FILTER, class SL, class DE>
GGNN_DEV int fetch(SL& sl, const DE& de, const WaveLds& lds, int cand,
                   const int32_t* translation)
{
  uint2 rows = make_uint2(0u, 0u);
  return fetch<MODE, FILTER>(sl, de, lds, cand, translation, NoPrescreen{}, rows);
}

// ---------------------------------------------------------------------------------------------
// Early rows (round 5).  A pop of fetch() above is a chain of DEPENDENT steps: pop bookkeeping ->
// graph row -> membership test -> compaction -> code rows (or the rows themselves) -> verdicts ->
// float rows -> replay, i.e. every memory round trip starts only after the instruction chain in
// front of it has run.  For graph rows of at most 24 neighbours (KBuild <= 24, every BASELINE
// configuration) whose FIRST row read is 128 bytes wide -- 8 lanes x 16 bytes: the pre-screen codes
// of a float base (Prescreen<8,1>) or the rows of a base with <= 128-byte rows (uint8 D <= 128,
// DistEngine<.,8,1> without pre-screen) -- the order becomes
//     decide the pop (peek) -> graph row (speculated: usually there) -> REQUEST the first-read
//     rows of all <= 24 neighbours -> pop bookkeeping + membership test (under that latency) ->
//     verdicts with the membership verdict applied as a mask -> [float rows -> distances] -> replay.
// Candidate s*8 + grp goes to the eight lanes of group grp in step s (ds_bpermute), so there is no
// compaction in front of the loads; known candidates are read too (21.4 of 24 survive the test
// anyway) and are masked out afterwards.  Decisions, counters and results are those of fetch():
// the set of evaluated candidates, their order and the criteria they meet are unchanged (the row
// counters keep counting what the algorithm needs, not the extra rows of known candidates).
// ---------------------------------------------------------------------------------------------
constexpr int kEarlySteps = 3;  // 24 candidates
// Candidate c = 3 * grp + s goes to the eight lanes of row group grp in step s, and its verdict is
// taken in lane 8 * grp + s: ascending lanes are ascending candidates, so ONE ballot over the three
// steps lists the candidates that pass in the order the replay needs (simple_knn_cache.cuh:268-286
// evaluates them one after the other).
template <class RD>  // RD: the reader of the first rows (Prescreen<8,1,.> or DistEngine<.,8,1>)
struct EarlyRows {
  static_assert(RD::LPR == 8 && RD::NCH == 1, "early rows: 8 lanes x one 16-byte chunk per row");
  int kk[kEarlySteps];                       // candidate key of this lane's row group, per step
  typename RD::Chunk v[kEarlySteps][1];
  // cand: lane j (< 24) holds candidate j or EMPTY; translation: optional map from candidate keys
  // to base rows (upper graph layers of the merge kernel)
  GGNN_DEV void issue(const RD& rd, const int cand, const int32_t* translation = nullptr)
  {
    const int grp = threadIdx.x >> 3;
    // the three crossbar reads first (one wait for all of them, not one round trip per step)
#pragma unroll
    for (int s = 0; s < kEarlySteps; ++s)
      kk[s] = __builtin_amdgcn_ds_bpermute((grp * kEarlySteps + s) << 2, cand);
    int m[kEarlySteps];
#pragma unroll
    for (int s = 0; s < kEarlySteps; ++s) {
      // EMPTY slots read row 0 (verdict ignored): no branch and no zero-fill around the loads
      m[s] = max(kk[s], 0);
      if (translation)
        m[s] = translation[m[s]];
    }
    // ONE wave-uniform branch for "every lane's chunk lies inside the row" (the common shapes)
    // instead of an exec-mask region and a zero-fill around each of the three loads
    if (rd.all_chunks) {
#pragma unroll
      for (int s = 0; s < kEarlySteps; ++s)
        v[s][0] = rd.load_chunk(rd.row_ptr(m[s]), 0);
    }
    else {
#pragma unroll
      for (int s = 0; s < kEarlySteps; ++s) {
        if (rd.chunk_valid(0))
          v[s][0] = rd.load_chunk(rd.row_ptr(m[s]), 0);
        else
          v[s][0] = typename RD::Chunk{};
      }
    }
  }
  // the value of step (lane & 7) in lanes whose (lane & 7) < 3 (their candidate's lane).  The three
  // values are taken BY VALUE: with a reference to the array the conditional operator reads the
  // element inside each branch, the optimiser sinks the three reads into one read of a selected
  // address, and the array (kk[] of every pop) goes to scratch memory -- three scratch stores and
  // a scratch load (vmcnt(0)) per pop, measured as +19 % memory traffic and +17 % kernel time.
  template <typename T>
  static GGNN_DEV T of_my_step(const T x0, const T x1, const T x2)
  {
    const int w = threadIdx.x & 7;
    T r = x2;
    r = (w == 1) ? x1 : r;
    r = (w == 0) ? x0 : r;
    return r;
  }
};

// Pushes, in ascending lane order, the candidates of `m` (lane j: key k_of, distance d_of) that
// still beat the criteria -- the replay of simple_knn_cache.cuh:268-286 (criteria() never increases
// during a fetch, so a candidate that fails it once fails it later too)
template <class SL>
GGNN_DEV void replay_lanes(SL& sl, unsigned long long m, const int k_of, const float d_of)
{
  while (m) {
    const int j = __ffsll(static_cast<long long>(m)) - 1;
    m &= m - 1;
    const float d = rdlanef(d_of, j);
    const int k = rdlane(k_of, j);
    if (d < sl.criteria())
      sl.push(k, d);
  }
}

// second half of a pop in the early-rows order: membership test, verdicts, exact phase, replay.
// cand: lane j (< 24) holds candidate j or EMPTY (the value issue() was given).
// COUNT = false (launches that do not collect work counters: every production call): the scan of
// the sorted part moves behind the verdicts (SortedList::drop_sorted) -- same candidates on the
// float rows, same pushes, same results; the return value and `rows` then count the survivors of
// the VISITED test only and are not used.
template <int MODE, bool COUNT = true, class SL, class DE, class PS, class ER, class HOOK>
GGNN_DEV int fetch_early(SL& sl, const DE& de, const WaveLds& lds, int cand, const ER& er,
                         const PS& ps, uint2& rows, HOOK&& after_filter,
                         const int32_t* translation = nullptr)
{
  const int lane = threadIdx.x;
  const int grp = lane >> 3, w = lane & 7;
  cand = sl.template filter<false, COUNT>(lower_half_to_both(cand), lds.known);
  const unsigned surv = static_cast<unsigned>(__ballot(lane < 32 && cand != kEmptyKey));
  const int nsurv = __popc(surv);
  after_filter();
  // this lane's candidate (w < 3) survived the membership test
  const bool alive = w < kEarlySteps && ((surv >> (grp * kEarlySteps + w)) & 1u);
  // rows.x / rows.y count the rows the ALGORITHM needs, as fetch() does (code rows of the
  // survivors of the membership test while the pre-screen is active, float rows of those that pass
  // it): the roofline's algorithmic bytes do not grow because this order also requests the rows of
  // the ~11 % known candidates (measured fabric traffic shows those)
  if constexpr (PS::enabled) {
    if (nsurv == 0)
      return 0;
    // +inf (list not full yet, pre-screen unusable): nothing is dropped, every survivor is evaluated
    const float s_thr = ps.threshold(sl.criteria());
    if (s_thr < inf_f())
      rows.y += nsurv;
    float S[kEarlySteps];
#pragma unroll
    for (int s = 0; s < kEarlySteps; ++s)
      S[s] = group_sum<8>(ps.partial(er.v[s]));  // (every lane of the group holds the sum)
    bool pass = alive && !(ER::of_my_step(S[0], S[1], S[2]) >= s_thr);
    unsigned long long pm = __ballot(pass);   // ascending lanes = ascending candidates
    const int mykey = ER::of_my_step(er.kk[0], er.kk[1], er.kk[2]);
    if constexpr (!COUNT) {
      pm = sl.drop_sorted(pm, mykey);
      pass = (pm >> lane) & 1ull;
    }
    const int neval = __popcll(pm);
    if (neval == 0)
      return nsurv;
    rows.x += neval;
    constexpr int kSteps = StepsOf<DE::LPR, DE::NCH>::value;
    constexpr int kExactSteps = (DE::NCH == 3 && DE::ROWS >= 8) ? 1 : (kSteps > 2) ? 2 : kSteps;
    // keys of the candidates that pass -> LDS in candidate order (one write: the ballot is already
    // in that order), float rows, distances -> LDS, replay.  (Routing the ~3 keys through scalar
    // registers to the row groups and keeping the distances in their lanes -- no LDS between the
    // verdicts and the replay -- was measured: identical results, 10k-query batch unchanged, 100k
    // batch and build -2 %; not worth a second code path.  DESIGN.md Appendix B.)
    if (pass)
      lds.ckeys[__popcll(pm & ((1ull << lane) - 1ull))] = mykey;
    __syncthreads();
    compute_distances<MODE, DE, kExactSteps>(de, lds, neval, translation);
    __syncthreads();
    const float cd = lane < neval ? lds.cd0[lane] : inf_f();
    const int ck = lane < neval ? lds.ckeys[lane] : kEmptyKey;
    replay_lanes(sl, __ballot(cd < sl.criteria()), ck, cd);
    return nsurv;
  }
  else {
    // the requested rows ARE the base rows: distances stay in the lanes that summed them (every
    // lane of a group holds the sum), the replay reads lane 8 * grp + s for candidate 3 * grp + s
    rows.x += nsurv;
    if (nsurv == 0)
      return 0;
    float dd[kEarlySteps];
#pragma unroll
    for (int s = 0; s < kEarlySteps; ++s) {
      float a, b;
      de.template partial<MODE>(er.v[s], a, b);
      a = group_sum<8>(a);
      if (MODE == kCos)
        b = group_sum<8>(b);
      dd[s] = (MODE == kCos) ? de.finish_cos(a, b) : a;
    }
    const float dmine = ER::of_my_step(dd[0], dd[1], dd[2]);
    const int mykey = ER::of_my_step(er.kk[0], er.kk[1], er.kk[2]);
    unsigned long long m = __ballot(alive && dmine < sl.criteria());
    if constexpr (!COUNT)
      m = sl.drop_sorted(m, mykey);
    replay_lanes(sl, m, mykey, dmine);
    return nsurv;
  }
}

// hook VIS_SLOTS = <1..8>: test hook that shrinks the buckets of the hashed visited set so that the
// stash, its overflow into the ring scan and the removal paths are exercised by ordinary searches
inline uint32_t vis_slots_hook()
{
  const int64_t v = hook(kHookVisSlots);
  return (v >= 1 && v <= kVisSlots) ? static_cast<uint32_t>(v) : static_cast<uint32_t>(kVisSlots);
}

// block-size / chunk configuration by dimension and element type (host side)
struct DistConfig {
  int lpr, nch;
};
inline DistConfig pick_dist_config(uint32_t D, ggnn_dtype dtype)
{
  const uint32_t epc = dtype == GGNN_F32 ? 4 : 16;
  const uint32_t chunks = (D + epc - 1) / epc;
  if (chunks <= 8)
    return {8, 1};
  if (chunks <= 16)
    return {8, 2};
  // 17..24 chunks (D = 96 float32, the DEEP shape): 8 lanes x 3 chunks cover the row exactly; the
  // {16, 2} layout would leave 8 of its 32 chunk slots -- a quarter of every load instruction --
  // on dead lanes
  if (chunks <= 24)
    return {8, 3};
  if (chunks <= 32)
    return {16, 2};
  if (chunks <= 64)
    return {16, 4};
  if (chunks <= 256)
    return {64, 4};
  return {64, 16};
}

// dispatch a functor templated on <BaseT, LPR, NCH>
#define GGNN_DISPATCH_DIST(dtype, D, F)                                           \
  do {                                                                            \
    const ::ggnn_amd::DistConfig _dc = ::ggnn_amd::pick_dist_config((D), (dtype)); \
    if ((dtype) == GGNN_F32) {                                                    \
      if (_dc.lpr == 8 && _dc.nch == 1) { F(float, 8, 1); }                       \
      else if (_dc.lpr == 8 && _dc.nch == 2) { F(float, 8, 2); }                  \
      else if (_dc.lpr == 8 && _dc.nch == 3) { F(float, 8, 3); }                  \
      else if (_dc.lpr == 16 && _dc.nch == 2) { F(float, 16, 2); }                \
      else if (_dc.lpr == 16 && _dc.nch == 4) { F(float, 16, 4); }                \
      else if (_dc.lpr == 64 && _dc.nch == 4) { F(float, 64, 4); }                \
      else { F(float, 64, 16); }                                                  \
    }                                                                             \
    else {                                                                        \
      if (_dc.lpr == 8 && _dc.nch == 1) { F(uint8_t, 8, 1); }                     \
      else if (_dc.lpr == 8 && _dc.nch == 2) { F(uint8_t, 8, 2); }                \
      else if (_dc.lpr == 8 && _dc.nch == 3) { F(uint8_t, 8, 3); }                \
      else if (_dc.lpr == 16 && _dc.nch == 2) { F(uint8_t, 16, 2); }              \
      else if (_dc.lpr == 16 && _dc.nch == 4) { F(uint8_t, 16, 4); }              \
      else if (_dc.lpr == 64 && _dc.nch == 4) { F(uint8_t, 64, 4); }              \
      else { F(uint8_t, 64, 16); }                                                \
    }                                                                             \
  } while (0)

inline void check_vector_layout(const void* base, uint32_t D, ggnn_dtype dtype)
{
  const uint32_t epc = dtype == GGNN_F32 ? 4 : 16;
  GGNN_REQUIRE(D >= 1 && D <= 4096, GGNN_INVALID_ARGUMENT, "D must be in [1, 4096]");
  GGNN_REQUIRE(D % epc == 0, GGNN_UNSUPPORTED,
               "this build needs D to be a multiple of 16 bytes per row "
               "(4 for float32, 16 for uint8)");
  GGNN_REQUIRE((reinterpret_cast<uintptr_t>(base) & 15u) == 0, GGNN_INVALID_ARGUMENT,
               "base/query pointers must be 16-byte aligned");
}

}  // namespace ggnn_amd
