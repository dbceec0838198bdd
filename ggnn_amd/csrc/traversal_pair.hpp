// TWO SEARCHES PER WAVE64 (round 6): device primitives of query_pair_kernel (query_pair.hip).
//
// A search of 257+ iterations gets the reference's 512+-key cache, whose sorted part (best list +
// priority queue) is 32 keys (query_kernels.cu:98-110) -- half a wave.  The one-search-per-wave
// kernels (traversal.hpp) then run every list operation (push, pop, duplicate test) and all the
// scalar bookkeeping of a pop on a half-empty wave, and such searches are the long ones: a wave's
// own dependent chain (~10 000 cycles per pop), not the memory system, bounds a 10 000-query
// batch, which does not even fit the chip in one round (7168 resident waves).
// Here lanes 0-31 run search 2n and lanes 32-63 search 2n+1 IN LOCKSTEP: every instruction serves
// both, what is a scalar of the search in traversal.hpp (criteria, queue head, counters, the popped
// key) is a value that is uniform within a half-wave, and the only wave-uniform branches are "any
// half needs this".  The state evolution of each search is the reference's (simple_knn_cache.cuh:
// 58-352, query_layer.cu:48-90) exactly as in SortedList<1, tag set, ring-less>: same candidate
// order, same criteria at every acceptance, same quirks (Q1 ring wrap of the queue, Q2 ties), same
// counters.
//
// Layout of a pop (the early-rows order of traversal.hpp):
//   peek (per half) -> graph row (speculated) -> REQUEST the code rows (or the uint8 rows) of all
//   <= 24 neighbours: 4 row groups of 8 lanes per half, 6 steps -> bookkeeping of the pop and
//   membership test under that latency -> verdicts -> float rows (2 or 4 rows per half and step)
//   -> replay of the acceptances, both halves' pushes in the same loop iterations.
// Visited set: the 16-bit tag set of traversal.hpp ("long rings") without a ring -- used only when
// the search cannot wrap its ring (max_iterations <= cache - sorted) and every key of the shard
// fits nb_bits + 16 bits; buckets of 4 or 8 tags, a count byte per bucket, a 32-entry stash, an
// overflow list in global memory (normally empty).
#pragma once
#include "traversal.hpp"

namespace ggnn_amd {

constexpr int kHalf = 32;

// value that lane `idx` (0..31, wave-uniform) of the caller's OWN half-wave holds
GGNN_DEV int half_bcast(int v, int idx, bool upper)
{
  const int a = rdlane(v, idx), b = rdlane(v, idx + kHalf);
  return upper ? b : a;
}
GGNN_DEV float half_bcast(float v, int idx, bool upper)
{
  const float a = rdlanef(v, idx), b = rdlanef(v, idx + kHalf);
  return upper ? b : a;
}
// this half's 32 bits of a wave-wide ballot
GGNN_DEV unsigned half_of(unsigned long long m, bool upper)
{
  return upper ? static_cast<unsigned>(m >> 32) : static_cast<unsigned>(m);
}
GGNN_DEV bool half_any(bool p, bool upper)
{
  return half_of(__ballot(p), upper) != 0u;
}

// LDS of ONE search (ints), two of these per wave:
//   known[32] | ckeys[32] | cd0[32] | stash[32] | tags[2^NB x SLOTS x 2 B] | counts[2^NB B] | query row
template <int NB, int SLOTS>
struct PairLayout {
  static_assert(SLOTS == 4 || SLOTS == 8, "buckets of 4 or 8 tags");
  static constexpr int kKnown = 0, kCkeys = 32, kCd0 = 64, kStash = 96, kTags = 128;
  static constexpr int kTagInts = (1 << NB) * SLOTS / 2;
  static constexpr int kCnt = kTags + kTagInts;
  static constexpr int kQrow = kCnt + (1 << NB) / 4;
  static constexpr size_t ints(size_t qrow_bytes)
  {
    return kQrow + qrow_bytes / 4;
  }
};

// Sorted part (one entry per lane of the half: SORTED == 32) + ring-less tag set of one search per
// half-wave.  Counterpart of SortedList<1, -NB, true>; act / go arguments are per-half predicates.
template <int NB, int SLOTS>
struct PairList {
  using L = PairLayout<NB, SLOTS>;
  int key;
  float dist;
  int BEST, P;      // wave-uniform (both searches of a launch share KQuery)
  int slots;        // usable tags per bucket (test hook VIS_SLOTS)
  float xi;         // per half
  int head_in;      // per half: r_prioQ_head - BEST
  int stash_n;      // per half
  int ovf_n;        // per half: keys in the overflow list (global memory)
  int* lds;         // this half's LDS region
  int* ovf;         // this search's overflow list
  int li;           // lane within the half
  bool upper;

  GGNN_DEV void init(int best, float xi_, int* lds_, int* ovf_, int usable_slots)
  {
    li = threadIdx.x & (kHalf - 1);
    upper = threadIdx.x >= kHalf;
    BEST = best;
    P = kHalf - best;
    slots = usable_slots < SLOTS ? usable_slots : SLOTS;
    xi = xi_;
    lds = lds_;
    ovf = ovf_;
    key = kEmptyKey;
    dist = inf_f();
    head_in = 0;
    stash_n = 0;
    ovf_n = 0;
    // counts to zero; tags need no clearing (masked by the counts).  ckeys[0] is read by lanes
    // without a row of their own before anything was written there: a valid key
    for (int i = li; i < ((1 << NB) >> 2); i += kHalf)
      lds[L::kCnt + i] = 0;
    if (li == 0)
      lds[L::kCkeys] = 0;
    __syncthreads();
  }

  GGNN_DEV float dist_at(int i) const { return half_bcast(dist, i, upper); }
  GGNN_DEV int key_at(int i) const { return half_bcast(key, i, upper); }
  // simple_knn_cache.cuh:121-124
  GGNN_DEV float criteria() const { return dist_at(BEST - 1) + xi; }

  static GGNN_DEV uint32_t tag_hash(uint32_t k)
  {
    const uint32_t h = k * kTagMul;
    return NB + 16 >= 32 ? h : (h & ((1u << (NB + 16)) - 1u));
  }

  // simple_knn_cache.cuh:126-213 in lane form (SortedList::push_step with R = 1, SORTED = 32);
  // k, d, act: uniform within the half
  GGNN_DEV void push(int k, float d, bool act)
  {
    act = act && !half_any(act && key == k, upper);
    // whole-wave shift: lane 32 receives search A's last entry, but entry 0 never looks left
    const int pk = lane_up1(key);
    const float pd = lane_up1(dist);
    // logical index of the entry in physical slot BEST; Q1: nothing shifts into it
    const int qlane = head_in ? BEST + (P - head_in) : -1;
    const bool first = (li == 0) || (li == BEST);
    const bool active = act && (dist >= d);
    const bool prev_active = !first && (pd >= d);
    if (active) {
      if (first || !prev_active) {
        key = k;
        dist = d;
      }
      else if (li != qlane && pk != kEmptyKey) {
        key = pk;
        dist = pd;
      }
    }
  }

  // k (uniform within the half) has just been popped: SortedList::tag_insert
  GGNN_DEV void tag_insert(int k, bool act)
  {
    const uint32_t h = tag_hash(static_cast<uint32_t>(k));
    const uint32_t b = h >> 16;
    const unsigned short t = static_cast<unsigned short>(h & 0xffffu);
    unsigned short* tg = reinterpret_cast<unsigned short*>(lds + L::kTags);
    unsigned char* cn = reinterpret_cast<unsigned char*>(lds + L::kCnt);
    const int c = cn[b];
    const unsigned short mine = tg[b * SLOTS + (li & (SLOTS - 1))];
    // a key can be popped more than once (quirk Q1 duplicates a queue entry): kept once
    const bool ins = act && !half_any(act && li < c && mine == t, upper);
    const bool to_bucket = ins && c < slots;
    if (to_bucket && li == 0) {
      tg[b * SLOTS + c] = t;
      cn[b] = static_cast<unsigned char>(c + 1);
    }
    const bool spill = ins && !to_bucket;
    if (__any(spill)) {  // rare: bucket full
      const bool to_stash = spill && stash_n < kVisStash;
      if (to_stash && li == 0)
        lds[L::kStash + stash_n] = k;
      stash_n += to_stash ? 1 : 0;
      const bool to_ovf = spill && !to_stash;
      if (to_ovf && li == 0)
        ovf[ovf_n] = k;  // (at most one entry per pop, pops <= ring length)
      ovf_n += to_ovf ? 1 : 0;
    }
  }

  // bookkeeping of a pop whose key was decided by the caller (simple_knn_cache.cuh:225-238)
  GGNN_DEV void pop_commit(int k0, bool go)
  {
    tag_insert(k0, go);
    // whole-wave shift: lane 31 receives search B's first entry, but the last entry is emptied
    const int nk = lane_down1(key);
    const float nd = lane_down1(dist);
    if (go && li >= BEST) {
      const bool last = li == kHalf - 1;
      key = last ? kEmptyKey : nk;
      dist = last ? inf_f() : nd;
    }
    const int nh = (head_in + 1 >= P) ? 0 : head_in + 1;
    head_in = go ? nh : head_in;
  }

  // membership test of simple_knn_cache.cuh:246-261: every lane tests ITS OWN candidate against
  // the sorted keys, the tag set, the stash and the overflow list of its half
  GGNN_DEV int filter(int cand) const
  {
    __syncthreads();
    lds[L::kKnown + li] = key;
    __syncthreads();
    const unsigned c = static_cast<unsigned>(cand);
    auto fold = [c](unsigned acc, const int4& e) {
      acc = min(min(acc, static_cast<unsigned>(e.x) ^ c), static_cast<unsigned>(e.y) ^ c);
      return min(min(acc, static_cast<unsigned>(e.z) ^ c), static_cast<unsigned>(e.w) ^ c);
    };
    unsigned acc0 = 0xffffffffu, acc1 = 0xffffffffu;
    // the tag probe first: its two reads travel while the sorted keys are folded
    const uint32_t hh = tag_hash(c);
    const uint32_t b = hh >> 16;
    const uint32_t tt = (hh & 0xffffu) * 0x10001u;
    const int v = reinterpret_cast<const unsigned char*>(lds + L::kCnt)[b];
    auto pair = [tt, v](unsigned a, int wv, int slot) {
      const unsigned x = static_cast<unsigned>(wv) ^ tt;
      const unsigned lo = slot < v ? (x & 0xffffu) : 1u;
      const unsigned hi = slot + 1 < v ? (x >> 16) : 1u;
      return min(a, min(lo, hi));
    };
    if constexpr (SLOTS == 8) {
      const int4 w = *reinterpret_cast<const int4*>(lds + L::kTags + b * 4);
      acc1 = pair(pair(pair(pair(acc1, w.x, 0), w.y, 2), w.z, 4), w.w, 6);
    }
    else {
      const int2 w = *reinterpret_cast<const int2*>(lds + L::kTags + b * 2);
      acc1 = pair(pair(acc1, w.x, 0), w.y, 2);
    }
    const int4* kp = reinterpret_cast<const int4*>(lds + L::kKnown);
#pragma unroll
    for (int t = 0; t < 8; t += 2) {
      acc0 = fold(acc0, kp[t]);
      acc1 = fold(acc1, kp[t + 1]);
    }
    // stash / overflow list: normally empty; the loops run to the larger count of the two halves
    const int smax = max(rdlane(stash_n, 0), rdlane(stash_n, kHalf));
    for (int t = 0; t < smax; ++t) {
      const unsigned x = static_cast<unsigned>(lds[L::kStash + t]) ^ c;
      acc0 = min(acc0, t < stash_n ? x : 0xffffffffu);
    }
    const int omax = max(rdlane(ovf_n, 0), rdlane(ovf_n, kHalf));
    if (omax) {
      // the keys were stored by lane 0 of the half: the vector L1 is invalidated first
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int t = 0; t < omax; ++t) {
        const unsigned x = static_cast<unsigned>(ovf[t < ovf_n ? t : 0]) ^ c;
        acc1 = min(acc1, t < ovf_n ? x : 0xffffffffu);
      }
    }
    return (min(acc0, acc1) == 0u) ? kEmptyKey : cand;
  }
};

// First-read rows of up to 4 x STEPS candidates per half (STEPS = 6: the <= 24 neighbours of a
// graph row; 8: a chunk of 32 start points).  Candidate c = STEPS * grp + s of a half goes to the
// eight lanes of its row group grp (0..3) in step s and is judged in lane 8 * grp + s of the half:
// ascending lanes are ascending candidates, so one ballot lists both halves' passing candidates in
// the order the replay needs.
template <class RD, int STEPS>
struct PairRows {
  static_assert(RD::LPR == 8 && RD::NCH == 1, "8 lanes x one 16-byte chunk per row");
  typename RD::Chunk v[STEPS][1];
  // lane of `cand` that holds the candidate this lane judges (valid for (lane & 7) < STEPS)
  static GGNN_DEV int my_candidate_lane()
  {
    const int lane = threadIdx.x;
    return (lane & kHalf) + ((lane & (kHalf - 1)) >> 3) * STEPS + (lane & 7);
  }
  // cand: lane li (< 4 * STEPS) of each half holds candidate li of its search or EMPTY
  GGNN_DEV void issue(const RD& rd, const int cand)
  {
    const int lane = threadIdx.x;
    const int src0 = (lane & kHalf) + ((lane & (kHalf - 1)) >> 3) * STEPS;
    int kk[STEPS];
    // the crossbar reads first (one wait for all of them)
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
      kk[s] = __builtin_amdgcn_ds_bpermute((src0 + s) << 2, cand);
    // EMPTY slots read row 0 (verdict ignored): no branch and no zero-fill around the loads
    if (rd.all_chunks) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s)
        v[s][0] = rd.load_chunk(rd.row_ptr(max(kk[s], 0)), 0);
    }
    else {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        if (rd.chunk_valid(0))
          v[s][0] = rd.load_chunk(rd.row_ptr(max(kk[s], 0)), 0);
        else
          v[s][0] = typename RD::Chunk{};
      }
    }
  }
  // x[lane & 7] for lanes with (lane & 7) < STEPS (constant indices after unrolling: registers)
  template <typename T>
  static GGNN_DEV T of_my_step(const T (&x)[STEPS])
  {
    const int w = threadIdx.x & 7;
    T r = x[STEPS - 1];
#pragma unroll
    for (int s = STEPS - 2; s >= 0; --s)
      r = (w == s) ? x[s] : r;
    return r;
  }
};

// replay of simple_knn_cache.cuh:268-286 for both halves: candidates of `m` (lane j: key k_of,
// distance d_of) in ascending lane order per half, the criteria re-read after every push
template <class PL>
GGNN_DEV void replay_pair(PL& sl, unsigned long long m, const int k_of, const float d_of)
{
  unsigned mlo = static_cast<unsigned>(m), mhi = static_cast<unsigned>(m >> 32);
  while (mlo | mhi) {
    const int ja = mlo ? __ffs(static_cast<int>(mlo)) - 1 : 0;
    const int jb = (mhi ? __ffs(static_cast<int>(mhi)) - 1 : 0) + kHalf;
    const float da = mlo ? rdlanef(d_of, ja) : inf_f();
    const float db = mhi ? rdlanef(d_of, jb) : inf_f();
    const int ka = rdlane(k_of, ja), kb = rdlane(k_of, jb);
    mlo &= mlo - 1;
    mhi &= mhi - 1;
    const float d = sl.upper ? db : da;
    const int k = sl.upper ? kb : ka;
    sl.push(k, d, d < sl.criteria());
  }
}

// distances of the neval compacted candidates of each half (keys in ckeys[0, neval)) -> cd0
template <int MODE, class DE, class L, int STEPS>
GGNN_DEV void pair_distances(const DE& de, int* lds, const int neval, const int nmax)
{
  constexpr int ROWS = kHalf / DE::LPR;  // rows per half and step
  using Chunk = typename DE::Chunk;
  const int li = threadIdx.x & (kHalf - 1);
  const int grp = li / DE::LPR;
  float* cd0 = reinterpret_cast<float*>(lds + L::kCd0);
  for (int s0 = 0; s0 < nmax; s0 += ROWS * STEPS) {
    Chunk v[STEPS][DE::NCH];
    int rr[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s > 0 && s0 + s * ROWS >= nmax)
        break;  // wave-uniform: no rows left for this and the following steps
      const int r = s0 + s * ROWS + grp;
      const bool valid = r < neval;
      rr[s] = valid ? r : -1;
      // slots past the end read the row of an earlier candidate (cached, result never stored)
      const int m = lds[L::kCkeys + (valid ? r : 0)];
      const auto* row = de.row_ptr(m);
#pragma unroll
      for (int c = 0; c < DE::NCH; ++c) {
        if (de.all_chunks || de.chunk_valid(c))
          v[s][c] = de.load_chunk(row, c);
        else
          v[s][c] = ChunkOf<typename DE::Base>::zero();
      }
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s0 + s * ROWS >= nmax)
        break;
      if (s > 0)
        asm volatile("" ::: "memory");  // keep the branch: the later steps are usually empty
      float a, b;
      de.template partial<MODE>(v[s], a, b);
      a = group_sum<DE::LPR>(a);
      if (MODE == kCos)
        b = group_sum<DE::LPR>(b);
      if (rr[s] >= 0 && de.g == 0)
        cd0[rr[s]] = (MODE == kCos) ? de.finish_cos(a, b) : a;
    }
  }
}

// per-half work counters (uniform within the half)
struct PairCounters {
  uint32_t n_dist, n_pop, float_rows, code_rows;
};

// fetch() of simple_knn_cache.cuh:241-289 for both halves, rows already requested into `er`.
// cand: lane li (< 4 * STEPS) of each half holds candidate li of its search or EMPTY.
template <int MODE, bool FILTER, int STEPS, class PL, class DE, class PS, class ER, class HOOK>
GGNN_DEV void fetch_pair(PL& sl, const DE& de, int cand, const ER& er, const PS& ps,
                         PairCounters& cnt, HOOK&& after_filter)
{
  using L = typename PL::L;
  const int li = sl.li;
  const int grp = li >> 3, w = li & 7;
  if constexpr (FILTER)
    cand = sl.filter(cand);
  const unsigned long long surv = __ballot(cand != kEmptyKey);
  const unsigned surv_h = half_of(surv, sl.upper);
  const int nsurv = __popc(surv_h);
  after_filter();
  if (surv == 0ull)
    return;
  cnt.n_dist += nsurv;
  // this lane's candidate (w < STEPS) survived the membership test
  const bool alive = w < STEPS && ((surv_h >> (grp * STEPS + w)) & 1u);
  const int mykey = __builtin_amdgcn_ds_bpermute(ER::my_candidate_lane() << 2, cand);
  if constexpr (PS::enabled) {
    // +inf (list not full yet, pre-screen unusable): nothing is dropped, every survivor is evaluated
    const float s_thr = ps.threshold(sl.criteria());
    cnt.code_rows += (s_thr < inf_f()) ? nsurv : 0;
    float S[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
      S[s] = group_sum<8>(ps.partial(er.v[s]));  // (every lane of the group holds the sum)
    const bool pass = alive && !(ER::of_my_step(S) >= s_thr);
    const unsigned long long pm = __ballot(pass);  // ascending lanes = ascending candidates
    if (pm == 0ull)
      return;
    const unsigned pm_h = half_of(pm, sl.upper);
    const int neval = __popc(pm_h);
    cnt.float_rows += neval;
    const int nmax = max(__popc(static_cast<unsigned>(pm)), __popc(static_cast<unsigned>(pm >> 32)));
    constexpr int kSteps = StepsOf<DE::LPR, DE::NCH>::value;
    constexpr int kExactSteps = (DE::NCH == 3) ? 1 : (kSteps > 2) ? 2 : kSteps;
    if (pass)
      sl.lds[L::kCkeys + __popc(pm_h & ((1u << li) - 1u))] = mykey;
    __syncthreads();
    pair_distances<MODE, DE, L, kExactSteps>(de, sl.lds, neval, nmax);
    __syncthreads();
    const float cd = li < neval ? reinterpret_cast<const float*>(sl.lds + L::kCd0)[li] : inf_f();
    const int ck = li < neval ? sl.lds[L::kCkeys + li] : kEmptyKey;
    replay_pair(sl, __ballot(cd < sl.criteria()), ck, cd);
  }
  else {
    // the requested rows ARE the base rows: distances stay in the lanes that summed them
    cnt.float_rows += nsurv;
    float dd[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      float a, b;
      de.template partial<MODE>(er.v[s], a, b);
      a = group_sum<8>(a);
      if (MODE == kCos)
        b = group_sum<8>(b);
      dd[s] = (MODE == kCos) ? de.finish_cos(a, b) : a;
    }
    const float dmine = ER::of_my_step(dd);
    replay_pair(sl, __ballot(alive && dmine < sl.criteria()), mykey, dmine);
  }
}

}  // namespace ggnn_amd
