// TEST INFRASTRUCTURE ONLY -- see ggnn_oracle.hpp for the contract ("parity unpinned").
//
// CPU restatement of the reference hot path.  File:line citations are relative to
// /root/reference.  Float arithmetic notes:
//  * per-thread partial sums use fmaf() where nvcc's default -fmad=true would contract
//    "acc += a*b" (distance.cuh:134,144-149); this file is compiled with -ffp-contract=off so
//    nothing else is contracted.
//  * cub::BlockReduce is a third-party dependency that is not part of /root/reference (CUDA
//    Toolkit >= 12, version unpinned, CMakeLists.txt:13).  Its published default algorithm
//    (BLOCK_REDUCE_WARP_REDUCTIONS: shuffle-down tree inside each warp, then lane 0 adds the
//    warp aggregates in warp order) is restated in block_reduce_sum().  The summation order
//    is NOT pinned by any reference test => float distances are compared with 1e-4 relative
//    tolerance, index parity is defined on integer-valued inputs where every order is exact.
#include "ggnn_oracle.hpp"
#include "wave_model.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

constexpr int32_t EMPTY_KEY = -1;
constexpr float INF = std::numeric_limits<float>::infinity();
constexpr uint32_t L = 4;  // graph_config.h:42-44

double g_margin_min = std::numeric_limits<double>::infinity();
// bench baseline only: plain 8-accumulator loops instead of the thread/tree emulation
// (identical results on integer-valued data, where every summation order is exact)
int g_fast_distance = 0;
// 0: the reference's summation order (thread-strided partials + cub::BlockReduce, restated);
// 1: the summation order of the product kernels' DistEngine (ggnn_amd/csrc/traversal.hpp:
//    LPR lanes per row, NCH 16-byte chunks per lane, fmaf chains per lane, pairwise tree over
//    the lanes) -- with it float results of query / merge / top can be compared BIT FOR BIT,
//    which separates "same algorithm" from "same rounding" (the reference's own order is not
//    pinned by anything, see the file header)
int g_wave_order = 0;
std::atomic<uint64_t> g_accept_total{0};
std::atomic<uint64_t> g_eval_total{0};

// include/ggnn/base/def.h:37-56
inline uint32_t bit_ceil_u32(uint32_t v)
{
  if (v <= 1)
    return 1;
  v--;
  v |= v >> 1;
  v |= v >> 2;
  v |= v >> 4;
  v |= v >> 8;
  v |= v >> 16;
  v++;
  return v;
}
// include/ggnn/base/def.h:58-62
inline uint32_t next_multiple32(uint32_t v)
{
  return v % 32 == 0 ? v : 32 * (v / 32 + 1);
}

template <typename F>
void parallel_for(uint32_t n, int threads, F&& f)
{
  // same pattern as src/ggnn/base/result_merger.cpp:131-142
  uint32_t nt = threads <= 0 ? std::thread::hardware_concurrency() : (uint32_t)threads;
  nt = std::max(1u, std::min(nt, n));
  if (nt <= 1) {
    for (uint32_t i = 0; i < n; ++i)
      f(i);
    return;
  }
  std::atomic<uint32_t> next{0};
  std::vector<std::thread> pool;
  const uint32_t chunk = std::max(1u, std::min(16u, n / (nt * 4)));
  for (uint32_t t = 0; t < nt; ++t)
    pool.emplace_back([&]() {
      for (;;) {
        const uint32_t begin = next.fetch_add(chunk);
        if (begin >= n)
          break;
        const uint32_t end = std::min(n, begin + chunk);
        for (uint32_t i = begin; i < end; ++i)
          f(i);
      }
    });
  for (auto& t : pool)
    t.join();
}

// cub::BlockReduce<float, BLOCK>::Sum restated (see file header).  v has `block` entries.
float block_reduce_sum(const float* v, uint32_t block)
{
  float total = 0.f;
  const uint32_t nw = (block + 31) / 32;
  for (uint32_t w = 0; w < nw; ++w) {
    float lane[32], tmp[32];
    for (uint32_t l = 0; l < 32; ++l)
      lane[l] = (w * 32 + l < block) ? v[w * 32 + l] : 0.f;
    const uint32_t last = std::min(32u, block - w * 32) - 1;
    for (uint32_t off = 1; off < 32; off <<= 1) {
      for (uint32_t l = 0; l < 32; ++l)
        tmp[l] = (l + off <= last) ? lane[l] + lane[l + off] : lane[l];
      std::memcpy(lane, tmp, sizeof(lane));
    }
    total = (w == 0) ? lane[0] : total + lane[0];
  }
  return total;
}

struct BaseView {
  const void* p;
  int dtype;
  uint32_t D;
  inline float at(uint64_t row, uint32_t d) const
  {
    // static_cast<ValueT>(BaseT) -- distance.cuh:131-134
    return dtype == ORC_F32 ? static_cast<const float*>(p)[row * D + d]
                            : static_cast<float>(static_cast<const uint8_t*>(p)[row * D + d]);
  }
};

// ------------------------------------------------------------------------------------------
// Row A: Distance  (include/ggnn/cuda_utils/distance.cuh:34-164)
// ------------------------------------------------------------------------------------------
  // DistEngine's order (traversal.hpp:696-860, pick_dist_config): lane g of LPR owns the 16-byte
  // chunks c*LPR + g, c < NCH, accumulated in (c, element) order; lanes are summed pairwise
  // (quad_perm, row_half_mirror, row_mirror, permlane16/32 swaps = a balanced tree in lane order)
void wave_layout(int dtype, uint32_t D, uint32_t& lpr, uint32_t& nch, uint32_t& epc)
  {
    epc = dtype == ORC_F32 ? 4 : 16;
    const uint32_t chunks = (D + epc - 1) / epc;
    if (chunks <= 8) { lpr = 8; nch = 1; }
    else if (chunks <= 16) { lpr = 8; nch = 2; }
    else if (chunks <= 24) { lpr = 8; nch = 3; }
    else if (chunks <= 32) { lpr = 16; nch = 2; }
    else if (chunks <= 64) { lpr = 16; nch = 4; }
    else if (chunks <= 256) { lpr = 64; nch = 4; }
    else { lpr = 64; nch = 16; }
  }
float tree_sum(float* v, uint32_t n)
  {
    for (uint32_t w = 1; w < n; w <<= 1)
      for (uint32_t i = 0; i + w < n; i += 2 * w)
        v[i] = v[i] + v[i + w];
    return v[0];
  }

struct DistCalc {
  BaseView base;
  int measure;
  uint32_t block, items;
  std::vector<float> q;        // r_query of all threads, indexed by dimension
  float q_norm{0.f};           // r_query_norm (thread 0)
  std::vector<float> pa, pb;   // per-thread partials
  uint64_t n_calls{0};
  uint64_t n_accepted{0};  // distance evaluations that passed the criteria (statistics only)

  DistCalc(const BaseView& b, int measure_, uint32_t block_, uint32_t items_)
      : base(b), measure(measure_), block(block_), items(items_), q(b.D), pa(block_), pb(block_)
  {
  }

  // distance.cuh:104-117
  void load_query(const BaseView& src, uint64_t row)
  {
    for (uint32_t d = 0; d < base.D; ++d)
      q[d] = src.at(row, d);
    if (measure == ORC_COSINE) {
      for (uint32_t t = 0; t < block; ++t) {
        float acc = 0.f;
        for (uint32_t item = 0; item < items; ++item) {
          const uint32_t d = item * block + t;
          const float v = d < base.D ? q[d] : 0.f;
          acc = fmaf(v, v, acc);
        }
        pa[t] = acc;
      }
      q_norm = block_reduce_sum(pa.data(), block);
      if (g_wave_order)
        q_norm = wave_query_norm();
    }
  }

  // distance.cuh:119-163
  float distance_fast(uint64_t other) const
  {
    const uint32_t D = base.D;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nrm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (base.dtype == ORC_F32) {
      const float* o = static_cast<const float*>(base.p) + other * D;
      for (uint32_t d = 0; d < D; ++d) {
        if (measure == ORC_EUCLIDEAN) {
          const float diff = o[d] - q[d];
          acc[d & 7] += diff * diff;
        }
        else {
          acc[d & 7] += o[d] * q[d];
          nrm[d & 7] += o[d] * o[d];
        }
      }
    }
    else {
      const uint8_t* o = static_cast<const uint8_t*>(base.p) + other * D;
      for (uint32_t d = 0; d < D; ++d) {
        const float ov = static_cast<float>(o[d]);
        if (measure == ORC_EUCLIDEAN) {
          const float diff = ov - q[d];
          acc[d & 7] += diff * diff;
        }
        else {
          acc[d & 7] += ov * q[d];
          nrm[d & 7] += ov * ov;
        }
      }
    }
    const float a = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    if (measure == ORC_EUCLIDEAN)
      return a;
    const float n = ((nrm[0] + nrm[1]) + (nrm[2] + nrm[3])) + ((nrm[4] + nrm[5]) + (nrm[6] + nrm[7]));
    const float norm_sqr = q_norm * n;
    return norm_sqr > 0.f ? std::fabs(1.0f - a / std::sqrt(norm_sqr)) : 1.0f;
  }

  float wave_query_norm() const
  {
    uint32_t lpr, nch, epc;
    wave_layout(base.dtype, base.D, lpr, nch, epc);
    float lane[64];
    for (uint32_t g = 0; g < lpr; ++g) {
      float nrm = 0.f;
      for (uint32_t c = 0; c < nch; ++c)
        for (uint32_t e = 0; e < epc; ++e) {
          const uint32_t d = (c * lpr + g) * epc + e;
          const float v = d < base.D ? q[d] : 0.f;
          nrm = fmaf(v, v, nrm);
        }
      lane[g] = nrm;
    }
    return tree_sum(lane, lpr);
  }
  float distance_wave(uint64_t other) const
  {
    uint32_t lpr, nch, epc;
    wave_layout(base.dtype, base.D, lpr, nch, epc);
    const uint32_t D = base.D;
    float la[64], lb[64];
    for (uint32_t g = 0; g < lpr; ++g) {
      if (base.dtype == ORC_U8) {
        // packed integer arithmetic (v_dot4_u32_u8), exact per lane
        uint32_t ab = 0, bb = 0, qq = 0;
        for (uint32_t c = 0; c < nch; ++c)
          for (uint32_t e = 0; e < epc; ++e) {
            const uint32_t d = (c * lpr + g) * epc + e;
            if (d < D) {
              const uint32_t o = static_cast<uint32_t>(base.at(other, d));
              const uint32_t qv = static_cast<uint32_t>(q[d]);
              ab += o * qv;
              bb += o * o;
              qq += qv * qv;
            }
          }
        if (measure == ORC_EUCLIDEAN) {
          la[g] = static_cast<float>((qq + bb) - 2u * ab);
          lb[g] = 0.f;
        }
        else {
          la[g] = static_cast<float>(ab);
          lb[g] = static_cast<float>(bb);
        }
        continue;
      }
      float a = 0.f, b = 0.f;
      for (uint32_t c = 0; c < nch; ++c)
        for (uint32_t e = 0; e < epc; ++e) {
          const uint32_t d = (c * lpr + g) * epc + e;
          const float o = d < D ? base.at(other, d) : 0.f;
          const float qq = d < D ? q[d] : 0.f;
          if (measure == ORC_EUCLIDEAN) {
            const float diff = o - qq;
            a = fmaf(diff, diff, a);
          }
          else {
            a = fmaf(o, qq, a);
            b = fmaf(o, o, b);
          }
        }
      la[g] = a;
      lb[g] = b;
    }
    const float a = tree_sum(la, lpr);
    if (measure == ORC_EUCLIDEAN)
      return a;
    const float nrm = tree_sum(lb, lpr);
    const float norm_sqr = q_norm * nrm;
    return norm_sqr > 0.f ? std::fabs(1.0f - a / std::sqrt(norm_sqr)) : 1.0f;
  }

  float distance(uint64_t other)
  {
    ++n_calls;
    if (g_wave_order)
      return distance_wave(other);
    if (g_fast_distance)
      return distance_fast(other);
    const uint32_t D = base.D;
    if (measure == ORC_EUCLIDEAN) {
      for (uint32_t t = 0; t < block; ++t) {
        float acc = 0.f;
        for (uint32_t item = 0; item < items; ++item) {
          const uint32_t d = item * block + t;
          const float diff = d < D ? base.at(other, d) - q[d] : 0.f;
          acc = fmaf(diff, diff, acc);
        }
        pa[t] = acc;
      }
      return block_reduce_sum(pa.data(), block);
    }
    for (uint32_t t = 0; t < block; ++t) {
      float dot = 0.f, nrm = 0.f;
      for (uint32_t item = 0; item < items; ++item) {
        const uint32_t d = item * block + t;
        if (d < D) {
          const float o = base.at(other, d);
          dot = fmaf(o, q[d], dot);
          nrm = fmaf(o, o, nrm);
        }
      }
      pa[t] = dot;
      pb[t] = nrm;
    }
    const float dot = block_reduce_sum(pa.data(), block);
    const float nrm = block_reduce_sum(pb.data(), block);
    const float norm_sqr = q_norm * nrm;
    return norm_sqr > 0.f ? std::fabs(1.0f - dot / std::sqrt(norm_sqr)) : 1.0f;
  }
};

// ------------------------------------------------------------------------------------------
// Row B: KBestList  (include/ggnn/cuda_utils/k_best_list.cuh:29-142), lockstep emulation
// ------------------------------------------------------------------------------------------
struct KBest {
  uint32_t BEST, BLOCK;
  std::vector<float> d;
  std::vector<int32_t> id;
  std::vector<float> r_dist;
  std::vector<int32_t> r_id;
  std::vector<uint8_t> cond;

  KBest(uint32_t best, uint32_t block)
      : BEST(best), BLOCK(block), d(best, INF), id(best, EMPTY_KEY), r_dist(block), r_id(block),
        cond(block)
  {
  }
  float worst() const { return d[BEST - 1]; }

  // k_best_list.cuh:77-109
  void add_unique(float dist, int32_t key)
  {
    for (uint32_t i = ((BEST - 1) / BLOCK) * BLOCK;; i -= BLOCK) {
      for (uint32_t t = 0; t < BLOCK; ++t) {
        const uint32_t k = i + t;
        if (k < BEST) {
          r_dist[t] = d[k];
          r_id[t] = id[k];
        }
      }
      // __syncthreads(); then: all shifts, all neighbour reads, all inserts
      for (uint32_t t = 0; t < BLOCK; ++t) {
        const uint32_t k = i + t;
        if (k < BEST && dist < r_dist[t] && k < BEST - 1) {
          d[k + 1] = r_dist[t];
          id[k + 1] = r_id[t];
        }
      }
      for (uint32_t t = 0; t < BLOCK; ++t) {
        const uint32_t k = i + t;
        cond[t] = (k < BEST && dist < r_dist[t]) && (!k || d[k - 1] <= dist);
      }
      for (uint32_t t = 0; t < BLOCK; ++t) {
        const uint32_t k = i + t;
        if (cond[t]) {
          d[k] = dist;
          id[k] = key;
        }
      }
      if (!i)
        break;
    }
  }
};

// ------------------------------------------------------------------------------------------
// Row C: SimpleKNNCache  (include/ggnn/cuda_utils/simple_knn_cache.cuh:31-396)
// ------------------------------------------------------------------------------------------
struct Cache {
  uint32_t BEST, SORTED, CACHE, BLOCK;
  std::vector<int32_t> key;  // s_cache
  std::vector<float> dist;   // s_dists
  uint32_t pq_head{0}, vis_head{0};
  float xi{0.f};
  // per-thread registers of push()
  std::vector<int32_t> r_key;
  std::vector<float> r_dist;
  std::vector<uint32_t> idx;
  std::vector<uint8_t> active, ins;

  Cache(uint32_t best, uint32_t sorted, uint32_t cache, uint32_t block, float xi_)
      : BEST(best), SORTED(sorted), CACHE(cache), BLOCK(block), key(cache), dist(sorted), xi(xi_),
        r_key(block), r_dist(block), idx(block), active(block), ins(block)
  {
    init();
  }

  // simple_knn_cache.cuh:73-87
  void init()
  {
    std::fill(key.begin(), key.end(), EMPTY_KEY);
    std::fill(dist.begin(), dist.end(), INF);
    pq_head = BEST;
    vis_head = SORTED;
  }

  // simple_knn_cache.cuh:121-124
  float criteria() const { return dist[BEST - 1] + xi; }

  // simple_knn_cache.cuh:126-213 (identical in simple_knn_sym_cache.cuh:290-377)
  void push(int32_t k, float d)
  {
    for (uint32_t i = 0; i < SORTED; ++i)
      if (key[i] == k)
        return;  // :131-146
    const uint32_t head = pq_head;
    const uint32_t head_in = head - BEST;
    std::fill(active.begin(), active.end(), 0);
    uint32_t block_start = ((SORTED + BLOCK - 1) / BLOCK) * BLOCK;
    for (;;) {
      // shift (:164-173)
      for (uint32_t t = 0; t < BLOCK; ++t) {
        if (active[t] && r_key[t] != EMPTY_KEY) {
          const uint32_t idx_next = (idx[t] + 1 == SORTED) ? BEST : idx[t] + 1;
          const bool has_next = idx_next != BEST && idx_next != head;  // Q1
          if (has_next) {
            key[idx_next] = r_key[t];
            dist[idx_next] = r_dist[t];
          }
        }
      }
      // find insert points (:175-178)
      for (uint32_t t = 0; t < BLOCK; ++t) {
        ins[t] = 0;
        if (active[t]) {
          const bool has_prev = idx[t] != 0 && idx[t] != head;
          const uint32_t idx_prev = idx[t] != BEST ? idx[t] - 1 : SORTED - 1;
          ins[t] = !has_prev || dist[idx_prev] < d;
        }
      }
      // insert (:179-182)
      for (uint32_t t = 0; t < BLOCK; ++t) {
        if (ins[t]) {
          key[idx[t]] = k;
          dist[idx[t]] = d;
        }
      }
      if (!block_start)
        break;
      block_start -= BLOCK;
      // read (:188-208)
      for (uint32_t t = 0; t < BLOCK; ++t) {
        uint32_t i = block_start + t;
        active[t] = i < SORTED;
        if (active[t]) {
          if (i >= BEST)
            i = (i + head_in < SORTED) ? i + head_in : i + head_in - SORTED + BEST;
          idx[t] = i;
          r_key[t] = key[i];
          r_dist[t] = dist[i];
          active[t] = r_dist[t] >= d;  // Q2: >=
        }
      }
    }
  }

  // simple_knn_cache.cuh:215-239 ; crit passed in because the sym cache uses criteria_sym()
  int32_t pop_with(float crit)
  {
    const int32_t k = key[pq_head];
    const float d = dist[pq_head];
    if (k == EMPTY_KEY || d >= crit)
      return EMPTY_KEY;
    key[vis_head] = k;
    vis_head = (vis_head + 1) >= CACHE ? SORTED : vis_head + 1;
    key[pq_head] = EMPTY_KEY;
    dist[pq_head] = INF;
    pq_head = (pq_head + 1) >= SORTED ? BEST : pq_head + 1;
    return k;
  }
  int32_t pop() { return pop_with(criteria()); }

  // filter part of fetch(), simple_knn_cache.cuh:246-261 (per-thread strided scan with the
  // early break in the visited region)
  void filter(int32_t* keys, uint32_t len) const
  {
    for (uint32_t t = 0; t < BLOCK; ++t) {
      for (uint32_t i = t; i < CACHE; i += BLOCK) {
        const int32_t n = key[i];
        if (n == EMPTY_KEY) {
          if (i >= SORTED)
            break;
          continue;
        }
        for (uint32_t k = 0; k < len; ++k)
          if (keys[k] == n)
            keys[k] = EMPTY_KEY;
      }
    }
  }
  // simple_knn_sym_cache.cuh:408-419 (no early break)
  void filter_sym(int32_t* keys, uint32_t len) const
  {
    for (uint32_t i = 0; i < CACHE; ++i) {
      const int32_t n = key[i];
      if (n == EMPTY_KEY)
        continue;
      for (uint32_t k = 0; k < len; ++k)
        if (keys[k] == n)
          keys[k] = EMPTY_KEY;
    }
  }

  // simple_knn_cache.cuh:297-333
  void transform(const int32_t* tr)
  {
    for (uint32_t i = 0; i < CACHE; ++i) {
      if (i < BEST) {
        int32_t k = key[i];
        if (k != EMPTY_KEY)
          k = tr[k];
        key[i] = k;
        if (i + BEST < SORTED) {
          key[i + BEST] = k;
          dist[i + BEST] = dist[i];
        }
      }
      else if (i < 2 * BEST && i < SORTED) {
      }
      else {
        key[i] = EMPTY_KEY;
        if (i < SORTED)
          dist[i] = INF;
      }
    }
    pq_head = BEST;
    vis_head = SORTED;
  }
};

// fetch(), simple_knn_cache.cuh:241-289: optional filter, then visit non-empty keys in array
// order, distance, push if below the (re-read) criteria.
template <bool FILTER>
void cache_fetch(Cache& c, DistCalc& dc, int32_t* keys, const int32_t* translation, uint32_t len)
{
  if (FILTER)
    c.filter(keys, len);
  for (uint32_t k = 0; k < len; ++k) {
    const int32_t other_n = keys[k];
    if (other_n == EMPTY_KEY)
      continue;
    const int32_t other_m = translation ? translation[other_n] : other_n;
    const float d = dc.distance((uint64_t)other_m);
    if (d < c.criteria()) {
      ++dc.n_accepted;
      c.push(other_n, d);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Row K (cache part): SimpleKNNSymCache distance with the half point
// (include/ggnn/cuda_utils/simple_knn_sym_cache.cuh:143-283)
// ------------------------------------------------------------------------------------------
struct SymDist {
  BaseView base;
  int measure;
  uint32_t block, items;
  std::vector<float> q, half;
  float q_norm{0.f}, half_norm{0.f};
  std::vector<float> pa, pb, pc;

  SymDist(const BaseView& b, int measure_, uint32_t block_, uint32_t items_)
      : base(b), measure(measure_), block(block_), items(items_), q(b.D), half(b.D), pa(block_),
        pb(block_), pc(block_)
  {
  }
  // :143-157
  void load_query(uint64_t row)
  {
    for (uint32_t d = 0; d < base.D; ++d)
      q[d] = base.at(row, d);
    if (measure == ORC_COSINE) {
      for (uint32_t t = 0; t < block; ++t) {
        float acc = 0.f;
        for (uint32_t item = 0; item < items; ++item) {
          const uint32_t d = item * block + t;
          const float v = d < base.D ? q[d] : 0.f;
          acc = fmaf(v, v, acc);
        }
        pa[t] = acc;
      }
      q_norm = block_reduce_sum(pa.data(), block);
    }
  }
  // :159-187 (half point and norms)
  void set_half(uint64_t other_m)
  {
    const float w = 0.5f - 0.1f;  // (0.5f - EPS), :39,171
    for (uint32_t d = 0; d < base.D; ++d)
      half[d] = fmaf(w, base.at(other_m, d) - q[d], q[d]);
    if (measure == ORC_COSINE) {
      for (uint32_t t = 0; t < block; ++t) {
        float a = 0.f, b = 0.f;
        for (uint32_t item = 0; item < items; ++item) {
          const uint32_t d = item * block + t;
          if (d < base.D) {
            a = fmaf(q[d], q[d], a);
            b = fmaf(half[d], half[d], b);
          }
        }
        pa[t] = a;
        pb[t] = b;
      }
      q_norm = block_reduce_sum(pa.data(), block);
      half_norm = block_reduce_sum(pb.data(), block);
      if (g_wave_order) {
        // SymEngine::set_half, ggnn_amd/csrc/sym.hip: per-lane fmaf chains + lane tree
        uint32_t lpr, nch, epc;
        wave_layout(base.dtype, base.D, lpr, nch, epc);
        float la[64], lb[64];
        for (uint32_t g = 0; g < lpr; ++g) {
          float a = 0.f, b = 0.f;
          for (uint32_t c = 0; c < nch; ++c)
            for (uint32_t e = 0; e < epc; ++e) {
              const uint32_t d = (c * lpr + g) * epc + e;
              const float qq = d < base.D ? q[d] : 0.f;
              const float hh = d < base.D ? half[d] : 0.f;
              a = fmaf(qq, qq, a);
              b = fmaf(hh, hh, b);
            }
          la[g] = a;
          lb[g] = b;
        }
        q_norm = tree_sum(la, lpr);
        half_norm = tree_sum(lb, lpr);
      }
    }
  }
  // :214-283
  void distance(uint64_t other, float& d_query, float& d_half)
  {
    const uint32_t D = base.D;
    if (g_wave_order) {
      // SymEngine::partial2 + finish, ggnn_amd/csrc/sym.hip
      uint32_t lpr, nch, epc;
      wave_layout(base.dtype, D, lpr, nch, epc);
      float la[64], lb[64], lc[64];
      for (uint32_t g = 0; g < lpr; ++g) {
        float a = 0.f, b = 0.f, n = 0.f;
        for (uint32_t c = 0; c < nch; ++c)
          for (uint32_t e = 0; e < epc; ++e) {
            const uint32_t d = (c * lpr + g) * epc + e;
            const float o = d < D ? base.at(other, d) : 0.f;
            const float qq = d < D ? q[d] : 0.f;
            const float hh = d < D ? half[d] : 0.f;
            if (measure == ORC_EUCLIDEAN) {
              const float dq = qq - o;
              a = fmaf(dq, dq, a);
              const float dh = hh - o;
              b = fmaf(dh, dh, b);
            }
            else {
              a = fmaf(qq, o, a);
              b = fmaf(hh, o, b);
              n = fmaf(o, o, n);
            }
          }
        la[g] = a;
        lb[g] = b;
        lc[g] = n;
      }
      d_query = tree_sum(la, lpr);
      d_half = tree_sum(lb, lpr);
      if (measure == ORC_COSINE) {
        const float n = tree_sum(lc, lpr);
        const float qn = n * q_norm;
        const float hn = n * half_norm;
        d_query = qn > 0.f ? std::fabs(1.0f - d_query / std::sqrt(qn)) : 1.0f;
        d_half = hn > 0.f ? std::fabs(1.0f - d_half / std::sqrt(hn)) : 1.0f;
      }
      return;
    }
    for (uint32_t t = 0; t < block; ++t) {
      float a = 0.f, b = 0.f, n = 0.f;
      for (uint32_t item = 0; item < items; ++item) {
        const uint32_t d = item * block + t;
        if (d < D) {
          const float o = base.at(other, d);
          if (measure == ORC_EUCLIDEAN) {
            const float dq = q[d] - o;
            a = fmaf(dq, dq, a);
            const float dh = half[d] - o;
            b = fmaf(dh, dh, b);
          }
          else {
            a = fmaf(q[d], o, a);
            b = fmaf(half[d], o, b);
            n = fmaf(o, o, n);
          }
        }
      }
      pa[t] = a;
      pb[t] = b;
      pc[t] = n;
    }
    d_query = block_reduce_sum(pa.data(), block);
    d_half = block_reduce_sum(pb.data(), block);
    if (measure == ORC_COSINE) {
      const float n = block_reduce_sum(pc.data(), block);
      const float qn = n * q_norm;
      const float hn = n * half_norm;
      d_query = qn > 0.f ? std::fabs(1.0f - d_query / std::sqrt(qn)) : 1.0f;
      d_half = hn > 0.f ? std::fabs(1.0f - d_half / std::sqrt(hn)) : 1.0f;
    }
  }
};

inline float xi_from(int measure, float nn1, float tau)
{
  // query_layer.cu:48-50 / merge_layer.cu:74-76 / sym_query_layer.cu:55-57 (same op order)
  return measure == ORC_EUCLIDEAN ? (nn1 * nn1) * tau * tau : nn1 * tau;
}

// radix-sort key order of floats as cub sorts them (CUDA 12 / CUB 2.x: -0.0 == +0.0)
inline uint32_t radix_key(float f)
{
  if (f == 0.0f)
    f = 0.0f;  // folds -0.0 to +0.0
  uint32_t b;
  std::memcpy(&b, &f, 4);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

}  // namespace

extern "C" {

void orc_set_wave_order(int enable)
{
  g_wave_order = enable;
}
void orc_set_fast_distance(int enable)
{
  g_fast_distance = enable;
}

uint64_t orc_eval_total(int reset)
{
  return reset ? g_eval_total.exchange(0) : g_eval_total.load();
}

uint64_t orc_accept_total(int reset)
{
  return reset ? g_accept_total.exchange(0) : g_accept_total.load();
}

void orc_margin_reset()
{
  g_margin_min = std::numeric_limits<double>::infinity();
}
double orc_margin_min()
{
  return g_margin_min;
}

// ------------------------------------------------------------------------------------------
// Row L: GraphConfig  (src/ggnn/base/graph_config.cpp:39-98, include/ggnn/base/graph_config.h)
// ------------------------------------------------------------------------------------------
void orc_graph_config(uint32_t N, uint32_t D, uint32_t KBuild, OrcGraphConfig* c)
{
  std::memset(c, 0, sizeof(*c));
  c->N = N;
  c->D = D;
  c->KBuild = KBuild;
  c->KF = KBuild / 2;                  // graph_config.h:56
  c->S = next_multiple32(c->KF + 1);   // graph_config.h:62
  // graph_config.cpp:71-92
  const float growth =
      std::pow(static_cast<float>(N) / static_cast<float>(c->S), 1.f / (L - 1));
  const uint32_t Gf = static_cast<uint32_t>(growth);
  const uint32_t Gc = Gf + 1;
  const float S0f = static_cast<float>(N) / (std::pow(static_cast<float>(Gf), (L - 1.0f)));
  const float S0c = static_cast<float>(N) / (std::pow(static_cast<float>(Gc), (L - 1.0f)));
  const bool is_floor =
      (static_cast<uint32_t>(S0c) < KBuild) ||
      (std::abs(S0f - static_cast<float>(c->S)) < std::abs(S0c - static_cast<float>(c->S)));
  c->G = is_floor ? Gf : Gc;
  c->S0 = is_floor ? static_cast<uint32_t>(S0f) : static_cast<uint32_t>(S0c);
  c->S0_off = N - c->G * c->G * c->G * c->S0;
  c->SG = c->S / c->G;
  c->SG_off = c->S - c->SG * c->G;
  // graph_config.cpp:39-63
  uint32_t B = 1;
  for (uint32_t l = L - 1; l != 0xffffffffu; --l, B *= c->G) {
    c->Bs[l] = B;
    c->Ns[l] = B * c->S;
  }
  c->Ns[0] = N;
  c->Ns_offsets[0] = 0;
  c->STs_offsets[0] = 0;
  c->STs_offsets[1] = 0;
  c->Ns_offsets[1] = N;
  for (uint32_t l = 2; l < L; ++l) {
    c->Ns_offsets[l] = c->Ns_offsets[l - 1] + c->Ns[l - 1];
    c->STs_offsets[l] = c->STs_offsets[l - 1] + c->Ns[l - 1];
  }
  c->N_all = c->Ns_offsets[L - 1] + c->Ns[L - 1];
  c->ST_all = c->STs_offsets[L - 1] + c->Ns[L - 1];
}

// Row E: src/ggnn/query/query_kernels.cu:55-110
int orc_query_sizing(uint32_t D, uint32_t KQuery, uint32_t max_iters, OrcQuerySizing* out)
{
  if (KQuery > 6000 || max_iters > 8192 || D > 4096)
    return 1;
  const uint32_t required_sorted = next_multiple32(KQuery + 1 + 16);
  const uint32_t cache_size =
      std::max({256u, required_sorted + 32u, bit_ceil_u32(max_iters)});
  const uint32_t cache_block = bit_ceil_u32((cache_size + 15) / 16);
  const uint32_t dim_block = bit_ceil_u32((D + 3) / 4);
  out->block_dim_x = std::max({32u, cache_block, dim_block});
  out->cache_size = cache_size;
  out->sorted_size = std::max(cache_size < 512u ? 64u : 32u, required_sorted);
  if (cache_size > 8192 || out->block_dim_x > 1024)
    return 1;
  return 0;
}

// src/ggnn/construction/graph_construction.cu:154-161
uint32_t orc_construction_block(uint32_t D, uint32_t min_block, uint32_t* items_per_thread)
{
  const uint32_t items = D <= 1024 ? 4u : 8u;
  if (items_per_thread)
    *items_per_thread = items;
  return std::max(min_block, bit_ceil_u32((D + items - 1) / items));
}

float orc_distance(const void* base, const void* query_row, uint32_t D, int dtype, int measure,
                   uint64_t other_id, uint32_t block, uint32_t items)
{
  BaseView b{base, dtype, D};
  BaseView q{query_row, dtype, D};
  DistCalc dc(b, measure, block, items);
  dc.load_query(q, 0);
  return dc.distance(other_id);
}

// ------------------------------------------------------------------------------------------
// Row F: bf_query  (src/ggnn/query/bf_query_layer.cu:39-65; block size query_kernels.cu:209-211)
// ------------------------------------------------------------------------------------------
void orc_bf_query(const void* base, uint32_t N, uint32_t D, int dtype, const void* query,
                  uint32_t Nq, uint32_t K, int measure, int32_t* out_ids, float* out_dists,
                  int threads)
{
  const uint32_t block = std::max(32u, bit_ceil_u32((D + 3) / 4));
  BaseView b{base, dtype, D};
  BaseView qv{query, dtype, D};
  if (g_fast_distance && dtype == ORC_F32 && measure == ORC_EUCLIDEAN) {
    // bench baseline (cpu_baseline of bench.py): cache-blocked scan.  A tile of base rows stays
    // in the core's L2 while a group of queries visits it, four rows share every query load and
    // the inner loop is eight independent lanes (auto-vectorised to AVX2).  Per query the rows
    // are still offered to the KBestList in base order with the reference's worst() guard
    // (bf_query_layer.cu:52-57), and the eight-lane sum + tree equals distance_fast(), so the
    // results are the ones of the plain port.
    constexpr uint32_t G = 32, T = 512;
    parallel_for((Nq + G - 1) / G, threads, [&](uint32_t grp) {
      const uint32_t q0 = grp * G, qn = std::min(G, Nq - q0);
      const float* bq = static_cast<const float*>(query);
      const float* bb = static_cast<const float*>(base);
      std::vector<KBest> bests;
      for (uint32_t g = 0; g < qn; ++g)
        bests.emplace_back(K, block);
      std::vector<float> dt(T);
      auto hsum = [](const auto& a) {
        return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      };
      for (uint32_t i0 = 0; i0 < N; i0 += T) {
        const uint32_t rows = std::min(T, N - i0);
        for (uint32_t g = 0; g < qn; ++g) {
          const float* q = bq + (size_t)(q0 + g) * D;
          uint32_t r = 0;
          for (; r + 4 <= rows; r += 4) {
            const float* p0 = bb + (size_t)(i0 + r) * D;
            const float *p1 = p0 + D, *p2 = p1 + D, *p3 = p2 + D;
            typedef float v8 __attribute__((vector_size(32), aligned(4)));
            v8 a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            uint32_t d = 0;
            for (; d + 8 <= D; d += 8) {
              const v8 qv = *reinterpret_cast<const v8*>(q + d);
              const v8 x0 = *reinterpret_cast<const v8*>(p0 + d) - qv;
              const v8 x1 = *reinterpret_cast<const v8*>(p1 + d) - qv;
              const v8 x2 = *reinterpret_cast<const v8*>(p2 + d) - qv;
              const v8 x3 = *reinterpret_cast<const v8*>(p3 + d) - qv;
              a0 += x0 * x0;
              a1 += x1 * x1;
              a2 += x2 * x2;
              a3 += x3 * x3;
            }
            for (; d < D; ++d) {
              const float qv = q[d];
              a0[d & 7] += (p0[d] - qv) * (p0[d] - qv);
              a1[d & 7] += (p1[d] - qv) * (p1[d] - qv);
              a2[d & 7] += (p2[d] - qv) * (p2[d] - qv);
              a3[d & 7] += (p3[d] - qv) * (p3[d] - qv);
            }
            dt[r] = hsum(a0);
            dt[r + 1] = hsum(a1);
            dt[r + 2] = hsum(a2);
            dt[r + 3] = hsum(a3);
          }
          for (; r < rows; ++r) {
            const float* p0 = bb + (size_t)(i0 + r) * D;
            float a0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t d = 0; d < D; ++d)
              a0[d & 7] += (p0[d] - q[d]) * (p0[d] - q[d]);
            dt[r] = hsum(a0);
          }
          KBest& best = bests[g];
          for (uint32_t t = 0; t < rows; ++t)
            if (dt[t] < best.worst())
              best.add_unique(dt[t], (int32_t)(i0 + t));
        }
      }
      for (uint32_t g = 0; g < qn; ++g)
        for (uint32_t k = 0; k < K; ++k) {
          out_ids[(size_t)(q0 + g) * K + k] = bests[g].id[k];
          out_dists[(size_t)(q0 + g) * K + k] = bests[g].d[k];
        }
    });
    return;
  }
  if (g_fast_distance) {
    // other element types / measures: every base row is reused for a group of queries
    constexpr uint32_t G = 16;
    parallel_for((Nq + G - 1) / G, threads, [&](uint32_t grp) {
      const uint32_t q0 = grp * G, qn = std::min(G, Nq - q0);
      std::vector<DistCalc> dcs;
      std::vector<KBest> bests;
      for (uint32_t g = 0; g < qn; ++g) {
        dcs.emplace_back(b, measure, block, 4);
        dcs.back().load_query(qv, q0 + g);
        bests.emplace_back(K, block);
      }
      for (uint32_t i = 0; i < N; ++i)
        for (uint32_t g = 0; g < qn; ++g) {
          const float d = dcs[g].distance_fast(i);
          if (d < bests[g].worst())
            bests[g].add_unique(d, (int32_t)i);
        }
      for (uint32_t g = 0; g < qn; ++g)
        for (uint32_t k = 0; k < K; ++k) {
          out_ids[(size_t)(q0 + g) * K + k] = bests[g].id[k];
          out_dists[(size_t)(q0 + g) * K + k] = bests[g].d[k];
        }
    });
    return;
  }
  parallel_for(Nq, threads, [&](uint32_t n) {
    DistCalc dc(b, measure, block, 4);
    dc.load_query(qv, n);
    KBest best(K, block);
    for (uint32_t i = 0; i < N; ++i) {
      const float d = dc.distance(i);
      if (d < best.worst())
        best.add_unique(d, (int32_t)i);
    }
    for (uint32_t k = 0; k < K; ++k) {
      out_ids[(size_t)n * K + k] = best.id[k];
      out_dists[(size_t)n * K + k] = best.d[k];
    }
  });
}

// ------------------------------------------------------------------------------------------
// Row D: query  (src/ggnn/query/query_layer.cu:39-97)
// ------------------------------------------------------------------------------------------
void orc_query(const void* base, uint32_t N, uint32_t D, int dtype, const void* query,
               uint32_t Nq, const int32_t* graph0, uint32_t KBuild, const int32_t* start,
               uint32_t num_start, const float* nn1_stats, uint32_t KQuery, float tau_query,
               uint32_t max_iters, int measure, uint32_t shards_per_gpu, uint32_t on_gpu_shard,
               int32_t* out_ids, float* out_dists, uint32_t* n_dist, uint32_t* n_pop,
               int threads)
{
  OrcQuerySizing sz;
  if (orc_query_sizing(D, KQuery, max_iters, &sz))
    return;
  BaseView b{base, dtype, D};
  BaseView qv{query, dtype, D};
  const float xi = xi_from(measure, nn1_stats[1], tau_query);  // Q4: max
  constexpr uint32_t K_BLOCK = 32;
  parallel_for(Nq, threads, [&](uint32_t n) {
    DistCalc dc(b, measure, sz.block_dim_x, 4);
    dc.load_query(qv, n);
    Cache cache(KQuery, sz.sorted_size, sz.cache_size, sz.block_dim_x, xi);
    {
      // fetch_unfiltered(d_starting_points, nullptr, S): keys are read 32 at a time (:268-275)
      std::vector<int32_t> sp(start, start + num_start);
      cache_fetch<false>(cache, dc, sp.data(), nullptr, num_start);
    }
    uint32_t pops = 0;
    for (uint32_t ite = 0; ite < max_iters; ++ite) {
      cache.xi = (measure == ORC_EUCLIDEAN) ? std::min(xi, cache.dist[0] * tau_query * tau_query)
                                            : std::min(xi, cache.dist[0] * tau_query);
      const int32_t anchor = cache.pop();
      if (anchor == EMPTY_KEY)
        break;
      ++pops;
      int32_t s_knn[K_BLOCK];
      for (uint32_t i = 0; i < KBuild; i += K_BLOCK) {
        for (uint32_t t = 0; t < K_BLOCK; ++t)
          s_knn[t] = (i + t < KBuild) ? graph0[(size_t)anchor * KBuild + i + t] : EMPTY_KEY;
        cache_fetch<true>(cache, dc, s_knn, nullptr, K_BLOCK);
      }
    }
    const size_t row = (size_t)n * shards_per_gpu + on_gpu_shard;
    for (uint32_t k = 0; k < KQuery; ++k) {
      // write_best with idx_offset (simple_knn_cache.cuh:344-352): EMPTY becomes -1+offset
      out_ids[row * KQuery + k] = cache.key[k] + (int32_t)(on_gpu_shard * N);
      out_dists[row * KQuery + k] = cache.dist[k];
    }
    if (n_dist)
      n_dist[n] = (uint32_t)dc.n_calls;
    if (n_pop)
      n_pop[n] = pops;
    g_accept_total.fetch_add(dc.n_accepted);
    g_eval_total.fetch_add(dc.n_calls);
  });
}

// ------------------------------------------------------------------------------------------
// Row I: top  (src/ggnn/construction/top_merge_layer.cu:40-82)
// ------------------------------------------------------------------------------------------
void orc_top(const void* base, uint32_t D, int dtype, int measure, uint32_t KBuild,
             const int32_t* translation, uint32_t Nlayer, uint32_t S, uint32_t S_offset,
             uint32_t layer, int32_t* graph_layer, float* nn1_dist_buffer, int threads)
{
  uint32_t items;
  const uint32_t block = orc_construction_block(D, 128, &items);
  BaseView b{base, dtype, D};
  parallel_for(Nlayer, threads, [&](uint32_t n) {
    const int32_t m = (!layer) ? (int32_t)n : translation[n];
    DistCalc dc(b, measure, block, items);
    dc.load_query(b, (uint64_t)m);
    KBest best(KBuild, block);
    const uint32_t S_plus_offset = S_offset * (S + 1);
    const uint32_t S_actual = (!layer && n < S_plus_offset) ? S + 1 : S;
    const uint32_t start = (layer || n < S_plus_offset)
                               ? (n / S_actual) * S_actual
                               : S_plus_offset + ((n - S_plus_offset) / S_actual) * S_actual;
    const uint32_t end = start + S_actual;
    for (uint32_t other_n = start; other_n < end; ++other_n) {
      const int32_t other_m = layer ? translation[other_n] : (int32_t)other_n;
      if (m == other_m)
        continue;
      const float d = dc.distance((uint64_t)other_m);
      best.add_unique(d, (int32_t)other_n);
    }
    for (uint32_t k = 0; k < KBuild; ++k)
      graph_layer[(size_t)n * KBuild + k] = best.id[k];
    float nn1 = best.d[1];  // Q4
    if (measure == ORC_EUCLIDEAN)
      nn1 = std::sqrt(nn1);
    nn1_dist_buffer[n] = nn1;
  });
}

// ------------------------------------------------------------------------------------------
// Row M: select  (src/ggnn/construction/wrs_select_layer.cu:41-102), rng injected
// ------------------------------------------------------------------------------------------
void orc_select(const OrcGraphConfig* cfg, uint32_t layer, const float* nn1_dist_buffer,
                const float* rng, int32_t* translation_all, int32_t* selection_all)
{
  constexpr uint32_t BLOCK = 128;
  const uint32_t S = layer ? cfg->S : cfg->S0;
  const uint32_t S_offset = layer ? 0 : cfg->S0_off;
  int32_t* sel = selection_all + cfg->STs_offsets[layer + 1];
  int32_t* tr = translation_all + cfg->STs_offsets[layer + 1];
  const int32_t* tr_layer = translation_all + cfg->STs_offsets[layer];
  struct Item {
    uint32_t rk;
    uint32_t pos;  // blocked-arrangement position (stability order)
    int32_t n;
  };
  for (uint32_t b = 0; b < cfg->Bs[layer]; ++b) {
    const uint32_t S_current = S + (b < S_offset);
    const uint32_t start = b * S + std::min(b, S_offset);
    std::vector<Item> items;
    for (uint32_t i = 0; i < 2 * BLOCK; ++i) {
      const uint32_t t = i % BLOCK, item = i / BLOCK;
      float e = -1.f;
      int32_t v = -1;
      if (i < S_current) {
        const uint32_t n = start + i;
        e = (-1 * std::log(rng[n])) /
            (nn1_dist_buffer[n] + std::numeric_limits<float>::epsilon());
        v = (int32_t)n;
      }
      items.push_back({radix_key(e), t * 2 + item, v});
    }
    std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& c) {
      if (a.rk != c.rk)
        return a.rk > c.rk;  // descending
      return a.pos < c.pos;
    });
    const uint32_t upper_segment = b / cfg->G;
    const uint32_t nth = b - upper_segment * cfg->G;
    const uint32_t num_selected = cfg->SG + (nth < cfg->SG_off);
    const uint32_t dest = upper_segment * cfg->S + nth * cfg->SG + std::min(nth, cfg->SG_off);
    for (uint32_t s = 0; s < num_selected; ++s) {
      const int32_t n = items[s].n;
      sel[dest + s] = n;
      tr[dest + s] = (!layer) ? n : tr_layer[n];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Row J: merge  (src/ggnn/construction/merge_layer.cu:40-158; sizing merge_layer.cuh:40-65)
// ------------------------------------------------------------------------------------------
void orc_merge(const void* base, int dtype, int measure, const OrcGraphConfig* cfg,
               const int32_t* graph_all, const int32_t* translation_all,
               const int32_t* selection_all, const float* nn1_stats, float tau_build,
               uint32_t layer_top, uint32_t layer_btm, int32_t* graph_buffer,
               float* nn1_dist_buffer, uint32_t* n_dist, int threads)
{
  constexpr uint32_t K_BLOCK = 32, MAX_ITERATIONS = 200, CACHE_SIZE = 256;
  const uint32_t D = cfg->D, KBuild = cfg->KBuild, S = cfg->S;
  const uint32_t SORTED = std::max(64u, next_multiple32(KBuild + 1 + 16));
  uint32_t items;
  const uint32_t block = orc_construction_block(D, 32, &items);
  BaseView b{base, dtype, D};
  const float xi = xi_from(measure, nn1_stats[0], tau_build);  // Q4: mean
  parallel_for(cfg->Ns[layer_btm], threads, [&](uint32_t un) {
    const int32_t n = (int32_t)un;
    const int32_t m = (!layer_btm) ? n : translation_all[cfg->STs_offsets[layer_btm] + n];
    DistCalc dc(b, measure, block, items);
    dc.load_query(b, (uint64_t)m);
    Cache cache(KBuild + 1, SORTED, CACHE_SIZE, block, xi);
    int32_t s_knn[K_BLOCK];
    {
      // get_top_seg_offset, :40-61
      uint32_t seg_btm = un / S;
      if (!layer_btm) {
        const uint32_t offset_points = cfg->S0_off * (cfg->S0 + 1);
        seg_btm = (un < offset_points) ? un / (cfg->S0 + 1)
                                       : cfg->S0_off + (un - offset_points) / cfg->S0;
      }
      uint32_t powG = cfg->G;
      for (uint32_t i = 1; i < layer_top - layer_btm; ++i)
        powG *= cfg->G;
      const uint32_t s_offset = (seg_btm / powG) * S;
      for (uint32_t i = 0; i < S; i += K_BLOCK) {
        for (uint32_t t = 0; t < K_BLOCK; ++t)
          s_knn[t] = (i + t < S) ? (int32_t)(s_offset + i + t) : EMPTY_KEY;
        cache_fetch<false>(cache, dc, s_knn, translation_all + cfg->STs_offsets[layer_top],
                           K_BLOCK);
      }
    }
    for (uint32_t layer = layer_top - 1; layer >= layer_btm && layer != 0xffffffffu; layer--) {
      cache.transform(selection_all + cfg->STs_offsets[layer + 1]);
      const int32_t* tr = (!layer) ? nullptr : translation_all + cfg->STs_offsets[layer];
      if (layer == layer_btm) {
        int32_t self = n;
        cache_fetch<false>(cache, dc, &self, tr, 1);
      }
      for (uint32_t ite = 0; ite < MAX_ITERATIONS; ++ite) {
        const int32_t anchor = cache.pop();
        if (anchor == EMPTY_KEY)
          break;
        for (uint32_t j = 0; j < KBuild; j += K_BLOCK) {
          for (uint32_t t = 0; t < K_BLOCK; ++t) {
            const uint32_t k = j + t;
            s_knn[t] = (k < KBuild)
                           ? graph_all[((size_t)cfg->Ns_offsets[layer] + anchor) * KBuild + k]
                           : EMPTY_KEY;
          }
          cache_fetch<true>(cache, dc, s_knn, tr, K_BLOCK);
        }
      }
    }
    // :123-145 (Q3)
    int32_t s_own_idx = -1;
    for (uint32_t k = 0; k < KBuild; ++k)
      if (cache.key[k] == n)
        s_own_idx = (int32_t)k;
    for (uint32_t k = 0; k < KBuild; ++k) {
      const int32_t idx = cache.key[k + ((int32_t)k >= s_own_idx)];
      graph_buffer[(size_t)n * KBuild + k] = (idx != EMPTY_KEY) ? idx : n;
    }
    if (!layer_btm) {
      // :147-157
      uint32_t i = (uint32_t)(s_own_idx + 1);
      float dist;
      do {
        dist = cache.dist[i];
        ++i;
      } while (dist == 0.0f && i < cache.BEST);
      if (measure == ORC_EUCLIDEAN)
        dist = std::sqrt(dist);
      nn1_dist_buffer[n] = dist;
    }
    if (n_dist)
      n_dist[n] = (uint32_t)dc.n_calls;
    g_accept_total.fetch_add(dc.n_accepted);
    g_eval_total.fetch_add(dc.n_calls);
  });
}

// ------------------------------------------------------------------------------------------
// Row K: sym  (src/ggnn/construction/sym_query_layer.cu:39-145; sizing sym_query_layer.cuh)
// ------------------------------------------------------------------------------------------
void orc_sym(const void* base, int dtype, int measure, uint32_t D, uint32_t KBuild,
             const int32_t* graph_layer, const int32_t* translation, uint32_t Nlayer,
             const float* nn1_stats, float tau_build, int32_t* sym_buffer, uint32_t* sym_atomic,
             uint32_t first_n, uint32_t count)
{
  constexpr uint32_t K_BLOCK = 32, MAX_PER_PATH_ITERATIONS = 20, CACHE_SIZE = 128;
  const uint32_t KF = KBuild / 2, KL = KBuild - KF;
  const uint32_t sorted_size = std::max(64u, next_multiple32(KBuild / 2 + 16));
  uint32_t items;
  const uint32_t block = orc_construction_block(D, 64, &items);
  BaseView b{base, dtype, D};
  const float xi = xi_from(measure, nn1_stats[0], tau_build);
  const uint32_t end = std::min(Nlayer, first_n + count);
  for (uint32_t un = first_n; un < end; ++un) {
    const int32_t n = (int32_t)un;
    SymDist sd(b, measure, block, items);
    sd.load_query((uint64_t)(translation ? translation[n] : n));
    Cache cache(KF, sorted_size, CACHE_SIZE, block, xi);
    float criteria_half = 0.f;
    for (uint32_t i = 0; i < KL; i += K_BLOCK) {
      int32_t s_sym_ids[K_BLOCK];
      for (uint32_t t = 0; t < K_BLOCK && i + t < KL; ++t)
        s_sym_ids[t] = graph_layer[(size_t)n * KBuild + i + t];
      for (uint32_t k = 0; i + k < KL && k < K_BLOCK; ++k) {
        bool connected = false;
        {
          // init_start_point, simple_knn_sym_cache.cuh:159-201
          const int32_t other_n = s_sym_ids[k];
          const int32_t other_m = translation ? translation[other_n] : other_n;
          sd.set_half((uint64_t)other_m);
          float dq, dh;
          sd.distance((uint64_t)other_m, dq, dh);
          criteria_half = dh + xi;
          cache.init();
          cache.key[0] = cache.key[cache.BEST] = other_n;
          cache.dist[0] = cache.dist[cache.BEST] = dq;
        }
        bool found = false;
        for (uint32_t ite = 0; ite < MAX_PER_PATH_ITERATIONS && !found; ++ite) {
          // pop with criteria_sym() = s_dists[0] + xi (:285-288, :387)
          const int32_t anchor = cache.pop_with(cache.dist[0] + cache.xi);
          if (anchor == EMPTY_KEY)
            break;
          int32_t s_knn[K_BLOCK];
          for (uint32_t i2 = 0; i2 < KBuild; i2 += K_BLOCK) {
            for (uint32_t t = 0; t < K_BLOCK; ++t) {
              const uint32_t k2 = i2 + t;
              if (k2 < KBuild) {
                const int32_t other_id =
                    (k2 < KL) ? graph_layer[(size_t)anchor * KBuild + k2]
                              : sym_buffer[(size_t)anchor * KF + k2 - KL];
                if (other_id == n)
                  connected = true;
                s_knn[t] = other_id;
              }
              else
                s_knn[t] = EMPTY_KEY;
            }
            if (connected) {
              found = true;
              break;
            }
            // fetch, simple_knn_sym_cache.cuh:405-436
            cache.filter_sym(s_knn, K_BLOCK);
            for (uint32_t kk = 0; kk < K_BLOCK; ++kk) {
              const int32_t other_n = s_knn[kk];
              if (other_n == EMPTY_KEY)
                continue;
              const int32_t other_m = translation ? translation[other_n] : other_n;
              float dq, dh;
              sd.distance((uint64_t)other_m, dq, dh);
              const float crit = cache.dist[0] + cache.xi;
              if (dq < crit) {
                const double mg = std::fabs((double)dh - (double)criteria_half) /
                                  std::max(1e-30, (double)std::fabs(criteria_half));
                g_margin_min = std::min(g_margin_min, mg);
              }
              if (dq < crit && dh < criteria_half)
                cache.push(other_n, dq);
            }
          }
        }
        if (!found) {
          // :121-141
          for (uint32_t i3 = 0; i3 < KF; ++i3) {
            const int32_t other_n = cache.key[i3];
            if (other_n == EMPTY_KEY)
              break;
            const uint32_t pos = sym_atomic[other_n]++;
            if (pos < KF) {
              sym_buffer[(size_t)other_n * KF + pos] = n;
              break;
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Row K (merge part): src/ggnn/construction/sym_buffer_merge_layer.cu:36-99
// ------------------------------------------------------------------------------------------
void orc_sym_buffer_merge(uint32_t KBuild, uint32_t Nlayer, const int32_t* sym_buffer,
                          const uint32_t* sym_atomic, int32_t* graph_layer)
{
  const uint32_t KF = KBuild / 2, KL = KBuild - KF;
  std::vector<int32_t> s_sym(KF), s_graph(KF);
  for (uint32_t n = 0; n < Nlayer; ++n) {
    uint32_t r_num_links = sym_atomic[n];
    for (uint32_t kf = 0; kf < KF; ++kf) {
      s_sym[kf] = sym_buffer[(size_t)n * KF + kf];
      s_graph[kf] = graph_layer[(size_t)n * KBuild + KL + kf];
    }
    for (uint32_t i = 0; i < KF; ++i) {
      bool found = r_num_links >= KF;
      if (!found) {
        const int32_t r_graph = s_graph[i];
        for (uint32_t kf = 0; kf < KF; ++kf)
          if (r_graph == s_sym[kf])
            found = true;
        if (!found) {
          s_sym[r_num_links] = r_graph;
          ++r_num_links;
        }
      }
    }
    for (uint32_t kf = 0; kf < KF; ++kf) {
      const int32_t res = s_sym[kf];
      graph_layer[(size_t)n * KBuild + KL + kf] = (res >= 0) ? res : (int32_t)n;
    }
  }
}

// Row N: src/ggnn/construction/graph_construction.cu:381-393, 79-83.
// cub::DeviceReduce's float summation order is third-party and unpinned; the sum is taken in
// float64 here (order-insensitive far below one float32 ulp), so that any summation order --
// this serial loop, the engine's tree -- rounds to the same float mean.
void orc_nn1_stats(const float* v, uint32_t N, float* out)
{
  double sum = 0.0;
  float mx = N ? v[0] : 0.f;
  for (uint32_t i = 0; i < N; ++i) {
    sum += static_cast<double>(v[i]);
    mx = std::max(mx, v[i]);
  }
  out[0] = static_cast<float>(sum / static_cast<double>(N));
  out[1] = mx;
}

// ------------------------------------------------------------------------------------------
// Row O: build/refine schedule  (src/ggnn/construction/graph_construction.cu:128-147,177-199)
// ------------------------------------------------------------------------------------------
void orc_build(const void* base, int dtype, int measure, const OrcGraphConfig* cfg,
               float tau_build, uint32_t refinement_iterations, const float* rng,
               int32_t* graph_all, int32_t* translation_all, int32_t* selection_all,
               float* nn1_stats, int threads)
{
  const uint32_t N = cfg->N, K = cfg->KBuild, KF = cfg->KF, D = cfg->D;
  std::vector<float> nn1_dist(N);
  std::vector<int32_t> graph_buffer((size_t)N * K);
  std::vector<int32_t> sym_buffer((size_t)N * KF);
  std::vector<uint32_t> sym_atomic(N);

  auto layer_graph = [&](uint32_t l) { return graph_all + (size_t)cfg->Ns_offsets[l] * K; };
  auto layer_tr = [&](uint32_t l) -> int32_t* {
    return l ? translation_all + cfg->STs_offsets[l] : nullptr;
  };
  auto do_merge = [&](uint32_t top, uint32_t btm) {
    if (top == btm) {
      orc_top(base, D, dtype, measure, K, layer_tr(btm), cfg->Ns[btm], btm ? cfg->S : cfg->S0,
              btm ? 0 : cfg->S0_off, btm, layer_graph(btm), nn1_dist.data(), threads);
    }
    else {
      orc_merge(base, dtype, measure, cfg, graph_all, translation_all, selection_all, nn1_stats,
                tau_build, top, btm, graph_buffer.data(), nn1_dist.data(), nullptr, threads);
      std::memcpy(layer_graph(btm), graph_buffer.data(),
                  (size_t)cfg->Ns[btm] * K * sizeof(int32_t));
    }
    if (!btm)
      orc_nn1_stats(nn1_dist.data(), N, nn1_stats);
  };
  auto do_sym = [&](uint32_t l) {
    std::fill(sym_buffer.begin(), sym_buffer.begin() + (size_t)cfg->Ns[l] * KF, -1);
    std::fill(sym_atomic.begin(), sym_atomic.begin() + cfg->Ns[l], 0u);
    orc_sym(base, dtype, measure, D, K, layer_graph(l), layer_tr(l), cfg->Ns[l], nn1_stats,
            tau_build, sym_buffer.data(), sym_atomic.data(), 0, cfg->Ns[l]);
    orc_sym_buffer_merge(K, cfg->Ns[l], sym_buffer.data(), sym_atomic.data(), layer_graph(l));
  };

  for (uint32_t top = 0; top < L; ++top) {
    for (uint32_t btm = top; btm != 0xffffffffu; --btm) {
      do_merge(top, btm);
      if (top < L - 1 && top == btm)
        orc_select(cfg, top, nn1_dist.data(), rng + (size_t)top * N, translation_all,
                   selection_all);
      do_sym(btm);
    }
  }
  for (uint32_t r = 0; r < refinement_iterations; ++r) {
    for (uint32_t layer = L - 2; layer != 0xffffffffu; --layer) {
      do_merge(L - 1, layer);
      do_sym(layer);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Row G: sortQueryResults  (src/ggnn/base/gpu_instance.cu:745-790) -- stable ascending radix
// order of the float distances within each row.
// ------------------------------------------------------------------------------------------
void orc_sort_shard_results(uint32_t Nq, uint32_t row_len, int32_t* ids, float* dists)
{
  std::vector<uint32_t> order(row_len);
  std::vector<int32_t> ti(row_len);
  std::vector<float> td(row_len);
  for (uint32_t n = 0; n < Nq; ++n) {
    int32_t* ri = ids + (size_t)n * row_len;
    float* rd = dists + (size_t)n * row_len;
    for (uint32_t i = 0; i < row_len; ++i)
      order[i] = i;
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t c) { return radix_key(rd[a]) < radix_key(rd[c]); });
    for (uint32_t i = 0; i < row_len; ++i) {
      ti[i] = ri[order[i]];
      td[i] = rd[order[i]];
    }
    std::memcpy(ri, ti.data(), row_len * sizeof(int32_t));
    std::memcpy(rd, td.data(), row_len * sizeof(float));
  }
}

// ------------------------------------------------------------------------------------------
// Row H: ResultMerger::merge  (src/ggnn/base/result_merger.cpp:51-149)
// ------------------------------------------------------------------------------------------
void orc_merge_results(uint32_t Nq, uint32_t K, uint32_t num_gpus, uint32_t shards_per_gpu,
                       uint32_t N_shard, const int32_t* const* part_ids,
                       const float* const* part_dists, int32_t* out_ids, float* out_dists)
{
  const uint32_t stride = K * shards_per_gpu;
  if (num_gpus == 1) {
    // :55-73 (pass-through / first K of each pre-sorted row)
    for (uint32_t n = 0; n < Nq; ++n)
      for (uint32_t k = 0; k < K; ++k) {
        out_ids[(size_t)n * K + k] = part_ids[0][(size_t)n * stride + k];
        out_dists[(size_t)n * K + k] = part_dists[0][(size_t)n * stride + k];
      }
    return;
  }
  struct KDP {
    int32_t key;
    float dist;
    uint32_t partition;
  };
  auto cmp = [](const KDP& a, const KDP& b) { return a.dist >= b.dist; };
  std::vector<KDP> heap;
  std::vector<uint32_t> part_offsets(num_gpus);
  for (uint32_t n = 0; n < Nq; ++n) {
    heap.clear();
    std::fill(part_offsets.begin(), part_offsets.end(), 1u);
    for (uint32_t g = 0; g < num_gpus; ++g) {
      const size_t pos = (size_t)n * stride;
      heap.push_back({part_ids[g][pos], part_dists[g][pos], g});
    }
    std::make_heap(heap.begin(), heap.end(), cmp);
    for (uint32_t k = 0; k < K; ++k) {
      const KDP top = heap.front();
      out_ids[(size_t)n * K + k] =
          static_cast<int32_t>(top.partition * shards_per_gpu * N_shard) + top.key;
      out_dists[(size_t)n * K + k] = top.dist;
      if (k == K - 1)
        break;
      std::pop_heap(heap.begin(), heap.end(), cmp);
      heap.pop_back();
      const size_t pos = (size_t)n * stride + part_offsets[top.partition];
      ++part_offsets[top.partition];
      heap.push_back({part_ids[top.partition][pos], part_dists[top.partition][pos],
                      top.partition});
      std::push_heap(heap.begin(), heap.end(), cmp);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Evaluator  (src/ggnn/base/eval.cpp:37-65, 88-242), including quirk Q5 (norm of `a` twice,
// sqrt for Euclidean).
// ------------------------------------------------------------------------------------------
static float eval_distance(const BaseView& a, uint64_t ra, const BaseView& b, uint64_t rb,
                           int measure)
{
  float distance = 0.f, a_norm = 0.f, b_norm = 0.f;
  for (uint32_t d = 0; d < a.D; ++d) {
    const float x = a.at(ra, d), y = b.at(rb, d);
    if (measure == ORC_EUCLIDEAN)
      distance += (x - y) * (x - y);
    else {
      distance += x * y;
      a_norm += x * x;
      b_norm += x * x;  // eval.cpp:52 uses a for both norms
    }
  }
  if (measure == ORC_EUCLIDEAN)
    return std::sqrt(distance);
  if (a_norm * b_norm > 0.f)
    return std::fabs(1.0f - distance / std::sqrt(a_norm * b_norm));
  return 1.0f;
}

void orc_evaluate(const void* base, uint32_t N, const void* query, uint32_t Nq, uint32_t D,
                  int dtype, int measure, const int32_t* gt, uint32_t gt_D, uint32_t KQuery,
                  const int32_t* results, uint32_t Nres, float* out)
{
  std::vector<uint32_t> top1End, topKEnd;
  const bool has_dup = base && query && N && Nq;
  if (has_dup) {
    BaseView b{base, dtype, D}, q{query, dtype, D};
    const float Epsilon = 0.000001f;
    for (uint32_t n = 0; n < Nq; ++n) {
      const float gt_dist1 = eval_distance(b, (uint64_t)gt[(size_t)n * gt_D], q, n, measure);
      uint32_t dup1 = 0, dupk = 0;
      for (uint32_t k = 1; k < gt_D; ++k) {
        const float dk = eval_distance(b, (uint64_t)gt[(size_t)n * gt_D + k], q, n, measure);
        if (dk - gt_dist1 > Epsilon)
          break;
        ++dup1;
      }
      top1End.push_back(1 + dup1);
      if (KQuery <= gt_D) {
        const float dK =
            eval_distance(b, (uint64_t)gt[(size_t)n * gt_D + KQuery - 1], q, n, measure);
        for (uint32_t k = KQuery; k < gt_D; ++k) {
          const float dk = eval_distance(b, (uint64_t)gt[(size_t)n * gt_D + k], q, n, measure);
          if (dk - dK > Epsilon)
            break;
          ++dupk;
        }
        topKEnd.push_back(KQuery + dupk);
      }
      else
        topKEnd.push_back(gt_D);
    }
  }
  uint32_t c1 = 0, c1_dup = 0, cK = 0, cK_dup = 0, rK = 0, rK_dup = 0;
  for (uint32_t n = 0; n < Nres; ++n) {
    const uint32_t endTop1 = has_dup ? top1End[n] : 1;
    const uint32_t endTopK = has_dup ? topKEnd[n] : KQuery;
    for (uint32_t kr = 0; kr < KQuery; ++kr) {
      const int32_t qk = results[(size_t)n * KQuery + kr];
      for (uint32_t kg = 0; kg < endTopK; ++kg) {
        if (qk == gt[(size_t)n * gt_D + kg]) {
          if (!kg) {
            if (!kr)
              ++c1;
            if (kg < KQuery)
              ++rK;
            ++rK_dup;
          }
          if (kg < endTop1 && !kr)
            ++c1_dup;
          if (kg < KQuery)
            ++cK;
          ++cK_dup;
        }
      }
    }
  }
  const float inv_q = 1.0f / static_cast<float>(Nres);
  const float inv_r = 1.0f / static_cast<float>(Nres * KQuery);
  const float nan = std::numeric_limits<float>::quiet_NaN();
  out[0] = c1 * inv_q;
  out[1] = has_dup ? c1_dup * inv_q : nan;
  out[2] = cK * inv_r;
  out[3] = has_dup ? cK_dup * inv_r : nan;
  out[4] = rK * inv_q;
  out[5] = has_dup ? rK_dup * inv_q : nan;
  out[6] = has_dup ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------
// KAT / design-validation helpers
// ------------------------------------------------------------------------------------------
static float bits_to_float(int32_t b)
{
  float f;
  std::memcpy(&f, &b, 4);
  return f;
}

// KBestList driven by a (dist, id) stream, as bf_query (check_worst) or top (no check) use it
void orc_kbest_script(uint32_t BEST, uint32_t BLOCK, const float* dists, const int32_t* ids,
                      uint32_t n, int check_worst, float* out_d, int32_t* out_i)
{
  KBest best(BEST, BLOCK);
  for (uint32_t i = 0; i < n; ++i)
    if (!check_worst || dists[i] < best.worst())
      best.add_unique(dists[i], ids[i]);
  std::memcpy(out_d, best.d.data(), BEST * sizeof(float));
  std::memcpy(out_i, best.id.data(), BEST * sizeof(int32_t));
}

void orc_cache_script(uint32_t BEST, uint32_t SORTED, uint32_t CACHE, uint32_t BLOCK, float xi,
                      const int32_t* ops, uint32_t n_ops, int32_t* out_keys, float* out_dists,
                      int32_t* out_pops, uint32_t* out_heads)
{
  Cache c(BEST, SORTED, CACHE, BLOCK, xi);
  std::vector<int32_t> ident(1 << 20);
  for (size_t i = 0; i < ident.size(); ++i)
    ident[i] = (int32_t)i;
  for (uint32_t i = 0; i < n_ops; ++i) {
    const int32_t op = ops[3 * i], key = ops[3 * i + 1];
    const float d = bits_to_float(ops[3 * i + 2]);
    out_pops[i] = -2;
    if (op == 0)
      c.push(key, d);
    else if (op == 1)
      out_pops[i] = c.pop();
    else if (op == 2)
      c.xi = d;
    else if (op == 3)
      c.transform(ident.data());
  }
  std::memcpy(out_keys, c.key.data(), CACHE * sizeof(int32_t));
  std::memcpy(out_dists, c.dist.data(), SORTED * sizeof(float));
  out_heads[0] = c.pq_head;
  out_heads[1] = c.vis_head;
}

void orc_wave_model_script(uint32_t BEST, uint32_t SORTED, uint32_t CACHE, float xi,
                           const int32_t* ops, uint32_t n_ops, int32_t* out_keys,
                           float* out_dists, int32_t* out_pops, uint32_t* out_heads)
{
  wave_model::WaveCache c(BEST, SORTED, CACHE);
  c.xi = xi;
  std::vector<int32_t> ident(1 << 20);
  for (size_t i = 0; i < ident.size(); ++i)
    ident[i] = (int32_t)i;
  for (uint32_t i = 0; i < n_ops; ++i) {
    const int32_t op = ops[3 * i], key = ops[3 * i + 1];
    const float d = bits_to_float(ops[3 * i + 2]);
    out_pops[i] = -2;
    if (op == 0)
      c.push(key, d);
    else if (op == 1)
      out_pops[i] = c.pop();
    else if (op == 2)
      c.xi = d;
    else if (op == 3)
      c.transform(ident.data());
  }
  c.to_physical(out_keys, out_dists, out_heads);
}

}  // extern "C"

extern "C" {
uint32_t orc_bit_ceil(uint32_t v)
{
  return bit_ceil_u32(v);
}
uint32_t orc_next_multiple32(uint32_t v)
{
  return next_multiple32(v);
}
size_t orc_align8(size_t v)
{
  return ((v + 7) / 8) * 8;  // include/ggnn/base/def.h:64-68
}
}
