// TEST INFRASTRUCTURE ONLY (see ggnn_oracle.hpp).
//
// CPU model of the data structure the HIP kernels use for the reference's SimpleKNNCache
// (include/ggnn/cuda_utils/simple_knn_cache.cuh:58-352): the sorted part of the cache is kept
// "one entry per lane" in LOGICAL order -- entries [0,BEST) are the best list, entries
// [BEST,SORTED) are the priority queue starting at its head -- so the ring buffer of the
// reference becomes a plain shift.  The physical ring position only matters for quirk Q1
// (the element in physical slot SORTED-1 is never moved across the wrap), which is tracked
// through head_in (= r_prioQ_head - BEST).
//
// tests/test_wave_model.py drives this model and the literal emulation with the same random
// op scripts and requires identical physical state, i.e. it validates the kernel design on CPU.
#pragma once
#include <cstdint>
#include <limits>
#include <vector>

namespace wave_model {

constexpr int32_t EMPTY_KEY = -1;

struct WaveCache {
  uint32_t BEST, SORTED, CACHE, P, VIS;
  std::vector<int32_t> key;   // [SORTED] logical order
  std::vector<float> dist;    // [SORTED]
  std::vector<int32_t> vis;   // [VIS] physical ring (slot SORTED+i of the reference cache)
  uint32_t head_in{0};        // r_prioQ_head - BEST
  uint32_t vis_head{0};       // r0_visited_head - SORTED
  float xi{0.f};

  WaveCache(uint32_t best, uint32_t sorted, uint32_t cache)
      : BEST(best), SORTED(sorted), CACHE(cache), P(sorted - best), VIS(cache - sorted)
  {
    reset();
  }

  void reset()
  {
    key.assign(SORTED, EMPTY_KEY);
    dist.assign(SORTED, std::numeric_limits<float>::infinity());
    vis.assign(VIS, EMPTY_KEY);
    head_in = 0;
    vis_head = 0;
  }

  float criteria() const { return dist[BEST - 1] + xi; }

  // simple_knn_cache.cuh:126-213 in lane form
  void push(int32_t k, float d)
  {
    for (uint32_t i = 0; i < SORTED; ++i)
      if (key[i] == k)
        return;
    // logical index of the entry that lives in physical slot BEST (only special if wrapped)
    const uint32_t qlane = head_in ? BEST + (P - head_in) : 0xffffffffu;
    std::vector<int32_t> nk(key);
    std::vector<float> nd(dist);
    for (uint32_t i = 0; i < SORTED; ++i) {
      const bool active = dist[i] >= d;
      if (!active)
        continue;
      const bool first = (i == 0 || i == BEST);
      const bool prev_active = !first && dist[i - 1] >= d;
      if (first || !prev_active) {
        nk[i] = k;
        nd[i] = d;
      }
      else if (i == qlane || key[i - 1] == EMPTY_KEY) {
        // Q1: nothing is shifted across the physical wrap; EMPTY entries are never shifted
      }
      else {
        nk[i] = key[i - 1];
        nd[i] = dist[i - 1];
      }
    }
    key.swap(nk);
    dist.swap(nd);
  }

  // simple_knn_cache.cuh:215-239
  int32_t pop()
  {
    const int32_t k0 = key[BEST];
    const float d0 = dist[BEST];
    if (k0 == EMPTY_KEY || d0 >= criteria())
      return EMPTY_KEY;
    vis[vis_head] = k0;
    vis_head = (vis_head + 1 >= VIS) ? 0 : vis_head + 1;
    for (uint32_t i = BEST; i + 1 < SORTED; ++i) {
      key[i] = key[i + 1];
      dist[i] = dist[i + 1];
    }
    key[SORTED - 1] = EMPTY_KEY;
    dist[SORTED - 1] = std::numeric_limits<float>::infinity();
    head_in = (head_in + 1 >= P) ? 0 : head_in + 1;
    return k0;
  }

  // simple_knn_cache.cuh:297-333
  void transform(const int32_t* selection)
  {
    // the reference works on the physical layout; best entries are physical == logical
    std::vector<int32_t> nk(SORTED, EMPTY_KEY);
    std::vector<float> nd(SORTED, std::numeric_limits<float>::infinity());
    for (uint32_t i = 0; i < BEST; ++i) {
      int32_t k = key[i];
      if (k != EMPTY_KEY)
        k = selection[k];
      nk[i] = k;
      nd[i] = dist[i];
      if (i + BEST < SORTED) {
        nk[i + BEST] = k;
        nd[i + BEST] = dist[i];
      }
    }
    key.swap(nk);
    dist.swap(nd);
    vis.assign(VIS, EMPTY_KEY);
    head_in = 0;
    vis_head = 0;
  }

  bool known(int32_t k) const
  {
    for (uint32_t i = 0; i < SORTED; ++i)
      if (key[i] == k)
        return true;
    for (uint32_t i = 0; i < VIS; ++i)
      if (vis[i] == k)
        return true;
    return false;
  }

  // convert to the reference's physical layout
  void to_physical(int32_t* out_keys /*CACHE*/, float* out_dists /*SORTED*/,
                   uint32_t* heads /*2*/) const
  {
    for (uint32_t i = 0; i < SORTED; ++i) {
      const uint32_t p = (i < BEST) ? i : BEST + ((i - BEST + head_in) % P);
      out_keys[p] = key[i];
      out_dists[p] = dist[i];
    }
    for (uint32_t i = 0; i < VIS; ++i)
      out_keys[SORTED + i] = vis[i];
    heads[0] = BEST + head_in;
    heads[1] = SORTED + vis_head;
  }
};

}  // namespace wave_model
