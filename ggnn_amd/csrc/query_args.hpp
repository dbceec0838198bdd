// Kernel arguments of the query kernels (query.hip).
#pragma once
#include "common.hpp"

namespace ggnn_amd {

struct QueryArgs {
  const void* base;
  const void* query;
  const int32_t* graph0;
  const int32_t* start;
  const float* nn1_stats;
  int32_t* ids;
  float* dists;
  uint32_t* n_dist;
  uint32_t* n_pop;
  uint2* n_rows;
  uint32_t D, Nq, N_base, KBuild, num_start, KQuery, sorted, cache, max_iters;
  uint32_t shards_per_gpu, on_gpu_shard;
  float tau;
  // optional pre-screen copy of the base coded for this measure (prescreen.hip); float32 only
  const uint8_t* ps_codes;
  const float* ps_params;
  uint32_t ps_Dc;
  uint32_t vis_slots;  // usable keys per bucket of the hashed visited set (kVisSlots; test hook)
  // tag-set form (long rings, traversal.hpp kTagSet): [Nq x (cache - sorted)] visited rings in
  // global memory (scratch of the launch) and the bucket bits of the set
  int32_t* ring;
  uint32_t tag_bits;
};

}  // namespace ggnn_amd
