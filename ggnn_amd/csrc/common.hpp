// Shared host/device definitions of the MI355X GGNN engine (libggnn_amd.so).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../include/ggnn_c.h"

namespace ggnn_amd {

constexpr int kWave = 64;
constexpr int32_t kEmptyKey = -1;
constexpr uint32_t kLayers = 4;  // reference: graph_config.h:42-44
constexpr uint32_t kKBlock = 32; // reference: K_BLOCK in query_layer.cu:42, merge_layer.cu:67

struct Error : std::runtime_error {
  ggnn_status status;
  Error(ggnn_status s, const std::string& msg) : std::runtime_error(msg), status(s) {}
};

#define GGNN_HIP_CHECK(expr)                                                                  \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      throw ::ggnn_amd::Error(_e == hipErrorOutOfMemory ? GGNN_OUT_OF_MEMORY                  \
                                                        : GGNN_DEVICE_ERROR,                  \
                              std::string(#expr) + ": " + hipGetErrorString(_e));            \
  } while (0)

#define GGNN_REQUIRE(cond, status, msg)        \
  do {                                         \
    if (!(cond))                               \
      throw ::ggnn_amd::Error((status), (msg)); \
  } while (0)

// include/ggnn/base/def.h:37-62 of the reference (host helpers, restated)
inline uint32_t bit_ceil_u32(uint32_t v)
{
  if (v <= 1)
    return 1;
  --v;
  v |= v >> 1;
  v |= v >> 2;
  v |= v >> 4;
  v |= v >> 8;
  v |= v >> 16;
  return v + 1;
}
inline uint32_t next_multiple32(uint32_t v)
{
  return (v + 31u) / 32u * 32u;
}
inline size_t align8(size_t s)
{
  return (s + 7) / 8 * 8;
}

// The AQL dispatch packet stores the grid size per dimension in WORK-ITEMS as 32 bits, so a 1-D
// launch of more than 2^32 / block_threads workgroups is silently truncated (one wave per point
// breaks beyond 2^26 points).  Large launches therefore use a 2-D grid; kernels recover the
// linear workgroup index with block_linear_index() and guard against the rounded-up tail.
constexpr uint32_t kMaxGridX = 1u << 20;
inline dim3 grid_for(uint64_t blocks)
{
  if (blocks <= kMaxGridX)
    return dim3(static_cast<uint32_t>(blocks ? blocks : 1));
  return dim3(kMaxGridX, static_cast<uint32_t>((blocks + kMaxGridX - 1) / kMaxGridX));
}
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t block_linear_index()
{
  return blockIdx.y * gridDim.x + blockIdx.x;
}
// XCD-aware block -> item mapping for kernels whose consecutive items share data (the construction
// kernels: consecutive points share their neighbourhoods).  Workgroup b runs on XCD b % 8 (observed,
// MI355X_MICROARCH.md; only speed depends on it) and every XCD has its own L2: with the identity
// mapping each XCD works on every 8th point and all eight L2s fetch the same rows.  Here XCD x gets
// the CONTIGUOUS items [x * per, (x + 1) * per), per = ceil(n / 8); launch xcd_grid_blocks(n)
// workgroups and drop the indices >= n.
__device__ __forceinline__ uint32_t xcd_contiguous_index(uint32_t b, uint32_t n, bool enabled)
{
  if (!enabled)
    return b;
  const uint32_t per = (n + 7u) >> 3;
  const uint32_t i = b >> 3;
  return i < per ? (b & 7u) * per + i : 0xffffffffu;
}
#endif
inline uint64_t xcd_grid_blocks(uint64_t n, bool enabled)
{
  return enabled ? ((n + 7) / 8) * 8 : n;
}

extern int g_log_level;
#define GGNN_LOG(level, ...)                 \
  do {                                       \
    if (::ggnn_amd::g_log_level >= (level)) { \
      std::fprintf(stderr, "[ggnn_amd] ");   \
      std::fprintf(stderr, __VA_ARGS__);     \
      std::fprintf(stderr, "\n");            \
    }                                        \
  } while (0)

// ---- host-side launchers implemented in the .hip files ---------------------------------------
struct QueryLaunch {
  const void* base;
  const void* query;
  ggnn_dtype dtype;
  uint32_t N_base, D, Nq;
  const int32_t* graph0;
  uint32_t KBuild;
  const int32_t* start;
  uint32_t num_start;
  const float* nn1_stats;
  uint32_t k_query;
  float tau_query;
  uint32_t max_iterations;
  ggnn_measure measure;
  uint32_t shards_per_gpu, on_gpu_shard;
  int32_t* ids;
  float* dists;
  uint32_t* n_dist;
  uint32_t* n_pop;
  // optional pre-screen copy of the base (launch_prescreen_encode, same measure); ignored unless float32
  const uint8_t* ps_codes{nullptr};
  const float* ps_params{nullptr};
  uint32_t ps_Dc{0};
  uint32_t* n_rows{nullptr};  // optional [Nq x 2]: float rows and code rows read per query
};
void launch_query(const QueryLaunch& a, hipStream_t stream);

// 8-bit pre-screen copy of a float32 base for one distance measure (prescreen.hip): codes
// [N x Dc], Dc = D rounded up to 16; params [prescreen_param_floats(D)] floats; scratch
// [prescreen_scratch_floats(N, D, measure)] floats.
constexpr int kPsHeaderFloats = 8;
// Code-row length: rows must not straddle more memory lines than their length needs -- a
// 96-byte code row at a 96-byte pitch touches 2.5 64-byte granules on average instead of 2 and a
// 128-byte line boundary three times in four (profiles/r03_d96: 1.40x the algorithmic bytes on the
// fabric).  Up to 64 dimensions the pitch is a power of two (several rows per line, none split),
// above that a multiple of 64 bytes; padding codes are zero on both sides of every difference.
inline uint32_t prescreen_code_dim(uint32_t D)
{
  if (D <= 64) {
    uint32_t p = 16;
    while (p < D)
      p <<= 1;
    return p;
  }
  return (D + 63u) / 64u * 64u;
}
size_t prescreen_param_floats(uint32_t D);
size_t prescreen_scratch_floats(uint32_t N, uint32_t D, ggnn_measure measure);
void launch_prescreen_encode(const float* base, uint32_t N, uint32_t D, ggnn_measure measure,
                             uint8_t* codes, float* params, float* scratch, hipStream_t stream);
void launch_prescreen_probe(const uint8_t* codes, const float* params, uint32_t D,
                            ggnn_measure measure, const float* query, uint32_t Nq,
                            const int32_t* cand, uint32_t M, const float* crit, int32_t* reject,
                            float* s_out, hipStream_t stream);

struct BfLaunch {
  const void* base;
  const void* query;
  ggnn_dtype dtype;
  uint32_t N_base, D, Nq, k_query;
  ggnn_measure measure;
  int32_t* ids;
  float* dists;
  // optional (device): number of queries the matrix-core path could not certify and answered
  // with the scan kernel instead (0 on the scan path)
  uint32_t* n_rescanned{nullptr};
};
void launch_bf_query(const BfLaunch& a, hipStream_t stream);

struct TopLaunch {
  const void* base;
  ggnn_dtype dtype;
  uint32_t D;
  ggnn_measure measure;
  uint32_t KBuild;
  const int32_t* translation;
  uint32_t N_layer, S, S_offset, layer;
  int32_t* graph_layer;
  float* nn1_dist_buffer;
};
void launch_top(const TopLaunch& a, hipStream_t stream);

struct MergeLaunch {
  const void* base;
  ggnn_dtype dtype;
  ggnn_measure measure;
  ggnn_graph_config cfg;
  const int32_t* graph_all;
  const int32_t* translation_all;
  const int32_t* selection_all;
  const float* nn1_stats;
  float tau_build;
  uint32_t layer_top, layer_btm;
  int32_t* graph_buffer;
  float* nn1_dist_buffer;
  uint32_t* n_dist;
  // optional pre-screen copy of the base (launch_prescreen_encode, same measure); ignored unless float32
  const uint8_t* ps_codes{nullptr};
  const float* ps_params{nullptr};
  uint32_t ps_Dc{0};
  // optional [N_btm x 4]: distance evaluations, float rows, code rows, pops per point
  uint32_t* n_work{nullptr};
};
void launch_merge(const MergeLaunch& a, hipStream_t stream);

struct SymLaunch {
  const void* base;
  ggnn_dtype dtype;
  ggnn_measure measure;
  uint32_t D, KBuild;
  const int32_t* graph_layer;
  const int32_t* translation;
  uint32_t N_layer;
  const float* nn1_stats;
  float tau_build;
  int32_t* sym_buffer;
  uint32_t* sym_atomic;
  uint32_t first_n, count;
  // optional pre-screen copy of the base (launch_prescreen_encode, same measure); ignored unless float32
  const uint8_t* ps_codes{nullptr};
  const float* ps_params{nullptr};
  uint32_t ps_Dc{0};
  // optional [N_layer x 4]: distance evaluations, float rows, code rows, pops per point
  uint32_t* n_work{nullptr};
};
void launch_sym(const SymLaunch& a, hipStream_t stream);

void launch_select(const ggnn_graph_config& cfg, uint32_t layer, const float* nn1_dist_buffer,
                   const float* rng, int32_t* translation_all, int32_t* selection_all,
                   hipStream_t stream);
void launch_uniform(float* out, uint32_t n, uint64_t seed, uint64_t stream_id,
                    hipStream_t stream);
void launch_sym_buffer_merge(uint32_t KBuild, uint32_t N_layer, int32_t* sym_buffer,
                             const uint32_t* sym_atomic, int32_t* graph_layer,
                             hipStream_t stream);
constexpr uint32_t kStatsBlocks = 1024;
void launch_nn1_stats(const float* nn1, uint32_t N, float* scratch, float* out,
                      hipStream_t stream);
void launch_sort_shard_results(uint32_t Nq, uint32_t row_len, int32_t* ids, float* dists,
                               hipStream_t stream);
void launch_merge_results(uint32_t Nq, uint32_t k, uint32_t num_parts, uint32_t stride,
                          uint32_t id_offset_per_part, const int32_t* parts_ids,
                          const float* parts_dists, int32_t* ids_out, float* dists_out,
                          hipStream_t stream);

// merge_results for the queries qlist[0, *qcount) only (device memory; null = all queries)
void launch_merge_results_subset(uint32_t Nq, uint32_t k, uint32_t num_parts, uint32_t stride,
                                 uint32_t id_offset_per_part, const int32_t* parts_ids,
                                 const float* parts_dists, int32_t* ids_out, float* dists_out,
                                 const uint32_t* qlist, const uint32_t* qcount,
                                 hipStream_t stream);

// ... and for the queries [first, first + count) only (every GPU of a multi-GPU handle merges its
// own slice of the query set); outputs are indexed by the query number.  part_elems: distance
// (in elements) between the rows of consecutive parts, 0 = Nq * stride (parts stored back to
// back); the packed exchange keeps every part's ids and distances in one block of 2 * Nq * stride
void launch_merge_results_range(uint32_t Nq, uint32_t k, uint32_t num_parts, uint32_t stride,
                                uint32_t id_offset_per_part, const int32_t* parts_ids,
                                const float* parts_dists, int32_t* ids_out, float* dists_out,
                                const uint32_t* qlist, const uint32_t* qcount, uint32_t first,
                                uint32_t count, hipStream_t stream, size_t part_elems = 0);

// stream-ordered scratch of one launch from a private, bounded pool per device (scratch.cpp)
void* scratch_alloc(size_t bytes, hipStream_t stream);
void scratch_free(void* p, hipStream_t stream);

// host layout math (graph_config.cpp)
void graph_config_init(uint32_t N, uint32_t D, uint32_t KBuild, ggnn_graph_config* out);
void query_sizing(uint32_t D, uint32_t k_query, uint32_t max_iterations, uint32_t* cache_size,
                  uint32_t* sorted_size);
uint32_t merge_sorted_size(uint32_t KBuild);
uint32_t sym_sorted_size(uint32_t KBuild);

}  // namespace ggnn_amd
