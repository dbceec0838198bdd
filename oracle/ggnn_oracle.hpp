// TEST INFRASTRUCTURE ONLY -- CPU oracle for the GGNN hot path.
//
// This is a from-scratch CPU restatement of the algorithms of the reference
// (cgtuebingen/ggnn v0.9.1) hot path, written to be the *checker* for the HIP
// kernels in ggnn_amd/csrc.  Nothing under ggnn_amd/ may include, link or call
// it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// PARITY MOSTLY UNPINNED: the reference has no tests, no golden vectors and no
// CPU implementation of this path, and none of its translation units can be
// built in this image without stand-ins (needs nvcc, glog, CUB, nanobind).
// Pinned: (a) KBestList -- the reference's own k_best_list.cuh compiles
// unchanged with hipcc (oracle/_ref, tests/test_gpu_ref_kbest.py runs it on
// the GPU against the emulation here); (b) the GraphConfig numbers recorded in
// SURVEY.md section 8(a) row L (tests/golden/graph_config.json); (c) hand-
// derived known-answer tests written from the cited reference lines.
// Everything that depends on CUB / cuRAND ordering stays unpinned.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference).  Block-lockstep semantics are emulated phase by phase
// ("all lanes read, barrier, all lanes shift, all lanes read prev, all lanes
// insert"), including the reference quirks Q1-Q4 of SURVEY.md section 8.
#pragma once
#include <cstdint>
#include <cstddef>

extern "C" {

// include/ggnn/base/def.h:27-30
enum { ORC_EUCLIDEAN = 0, ORC_COSINE = 1 };
enum { ORC_F32 = 0, ORC_U8 = 1 };

// src/ggnn/base/graph_config.cpp:39-98 (GraphDimensions + GraphDerivedParameters)
struct OrcGraphConfig {
  uint32_t N, D, KBuild;
  uint32_t KF, G, S, S0, S0_off, SG, SG_off;
  uint32_t N_all, ST_all;
  uint32_t Bs[4], Ns[4], Ns_offsets[4], STs_offsets[4];
};
void orc_graph_config(uint32_t N, uint32_t D, uint32_t KBuild, OrcGraphConfig* out);

// include/ggnn/base/def.h:37-68 (the helpers every sizing rule is built on)
uint32_t orc_bit_ceil(uint32_t v);
uint32_t orc_next_multiple32(uint32_t v);
size_t orc_align8(size_t v);

// host sizing rules, src/ggnn/query/query_kernels.cu:55-110
struct OrcQuerySizing {
  uint32_t cache_size, sorted_size, block_dim_x;
};
int orc_query_sizing(uint32_t D, uint32_t KQuery, uint32_t max_iters, OrcQuerySizing* out);
// src/ggnn/construction/graph_construction.cu:154-161
uint32_t orc_construction_block(uint32_t D, uint32_t min_block, uint32_t* items_per_thread);

// single distance as the reference computes it (block-strided partials + CUB-style tree);
// include/ggnn/cuda_utils/distance.cuh:104-163
float orc_distance(const void* base, const void* query_row, uint32_t D, int dtype, int measure,
                   uint64_t other_id, uint32_t block, uint32_t items);

// src/ggnn/query/bf_query_layer.cu:46-64
void orc_bf_query(const void* base, uint32_t N, uint32_t D, int dtype, const void* query,
                  uint32_t Nq, uint32_t K, int measure, int32_t* out_ids, float* out_dists,
                  int threads);

// src/ggnn/query/query_layer.cu:48-90 ; one "shard".
// graph0: [N x KBuild] layer-0 rows, start: [S] starting point ids (translation[L-1]).
// n_dist / n_pop (optional) receive per-query counters.
void orc_query(const void* base, uint32_t N, uint32_t D, int dtype, const void* query,
               uint32_t Nq, const int32_t* graph0, uint32_t KBuild, const int32_t* start,
               uint32_t num_start, const float* nn1_stats, uint32_t KQuery, float tau_query,
               uint32_t max_iters, int measure, uint32_t shards_per_gpu, uint32_t on_gpu_shard,
               int32_t* out_ids, float* out_dists, uint32_t* n_dist, uint32_t* n_pop,
               int threads);

// src/ggnn/construction/top_merge_layer.cu:40-82
void orc_top(const void* base, uint32_t D, int dtype, int measure, uint32_t KBuild,
             const int32_t* translation /*layer, may be null for layer 0*/, uint32_t Nlayer,
             uint32_t S, uint32_t S_offset, uint32_t layer, int32_t* graph_layer,
             float* nn1_dist_buffer, int threads);

// src/ggnn/construction/wrs_select_layer.cu:41-102 (rng injected)
void orc_select(const OrcGraphConfig* cfg, uint32_t layer, const float* nn1_dist_buffer,
                const float* rng, int32_t* translation_all /*[ST_all]*/,
                int32_t* selection_all /*[ST_all]*/);

// src/ggnn/construction/merge_layer.cu:40-158 ; writes graph_buffer [Ns[btm] x K]
void orc_merge(const void* base, int dtype, int measure, const OrcGraphConfig* cfg,
               const int32_t* graph_all, const int32_t* translation_all,
               const int32_t* selection_all, const float* nn1_stats, float tau_build,
               uint32_t layer_top, uint32_t layer_btm, int32_t* graph_buffer,
               float* nn1_dist_buffer, uint32_t* n_dist, int threads);

// src/ggnn/construction/sym_query_layer.cu:39-145 (points processed in ascending n; one
// valid serialisation of the reference's racy schedule).  first_n/count allow partial runs.
void orc_sym(const void* base, int dtype, int measure, uint32_t D, uint32_t KBuild,
             const int32_t* graph_layer, const int32_t* translation_layer /*null on layer 0*/,
             uint32_t Nlayer, const float* nn1_stats, float tau_build, int32_t* sym_buffer,
             uint32_t* sym_atomic, uint32_t first_n, uint32_t count);

// src/ggnn/construction/sym_buffer_merge_layer.cu:36-99
void orc_sym_buffer_merge(uint32_t KBuild, uint32_t Nlayer, const int32_t* sym_buffer,
                          const uint32_t* sym_atomic, int32_t* graph_layer);

// src/ggnn/construction/graph_construction.cu:381-393 ; out = [mean, max]
void orc_nn1_stats(const float* nn1_dist_buffer, uint32_t N, float* out);

// full schedule, src/ggnn/construction/graph_construction.cu:128-147 ; rng: [4 x N] uniform
// (0,1] numbers, row l used by select(l).  graph_all [N_all x K], translation/selection [ST_all].
void orc_build(const void* base, int dtype, int measure, const OrcGraphConfig* cfg,
               float tau_build, uint32_t refinement_iterations, const float* rng,
               int32_t* graph_all, int32_t* translation_all, int32_t* selection_all,
               float* nn1_stats, int threads);

// src/ggnn/base/gpu_instance.cu:745-790 (stable ascending sort of each [shards*K] row by dist)
void orc_sort_shard_results(uint32_t Nq, uint32_t row_len, int32_t* ids, float* dists);
// src/ggnn/base/result_merger.cpp:51-149 ; parts: num_gpus arrays [Nq x K*spg]
void orc_merge_results(uint32_t Nq, uint32_t K, uint32_t num_gpus, uint32_t shards_per_gpu,
                       uint32_t N_shard, const int32_t* const* part_ids,
                       const float* const* part_dists, int32_t* out_ids, float* out_dists);

// src/ggnn/base/eval.cpp:88-242 ; out[7] = {c1,c1_dup,cK,cK_dup,rK,rK_dup,has_dup}
void orc_evaluate(const void* base, uint32_t N, const void* query, uint32_t Nq, uint32_t D,
                  int dtype, int measure, const int32_t* gt, uint32_t gt_D, uint32_t KQuery,
                  const int32_t* results, uint32_t Nres, float* out);

// KBestList (k_best_list.cuh:29-142) driven by a stream of (dist, id) pairs; checked against the
// reference's own device code where oracle/_ref is available (tests/test_gpu_ref_kbest.py)
void orc_kbest_script(uint32_t BEST, uint32_t BLOCK, const float* dists, const int32_t* ids,
                      uint32_t n, int check_worst, float* out_d, int32_t* out_i);

// ---- known-answer / design-validation helpers --------------------------------------------
// Literal emulation of SimpleKNNCache (simple_knn_cache.cuh:58-352) driven by an op script.
// ops: [n_ops x 3] int32 {op, key, dist_bits}; op 0=push(key,dist) 1=pop 2=set_xi(dist)
// 3=transform(identity map, i.e. selection[i]=i).  Returns final state.
void orc_cache_script(uint32_t BEST, uint32_t SORTED, uint32_t CACHE, uint32_t BLOCK, float xi,
                      const int32_t* ops, uint32_t n_ops, int32_t* out_keys /*CACHE*/,
                      float* out_dists /*SORTED*/, int32_t* out_pops /*n_ops*/,
                      uint32_t* out_heads /*2*/);
// Same script through the wave64 "one entry per lane, logical order" model that the HIP
// kernels implement (oracle/wave_model.hpp).  Output is converted to the physical layout so
// that it can be compared 1:1 with orc_cache_script.
void orc_wave_model_script(uint32_t BEST, uint32_t SORTED, uint32_t CACHE, float xi,
                           const int32_t* ops, uint32_t n_ops, int32_t* out_keys,
                           float* out_dists, int32_t* out_pops, uint32_t* out_heads);

// smallest relative margin seen in inexact float decisions since the last reset (sym half
// test); lets tests assert that a seeded input is "decision tie-free".
// bench cpu_baseline only: compute distances with plain multi-accumulator loops (a fair scalar/
// SIMD port) instead of the thread-by-thread emulation.  Default off (parity tests).
void orc_set_fast_distance(int enable);
// float distances of query / merge / top summed in the order of the product kernels' DistEngine
// instead of the reference's (restated) cub::BlockReduce order; see g_wave_order
void orc_set_wave_order(int enable);
// statistics: number of distance evaluations accepted (d < criteria) by orc_query since reset
uint64_t orc_accept_total(int reset);
// statistics: number of distance evaluations of orc_query / orc_merge since reset
uint64_t orc_eval_total(int reset);
void orc_margin_reset();
double orc_margin_min();

}  // extern "C"
