/*
 * ggnn_c.h -- C-ABI of the MI355X-native GGNN query+build engine (libggnn_amd.so).
 *
 * The reference (cgtuebingen/ggnn v0.9.1) has no C-ABI: its nanobind module wraps the C++
 * class ggnn::GGNN<int32_t,float> directly (src/ggnn/python/nanobind.cu:184-268).  This header
 * is the thin C boundary a binding for that class binds instead; every entry point cites the
 * reference interface it replaces (paths relative to the reference tree).
 *
 *  - Section 1 mirrors the public class           include/ggnn/base/ggnn.cuh:42-182
 *  - Section 2 mirrors the internal operator seam  include/ggnn/query/query_kernels.cuh:33-61
 *                                                  include/ggnn/construction/graph_construction.cuh:35-59
 *    (one function per kernel, device pointers + HIP stream; used by the engine itself and by
 *    the per-kernel parity tests with injected inputs)
 *
 * Conventions: plain pointers and sizes only.  Every call returns a ggnn_status; the message
 * of the last failure is available through ggnn_last_error().  API misuse maps to the
 * exceptions the reference throws (INVALID_STATE/INVALID_ARGUMENT -> std::runtime_error,
 * OUT_OF_RANGE -> std::out_of_range); see INTEGRATION.md.
 * Distances are squared L2 (no sqrt) or |1 - cos|, ids are int32, as in the reference.
 */
#ifndef GGNN_C_H
#define GGNN_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GGNN_OK = 0,
  GGNN_INVALID_ARGUMENT = 1, /* std::runtime_error / glog CHECK in the reference */
  GGNN_INVALID_STATE = 2,    /* std::runtime_error ("base needs to be set", ...) */
  GGNN_OUT_OF_RANGE = 3,     /* std::out_of_range (ggnn.cu:96) */
  GGNN_OUT_OF_MEMORY = 4,
  GGNN_DEVICE_ERROR = 5,
  GGNN_UNSUPPORTED = 6,
  GGNN_IO_ERROR = 7
} ggnn_status;

/* include/ggnn/base/def.h:27-30 */
typedef enum { GGNN_EUCLIDEAN = 0, GGNN_COSINE = 1 } ggnn_measure;
/* include/ggnn/base/dataset.cuh (DataType): base / query element type */
typedef enum { GGNN_F32 = 0, GGNN_U8 = 1 } ggnn_dtype;
/* include/ggnn/base/data.cuh (MemoryLocation) reduced to what the boundary needs */
typedef enum { GGNN_CPU = 0, GGNN_GPU = 1 } ggnn_location;

typedef struct ggnn_handle ggnn_t;

/* ------------------------------------------------------------------------------------------
 * Section 1: ggnn::GGNN<int32_t,float>
 * ---------------------------------------------------------------------------------------- */

/* GGNN::GGNN() ggnn.cu:415-418 */
ggnn_status ggnn_create(ggnn_t** out);
void ggnn_destroy(ggnn_t* h);
/* message of the last failed call on this handle (or of ggnn_create when h == NULL) */
const char* ggnn_last_error(const ggnn_t* h);
const char* ggnn_version(void);

/* setWorkingDirectory ggnn.cuh:66-71 */
ggnn_status ggnn_set_working_directory(ggnn_t* h, const char* dir);
/* setCPUMemoryLimit ggnn.cuh:72-77 (accepted; all shards stay resident in 288 GB HBM) */
ggnn_status ggnn_set_cpu_memory_limit(ggnn_t* h, size_t memory_limit);
/* setReservedGPUMemory ggnn.cuh:79-84 */
ggnn_status ggnn_set_reserved_gpu_memory(ggnn_t* h, size_t reserved_memory);
/* setGPUs ggnn.cuh:86-98, ggnn.cu:89-100 (same range check, incl. quirk Q5) */
ggnn_status ggnn_set_gpus(ggnn_t* h, const int* gpu_ids, size_t num_gpus);
/* setShardSize ggnn.cuh:100-106 */
ggnn_status ggnn_set_shard_size(ggnn_t* h, uint32_t n_shard);
/* setReturnResultsOnGPU ggnn.cuh:108-114 */
ggnn_status ggnn_set_return_results_on_gpu(ggnn_t* h, int return_results_on_gpu);

/* setBase / setBaseReference ggnn.cuh:116-128, ggnn.cu:456-497.
 * take_copy != 0: the engine copies the data (setBase with an owning dataset, and what the
 * Python binding does, nanobind.cu:102-110).  take_copy == 0: borrow (setBaseReference); the
 * caller keeps the memory alive.  Host data is staged to the device at build()/bf_query(). */
ggnn_status ggnn_set_base(ggnn_t* h, const void* data, uint64_t N, uint32_t D, ggnn_dtype dtype,
                          ggnn_location location, int gpu_id, int take_copy);

/* build ggnn.cuh:130-137, ggnn.cu:205-240 */
ggnn_status ggnn_build(ggnn_t* h, uint32_t k_build, float tau_build,
                       uint32_t refinement_iterations, ggnn_measure measure);
/* store / load ggnn.cuh:138-147, ggnn.cu:242-276 ; files <workdir>/part_<shard>.ggnn with the
 * reference's pool layout (graph.cpp:48-91) */
ggnn_status ggnn_store(ggnn_t* h);
ggnn_status ggnn_load(ggnn_t* h, uint32_t k_build);

/* query ggnn.cuh:149-160, ggnn.cu:518-541.
 * ids_out/dists_out: [Nq x KQuery] (or [Nq x KQuery*shards] unmerged when results are returned
 * on the GPU, ggnn.cuh:108-113) in memory of kind out_location. */
ggnn_status ggnn_query(ggnn_t* h, const void* query, uint64_t Nq, uint32_t D, ggnn_dtype dtype,
                       ggnn_location location, int gpu_id, uint32_t k_query, float tau_query,
                       uint32_t max_iterations, ggnn_measure measure, int32_t* ids_out,
                       float* dists_out, ggnn_location out_location);

/* Extension for serving (no counterpart in the reference, whose query() blocks,
 * gpu_instance.cu:687-712): enqueue one batch and return.  Batches given different `slot`s run on
 * different streams, so the thin tail of one batch's launch overlaps with the next batch.
 * Handle on one GPU: `query`, `ids_out`, `dists_out` are device memory on that GPU (gpu_id),
 * rows 16-byte aligned; the results are the sorted [Nq, k_query * shards] rows of
 * results-on-GPU mode.
 * Handle on several GPUs (ggnn_set_gpus): `query` is device memory of GPU gpu_id (any GPU of
 * the node) or, with gpu_id < 0, page-locked host memory; it is copied to every GPU on the
 * slot's stream.  The results are the MERGED [Nq, k_query] arrays (local searches, one packed
 * RCCL all-gather, per-GPU slice merges: replaces ggnn.cu:308-329 + result_merger.cpp:51-149),
 * written by asynchronous copies: ids_out / dists_out must be device memory or page-locked host
 * memory (pageable memory works but makes the copies synchronous).
 * Everything is valid after ggnn_synchronize() / ggnn_synchronize_slot(slot); `query`,
 * `ids_out` and `dists_out` must stay alive and untouched until then.
 * Ordering with ggnn_query(): both may be used on one handle from one thread; a blocking call
 * has its own stream and staging and does not wait for batches in flight.  A call whose
 * measure differs from the one the pre-screen copy was coded for drains every slot first. */
ggnn_status ggnn_query_async(ggnn_t* h, const void* query, uint64_t Nq, uint32_t D,
                             ggnn_dtype dtype, int gpu_id, uint32_t k_query, float tau_query,
                             uint32_t max_iterations, ggnn_measure measure, int32_t* ids_out,
                             float* dists_out, uint32_t slot);
ggnn_status ggnn_synchronize(ggnn_t* h);
/* wait for the batches of one slot only (slots are taken modulo the number of streams, 4) */
ggnn_status ggnn_synchronize_slot(ggnn_t* h, uint32_t slot);

/* bfQuery ggnn.cuh:162-172, ggnn.cu:543-564 */
ggnn_status ggnn_bf_query(ggnn_t* h, const void* query, uint64_t Nq, uint32_t D,
                          ggnn_dtype dtype, ggnn_location location, int gpu_id, uint32_t k_gt,
                          ggnn_measure measure, int32_t* ids_out, float* dists_out,
                          ggnn_location out_location);

/* layout of one graph shard, include/ggnn/base/graph_config.h:31-112 */
typedef struct {
  uint32_t N, D, KBuild;
  uint32_t KF, G, S, S0, S0_off, SG, SG_off;
  uint32_t N_all, ST_all;
  uint32_t Bs[4], Ns[4], Ns_offsets[4], STs_offsets[4];
} ggnn_graph_config;

/* getGraph ggnn.cuh:174-175 + Graph (graph.h:38-71): device views of one shard, valid until
 * the handle is destroyed.  graph [N_all x KBuild], translation/selection [ST_all],
 * nn1_stats [2] = {mean, max}. */
typedef struct {
  ggnn_graph_config config;
  const int32_t* graph;
  const int32_t* translation;
  const int32_t* selection;
  const float* nn1_stats;
  int gpu_id;
} ggnn_graph_view;
ggnn_status ggnn_get_graph(ggnn_t* h, uint32_t global_shard_id, ggnn_graph_view* out);

/* tracing (reference: cudaEvent timings printed through glog, gpu_instance.cu:536-545,687-712).
 * Sum of the HIP-event durations (ms) of the kernels of the last build / query / bf_query. */
ggnn_status ggnn_last_timing_ms(const ggnn_t* h, float* build_ms, float* query_ms, float* bf_ms);
/* per-query work counters of the last ggnn_query (sum over queries and shards): number of
 * distance evaluations and of successful pops; used for the roofline figure. */
ggnn_status ggnn_last_query_counters(const ggnn_t* h, uint64_t* n_dist, uint64_t* n_pop);
/* how the last ggnn_query combined the per-GPU results (replaces the D2H + CPU heap merge of
 * ggnn.cu:308-329 / result_merger.cpp:51-149): "none" (one GPU), "rccl" (grouped ncclAllGather
 * over xGMI + per-GPU slice merge) "copy" (peer copies to the first GPU: contexts sharing one
 * device, or no librccl) or "gather" (hook EXCHANGE = 3).  Hook EXCHANGE (ggnn_set_hook) forces
 * one of them. */
const char* ggnn_last_exchange(const ggnn_t* h);
/* ranks of the RCCL communicator the handle's exchange runs on (ncclCommCount of its first
 * communicator): the number of GPUs of the handle once an "rccl" exchange has run, 0 before that,
 * without librccl, or after a fallback to peer copies.  A multi-GPU benchmark line that says
 * "rccl" must carry this number equal to its GPU count. */
ggnn_status ggnn_rccl_ranks(const ggnn_t* h, uint32_t* ranks);
/* number of half-batches the last blocking multi-GPU ggnn_query was searched in: 2 when the search
 * of the second half overlapped the exchange and merge of the first (hook QUERY_SPLIT), else 1 */
ggnn_status ggnn_last_query_parts(const ggnn_t* h, uint32_t* parts);
/* queries of the last ggnn_bf_query that were answered by the exhaustive scan because the
 * matrix-core pre-selection could not be certified exact (tracing; results are exact either way) */
ggnn_status ggnn_last_bf_query_rescanned(const ggnn_t* h, uint32_t* n_rescanned);
ggnn_status ggnn_set_collect_counters(ggnn_t* h, int enable);
/* rows the last ggnn_query read for those evaluations (needs collect_counters): float rows
 * (4*D bytes each) and, with the pre-screen, 8-bit code rows (prescreen_code_dim(D) bytes each: a power of two up to 64, multiples of 64 above). */
ggnn_status ggnn_last_query_rows_read(const ggnn_t* h, uint64_t* float_rows, uint64_t* code_rows);
/* Work of the construction kernels of the last ggnn_build, summed over every launch and shard
 * (needs ggnn_set_collect_counters before the build; the build then synchronises after every
 * merge / sym launch to time it with its own HIP events and to read the per-point counters, so it
 * is a diagnostic mode).  For SURVEY 8(d)'s build roofline ("sum over kernel launches"):
 * bytes of a kernel = float_rows x 4D + code_rows x prescreen_code_dim(D) + pops x graph-row bytes
 * (merge: KBuild x 4; sym: (KBuild + KBuild/2) x 4: graph row + pending inverse links)
 * + points x (own row 4D + result row). */
typedef struct {
  uint64_t launches, points;   /* kernel launches and points (= waves) they processed */
  uint64_t n_dist;             /* distance evaluations of the reference's algorithm */
  uint64_t float_rows;         /* 4D-byte rows actually read */
  uint64_t code_rows;          /* pre-screen code rows read */
  uint64_t pops;               /* graph rows read */
  double ms;                   /* sum of the launches' HIP-event durations */
} ggnn_kernel_work;
typedef struct {
  ggnn_kernel_work merge, sym;
} ggnn_build_work;
ggnn_status ggnn_last_build_work(const ggnn_t* h, ggnn_build_work* out);
/* Exact pre-screen of the float32 query and merge kernels (no reference counterpart; results
 * are identical with it on or off, see ggnn_op_prescreen_encode).  On by default; hook PRESCREEN = 0
 * (ggnn_set_hook) turns the default of handles created afterwards off. */
ggnn_status ggnn_set_prescreen(ggnn_t* h, int enable);
/* Deterministic build (no reference counterpart; the reference's build is not reproducible:
 * cuRAND stream graph_construction.cu:96-102,168-169, atomics and cross-block reads in sym
 * sym_query_layer.cu:102-104,124-141).  `rng` (host, may be null with n_rng = 0): uniform (0,1]
 * numbers for the selection kernel, [3][N_shard], layer l of every shard reads rng[l * N_shard
 * + n] instead of the engine's generator; `serial_sym` != 0 launches the sym kernel one point
 * at a time in ascending order.  With both, ggnn_build follows one fixed serialisation of
 * GraphConstructionImpl::build/refine (graph_construction.cu:128-147) that a CPU restatement can
 * be compared with bit for bit.  Slow; off by default. */
ggnn_status ggnn_set_build_hooks(ggnn_t* h, const float* rng, uint64_t n_rng, int serial_sym);
/* shard layout chosen by ggnn_build / ggnn_load (GGNNImpl::prepare, ggnn.cu:154-203): total
 * number of shards, shards per GPU (the width factor of results returned on the GPU,
 * ggnn.cu:299-306) and points per shard.  GGNN_INVALID_STATE without a graph. */
ggnn_status ggnn_get_shard_layout(const ggnn_t* h, uint32_t* num_shards, uint32_t* shards_per_gpu,
                                  uint32_t* n_shard);
/* tracing: peak engine clock the device reports (hipDeviceAttributeClockRate), in Hz; used to
 * turn SQ cycle counters into utilisation figures instead of assuming a nominal clock */
ggnn_status ggnn_device_clock_hz(int device, double* clock_hz);
/* nanobind.cu:151 set_log_level */
void ggnn_set_log_level(int level);

/* Test and tuning hooks (no reference counterpart): process-wide named integers that select a
 * code path or a tuning constant.  NONE changes a result -- every value yields the same ids and
 * distances; they exist so that tests can force rarely taken paths and so that A/B measurements
 * need no rebuild.  Precedence: ggnn_set_hook > the environment variable GGNN_<NAME>, which is
 * consulted ONLY while GGNN_TEST_HOOKS=1 is set (a production process that merely inherits such a
 * variable is not steered by it) > the default.  Unknown name: GGNN_INVALID_ARGUMENT.
 *
 *   name             default  values
 *   PRESCREEN           1     default of ggnn_set_prescreen for handles created afterwards
 *   EXCHANGE            0     0 auto | 1 ("rccl") RCCL all-gather, also on one GPU (1-rank world)
 *                             | 2 ("copy") peer copies to the first GPU | 3 ("gather", blocking
 *                             calls) every GPU gathers all rows by peer copies and merges its
 *                             1/G slice: the RCCL path's structure without RCCL, for handles whose
 *                             contexts share a device
 *   SYM_PRESCREEN      -1     sym kernel with the pre-screen: -1 auto (rows >= 1 KB) | 0 | 1
 *   SHARD_OVERLAP       1     0 = resident shards of one GPU searched one launch at a time
 *   VIS_SLOTS           8     usable keys per bucket of the hashed visited set, 1..8 (small values
 *                             exercise stash, overflow and removal paths)
 *   VIS_TAG_SET         1     0 = visited rings of 481..2016 keys (searches of 513..2048 iterations)
 *                             are scanned instead of probed through the 16-bit tag set
 *   QUERY_SPLIT        -1     blocking ggnn_query on several GPUs as two half-batches in flight:
 *                             -1 auto (from 4096 queries) | 0 never | 1 always (from 2 queries)
 *   RESIDENT_SHARDS     0     GPU slots per device for its shards: 0 = decided from the free device
 *                             memory (all resident whenever they fit); n < shards per GPU forces the
 *                             out-of-core mode (shards take turns: GPU <-> pinned host <-> part files)
 *   XCD_MAP             3     bit 0: merge kernel, bit 1: sym kernel -- workgroups of one XCD take a
 *                             contiguous range of points (consecutive points share neighbourhoods)
 *   BF_POOL_KEEP_MB  1024     bytes the private bf_query scratch pool keeps between calls
 *   BF_NO_I8            0     1 = uint8 bf_query through the float matrix-core kernels
 *   BF_I8_V1            0     1 = LDS-list i8 kernel instead of the register-set one
 *   BF_SLICES           0     0 auto | base slices per query block (1..64)
 *   BF_NO_CENTER        0     1 = float32 rows not shifted by the column mean (exercises the re-scan)
 *   BF_TILES            3     2 | 3 | 4 base tiles per accumulator group (D > 128)
 *   BF_I8_NOSHARE       0     1 = slices of the i8 kernel do not share their bound
 *   BF_I8_RANKS        -1     bit mask of the K-best set positions the slices of the i8 kernel
 *                             exchange (bit i = the i-th of {0,1,2,4,9} for sets of 10); -1 = auto
 *                             (ONE position chosen from the number of slices: the default), 31 = all
 *                             five, 16 = only the last entry (the single shared bound of rounds 3-4)
 *   BF_I8_REFRESH      64     stages of 128 rows between those exchanges (before that: at stages
 *                             1, 2, 4, ...)
 *   BF_I8_SEED          0     rows of an optional seeding launch of the i8 kernel (the K-th best
 *                             distance over the head of the base as every slice's first bound);
 *                             0 = none (measured: the launch costs what its fewer hits save)
 *   BF_SCAN             0     1 = scan kernels instead of the matrix-core brute force
 *   RCCL_FAIL_AFTER     0     fault injection: the n-th multi-GPU exchange of the process reports an
 *                             RCCL failure (exercises the peer-copy fallback); 0 = never
 *   QUERY_EARLY         1     query kernel, graphs with KBuild <= 24: 1 = the first-read rows of a
 *                             pop's neighbours (pre-screen codes, or rows of <= 128 bytes) are
 *                             requested BEFORE the pop's bookkeeping and the membership test;
 *                             0 = after them (the order of rounds 1-4; same results)
 *   MERGE_EARLY         1     the same switch for the merge kernel
 *   QUERY_LDS_PAD       0     extra bytes of LDS per wave of the early-rows query kernels (lowers the
 *                             occupancy: measurements of its effect without a rebuild)
 *   QUERY_GLOBAL_RING   1     early-rows query kernels whose search cannot wrap its visited ring
 *                             (max_iterations <= ring length): 1 = no ring at all, the hashed set's
 *                             buckets and stash ARE the visited keys (overflow list in global
 *                             memory); 0 = ring in LDS mirrored by the set
 *   MERGE_COUNTING      0     merge launches WITHOUT work counters (every production build) run a
 *                             kernel that tests a candidate against the sorted part of the cache
 *                             only once it has passed the pre-screen; 1 = they run the counting
 *                             kernel, which scans the sorted part for every candidate first (same
 *                             graph; A/B and test hook) */
ggnn_status ggnn_set_hook(const char* name, int64_t value);
/* back to environment / default */
ggnn_status ggnn_reset_hook(const char* name);
/* the value a use of the hook would see right now */
ggnn_status ggnn_get_hook(const char* name, int64_t* value);

/* ------------------------------------------------------------------------------------------
 * Section 2: operator seam (device pointers, explicit stream).  `stream` is a hipStream_t.
 * ---------------------------------------------------------------------------------------- */

/* GraphConfig(params) src/ggnn/base/graph_config.cpp:30-113 (pure host) */
ggnn_status ggnn_graph_config_init(uint32_t N, uint32_t D, uint32_t KBuild,
                                   ggnn_graph_config* out);
/* host sizing of the query kernel, src/ggnn/query/query_kernels.cu:55-110 */
ggnn_status ggnn_query_sizing(uint32_t D, uint32_t k_query, uint32_t max_iterations,
                              uint32_t* cache_size, uint32_t* sorted_size);

/* QueryKernels::query  query_kernels.cu:50-186 -> query_layer.cu:39-97 (one shard).
 * graph0 [N_base x KBuild], start [num_start] (= translation[L-1]), nn1_stats [2].
 * Writes ids/dists rows (n*shards_per_gpu + on_gpu_shard) of width k_query, ids offset by
 * on_gpu_shard*N_base.  n_dist/n_pop: optional per-query counters [Nq]. */
ggnn_status ggnn_op_query(const void* base, ggnn_dtype dtype, uint32_t N_base, uint32_t D,
                          const void* query, uint32_t Nq, const int32_t* graph0,
                          uint32_t KBuild, const int32_t* start, uint32_t num_start,
                          const float* nn1_stats, uint32_t k_query, float tau_query,
                          uint32_t max_iterations, ggnn_measure measure,
                          uint32_t shards_per_gpu, uint32_t on_gpu_shard, int32_t* ids,
                          float* dists, uint32_t* n_dist, uint32_t* n_pop, void* stream);

/* Pre-screen copy of a float32 base (no reference counterpart; an exact pruning aid of
 * query_layer.cu:69-77 / simple_knn_cache.cuh:268-286): every row is stored a second time as
 * 8-bit codes c with x^_d = o_d + s*c_d plus a bound e_max >= max_rows ||x - x^|| (cosine: of the
 * row normalised to unit length).  During a float32 query or merge a candidate whose lower
 * bound (||q^ - x^|| - e_q - e_max)^2 already reaches the criteria is dropped without reading
 * its float row -- exactly the candidates the reference evaluates and then discards -- so ids,
 * distances and counters do not change.  The copy is specific to the measure it was made for.
 * code_dim = prescreen_code_dim(D) (a power of two up to 64, multiples of 64 above: rows do not
 * straddle more memory lines than their length needs); codes [N_base x code_dim] bytes; params [param_floats] floats
 * ([0] s, [1] 1/s, [2] e_max, [3] ||o||, [4] usable (0 when the data holds non-finite values),
 * [5..6] internal, [7] measure, [8..] o_d); scratch [scratch_floats] floats.  D must be a
 * multiple of 4. */
ggnn_status ggnn_prescreen_sizes(uint32_t N_base, uint32_t D, ggnn_measure measure,
                                 uint32_t* code_dim, size_t* param_floats, size_t* scratch_floats);
ggnn_status ggnn_op_prescreen_encode(const float* base, uint32_t N_base, uint32_t D,
                                     ggnn_measure measure, uint8_t* codes, float* params,
                                     float* scratch, void* stream);
/* Validation probe: for query n and candidate cand[n*M+j], reject[n*M+j] = 1 iff the pre-screen
 * would drop the candidate at criteria crit[n*M+j]; s_out (optional) receives the coded squared
 * distance in code units.  A correct bound never rejects at a criteria above the float
 * distance of the pair. */
ggnn_status ggnn_op_prescreen_probe(const uint8_t* codes, const float* params, uint32_t D,
                                    ggnn_measure measure, const float* query, uint32_t Nq,
                                    const int32_t* cand, uint32_t M, const float* crit,
                                    int32_t* reject, float* s_out, void* stream);
/* ggnn_op_query for float32 with the pre-screen copy made for `measure` (params[4] must be 1).
 * n_rows: optional [Nq x 2], float rows and code rows read per query. */
ggnn_status ggnn_op_query_prescreened(const float* base, uint32_t N_base, uint32_t D,
                                      const uint8_t* codes, const float* params,
                                      const float* query, uint32_t Nq, const int32_t* graph0,
                                      uint32_t KBuild, const int32_t* start, uint32_t num_start,
                                      const float* nn1_stats, uint32_t k_query, float tau_query,
                                      uint32_t max_iterations, ggnn_measure measure,
                                      uint32_t shards_per_gpu, uint32_t on_gpu_shard, int32_t* ids,
                                      float* dists, uint32_t* n_dist, uint32_t* n_pop,
                                      uint32_t* n_rows, void* stream);

/* QueryKernels::bruteForceQuery  query_kernels.cu:188-264 -> bf_query_layer.cu:39-65 */
ggnn_status ggnn_op_bf_query(const void* base, ggnn_dtype dtype, uint32_t N_base, uint32_t D,
                             const void* query, uint32_t Nq, uint32_t k_query,
                             ggnn_measure measure, int32_t* ids, float* dists, void* stream);
/* same, and *n_rescanned (device memory, may be NULL) receives the number of queries whose
 * matrix-core pre-selection could not be certified exact and which were therefore answered by the
 * exhaustive scan kernel (0 on paths that scan anyway).  The answers are exact either way. */
ggnn_status ggnn_op_bf_query_certified(const void* base, ggnn_dtype dtype, uint32_t N_base,
                                       uint32_t D, const void* query, uint32_t Nq,
                                       uint32_t k_query, ggnn_measure measure, int32_t* ids,
                                       float* dists, uint32_t* n_rescanned, void* stream);

/* top  graph_construction.cu:201-237 -> top_merge_layer.cu:40-82.  graph_layer/translation are
 * the views of `layer` (translation NULL on layer 0). */
ggnn_status ggnn_op_top(const void* base, ggnn_dtype dtype, uint32_t D, ggnn_measure measure,
                        uint32_t KBuild, const int32_t* translation_layer, uint32_t N_layer,
                        uint32_t S, uint32_t S_offset, uint32_t layer, int32_t* graph_layer,
                        float* nn1_dist_buffer, void* stream);

/* mergeLayer  graph_construction.cu:239-296 -> merge_layer.cu:63-158 (without the final D2D
 * copy).  graph_all [N_all x K], translation_all/selection_all [ST_all] (views "starting at
 * layer 1", i.e. indexed with STs_offsets), writes graph_buffer [Ns[btm] x K]. */
ggnn_status ggnn_op_merge(const void* base, ggnn_dtype dtype, ggnn_measure measure,
                          const ggnn_graph_config* cfg, const int32_t* graph_all,
                          const int32_t* translation_all, const int32_t* selection_all,
                          const float* nn1_stats, float tau_build, uint32_t layer_top,
                          uint32_t layer_btm, int32_t* graph_buffer, float* nn1_dist_buffer,
                          uint32_t* n_dist, void* stream);
/* ggnn_op_merge for float32 with the pre-screen copy of the base made for `measure`
 * (ggnn_op_prescreen_encode, params[4] must be 1); same outputs. */
ggnn_status ggnn_op_merge_prescreened(const float* base, const uint8_t* codes, const float* params,
                                      ggnn_measure measure, const ggnn_graph_config* cfg,
                                      const int32_t* graph_all,
                                      const int32_t* translation_all,
                                      const int32_t* selection_all, const float* nn1_stats,
                                      float tau_build, uint32_t layer_top, uint32_t layer_btm,
                                      int32_t* graph_buffer, float* nn1_dist_buffer,
                                      uint32_t* n_dist, void* stream);

/* select  graph_construction.cu:163-187 -> wrs_select_layer.cu:41-102.  rng [Ns[layer]] uniform
 * (0,1] numbers (the reference draws them with cuRAND XORWOW; injected here). */
ggnn_status ggnn_op_select(const ggnn_graph_config* cfg, uint32_t layer,
                           const float* nn1_dist_buffer, const float* rng,
                           int32_t* translation_all, int32_t* selection_all, void* stream);
/* uniform (0,1] generator used by ggnn_build in place of curandGenerateUniform
 * (graph_construction.cu:96-102,168-169); counter based, seed 1234 by default. */
ggnn_status ggnn_op_uniform(float* out, uint32_t n, uint64_t seed, uint64_t stream_id,
                            void* stream);

/* sym  graph_construction.cu:298-379 -> sym_query_layer.cu:39-145 (blocks first_n ..
 * first_n+count-1; sym_buffer/sym_atomic must have been cleared by the caller) */
ggnn_status ggnn_op_sym(const void* base, ggnn_dtype dtype, ggnn_measure measure, uint32_t D,
                        uint32_t KBuild, const int32_t* graph_layer,
                        const int32_t* translation_layer, uint32_t N_layer,
                        const float* nn1_stats, float tau_build, int32_t* sym_buffer,
                        uint32_t* sym_atomic, uint32_t first_n, uint32_t count, void* stream);
/* ggnn_op_sym for float32 with the pre-screen copy of the base made for `measure`
 * (ggnn_op_prescreen_encode, params[4] must be 1); same outputs. */
ggnn_status ggnn_op_sym_prescreened(const float* base, const uint8_t* codes, const float* params,
                                    ggnn_measure measure, uint32_t D, uint32_t KBuild,
                                    const int32_t* graph_layer, const int32_t* translation_layer,
                                    uint32_t N_layer, const float* nn1_stats, float tau_build,
                                    int32_t* sym_buffer, uint32_t* sym_atomic, uint32_t first_n,
                                    uint32_t count, void* stream);

/* sym_buffer_merge  sym_buffer_merge_layer.cu:36-99 (sym_buffer is used as scratch) */
ggnn_status ggnn_op_sym_buffer_merge(uint32_t KBuild, uint32_t N_layer, int32_t* sym_buffer,
                                     const uint32_t* sym_atomic, int32_t* graph_layer,
                                     void* stream);
/* computeNN1Stats + divide  graph_construction.cu:381-393,79-83 ; out = {mean, max};
 * scratch: >= ggnn_nn1_stats_scratch_floats() floats of device memory */
size_t ggnn_nn1_stats_scratch_floats(void);
ggnn_status ggnn_op_nn1_stats(const float* nn1_dist_buffer, uint32_t N, float* scratch,
                              float* out, void* stream);

/* GPUInstance::sortQueryResults  gpu_instance.cu:745-790: stable ascending sort of every
 * [row_len] row by distance (in place). */
ggnn_status ggnn_op_sort_shard_results(uint32_t Nq, uint32_t row_len, int32_t* ids, float* dists,
                                       void* stream);
/* ResultMerger::merge  result_merger.cpp:51-149 on the device: `parts` = num_parts consecutive
 * [Nq x stride] blocks (e.g. the RCCL all-gather buffer), each row sorted ascending; writes the
 * k best per query with id + part*id_offset_per_part.  Ties: lower part first. */
ggnn_status ggnn_op_merge_results(uint32_t Nq, uint32_t k, uint32_t num_parts, uint32_t stride,
                                  uint32_t id_offset_per_part, const int32_t* parts_ids,
                                  const float* parts_dists, int32_t* ids_out, float* dists_out,
                                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GGNN_C_H */
