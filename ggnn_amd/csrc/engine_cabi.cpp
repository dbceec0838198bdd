// extern "C" entry points of libggnn_amd.so (include/ggnn_c.h): handle API and operator seam.
// Every call is guarded: exceptions become status codes + ggnn_last_error().
#include "engine.hpp"

namespace {
thread_local std::string g_create_error;

// the engine switches devices (hipSetDevice) while it works; callers such as PyTorch keep their
// own notion of the current device, so every entry point leaves it as it found it
struct DeviceRestore {
  int prev{-1};
  DeviceRestore() { (void)hipGetDevice(&prev); }
  ~DeviceRestore()
  {
    int now = -1;
    if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev)
      (void)hipSetDevice(prev);
  }
};

template <typename F>
ggnn_status guarded(ggnn_t* h, F&& f)
{
  DeviceRestore restore;
  try {
    f();
    return GGNN_OK;
  }
  catch (const Error& e) {
    (h ? h->last_error : g_create_error) = e.what();
    return e.status;
  }
  catch (const std::bad_alloc&) {
    (h ? h->last_error : g_create_error) = "out of host memory";
    return GGNN_OUT_OF_MEMORY;
  }
  catch (const std::exception& e) {
    (h ? h->last_error : g_create_error) = e.what();
    return GGNN_DEVICE_ERROR;
  }
}

#define GGNN_NEED_HANDLE(h) \
  if (!(h))                 \
  return GGNN_INVALID_ARGUMENT
}  // namespace

extern "C" {

const char* ggnn_version(void)
{
  return "ggnn_amd 0.1.0 (gfx950)";
}

ggnn_status ggnn_create(ggnn_t** out)
{
  if (!out)
    return GGNN_INVALID_ARGUMENT;
  return guarded(nullptr, [&] { *out = new ggnn_handle(); });
}

void ggnn_destroy(ggnn_t* h)
{
  DeviceRestore restore;
  delete h;
}

const char* ggnn_last_error(const ggnn_t* h)
{
  return h ? h->last_error.c_str() : g_create_error.c_str();
}

void ggnn_set_log_level(int level)
{
  g_log_level = level;
}

ggnn_status ggnn_set_working_directory(ggnn_t* h, const char* dir)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] {
    // ggnn.cu:69-76
    const std::filesystem::path p = dir ? dir : "";
    const auto target = p.empty() ? std::filesystem::current_path() : std::filesystem::absolute(p);
    // graph parts of out-of-core shards live in the directory they were written to: store() skips
    // them as "already on disk" and read_part() would look for them in the new place
    if (target != h->graph_dir)
      for (const auto& ctx : h->devs)
        if (ctx.swap)
          for (const uint8_t on_disk : ctx.swap->on_disk)
            GGNN_REQUIRE(!on_disk, GGNN_INVALID_STATE,
                         "the working directory cannot change while graph parts of out-of-core "
                         "shards live in " + h->graph_dir.string());
    h->graph_dir = target;
    std::error_code ec;
    std::filesystem::create_directories(h->graph_dir, ec);
    GGNN_REQUIRE(!ec, GGNN_IO_ERROR, "cannot create working directory " + h->graph_dir.string());
  });
}

ggnn_status ggnn_set_cpu_memory_limit(ggnn_t* h, size_t memory_limit)
{
  GGNN_NEED_HANDLE(h);
  h->cpu_memory_limit = memory_limit;
  return GGNN_OK;
}

ggnn_status ggnn_set_reserved_gpu_memory(ggnn_t* h, size_t reserved_memory)
{
  GGNN_NEED_HANDLE(h);
  h->reserved_gpu_memory = reserved_memory;
  return GGNN_OK;
}

ggnn_status ggnn_set_gpus(ggnn_t* h, const int* gpu_ids, size_t num_gpus)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] {
    int count = 0;
    (void)hipGetDeviceCount(&count);
    for (size_t i = 0; i < num_gpus; ++i) {
      // ggnn.cu:94-97 (accepts gpu_id == device count, quirk Q5)
      GGNN_REQUIRE(gpu_ids[i] >= 0 && gpu_ids[i] <= count, GGNN_OUT_OF_RANGE,
                   "Invalid GPU index " + std::to_string(gpu_ids[i]) + " given.");
    }
    h->gpu_ids.assign(gpu_ids, gpu_ids + num_gpus);
  });
}

ggnn_status ggnn_set_shard_size(ggnn_t* h, uint32_t n_shard)
{
  GGNN_NEED_HANDLE(h);
  h->N_shard = n_shard;
  return GGNN_OK;
}

ggnn_status ggnn_set_return_results_on_gpu(ggnn_t* h, int v)
{
  GGNN_NEED_HANDLE(h);
  h->return_results_on_gpu = v != 0;
  return GGNN_OK;
}

ggnn_status ggnn_last_query_rows_read(const ggnn_t* h, uint64_t* float_rows, uint64_t* code_rows)
{
  if (!h)
    return GGNN_INVALID_ARGUMENT;
  if (float_rows)
    *float_rows = h->last_float_rows;
  if (code_rows)
    *code_rows = h->last_code_rows;
  return GGNN_OK;
}

ggnn_status ggnn_set_prescreen(ggnn_t* h, int enable)
{
  GGNN_NEED_HANDLE(h);
  h->prescreen = enable != 0;
  return GGNN_OK;
}

ggnn_status ggnn_last_query_parts(const ggnn_t* h, uint32_t* parts)
{
  GGNN_NEED_HANDLE(h);
  if (!parts)
    return GGNN_INVALID_ARGUMENT;
  *parts = h->last_query_parts;
  return GGNN_OK;
}

ggnn_status ggnn_last_build_work(const ggnn_t* h, ggnn_build_work* out)
{
  GGNN_NEED_HANDLE(h);
  if (!out)
    return GGNN_INVALID_ARGUMENT;
  *out = h->build_work;
  return GGNN_OK;
}

ggnn_status ggnn_set_hook(const char* name, int64_t value)
{
  const int h = ggnn_amd::hook_by_name(name);
  if (h < 0)
    return GGNN_INVALID_ARGUMENT;
  ggnn_amd::hook_set(static_cast<ggnn_amd::Hook>(h), value);
  return GGNN_OK;
}

ggnn_status ggnn_reset_hook(const char* name)
{
  const int h = ggnn_amd::hook_by_name(name);
  if (h < 0)
    return GGNN_INVALID_ARGUMENT;
  ggnn_amd::hook_reset(static_cast<ggnn_amd::Hook>(h));
  return GGNN_OK;
}

ggnn_status ggnn_get_hook(const char* name, int64_t* value)
{
  const int h = ggnn_amd::hook_by_name(name);
  if (h < 0 || !value)
    return GGNN_INVALID_ARGUMENT;
  *value = ggnn_amd::hook(static_cast<ggnn_amd::Hook>(h));
  return GGNN_OK;
}

ggnn_status ggnn_get_shard_layout(const ggnn_t* h, uint32_t* num_shards, uint32_t* shards_per_gpu,
                                  uint32_t* n_shard)
{
  GGNN_NEED_HANDLE(h);
  if (!h->has_graph())
    return GGNN_INVALID_STATE;
  if (num_shards)
    *num_shards = h->num_shards();
  if (shards_per_gpu)
    *shards_per_gpu = h->shards_per_gpu;
  if (n_shard)
    *n_shard = h->cfg.N;
  return GGNN_OK;
}

ggnn_status ggnn_set_collect_counters(ggnn_t* h, int enable)
{
  GGNN_NEED_HANDLE(h);
  h->collect_counters = enable != 0;
  return GGNN_OK;
}

ggnn_status ggnn_set_base(ggnn_t* h, const void* data, uint64_t N, uint32_t D, ggnn_dtype dtype,
                          ggnn_location location, int gpu_id, int take_copy)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] {
    // ggnn.cu:146-152
    GGNN_REQUIRE(!h->prepared, GGNN_INVALID_STATE,
                 "The base cannot be changed once the GPU instances are setup.");
    GGNN_REQUIRE(dtype == GGNN_F32 || dtype == GGNN_U8, GGNN_INVALID_ARGUMENT,
                 "unsupported datatype for base");
    // ggnn.cu:466-487: the element type is fixed by the first set_base
    GGNN_REQUIRE(!h->base_set || h->base_dtype == dtype, GGNN_INVALID_ARGUMENT,
                 "base has already been set with a different data type");
    GGNN_REQUIRE(data != nullptr && N > 0 && D > 0, GGNN_INVALID_ARGUMENT, "empty base");
    const size_t bytes = N * D * dtype_size(dtype);
    h->drop_host_copy();
    h->base_dev_copy.release();
    h->devs.clear();  // a base staged for an earlier bf_query() is stale now
    h->base_src = data;
    h->base_loc = location;
    h->base_gpu = gpu_id;
    if (take_copy) {
      if (location == GGNN_CPU) {
        h->base_host_copy.assign(static_cast<const uint8_t*>(data),
                                 static_cast<const uint8_t*>(data) + bytes);
        h->base_src = h->base_host_copy.data();
      }
      else {
        GGNN_HIP_CHECK(hipSetDevice(gpu_id));
        h->base_dev_copy.alloc(bytes);
        GGNN_HIP_CHECK(hipMemcpy(h->base_dev_copy.p, data, bytes, hipMemcpyDeviceToDevice));
        h->base_src = h->base_dev_copy.p;
      }
    }
    h->base_N = N;
    h->base_D = D;
    const uint32_t epc = 16 / static_cast<uint32_t>(dtype_size(dtype));
    h->pad_D = (D + epc - 1) / epc * epc;
    h->base_dtype = dtype;
    h->base_set = true;
  });
}

ggnn_status ggnn_build(ggnn_t* h, uint32_t k_build, float tau_build,
                       uint32_t refinement_iterations, ggnn_measure measure)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] { h->build(k_build, tau_build, refinement_iterations, measure); });
}

ggnn_status ggnn_device_clock_hz(int device, double* clock_hz)
{
  return guarded(nullptr, [&] {
    GGNN_REQUIRE(clock_hz != nullptr, GGNN_INVALID_ARGUMENT, "null output");
    int khz = 0;
    GGNN_HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, device));
    *clock_hz = static_cast<double>(khz) * 1e3;
  });
}

ggnn_status ggnn_set_build_hooks(ggnn_t* h, const float* rng, uint64_t n_rng, int serial_sym)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] {
    GGNN_REQUIRE(rng != nullptr || n_rng == 0, GGNN_INVALID_ARGUMENT, "rng is null");
    h->hook_rng.assign(rng, rng + n_rng);
    h->hook_serial_sym = serial_sym != 0;
  });
}

ggnn_status ggnn_store(ggnn_t* h)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] { h->store(); });
}

ggnn_status ggnn_load(ggnn_t* h, uint32_t k_build)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] { h->load(k_build); });
}

ggnn_status ggnn_query(ggnn_t* h, const void* query, uint64_t Nq, uint32_t D, ggnn_dtype dtype,
                       ggnn_location location, int gpu_id, uint32_t k_query, float tau_query,
                       uint32_t max_iterations, ggnn_measure measure, int32_t* ids_out,
                       float* dists_out, ggnn_location out_location)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] {
    h->query(query, Nq, D, dtype, location, gpu_id, k_query, tau_query, max_iterations, measure,
             ids_out, dists_out, out_location);
  });
}

ggnn_status ggnn_query_async(ggnn_t* h, const void* query, uint64_t Nq, uint32_t D,
                             ggnn_dtype dtype, int gpu_id, uint32_t k_query, float tau_query,
                             uint32_t max_iterations, ggnn_measure measure, int32_t* ids_out,
                             float* dists_out, uint32_t slot)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] {
    h->query_async(query, Nq, D, dtype, gpu_id < 0 ? GGNN_CPU : GGNN_GPU, gpu_id, k_query,
                   tau_query, max_iterations, measure, ids_out, dists_out, slot);
  });
}

ggnn_status ggnn_synchronize(ggnn_t* h)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] { h->synchronize(); });
}

ggnn_status ggnn_synchronize_slot(ggnn_t* h, uint32_t slot)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] { h->synchronize_slot(slot); });
}

ggnn_status ggnn_bf_query(ggnn_t* h, const void* query, uint64_t Nq, uint32_t D,
                          ggnn_dtype dtype, ggnn_location location, int gpu_id,
                          uint32_t k_gt, ggnn_measure measure, int32_t* ids_out, float* dists_out,
                          ggnn_location out_location)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] {
    h->bf_query(query, Nq, D, dtype, location, gpu_id, k_gt, measure, ids_out, dists_out,
                out_location);
  });
}

ggnn_status ggnn_get_graph(ggnn_t* h, uint32_t global_shard_id, ggnn_graph_view* out)
{
  GGNN_NEED_HANDLE(h);
  return guarded(h, [&] {
    GGNN_REQUIRE(out != nullptr, GGNN_INVALID_ARGUMENT, "null output");
    // ggnn.cu:392-413
    GGNN_REQUIRE(h->has_graph(), GGNN_INVALID_STATE, "No graph has been built or loaded yet.");
    GGNN_REQUIRE(global_shard_id < h->num_shards(), GGNN_INVALID_STATE,
                 "Shard " + std::to_string(global_shard_id) + " does not exist.");
    DeviceCtx& ctx = h->devs[global_shard_id / h->shards_per_gpu];
    if (ctx.swap) {
      // out-of-core shards: the view is valid until another shard takes the slot
      ctx.activate();
      h->acquire_shard(ctx, global_shard_id % h->shards_per_gpu, ctx.stream);
      GGNN_HIP_CHECK(hipStreamSynchronize(ctx.stream));
    }
    const Shard& sh = ctx.shards[global_shard_id % h->shards_per_gpu];
    out->config = h->cfg;
    out->config.D = h->base_D;  // caller-visible dimension (rows are padded internally)
    out->graph = sh.graph;
    out->translation = sh.translation;
    out->selection = sh.selection;
    out->nn1_stats = sh.nn1_stats;
    out->gpu_id = ctx.device;
  });
}

ggnn_status ggnn_last_timing_ms(const ggnn_t* h, float* build_ms, float* query_ms, float* bf_ms)
{
  GGNN_NEED_HANDLE(h);
  if (build_ms)
    *build_ms = h->build_ms;
  if (query_ms)
    *query_ms = h->query_ms;
  if (bf_ms)
    *bf_ms = h->bf_ms;
  return GGNN_OK;
}

const char* ggnn_last_exchange(const ggnn_t* h)
{
  return h ? h->last_exchange : "none";
}

ggnn_status ggnn_rccl_ranks(const ggnn_t* h, uint32_t* ranks)
{
  GGNN_NEED_HANDLE(h);
  uint32_t n = 0;
  if (!h->comms.empty() && h->comms[0] && Rccl::get().ok && Rccl::get().CommCount) {
    int count = 0;
    if (Rccl::get().CommCount(h->comms[0], &count) == ncclSuccess && count > 0)
      n = static_cast<uint32_t>(count);
  }
  if (ranks)
    *ranks = n;
  return GGNN_OK;
}

ggnn_status ggnn_last_bf_query_rescanned(const ggnn_t* h, uint32_t* n_rescanned)
{
  GGNN_NEED_HANDLE(h);
  if (n_rescanned)
    *n_rescanned = h->last_bf_rescanned;
  return GGNN_OK;
}

ggnn_status ggnn_last_query_counters(const ggnn_t* h, uint64_t* n_dist, uint64_t* n_pop)
{
  GGNN_NEED_HANDLE(h);
  if (n_dist)
    *n_dist = h->last_n_dist;
  if (n_pop)
    *n_pop = h->last_n_pop;
  return GGNN_OK;
}

// ---- Section 2: operator seam ----------------------------------------------------------------

ggnn_status ggnn_graph_config_init(uint32_t N, uint32_t D, uint32_t KBuild,
                                   ggnn_graph_config* out)
{
  return guarded(nullptr, [&] { graph_config_init(N, D, KBuild, out); });
}

ggnn_status ggnn_query_sizing(uint32_t D, uint32_t k_query, uint32_t max_iterations,
                              uint32_t* cache_size, uint32_t* sorted_size)
{
  return guarded(nullptr, [&] { query_sizing(D, k_query, max_iterations, cache_size, sorted_size); });
}

ggnn_status ggnn_op_query(const void* base, ggnn_dtype dtype, uint32_t N_base, uint32_t D,
                          const void* query, uint32_t Nq, const int32_t* graph0,
                          uint32_t KBuild, const int32_t* start, uint32_t num_start,
                          const float* nn1_stats, uint32_t k_query, float tau_query,
                          uint32_t max_iterations, ggnn_measure measure,
                          uint32_t shards_per_gpu, uint32_t on_gpu_shard, int32_t* ids,
                          float* dists, uint32_t* n_dist, uint32_t* n_pop, void* stream)
{
  return guarded(nullptr, [&] {
    QueryLaunch q{base,      query,          dtype,         N_base,       D,       Nq,
                  graph0,    KBuild,         start,         num_start,    nn1_stats, k_query,
                  tau_query, max_iterations, measure,       shards_per_gpu, on_gpu_shard, ids,
                  dists,     n_dist,         n_pop};
    launch_query(q, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_prescreen_sizes(uint32_t N_base, uint32_t D, ggnn_measure measure,
                                 uint32_t* code_dim, size_t* param_floats, size_t* scratch_floats)
{
  return guarded(nullptr, [&] {
    GGNN_REQUIRE(D >= 1 && D <= 4096 && D % 4 == 0, GGNN_INVALID_ARGUMENT,
                 "D must be a multiple of 4 in [4, 4096]");
    if (code_dim)
      *code_dim = prescreen_code_dim(D);
    if (param_floats)
      *param_floats = prescreen_param_floats(D);
    if (scratch_floats)
      *scratch_floats = prescreen_scratch_floats(N_base, D, measure);
  });
}

ggnn_status ggnn_op_prescreen_encode(const float* base, uint32_t N_base, uint32_t D,
                                     ggnn_measure measure, uint8_t* codes, float* params,
                                     float* scratch, void* stream)
{
  return guarded(nullptr, [&] {
    launch_prescreen_encode(base, N_base, D, measure, codes, params, scratch,
                            static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_prescreen_probe(const uint8_t* codes, const float* params, uint32_t D,
                                    ggnn_measure measure, const float* query, uint32_t Nq,
                                    const int32_t* cand, uint32_t M, const float* crit,
                                    int32_t* reject, float* s_out, void* stream)
{
  return guarded(nullptr, [&] {
    launch_prescreen_probe(codes, params, D, measure, query, Nq, cand, M, crit, reject, s_out,
                           static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_query_prescreened(const float* base, uint32_t N_base, uint32_t D,
                                      const uint8_t* codes, const float* params,
                                      const float* query, uint32_t Nq, const int32_t* graph0,
                                      uint32_t KBuild, const int32_t* start, uint32_t num_start,
                                      const float* nn1_stats, uint32_t k_query, float tau_query,
                                      uint32_t max_iterations, ggnn_measure measure,
                                      uint32_t shards_per_gpu, uint32_t on_gpu_shard, int32_t* ids,
                                      float* dists, uint32_t* n_dist, uint32_t* n_pop,
                                      uint32_t* n_rows, void* stream)
{
  return guarded(nullptr, [&] {
    GGNN_REQUIRE(codes && params, GGNN_INVALID_ARGUMENT, "pre-screen buffers are null");
    QueryLaunch q{base,      query,          GGNN_F32, N_base,         D,         Nq,
                  graph0,    KBuild,         start,    num_start,      nn1_stats, k_query,
                  tau_query, max_iterations, measure,  shards_per_gpu, on_gpu_shard, ids,
                  dists,     n_dist,         n_pop};
    q.ps_codes = codes;
    q.ps_params = params;
    q.ps_Dc = prescreen_code_dim(D);
    q.n_rows = n_rows;
    launch_query(q, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_bf_query(const void* base, ggnn_dtype dtype, uint32_t N_base, uint32_t D,
                             const void* query, uint32_t Nq, uint32_t k_query,
                             ggnn_measure measure, int32_t* ids, float* dists, void* stream)
{
  return guarded(nullptr, [&] {
    BfLaunch b{base, query, dtype, N_base, D, Nq, k_query, measure, ids, dists};
    launch_bf_query(b, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_bf_query_certified(const void* base, ggnn_dtype dtype, uint32_t N_base,
                                       uint32_t D, const void* query, uint32_t Nq,
                                       uint32_t k_query, ggnn_measure measure, int32_t* ids,
                                       float* dists, uint32_t* n_rescanned, void* stream)
{
  return guarded(nullptr, [&] {
    BfLaunch b{base, query, dtype, N_base, D, Nq, k_query, measure, ids, dists, n_rescanned};
    launch_bf_query(b, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_top(const void* base, ggnn_dtype dtype, uint32_t D, ggnn_measure measure,
                        uint32_t KBuild, const int32_t* translation_layer, uint32_t N_layer,
                        uint32_t S, uint32_t S_offset, uint32_t layer, int32_t* graph_layer,
                        float* nn1_dist_buffer, void* stream)
{
  return guarded(nullptr, [&] {
    TopLaunch t{base, dtype, D, measure, KBuild, translation_layer, N_layer, S, S_offset, layer,
                graph_layer, nn1_dist_buffer};
    launch_top(t, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_merge(const void* base, ggnn_dtype dtype, ggnn_measure measure,
                          const ggnn_graph_config* cfg, const int32_t* graph_all,
                          const int32_t* translation_all, const int32_t* selection_all,
                          const float* nn1_stats, float tau_build, uint32_t layer_top,
                          uint32_t layer_btm, int32_t* graph_buffer, float* nn1_dist_buffer,
                          uint32_t* n_dist, void* stream)
{
  return guarded(nullptr, [&] {
    GGNN_REQUIRE(cfg != nullptr, GGNN_INVALID_ARGUMENT, "null graph config");
    MergeLaunch m{base,      dtype,     measure,   *cfg,         graph_all,       translation_all,
                  selection_all, nn1_stats, tau_build, layer_top, layer_btm,      graph_buffer,
                  nn1_dist_buffer, n_dist};
    launch_merge(m, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_merge_prescreened(const float* base, const uint8_t* codes, const float* params,
                                      ggnn_measure measure, const ggnn_graph_config* cfg,
                                      const int32_t* graph_all,
                                      const int32_t* translation_all,
                                      const int32_t* selection_all, const float* nn1_stats,
                                      float tau_build, uint32_t layer_top, uint32_t layer_btm,
                                      int32_t* graph_buffer, float* nn1_dist_buffer,
                                      uint32_t* n_dist, void* stream)
{
  return guarded(nullptr, [&] {
    GGNN_REQUIRE(cfg != nullptr, GGNN_INVALID_ARGUMENT, "null graph config");
    GGNN_REQUIRE(codes && params, GGNN_INVALID_ARGUMENT, "pre-screen buffers are null");
    MergeLaunch m{base,          GGNN_F32,  measure,        *cfg,      graph_all, translation_all,
                  selection_all, nn1_stats, tau_build,      layer_top, layer_btm, graph_buffer,
                  nn1_dist_buffer, n_dist};
    m.ps_codes = codes;
    m.ps_params = params;
    m.ps_Dc = prescreen_code_dim(cfg->D);
    launch_merge(m, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_select(const ggnn_graph_config* cfg, uint32_t layer,
                           const float* nn1_dist_buffer, const float* rng,
                           int32_t* translation_all, int32_t* selection_all, void* stream)
{
  return guarded(nullptr, [&] {
    GGNN_REQUIRE(cfg != nullptr, GGNN_INVALID_ARGUMENT, "null graph config");
    launch_select(*cfg, layer, nn1_dist_buffer, rng, translation_all, selection_all,
                  static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_uniform(float* out, uint32_t n, uint64_t seed, uint64_t stream_id,
                            void* stream)
{
  return guarded(nullptr,
                 [&] { launch_uniform(out, n, seed, stream_id, static_cast<hipStream_t>(stream)); });
}

ggnn_status ggnn_op_sym(const void* base, ggnn_dtype dtype, ggnn_measure measure, uint32_t D,
                        uint32_t KBuild, const int32_t* graph_layer,
                        const int32_t* translation_layer, uint32_t N_layer,
                        const float* nn1_stats, float tau_build, int32_t* sym_buffer,
                        uint32_t* sym_atomic, uint32_t first_n, uint32_t count, void* stream)
{
  return guarded(nullptr, [&] {
    SymLaunch s{base,      dtype,     measure,    D,          KBuild,  graph_layer, translation_layer,
                N_layer,   nn1_stats, tau_build,  sym_buffer, sym_atomic, first_n,  count};
    launch_sym(s, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_sym_prescreened(const float* base, const uint8_t* codes, const float* params,
                                    ggnn_measure measure, uint32_t D, uint32_t KBuild,
                                    const int32_t* graph_layer, const int32_t* translation_layer,
                                    uint32_t N_layer, const float* nn1_stats, float tau_build,
                                    int32_t* sym_buffer, uint32_t* sym_atomic, uint32_t first_n,
                                    uint32_t count, void* stream)
{
  return guarded(nullptr, [&] {
    GGNN_REQUIRE(codes && params, GGNN_INVALID_ARGUMENT, "pre-screen buffers are null");
    SymLaunch s{base,    GGNN_F32,  measure,   D,          KBuild,     graph_layer, translation_layer,
                N_layer, nn1_stats, tau_build, sym_buffer, sym_atomic, first_n,     count};
    s.ps_codes = codes;
    s.ps_params = params;
    s.ps_Dc = prescreen_code_dim(D);
    launch_sym(s, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_sym_buffer_merge(uint32_t KBuild, uint32_t N_layer, int32_t* sym_buffer,
                                     const uint32_t* sym_atomic, int32_t* graph_layer,
                                     void* stream)
{
  return guarded(nullptr, [&] {
    launch_sym_buffer_merge(KBuild, N_layer, sym_buffer, sym_atomic, graph_layer,
                            static_cast<hipStream_t>(stream));
  });
}

size_t ggnn_nn1_stats_scratch_floats(void)
{
  return 3 * kStatsBlocks;
}

ggnn_status ggnn_op_nn1_stats(const float* nn1_dist_buffer, uint32_t N, float* scratch,
                              float* out, void* stream)
{
  return guarded(nullptr, [&] {
    launch_nn1_stats(nn1_dist_buffer, N, scratch, out, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_sort_shard_results(uint32_t Nq, uint32_t row_len, int32_t* ids, float* dists,
                                       void* stream)
{
  return guarded(nullptr, [&] {
    launch_sort_shard_results(Nq, row_len, ids, dists, static_cast<hipStream_t>(stream));
  });
}

ggnn_status ggnn_op_merge_results(uint32_t Nq, uint32_t k, uint32_t num_parts, uint32_t stride,
                                  uint32_t id_offset_per_part, const int32_t* parts_ids,
                                  const float* parts_dists, int32_t* ids_out, float* dists_out,
                                  void* stream)
{
  return guarded(nullptr, [&] {
    launch_merge_results(Nq, k, num_parts, stride, id_offset_per_part, parts_ids, parts_dists,
                         ids_out, dists_out, static_cast<hipStream_t>(stream));
  });
}

}  // extern "C"
