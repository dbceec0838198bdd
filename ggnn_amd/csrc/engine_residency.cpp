// Engine behind the C-ABI, part "residency": base staging, graph residency (prepare), pre-screen copies, store / load.
// The handle is declared in engine.hpp.
#include "engine.hpp"

// out-of-core shards re-read their rows at every swap on a copy stream: from pageable memory
// that copy is staged and synchronous, i.e. it does not overlap the search (round-4 advisor
// finding) -- the engine's own host copy is page-locked for as long as it is swapped from
void ggnn_handle::pin_host_copy()
{
  if (base_host_copy_registered || base_host_copy.empty())
    return;
  if (hipHostRegister(base_host_copy.data(), base_host_copy.size(), hipHostRegisterDefault) ==
      hipSuccess)
    base_host_copy_registered = true;
  else
    (void)hipGetLastError();  // (still correct from pageable memory, only not overlapped)
}

void ggnn_handle::drop_host_copy()
{
  if (base_host_copy_registered)
    (void)hipHostUnregister(base_host_copy.data());
  base_host_copy_registered = false;
  base_host_copy.clear();
  base_host_copy.shrink_to_fit();
}

// every shard of every GPU is built or loaded (a partly loaded handle has no graph)
bool ggnn_handle::has_graph() const
{
  if (!prepared || devs.empty())
    return false;
  for (const DeviceCtx& ctx : devs) {
    if (ctx.shards.empty())
      return false;
    for (const Shard& sh : ctx.shards)
      if (!sh.ready)
        return false;
  }
  return true;
}

// a failed prepare / build / load leaves the handle as it was after set_base: no contexts, no
// half-initialised shards, and build()/load()/set_base() may be called again
void ggnn_handle::rollback_graph()
{
  DeviceRestoreGuard keep;
  destroy_comms();
  devs.clear();
  prepared = false;
  shards_per_gpu = 0;
  cfg = ggnn_graph_config{};
}

std::vector<int> ggnn_handle::resolve_gpus() const
{
  if (!gpu_ids.empty())
    return gpu_ids;
  int d = 0;
  GGNN_HIP_CHECK(hipGetDevice(&d));  // ggnn.cu:172-176
  return {d};
}

// base.referenceOnGPU (dataset.cu:236-300): rows [row0, row0+rows) resident on ctx's GPU
void ggnn_handle::stage_base_slice(DeviceCtx& ctx, uint64_t row0, uint64_t rows)
{
  GGNN_REQUIRE(base_set, GGNN_INVALID_STATE, "The base needs to be set first.");
  ctx.activate();
  const size_t es = dtype_size(base_dtype);
  const bool padded = pad_D != base_D;
  const uint8_t* src = static_cast<const uint8_t*>(base_src) + row0 * base_D * es;
  const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
  if (!padded && aligned && base_loc == GGNN_GPU && base_gpu == ctx.device) {
    ctx.d_base = src;  // device memory on the right GPU (borrowed or our own copy)
    return;
  }
  ctx.base_copy.alloc(rows * pad_D * es);
  const hipMemcpyKind kind = base_loc == GGNN_GPU ? hipMemcpyDefault : hipMemcpyHostToDevice;
  if (padded) {
    GGNN_HIP_CHECK(hipMemsetAsync(ctx.base_copy.p, 0, ctx.base_copy.bytes, ctx.stream));
    GGNN_HIP_CHECK(hipMemcpy2DAsync(ctx.base_copy.p, pad_D * es, src, base_D * es, base_D * es,
                                    rows, kind, ctx.stream));
  }
  else
    GGNN_HIP_CHECK(hipMemcpyAsync(ctx.base_copy.p, src, ctx.base_copy.bytes, kind, ctx.stream));
  GGNN_HIP_CHECK(hipStreamSynchronize(ctx.stream));
  ctx.d_base = ctx.base_copy.p;
}

const void* ggnn_handle::shard_base(const DeviceCtx& ctx, uint32_t local_shard) const
{
  if (ctx.swap && !ctx.swap->base_borrowed)  // (valid after acquire_shard of this shard)
    return ctx.swap->base[local_shard % ctx.swap->slots].p;
  return static_cast<const uint8_t*>(ctx.d_base) +
         static_cast<size_t>(local_shard) * cfg.N * row_bytes();
}

// GGNNImpl::prepare, ggnn.cu:154-203
void ggnn_handle::prepare(uint32_t KBuild)
{
  // the part files of out-of-core shards are written by one host thread per GPU: the directory
  // is fixed here, before any of them runs (round-4 advisor finding: it was assigned lazily,
  // unlocked, from those threads)
  if (graph_dir.empty())
    graph_dir = std::filesystem::current_path();
  GGNN_REQUIRE(!prepared, GGNN_INVALID_STATE, "A graph has already been built or loaded.");
  GGNN_REQUIRE(base_set, GGNN_INVALID_STATE,
               "The base needs to be set before building a graph.");
  uint64_t n = base_N;
  if (N_shard > 0) {
    GGNN_REQUIRE(base_N % N_shard == 0, GGNN_INVALID_ARGUMENT,
                 "The base dataset needs to be evenly divisible by the shard size.");
    n = N_shard;
  }
  GGNN_REQUIRE(n > 0 && n < 0x7fffffffull, GGNN_INVALID_ARGUMENT,
               "shard size must be in [1, 2^31-1)");
  GGNN_REQUIRE(base_D >= 1 && base_D <= 4096, GGNN_INVALID_ARGUMENT, "D must be in [1, 4096]");
  GGNN_REQUIRE(KBuild >= 2 && KBuild <= 512, GGNN_INVALID_ARGUMENT,
               "KBuild must be in [2, 512]");
  const std::vector<int> gpus = resolve_gpus();
  const uint64_t num_gpus = gpus.size();
  const uint64_t spg = base_N / (n * num_gpus);
  GGNN_REQUIRE(n * num_gpus * spg == base_N && spg > 0, GGNN_INVALID_ARGUMENT,
               "base.N needs to be evenly divisible by (N_shard x num_gpus).");
  GGNN_REQUIRE(base_N < 0x7fffffffull, GGNN_INVALID_ARGUMENT,
               "ids are int32: at most 2^31-1 base points");
  graph_config_init(static_cast<uint32_t>(n), pad_D, KBuild, &cfg);
  // every lower segment must be able to contribute its share of points to the layer above
  // (the reference would silently select padding entries, wrs_select_layer.cu:57-66)
  GGNN_REQUIRE(cfg.SG + (cfg.SG_off ? 1u : 0u) <= cfg.S0 && cfg.S0 >= 2, GGNN_INVALID_ARGUMENT,
               "shard too small for a 4-layer graph with this KBuild (need more points per "
               "bottom segment than are promoted to the next layer)");
  shards_per_gpu = static_cast<uint32_t>(spg);
  // reuse a context created by an earlier bf_query() when it fits
  const bool reuse = devs.size() == 1 && num_gpus == 1 && devs[0].device == gpus[0];
  try {
    if (!reuse) {
      devs.clear();
      devs.resize(num_gpus);
    }
    for (uint32_t i = 0; i < num_gpus; ++i) {
      DeviceCtx& ctx = devs[i];
      ctx.device = gpus[i];
      ctx.first_shard = i * shards_per_gpu;
      ctx.activate();
      ctx.swap.reset();
      // do the shards of this GPU fit next to each other?  (at every BASELINE configuration:
      // yes -- 288 GB; otherwise they take turns in a few GPU slots, SwapState)
      const uint8_t* slice = static_cast<const uint8_t*>(base_src) +
                             static_cast<uint64_t>(ctx.first_shard) * n * base_D * dtype_size(base_dtype);
      const bool base_here = reuse || (pad_D == base_D && base_loc == GGNN_GPU &&
                                       base_gpu == ctx.device &&
                                       (reinterpret_cast<uintptr_t>(slice) & 15u) == 0);
      const uint32_t slots = shards_per_gpu > 1 ? plan_gpu_slots(ctx, shards_per_gpu, base_here) : 0;
      if (!reuse && (!slots || base_here))
        stage_base_slice(ctx, static_cast<uint64_t>(ctx.first_shard) * n,
                         static_cast<uint64_t>(shards_per_gpu) * n);
      ctx.shards.clear();
      ctx.shards.resize(shards_per_gpu);
      for (uint32_t s = 0; s < shards_per_gpu; ++s) {
        ctx.shards[s].global_id = ctx.first_shard + s;
        if (!slots)
          ctx.shards[s].allocate(cfg);
      }
      if (slots)
        setup_swap(ctx, slots, base_here);
    }
  }
  catch (...) {
    rollback_graph();
    throw;
  }
  prepared = true;
  GGNN_LOG(1, "prepare: gpus=%zu N_shard=%u shards/gpu=%u D=%u K=%u G=%u S=%u S0=%u S0_off=%u",
           devs.size(), cfg.N, shards_per_gpu, cfg.D, cfg.KBuild, cfg.G, cfg.S, cfg.S0,
           cfg.S0_off);
}

// the engine's own host/device copy of the caller's base is no longer needed once every GPU
// holds its slice
void ggnn_handle::release_caller_copy()
{
  // out-of-core shards re-read their rows from the caller's / the engine's copy at every swap
  for (const DeviceCtx& ctx : devs)
    if (ctx.swap && !ctx.swap->base_borrowed)
      return;
  bool borrowed_from_copy = false;
  for (const DeviceCtx& ctx : devs)
    borrowed_from_copy |= (ctx.base_copy.p == nullptr);
  drop_host_copy();
  if (!borrowed_from_copy)
    base_dev_copy.release();
}

// Pre-screen copy of shard si (traversal.hpp "Exact pre-screen"): pays when a float row spans
// more cache lines than its code row, i.e. from 256 bytes per row on.
bool ggnn_handle::ensure_prescreen(DeviceCtx& ctx, uint32_t si, ggnn_measure measure)
{
  Shard& sh = ctx.shards[si];
  // (out-of-core shards: a per-shard copy that would have to be re-coded at every swap; the
  // kernels read the float rows, results are the same)
  if (!prescreen || base_dtype != GGNN_F32 || pad_D < 64 || ctx.swap)
    return false;
  if (sh.ps_state != 0 && sh.ps_measure != measure) {
    // the codes belong to the other measure: code again.  Batches still in flight on this GPU
    // (query_async lanes, overlapped shard launches) read the old codes: drained explicitly
    // (not left to the hipFree inside DeviceBuffer::alloc, which only happens to synchronise)
    ctx.activate();
    GGNN_HIP_CHECK(hipStreamSynchronize(ctx.stream));
    for (int i = 0; i < DeviceCtx::kShardStreams; ++i)
      if (ctx.shard_stream[i])
        GGNN_HIP_CHECK(hipStreamSynchronize(ctx.shard_stream[i]));
    sh.ps_state = 0;
  }
  if (sh.ps_state == 0) {
    const uint32_t Dc = prescreen_code_dim(pad_D);
    DeviceBuffer scratch;
    try {
      sh.ps_codes.alloc(static_cast<size_t>(cfg.N) * Dc);
      sh.ps_params.alloc(prescreen_param_floats(pad_D) * 4);
      scratch.alloc(prescreen_scratch_floats(cfg.N, pad_D, measure) * 4);
    }
    catch (const Error& e) {
      if (e.status != GGNN_OUT_OF_MEMORY)
        throw;
      // an optional copy: without room for it the kernels read the float rows as before
      (void)hipGetLastError();
      sh.ps_codes.release();
      sh.ps_params.release();
      sh.ps_state = -1;
      sh.ps_measure = measure;
      GGNN_LOG(0, "[GPU: %d] no memory for the pre-screen copy of part %u, continuing without",
               ctx.device, sh.global_id);
      return false;
    }
    launch_prescreen_encode(static_cast<const float*>(shard_base(ctx, si)), cfg.N, pad_D, measure,
                            sh.ps_codes.as<uint8_t>(), sh.ps_params.as<float>(),
                            scratch.as<float>(), ctx.stream);
    float header[kPsHeaderFloats];
    GGNN_HIP_CHECK(hipMemcpyAsync(header, sh.ps_params.p, sizeof(header), hipMemcpyDeviceToHost,
                                  ctx.stream));
    GGNN_HIP_CHECK(hipStreamSynchronize(ctx.stream));
    sh.ps_state = header[4] != 0.f ? 1 : -1;
    sh.ps_measure = measure;
    GGNN_LOG(1, "[GPU: %d] pre-screen copy of part %u (%s): scale %g, max coding error %g%s",
             ctx.device, sh.global_id, measure == GGNN_EUCLIDEAN ? "L2" : "cosine", header[0],
             header[2], sh.ps_state > 0 ? "" : " (unusable, disabled)");
    if (sh.ps_state < 0) {
      sh.ps_codes.release();
      sh.ps_params.release();
    }
  }
  return sh.ps_state > 0;
}

std::filesystem::path ggnn_handle::part_file(uint32_t shard) const
{
  // gpu_instance.cu:86-115 (part_<global_shard_id>.ggnn)
  return graph_dir / ("part_" + std::to_string(shard) + ".ggnn");
}

void ggnn_handle::store()
{
  GGNN_REQUIRE(has_graph(), GGNN_INVALID_STATE, "There is no graph to store.");
  if (graph_dir.empty())
    graph_dir = std::filesystem::current_path();
  for_each_device([&](DeviceCtx& ctx) {
    if (ctx.swap) {
      // out-of-core shards: every pool is in its host buffer or already in its part file
      for (uint32_t si = 0; si < ctx.shards.size(); ++si)
        if (!ctx.swap->on_disk[si]) {
          write_part(ctx.first_shard + si, host_pool_of(ctx, si));
          ctx.swap->on_disk[si] = 1;
        }
      return;
    }
    std::vector<char> host(Shard::pool_bytes(cfg));
    for (const Shard& sh : ctx.shards) {
      GGNN_HIP_CHECK(hipMemcpy(host.data(), sh.pool.p, host.size(), hipMemcpyDeviceToHost));
      std::ofstream f(part_file(sh.global_id), std::ios::binary | std::ios::trunc);
      GGNN_REQUIRE(f.good(), GGNN_IO_ERROR, "cannot open " + part_file(sh.global_id).string());
      f.write(host.data(), static_cast<std::streamsize>(host.size()));
      GGNN_REQUIRE(f.good(), GGNN_IO_ERROR,
                   "short write to " + part_file(sh.global_id).string());
    }
  });
}

void ggnn_handle::load(uint32_t KBuild)
{
  GGNN_REQUIRE(base_set, GGNN_INVALID_STATE,
               "The base needs to be set before loading a graph.");
  if (graph_dir.empty())
    graph_dir = std::filesystem::current_path();
  prepare(KBuild);
  try {
    load_shards();
  }
  catch (...) {
    rollback_graph();
    throw;
  }
  release_caller_copy();
}

void ggnn_handle::load_shards()
{
  for_each_device([&](DeviceCtx& ctx) {
    if (ctx.swap) {
      // out-of-core shards: the files are validated now and read when a shard is first needed
      for (uint32_t si = 0; si < ctx.shards.size(); ++si) {
        const auto file = part_file(ctx.first_shard + si);
        std::error_code ec;
        const auto sz = std::filesystem::file_size(file, ec);
        GGNN_REQUIRE(!ec && sz == Shard::pool_bytes(cfg), GGNN_IO_ERROR,
                     "missing or mismatching graph file " + file.string());
        ctx.swap->on_disk[si] = 1;
        ctx.shards[si].ready = true;
      }
      return;
    }
    std::vector<char> host(Shard::pool_bytes(cfg));
    for (Shard& sh : ctx.shards) {
      const auto file = part_file(sh.global_id);
      std::error_code ec;
      const auto sz = std::filesystem::file_size(file, ec);
      // the reference validates by file size only (gpu_instance.cu:413-415)
      GGNN_REQUIRE(!ec && sz == host.size(), GGNN_IO_ERROR,
                   "missing or mismatching graph file " + file.string());
      std::ifstream f(file, std::ios::binary);
      f.read(host.data(), static_cast<std::streamsize>(host.size()));
      GGNN_REQUIRE(f.good(), GGNN_IO_ERROR, "short read from " + file.string());
      GGNN_HIP_CHECK(hipMemcpy(sh.pool.p, host.data(), host.size(), hipMemcpyHostToDevice));
      sh.ready = true;
    }
  });
}
