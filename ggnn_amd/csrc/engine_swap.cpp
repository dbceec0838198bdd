// Engine behind the C-ABI, part "swap": out-of-core shards: GPU slots <-> page-locked host buffers <-> part files (SURVEY 8(f)4;
// reference: gpu_instance.cu:135-227, 371-497).
// The handle is declared in engine.hpp.
#include "engine.hpp"

// GPU slots per device: 0 = every shard resident (the normal case).  Hook RESIDENT_SHARDS forces
// a number (tests); otherwise the shards are counted against the free device memory minus
// ggnn_set_reserved_gpu_memory, as GPUInstance::allocateGraph does (gpu_instance.cu:157-186).
uint32_t ggnn_handle::plan_gpu_slots(const DeviceCtx& ctx, uint32_t spg, bool base_on_this_gpu) const
{
  const int64_t forced = hook(kHookResidentShards);
  if (forced > 0)
    return forced >= spg ? 0u : static_cast<uint32_t>(forced);
  size_t free_b = 0, total_b = 0;
  GGNN_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  GGNN_REQUIRE(free_b > reserved_gpu_memory, GGNN_OUT_OF_MEMORY,
               "GPU memory does not suffice for the reserved amount.");
  free_b -= reserved_gpu_memory;
  const size_t pool_b = align8(Shard::pool_bytes(cfg));
  const size_t base_b = base_on_this_gpu ? 0 : static_cast<size_t>(cfg.N) * row_bytes();
  // construction scratch of one shard (build_device) + the optional 8-bit copy of resident shards
  const size_t scratch_b = static_cast<size_t>(cfg.N) * (cfg.KBuild * 4 + cfg.KF * 4 + 16);
  const size_t codes_b = (prescreen && base_dtype == GGNN_F32 && pad_D >= 64)
                             ? static_cast<size_t>(cfg.N) * prescreen_code_dim(pad_D) : 0;
  const size_t resident_need = spg * (pool_b + base_b + codes_b) + scratch_b;
  (void)ctx;
  if (resident_need <= free_b)
    return 0;
  // out of core: the slots must leave room for what is allocated later -- query staging and
  // result buffers, the packed exchange blocks, the scratch pool of the brute force and of long
  // searches (round-4 advisor finding: the slots took everything but the build scratch)
  const size_t headroom = std::min<size_t>(free_b / 8, size_t{2} << 30);
  GGNN_REQUIRE(free_b > scratch_b + headroom + pool_b + base_b, GGNN_OUT_OF_MEMORY,
               "GPU memory does not suffice for a single shard. use smaller shards.");
  return static_cast<uint32_t>(
      std::min<size_t>(spg - 1, (free_b - scratch_b - headroom) / (pool_b + base_b)));
}

void ggnn_handle::setup_swap(DeviceCtx& ctx, uint32_t slots, bool base_on_this_gpu)
{
  auto sw = std::make_unique<SwapState>();
  sw->device = ctx.device;
  sw->slots = slots;
  sw->base_borrowed = base_on_this_gpu;
  if (!base_on_this_gpu && base_loc == GGNN_CPU && base_src == base_host_copy.data())
    pin_host_copy();
  const size_t pool_b = align8(Shard::pool_bytes(cfg));
  // host buffers first (fail early, as the reference does): ggnn_set_cpu_memory_limit bounds them
  const size_t host_n = std::max<size_t>(
      1, std::min<size_t>(shards_per_gpu, cpu_memory_limit / std::max<size_t>(1, pool_b)));
  GGNN_REQUIRE(cpu_memory_limit >= pool_b, GGNN_OUT_OF_MEMORY,
               "CPU memory does not suffice for a single shard. use smaller shards.");
  sw->host.resize(host_n);
  for (PinnedBuffer& b : sw->host)
    b.grow(pool_b);
  sw->host_shard.assign(host_n, -1);
  sw->on_disk.assign(shards_per_gpu, 0);
  sw->pool.resize(slots);
  sw->base.resize(slots);
  for (uint32_t k = 0; k < slots; ++k) {
    sw->pool[k].alloc(pool_b);
    if (!base_on_this_gpu)
      sw->base[k].alloc(static_cast<size_t>(cfg.N) * row_bytes());
  }
  sw->pool_shard.assign(slots, -1);
  sw->base_shard.assign(slots, -1);
  GGNN_HIP_CHECK(hipStreamCreateWithFlags(&sw->io, hipStreamNonBlocking));
  sw->uploaded.assign(slots, nullptr);
  sw->consumed.assign(slots, nullptr);
  for (uint32_t k = 0; k < slots; ++k) {
    GGNN_HIP_CHECK(hipEventCreateWithFlags(&sw->uploaded[k], hipEventDisableTiming));
    GGNN_HIP_CHECK(hipEventCreateWithFlags(&sw->consumed[k], hipEventDisableTiming));
  }
  ctx.swap = std::move(sw);
  GGNN_LOG(1, "[GPU: %d] out-of-core shards: %u GPU slot(s), %zu host buffer(s) for %u shards%s",
           ctx.device, slots, host_n, shards_per_gpu,
           host_n < shards_per_gpu ? ", the rest on disk" : "");
}

// rows of local shard si -> its slot's base buffer, on `st`
void ggnn_handle::upload_base_shard(DeviceCtx& ctx, uint32_t si, hipStream_t st)
{
  SwapState& sw = *ctx.swap;
  if (sw.base_borrowed)
    return;
  const uint32_t k = si % sw.slots;
  if (sw.base_shard[k] == static_cast<int>(si))
    return;
  const size_t es = dtype_size(base_dtype);
  const uint64_t row0 = (static_cast<uint64_t>(ctx.first_shard) + si) * cfg.N;
  const uint8_t* src = static_cast<const uint8_t*>(base_src) + row0 * base_D * es;
  const hipMemcpyKind kind = base_loc == GGNN_GPU ? hipMemcpyDefault : hipMemcpyHostToDevice;
  if (pad_D != base_D) {
    GGNN_HIP_CHECK(hipMemsetAsync(sw.base[k].p, 0, sw.base[k].bytes, st));
    GGNN_HIP_CHECK(hipMemcpy2DAsync(sw.base[k].p, pad_D * es, src, base_D * es, base_D * es, cfg.N,
                                    kind, st));
  }
  else
    GGNN_HIP_CHECK(hipMemcpyAsync(sw.base[k].p, src, static_cast<size_t>(cfg.N) * pad_D * es, kind, st));
  sw.base_shard[k] = static_cast<int>(si);
}

// graph pool of local shard si in its host buffer (read from its part file if it is not there)
void* ggnn_handle::host_pool_of(DeviceCtx& ctx, uint32_t si)
{
  SwapState& sw = *ctx.swap;
  const size_t h = si % sw.host.size();
  if (sw.host_shard[h] != static_cast<int>(si)) {
    GGNN_REQUIRE(sw.on_disk[si], GGNN_INVALID_STATE,
                 "graph part " + std::to_string(ctx.first_shard + si) + " is neither in memory nor on disk");
    // (the buffer may still feed an upload of the shard it held: uploads are synchronous w.r.t.
    // the host here because every acquire waits for `uploaded` before it returns to the loop)
    GGNN_HIP_CHECK(hipStreamSynchronize(sw.io));
    read_part(ctx.first_shard + si, sw.host[h].p);
    sw.host_shard[h] = static_cast<int>(si);
  }
  return sw.host[h].p;
}

// Makes local shard si usable on the GPU: graph pool and base rows in slot si % slots, the
// shard's pointers set.  Copies run on `st` (the io stream for a prefetch); the slot's previous
// user is waited for through its `consumed` event, the caller orders its kernels behind
// `uploaded`.
void ggnn_handle::acquire_shard(DeviceCtx& ctx, uint32_t si, hipStream_t st, bool with_graph)
{
  SwapState& sw = *ctx.swap;
  const uint32_t k = si % sw.slots;
  Shard& sh = ctx.shards[si];
  const bool need_pool = with_graph && sw.pool_shard[k] != static_cast<int>(si);
  const bool need_base = !sw.base_borrowed && sw.base_shard[k] != static_cast<int>(si);
  if (need_pool || need_base || !with_graph)
    GGNN_HIP_CHECK(hipStreamWaitEvent(st, sw.consumed[k], 0));
  if (need_pool) {
    void* hp = host_pool_of(ctx, si);
    GGNN_HIP_CHECK(hipMemcpyAsync(sw.pool[k].p, hp, Shard::pool_bytes(cfg), hipMemcpyHostToDevice, st));
    sw.pool_shard[k] = static_cast<int>(si);
  }
  if (!with_graph)
    sw.pool_shard[k] = static_cast<int>(si);  // about to be built in place
  upload_base_shard(ctx, si, st);
  sh.view(cfg, sw.pool[k].p);
  GGNN_HIP_CHECK(hipEventRecord(sw.uploaded[k], st));
}

// the kernels enqueued on `st` so far are the last users of shard si's slot
void ggnn_handle::shard_consumed(DeviceCtx& ctx, uint32_t si, hipStream_t st)
{
  SwapState& sw = *ctx.swap;
  GGNN_HIP_CHECK(hipEventRecord(sw.consumed[si % sw.slots], st));
}

// after build: the graph pool of shard si goes to its host buffer and, when the host buffers do
// not hold every shard, to its part file at once (swapOutPart, gpu_instance.cu:372-425)
void ggnn_handle::retire_built_shard(DeviceCtx& ctx, uint32_t si)
{
  SwapState& sw = *ctx.swap;
  const uint32_t k = si % sw.slots;
  const size_t h = si % sw.host.size();
  GGNN_HIP_CHECK(hipStreamSynchronize(ctx.stream));
  GGNN_HIP_CHECK(hipMemcpy(sw.host[h].p, sw.pool[k].p, Shard::pool_bytes(cfg), hipMemcpyDeviceToHost));
  sw.host_shard[h] = static_cast<int>(si);
  sw.on_disk[si] = 0;
  if (sw.host.size() < shards_per_gpu) {
    // (graph_dir was resolved in prepare(): this runs on one host thread per GPU)
    write_part(ctx.first_shard + si, sw.host[h].p);
    sw.on_disk[si] = 1;
  }
}

void ggnn_handle::write_part(uint32_t global_shard, const void* host)
{
  const auto file = part_file(global_shard);
  std::ofstream f(file, std::ios::binary | std::ios::trunc);
  GGNN_REQUIRE(f.good(), GGNN_IO_ERROR, "cannot open " + file.string());
  f.write(static_cast<const char*>(host), static_cast<std::streamsize>(Shard::pool_bytes(cfg)));
  GGNN_REQUIRE(f.good(), GGNN_IO_ERROR, "short write to " + file.string());
}

void ggnn_handle::read_part(uint32_t global_shard, void* host)
{
  const auto file = part_file(global_shard);
  std::error_code ec;
  const auto sz = std::filesystem::file_size(file, ec);
  // the reference validates by file size only (gpu_instance.cu:413-415)
  GGNN_REQUIRE(!ec && sz == Shard::pool_bytes(cfg), GGNN_IO_ERROR,
               "missing or mismatching graph file " + file.string());
  std::ifstream f(file, std::ios::binary);
  f.read(static_cast<char*>(host), static_cast<std::streamsize>(Shard::pool_bytes(cfg)));
  GGNN_REQUIRE(f.good(), GGNN_IO_ERROR, "short read from " + file.string());
}
