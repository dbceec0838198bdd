// Engine behind the C-ABI, part "query": query drivers: shard loop of one GPU, blocking / split / asynchronous batches, bf_query
// (gpu_instance.cu:626-790, ggnn.cu:278-330).
// The handle is declared in engine.hpp.
#include "engine.hpp"

void ggnn_handle::check_query(uint64_t Nq, uint32_t D, ggnn_dtype dtype, const void* q) const
{
  GGNN_REQUIRE(dtype == base_dtype, GGNN_INVALID_ARGUMENT,
               "query data type does not match base data type");
  GGNN_REQUIRE(D == base_D, GGNN_INVALID_ARGUMENT, "query dimension does not match the base");
  GGNN_REQUIRE(Nq < 0xffffffffull, GGNN_INVALID_ARGUMENT, "too many queries");
  GGNN_REQUIRE(!Nq || q != nullptr, GGNN_INVALID_ARGUMENT, "query pointer is null");
}

// query.referenceOnGPU (gpu_instance.cu:638-641): the full query set on ctx's GPU
ggnn_handle::Staged ggnn_handle::stage_query(DeviceCtx& ctx, const void* q, uint64_t Nq, uint32_t D, ggnn_dtype dtype,
                   ggnn_location loc, int q_gpu)
{
  Staged s;
  if (!Nq)
    return s;
  const size_t es = dtype_size(dtype);
  const bool padded = pad_D != base_D;
  if (loc == GGNN_GPU && q_gpu == ctx.device && !padded &&
      (reinterpret_cast<uintptr_t>(q) & 15u) == 0) {
    s.ptr = q;
    return s;
  }
  const hipMemcpyKind kind = loc == GGNN_GPU ? hipMemcpyDefault : hipMemcpyHostToDevice;
  s.owned.alloc(Nq * pad_D * es);
  if (padded) {
    GGNN_HIP_CHECK(hipMemsetAsync(s.owned.p, 0, s.owned.bytes, ctx.stream));
    GGNN_HIP_CHECK(
        hipMemcpy2DAsync(s.owned.p, pad_D * es, q, D * es, D * es, Nq, kind, ctx.stream));
  }
  else
    GGNN_HIP_CHECK(hipMemcpyAsync(s.owned.p, q, s.owned.bytes, kind, ctx.stream));
  s.ptr = s.owned.p;
  return s;
}

// GPUInstance::query, gpu_instance.cu:626-743: all shards of one GPU into d_ids/d_dists
// [Nq, K * shards_per_gpu]
void ggnn_handle::query_device(DeviceCtx& ctx, const void* d_query, uint32_t nq, uint32_t k_query,
                  float tau_query, uint32_t max_iterations, ggnn_measure measure,
                  int32_t* d_ids, float* d_dists)
{
  hipStream_t stream = ctx.stream;
  const uint32_t spg = shards_per_gpu;
  DeviceBuffer c_dist, c_pop, c_rows;
  if (collect_counters) {
    c_dist.alloc(static_cast<size_t>(nq) * 4);
    c_pop.alloc(static_cast<size_t>(nq) * 4);
    c_rows.alloc(static_cast<size_t>(nq) * 8);
  }
  ctx.query_ms = 0.f;
  ctx.n_dist = ctx.n_pop = ctx.n_float_rows = ctx.n_code_rows = 0;
  std::vector<uint32_t> h_cnt;
  // Several resident shards: one launch per shard, spread over a few streams and NOT separated
  // by host synchronisation, so the under-occupied tail of a 10k-wave launch is filled by the
  // next shard's waves (hook SHARD_OVERLAP = 0: one launch at a time, as for the work counters).
  const bool overlap = spg > 1 && !collect_counters && hook(kHookShardOverlap) != 0 && !ctx.swap;
  if (overlap) {
    for (uint32_t si = 0; si < spg; ++si)
      (void)ensure_prescreen(ctx, si, measure);  // may code a shard (synchronises): do it first
    ctx.ensure_shard_streams();
    GGNN_HIP_CHECK(hipEventRecord(ctx.ev_ready, stream));  // query staged, earlier work done
    for (int i = 0; i < DeviceCtx::kShardStreams; ++i)
      GGNN_HIP_CHECK(hipStreamWaitEvent(ctx.shard_stream[i], ctx.ev_ready, 0));
    GGNN_HIP_CHECK(hipEventRecord(ctx.ev_a, stream));
  }
  for (uint32_t si = 0; si < spg; ++si) {
    if (ctx.swap) {
      // out-of-core shards (swapInPart / waitForPart, gpu_instance.cu:661-688): this shard is
      // in its slot (uploaded as the previous one's prefetch, or right now), the next one
      // starts travelling on the copy stream while this one is searched
      SwapState& sw = *ctx.swap;
      acquire_shard(ctx, si, sw.io);
      GGNN_HIP_CHECK(hipStreamWaitEvent(stream, sw.uploaded[si % sw.slots], 0));
      if (si + 1 < spg && sw.slots > 1)
        acquire_shard(ctx, si + 1, sw.io);
    }
    const bool use_ps = ensure_prescreen(ctx, si, measure);
    const Shard& sh = ctx.shards[si];
    QueryLaunch ql{shard_base(ctx, si),
                   d_query,
                   base_dtype,
                   cfg.N,
                   pad_D,
                   nq,
                   sh.graph,
                   cfg.KBuild,
                   sh.translation + cfg.STs_offsets[kLayers - 1],
                   cfg.S,
                   sh.nn1_stats,
                   k_query,
                   tau_query,
                   max_iterations,
                   measure,
                   spg,
                   si,
                   d_ids,
                   d_dists,
                   c_dist.as<uint32_t>(),
                   c_pop.as<uint32_t>()};
    if (use_ps) {
      ql.ps_codes = sh.ps_codes.as<uint8_t>();
      ql.ps_params = sh.ps_params.as<float>();
      ql.ps_Dc = prescreen_code_dim(pad_D);
    }
    ql.n_rows = c_rows.as<uint32_t>();
    if (overlap) {
      launch_query(ql, ctx.shard_stream[si % DeviceCtx::kShardStreams]);
      continue;
    }
    EventTimer timer(stream, ctx.ev_a, ctx.ev_b);
    launch_query(ql, stream);
    if (ctx.swap)
      shard_consumed(ctx, si, stream);
    const float ms = timer.stop();
    ctx.query_ms += ms;
    GGNN_LOG(0, "[GPU: %d] query part %u => ms: %.3f [%u points query -> %.3f us/point]",
             ctx.device, sh.global_id, ms, nq, ms * 1000.f / static_cast<float>(nq));
    if (collect_counters) {
      h_cnt.resize(nq);
      GGNN_HIP_CHECK(hipMemcpy(h_cnt.data(), c_dist.p, nq * 4ull, hipMemcpyDeviceToHost));
      for (uint32_t v : h_cnt)
        ctx.n_dist += v;
      GGNN_HIP_CHECK(hipMemcpy(h_cnt.data(), c_pop.p, nq * 4ull, hipMemcpyDeviceToHost));
      for (uint32_t v : h_cnt)
        ctx.n_pop += v;
      h_cnt.resize(2ull * nq);
      GGNN_HIP_CHECK(hipMemcpy(h_cnt.data(), c_rows.p, nq * 8ull, hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < nq; ++i) {
        ctx.n_float_rows += h_cnt[2 * i];
        ctx.n_code_rows += h_cnt[2 * i + 1];
      }
    }
  }
  if (overlap) {
    // join: the main stream continues (timing event, row sort) after every shard stream
    for (int i = 0; i < DeviceCtx::kShardStreams; ++i) {
      GGNN_HIP_CHECK(hipEventRecord(ctx.shard_done[i], ctx.shard_stream[i]));
      GGNN_HIP_CHECK(hipStreamWaitEvent(stream, ctx.shard_done[i], 0));
    }
    GGNN_HIP_CHECK(hipEventRecord(ctx.ev_b, stream));
  }
  if (spg > 1)
    launch_sort_shard_results(nq, k_query * spg, d_ids, d_dists, stream);
  GGNN_HIP_CHECK(hipStreamSynchronize(stream));
  if (overlap) {
    GGNN_HIP_CHECK(hipEventElapsedTime(&ctx.query_ms, ctx.ev_a, ctx.ev_b));
    GGNN_LOG(0, "[GPU: %d] query parts %u..%u overlapped => ms: %.3f [%u points query]",
             ctx.device, ctx.first_shard, ctx.first_shard + spg - 1, ctx.query_ms, nq);
  }
}

// GGNNImpl::queryImpl, ggnn.cu:278-330
void ggnn_handle::query(const void* q, uint64_t Nq, uint32_t D, ggnn_dtype dtype, ggnn_location loc,
           int q_gpu, uint32_t k_query, float tau_query, uint32_t max_iterations,
           ggnn_measure measure, int32_t* ids_out, float* dists_out, ggnn_location out_loc)
{
  GGNN_REQUIRE(has_graph(), GGNN_INVALID_STATE, "There is no graph to query.");
  check_query(Nq, D, dtype, q);
  const bool direct = (out_loc == GGNN_GPU);
  GGNN_REQUIRE(!(direct && devs.size() > 1), GGNN_INVALID_STATE,
               "Returning query results on GPU is only possible when using a single GPU.");
  query_ms = 0.f;
  last_n_dist = last_n_pop = last_float_rows = last_code_rows = 0;
  if (!Nq)
    return;
  const uint32_t nq = static_cast<uint32_t>(Nq);
  const size_t row = static_cast<size_t>(k_query) * shards_per_gpu;
  const size_t part = nq * row;

  // Several GPUs: a blocking batch is searched as TWO half-batches in flight, so that the
  // all-gather, the slice merges and the result copies of the first half overlap the search of
  // the second (the step is latency-bound: one ~2 ms kernel per GPU, then the exchange) -- the
  // caller gets the pipelining of query_async without having to use it.  Hook QUERY_SPLIT = 0
  // switches it off, 1 forces it from 2 queries on (tests).
  last_query_parts = 1;
  {
    const int64_t split = hook(kHookQuerySplit);
    const bool several = devs.size() > 1 || hook(kHookExchange) == 1;
    const bool want = split >= 0 ? split == 1 : nq >= 4096;
    if (several && !direct && !collect_counters && want && nq >= 2 && !swapping()) {
      query_split(q, nq, D, dtype, loc, q_gpu, k_query, tau_query, max_iterations, measure,
                  ids_out, dists_out);
      return;
    }
  }

  constexpr int lane = DeviceCtx::kBlockingLane;
  for_each_device([&](DeviceCtx& ctx) {
    Staged sq = stage_query(ctx, q, Nq, D, dtype, loc, q_gpu);
    int32_t* d_ids = ids_out;
    float* d_dists = dists_out;
    if (!direct) {
      DeviceCtx::grow(ctx.xb[lane].r_pack, 2 * part * 4);
      d_ids = ctx.xb[lane].r_pack.as<int32_t>();
      d_dists = reinterpret_cast<float*>(d_ids + part);
    }
    query_device(ctx, sq.ptr, nq, k_query, tau_query, max_iterations, measure, d_ids, d_dists);
  });
  for (const DeviceCtx& ctx : devs) {
    query_ms = std::max(query_ms, ctx.query_ms);  // GPUs run concurrently
    last_n_dist += ctx.n_dist;
    last_n_pop += ctx.n_pop;
    last_float_rows += ctx.n_float_rows;
    last_code_rows += ctx.n_code_rows;
  }
  if (direct)
    return;

  const bool force_rccl = hook(kHookExchange) == 1;
  if (devs.size() == 1 && !force_rccl) {
    // ResultMerger::merge for one GPU: first K of each pre-sorted row (result_merger.cpp:55-73)
    DeviceCtx& d0 = devs[0];
    d0.activate();
    const int32_t* r = d0.xb[lane].r_pack.as<int32_t>();
    GGNN_HIP_CHECK(hipMemcpy2DAsync(ids_out, k_query * 4ull, r, row * 4, k_query * 4ull, nq,
                                    hipMemcpyDeviceToHost, d0.stream));
    GGNN_HIP_CHECK(hipMemcpy2DAsync(dists_out, k_query * 4ull, r + part, row * 4,
                                    k_query * 4ull, nq, hipMemcpyDeviceToHost, d0.stream));
    GGNN_HIP_CHECK(hipStreamSynchronize(d0.stream));
    last_exchange = "none";
    return;
  }
  exchange(lane, nq, k_query, row, ids_out, dists_out, /*blocking=*/true);
}

// blocking multi-GPU query as two half-batches on the asynchronous lanes 0 and 1 (see query())
void ggnn_handle::query_split(const void* q, uint32_t nq, uint32_t D, ggnn_dtype dtype, ggnn_location loc,
                 int q_gpu, uint32_t k_query, float tau_query, uint32_t max_iterations,
                 ggnn_measure measure, int32_t* ids_out, float* dists_out)
{
  const size_t row = static_cast<size_t>(k_query) * shards_per_gpu;
  const size_t es = dtype_size(dtype);
  const uint32_t first[2] = {0u, nq / 2};
  const uint32_t count[2] = {nq / 2, nq - nq / 2};
  std::vector<Staged> staged(devs.size());
  // the whole query set once per GPU, both halves enqueued on their lanes; nothing waits yet
  for (size_t g = 0; g < devs.size(); ++g) {
    DeviceCtx& ctx = devs[g];
    ctx.activate();
    for (uint32_t si = 0; si < shards_per_gpu; ++si)
      (void)ensure_prescreen(ctx, si, measure);
    ctx.ensure_shard_streams();
    staged[g] = stage_query(ctx, q, nq, D, dtype, loc, q_gpu);
    GGNN_HIP_CHECK(hipEventRecord(ctx.ev_ready, ctx.stream));
    GGNN_HIP_CHECK(hipEventRecord(ctx.ev_a, ctx.stream));
    for (int half = 0; half < 2; ++half) {
      if (!count[half])
        continue;
      const int lane = half;
      hipStream_t st = ctx.lane_stream(lane);
      GGNN_HIP_CHECK(hipStreamWaitEvent(st, ctx.ev_ready, 0));
      DeviceCtx::ExchangeBufs& x = ctx.xb[lane];
      if (&ctx != &devs[0] && devs[0].xb[lane].consumed)
        GGNN_HIP_CHECK(hipStreamWaitEvent(st, devs[0].xb[lane].consumed, 0));
      const size_t part = count[half] * row;
      grow_lane(ctx, lane, x.r_pack, 2 * part * 4);
      int32_t* r = x.r_pack.as<int32_t>();
      const uint8_t* qh = static_cast<const uint8_t*>(staged[g].ptr) +
                          static_cast<size_t>(first[half]) * pad_D * es;
      enqueue_local_search(ctx, lane, qh, count[half], k_query, tau_query, max_iterations,
                           measure, r, reinterpret_cast<float*>(r + part));
      GGNN_HIP_CHECK(hipEventRecord(ctx.shard_done[lane], st));
      GGNN_HIP_CHECK(hipStreamWaitEvent(ctx.stream, ctx.shard_done[lane], 0));
    }
    GGNN_HIP_CHECK(hipEventRecord(ctx.ev_b, ctx.stream));  // both halves searched on this GPU
  }
  // the first half is exchanged, merged and copied out while the second is still being searched
  for (int half = 0; half < 2; ++half)
    if (count[half])
      exchange(half, count[half], k_query, row, ids_out + static_cast<size_t>(first[half]) * k_query,
               dists_out + static_cast<size_t>(first[half]) * k_query, /*blocking=*/true);
  for (DeviceCtx& ctx : devs) {
    ctx.activate();
    GGNN_HIP_CHECK(hipStreamSynchronize(ctx.stream));
    float ms = 0.f;
    GGNN_HIP_CHECK(hipEventElapsedTime(&ms, ctx.ev_a, ctx.ev_b));
    ctx.query_ms = ms;
    query_ms = std::max(query_ms, ms);
    GGNN_LOG(0, "[GPU: %d] query parts %u..%u, two half-batches in flight => ms: %.3f [%u points "
                "query]", ctx.device, ctx.first_shard, ctx.first_shard + shards_per_gpu - 1, ms, nq);
  }
  last_query_parts = 2;
}

// Grows one exchange buffer of a lane.  Batches in flight on the lane may still use the old
// allocation -- also from ANOTHER GPU's stream (peer copies read r_pack, RCCL kernels write
// g_pack), which the hipFree of the owning device does not wait for: every GPU's stream of the
// lane is drained first.  Rare: the first batch on a lane, or a larger one than any before.
void ggnn_handle::grow_lane(DeviceCtx& owner, int lane, DeviceBuffer& b, size_t bytes)
{
  if (b.bytes >= bytes)
    return;
  for (DeviceCtx& ctx : devs) {
    hipStream_t st = ctx.lane_stream(lane);
    if (!st)
      continue;
    ctx.activate();
    GGNN_HIP_CHECK(hipStreamSynchronize(st));
  }
  owner.activate();
  b.alloc(bytes);
}

// Extension for serving: enqueue one query batch and return.  Consecutive batches given
// different slots run on different streams, so the under-occupied tail of one batch's launch
// overlaps with the head of the next (a lone 10k-query launch is latency-bound).
//
// One GPU: query and result arrays already on that GPU (nothing is staged); results are the
// sorted [Nq, K * shards] rows of results-on-GPU mode (ggnn.cuh:108-113).
// Several GPUs: the query may live on any GPU of the node or in page-locked host memory (it is
// copied to every GPU on the slot's stream), results are the MERGED [Nq, K] arrays, written
// by asynchronous copies (device memory of any GPU, or page-locked host memory; with pageable
// memory the copies degrade to synchronous ones).  Local search, the RCCL all-gather, the
// slice merges and the result copies of batch i+1 are all enqueued while batch i runs.
// Valid after synchronize() / synchronize_slot(slot).
//
// Ordering contract with the blocking query(): both may be used on one handle from one
// thread; a blocking call does not wait for batches in flight (its buffers and stream are its
// own), and a call that has to re-code the pre-screen copy for another measure first drains
// every slot.
void ggnn_handle::query_async(const void* d_query, uint64_t Nq, uint32_t D, ggnn_dtype dtype,
                 ggnn_location loc, int q_gpu, uint32_t k_query, float tau_query,
                 uint32_t max_iterations, ggnn_measure measure, int32_t* d_ids, float* d_dists,
                 uint32_t slot)
{
  GGNN_REQUIRE(has_graph(), GGNN_INVALID_STATE, "There is no graph to query.");
  check_query(Nq, D, dtype, d_query);
  GGNN_REQUIRE(!Nq || (d_ids != nullptr && d_dists != nullptr), GGNN_INVALID_ARGUMENT,
               "result pointers are null");
  GGNN_REQUIRE(!swapping(), GGNN_UNSUPPORTED,
               "asynchronous queries need every shard resident on its GPU (the shards of this "
               "handle take turns in GPU memory)");
  if (!Nq)
    return;
  const uint32_t nq = static_cast<uint32_t>(Nq);
  const int lane = static_cast<int>(slot % DeviceCtx::kShardStreams);
  // the pre-screen copy of another measure is replaced below: nothing may still read it
  bool recode = false;
  for (const DeviceCtx& ctx : devs)
    for (const Shard& sh : ctx.shards)
      recode = recode || (sh.ps_state != 0 && sh.ps_measure != measure);
  if (recode)
    synchronize();
  const bool force_rccl = hook(kHookExchange) == 1;
  if (devs.size() == 1 && !force_rccl) {
    DeviceCtx& ctx = devs[0];
    GGNN_REQUIRE(loc == GGNN_GPU && q_gpu == ctx.device, GGNN_INVALID_ARGUMENT,
                 "asynchronous queries need the query on the engine's GPU");
    GGNN_REQUIRE(pad_D == base_D && (reinterpret_cast<uintptr_t>(d_query) & 15u) == 0,
                 GGNN_UNSUPPORTED,
                 "asynchronous queries need 16-byte aligned rows (no padding is staged)");
    ctx.activate();
    for (uint32_t si = 0; si < shards_per_gpu; ++si)
      (void)ensure_prescreen(ctx, si, measure);
    ctx.ensure_shard_streams();
    enqueue_local_search(ctx, lane, d_query, nq, k_query, tau_query, max_iterations, measure,
                         d_ids, d_dists);
    return;
  }
  GGNN_REQUIRE(pad_D == base_D, GGNN_UNSUPPORTED,
               "asynchronous queries need 16-byte rows (no padding is staged)");
  const size_t row = static_cast<size_t>(k_query) * shards_per_gpu;
  const size_t part = nq * row;
  const size_t qbytes = Nq * static_cast<size_t>(pad_D) * dtype_size(dtype);
  for (DeviceCtx& ctx : devs) {
    ctx.activate();
    for (uint32_t si = 0; si < shards_per_gpu; ++si)
      (void)ensure_prescreen(ctx, si, measure);
    ctx.ensure_shard_streams();
    DeviceCtx::ExchangeBufs& x = ctx.xb[lane];
    hipStream_t st = ctx.lane_stream(lane);
    const void* q_here = d_query;
    if (!(loc == GGNN_GPU && q_gpu == ctx.device &&
          (reinterpret_cast<uintptr_t>(d_query) & 15u) == 0)) {
      grow_lane(ctx, lane, x.q_stage, qbytes);
      GGNN_HIP_CHECK(hipMemcpyAsync(x.q_stage.p, d_query, qbytes, hipMemcpyDefault, st));
      q_here = x.q_stage.p;
    }
    // (copy exchange: the first GPU may still be copying this lane's previous rows)
#ifndef GGNN_EXP_NO_CONSUMED_WAIT  // (test-the-test hook)
    if (&ctx != &devs[0] && devs[0].xb[lane].consumed)
      GGNN_HIP_CHECK(hipStreamWaitEvent(st, devs[0].xb[lane].consumed, 0));
#endif
    grow_lane(ctx, lane, x.r_pack, 2 * part * 4);
    int32_t* r = x.r_pack.as<int32_t>();
    enqueue_local_search(ctx, lane, q_here, nq, k_query, tau_query, max_iterations, measure, r,
                         reinterpret_cast<float*>(r + part));
  }
  exchange(lane, nq, k_query, row, d_ids, d_dists, /*blocking=*/false);
}

// the shards of one GPU on one lane's stream, nothing waits
void ggnn_handle::enqueue_local_search(DeviceCtx& ctx, int lane, const void* d_query, uint32_t nq,
                          uint32_t k_query, float tau_query, uint32_t max_iterations,
                          ggnn_measure measure, int32_t* d_ids, float* d_dists)
{
  hipStream_t stream = ctx.lane_stream(lane);
  for (uint32_t si = 0; si < shards_per_gpu; ++si) {
    const Shard& sh = ctx.shards[si];
    QueryLaunch ql{shard_base(ctx, si), d_query, base_dtype, cfg.N, pad_D, nq, sh.graph,
                   cfg.KBuild, sh.translation + cfg.STs_offsets[kLayers - 1], cfg.S,
                   sh.nn1_stats, k_query, tau_query, max_iterations, measure, shards_per_gpu, si,
                   d_ids, d_dists, nullptr, nullptr};
    if (sh.ps_state > 0 && sh.ps_measure == measure) {
      ql.ps_codes = sh.ps_codes.as<uint8_t>();
      ql.ps_params = sh.ps_params.as<float>();
      ql.ps_Dc = prescreen_code_dim(pad_D);
    }
    launch_query(ql, stream);
  }
  if (shards_per_gpu > 1)
    launch_sort_shard_results(nq, k_query * shards_per_gpu, d_ids, d_dists, stream);
}

// wait for the batches enqueued on one slot only (the other slots keep running)
void ggnn_handle::synchronize_slot(uint32_t slot)
{
  for (DeviceCtx& ctx : devs) {
    ctx.activate();
    hipStream_t st = ctx.shard_stream[slot % DeviceCtx::kShardStreams];
    if (st)
      GGNN_HIP_CHECK(hipStreamSynchronize(st));
  }
}

void ggnn_handle::synchronize()
{
  for (DeviceCtx& ctx : devs) {
    ctx.activate();
    GGNN_HIP_CHECK(hipStreamSynchronize(ctx.stream));
    for (int i = 0; i < DeviceCtx::kShardStreams; ++i)
      if (ctx.shard_stream[i])
        GGNN_HIP_CHECK(hipStreamSynchronize(ctx.shard_stream[i]));
  }
}

// GGNNImpl::bfQueryImpl, ggnn.cu:332-390
void ggnn_handle::bf_query(const void* q, uint64_t Nq, uint32_t D, ggnn_dtype dtype, ggnn_location loc,
              int q_gpu, uint32_t k_gt, ggnn_measure measure, int32_t* ids_out,
              float* dists_out, ggnn_location out_loc)
{
  GGNN_REQUIRE(base_set, GGNN_INVALID_STATE,
               "There is no base dataset loaded which could be queried.");
  GGNN_REQUIRE(devs.size() <= 1, GGNN_INVALID_STATE,
               "The brute-force query only supports a single GPU.");
  check_query(Nq, D, dtype, q);
  if (devs.empty()) {
    // no graph yet: make the whole base resident on the (first) selected GPU
    const std::vector<int> gpus = resolve_gpus();
    devs.resize(1);
    devs[0].device = gpus[0];
    stage_base_slice(devs[0], 0, base_N);
  }
  DeviceCtx& ctx = devs[0];
  ctx.activate();
  bf_ms = 0.f;
  if (!Nq)
    return;
  // out-of-core shards with the rows on the host: the exhaustive scan needs the whole base on
  // the GPU for the duration of the call (fails with GGNN_OUT_OF_MEMORY if that is too much)
  DeviceBuffer whole_base;
  const void* bf_base = ctx.d_base;
  if (ctx.swap && !ctx.swap->base_borrowed) {
    const size_t es = dtype_size(base_dtype);
    whole_base.alloc(base_N * pad_D * es);
    const hipMemcpyKind kind = base_loc == GGNN_GPU ? hipMemcpyDefault : hipMemcpyHostToDevice;
    if (pad_D != base_D) {
      GGNN_HIP_CHECK(hipMemsetAsync(whole_base.p, 0, whole_base.bytes, ctx.stream));
      GGNN_HIP_CHECK(hipMemcpy2DAsync(whole_base.p, pad_D * es, base_src, base_D * es, base_D * es,
                                      base_N, kind, ctx.stream));
    }
    else
      GGNN_HIP_CHECK(hipMemcpyAsync(whole_base.p, base_src, whole_base.bytes, kind, ctx.stream));
    bf_base = whole_base.p;
  }
  Staged sq = stage_query(ctx, q, Nq, D, dtype, loc, q_gpu);
  const uint32_t nq = static_cast<uint32_t>(Nq);
  const bool direct = (out_loc == GGNN_GPU);
  DeviceBuffer r_ids, r_dists;
  int32_t* d_ids = ids_out;
  float* d_dists = dists_out;
  if (!direct) {
    r_ids.alloc(static_cast<size_t>(nq) * k_gt * 4);
    r_dists.alloc(static_cast<size_t>(nq) * k_gt * 4);
    d_ids = r_ids.as<int32_t>();
    d_dists = r_dists.as<float>();
  }
  if (!ctx.bf_rescanned.p)
    ctx.bf_rescanned.alloc(sizeof(uint32_t));
  BfLaunch bl{bf_base, sq.ptr, base_dtype, static_cast<uint32_t>(base_N), pad_D, nq, k_gt,
              measure,    d_ids,  d_dists,    ctx.bf_rescanned.as<uint32_t>()};
  EventTimer timer(ctx.stream, ctx.ev_a, ctx.ev_b);
  launch_bf_query(bl, ctx.stream);
  bf_ms = timer.stop();
  GGNN_HIP_CHECK(hipMemcpyAsync(&last_bf_rescanned, ctx.bf_rescanned.p, sizeof(uint32_t),
                                hipMemcpyDeviceToHost, ctx.stream));
  GGNN_LOG(0, "[GPU: %d] brute-force query: => ms: %.3f [%u points query -> %.3f us/point]",
           ctx.device, bf_ms, nq, bf_ms * 1000.f / static_cast<float>(nq));
  if (!direct) {
    GGNN_HIP_CHECK(hipMemcpyAsync(ids_out, d_ids, static_cast<size_t>(nq) * k_gt * 4,
                                  hipMemcpyDeviceToHost, ctx.stream));
    GGNN_HIP_CHECK(hipMemcpyAsync(dists_out, d_dists, static_cast<size_t>(nq) * k_gt * 4,
                                  hipMemcpyDeviceToHost, ctx.stream));
  }
  GGNN_HIP_CHECK(hipStreamSynchronize(ctx.stream));
}
