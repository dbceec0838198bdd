// Engine behind the C-ABI, part "exchange": exchange of the sorted candidates between the GPUs of one handle: in-engine RCCL all-gather
// + per-GPU slice merges, peer-copy fallback (replaces result_merger.cpp:51-149).
// The handle is declared in engine.hpp.
#include "engine.hpp"

void ggnn_handle::destroy_comms()
{
  if (!comms.empty() && Rccl::get().ok)
    for (ncclComm_t c : comms)
      if (c)
        (void)Rccl::get().CommDestroy(c);
  comms.clear();
  rccl_state = 0;
}

// RCCL needs distinct devices per rank; a handle whose contexts share a device (tests on a
// one-GPU box) and builds without librccl exchange through peer copies instead.
// Hook EXCHANGE = 1 (rccl) | 2 (copy) forces one of the two (rccl also for a single GPU: a
// 1-rank world); 3 (gather): exchange_gather_copies below.
bool ggnn_handle::ensure_comms()
{
  if (rccl_state != 0)
    return rccl_state > 0;
  rccl_state = -1;
  if (hook(kHookExchange) >= 2)
    return false;
  std::vector<int> ids;
  for (const DeviceCtx& ctx : devs)
    ids.push_back(ctx.device);
  std::vector<int> uniq = ids;
  std::sort(uniq.begin(), uniq.end());
  if (std::adjacent_find(uniq.begin(), uniq.end()) != uniq.end())
    return false;
  if (!Rccl::get().ok) {
    GGNN_LOG(0, "librccl.so not found: exchanging shard results with peer copies");
    return false;
  }
  comms.assign(ids.size(), nullptr);
  const ncclResult_t r = Rccl::get().CommInitAll(comms.data(), static_cast<int>(ids.size()),
                                                 ids.data());
  if (r != ncclSuccess) {
    GGNN_LOG(0, "ncclCommInitAll failed (%s): exchanging shard results with peer copies",
             Rccl::get().GetErrorString(r));
    comms.clear();
    return false;
  }
  rccl_state = 1;
  return true;
}

// Combines the per-GPU rows of one lane into the caller's [Nq, K] arrays.  blocking: waits and
// copies through pinned staging; otherwise everything is only enqueued on the lane's streams
// (the caller's arrays are written by asynchronous copies: device or page-locked memory).
void ggnn_handle::exchange(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
              float* dists_out, bool blocking)
{
  if (ensure_comms()) {
    try {
      exchange_rccl(lane, nq, k_query, row, ids_out, dists_out, blocking);
      return;
    }
    catch (const RcclError& e) {
      // A failed collective leaves the communicators unusable: drop them for good and serve this
      // and every later call through peer copies.  Only RCCL's own failures take this path (an
      // out-of-memory or HIP error propagates to the caller).  Other lanes may still have
      // all-gathers enqueued on these communicators, and after a partial group failure some
      // ranks hold a collective that will never complete on its own: every stream of every GPU
      // is drained (best effort) before the communicators go.
      GGNN_LOG(0, "RCCL exchange failed (%s): falling back to peer copies", e.what());
      for (DeviceCtx& ctx : devs) {
        ctx.activate();
        (void)hipStreamSynchronize(ctx.stream);
        for (int i = 0; i < DeviceCtx::kShardStreams; ++i)
          if (ctx.shard_stream[i])
            (void)hipStreamSynchronize(ctx.shard_stream[i]);
      }
      (void)hipGetLastError();
      destroy_comms();
      rccl_state = -1;
      ++rccl_fallbacks;
    }
  }
  if (blocking && hook(kHookExchange) == 3) {
    exchange_gather_copies(lane, nq, k_query, row, ids_out, dists_out);
    return;
  }
  exchange_peer_copies(lane, nq, k_query, row, ids_out, dists_out, blocking);
}

// slice of the query set that GPU g merges and returns
void ggnn_handle::slice_of(uint32_t nq, size_t G, size_t g, uint32_t* first, uint32_t* count)
{
  const uint32_t per = (nq + static_cast<uint32_t>(G) - 1) / static_cast<uint32_t>(G);
  *first = std::min<uint32_t>(nq, static_cast<uint32_t>(g) * per);
  *count = std::min<uint32_t>(per, nq - *first);
}

// merged slice [first, first + count) of ctx.m_pack -> caller's arrays
void ggnn_handle::return_slice(DeviceCtx& ctx, int lane, uint32_t nq, uint32_t k_query, uint32_t first,
                  uint32_t count, int32_t* ids_out, float* dists_out, bool blocking)
{
  DeviceCtx::ExchangeBufs& x = ctx.xb[lane];
  hipStream_t st = ctx.lane_stream(lane);
  const size_t off = static_cast<size_t>(first) * k_query;
  const size_t n = static_cast<size_t>(count) * k_query;
  const int32_t* m_ids = x.m_pack.as<int32_t>() + off;
  const int32_t* m_dists = x.m_pack.as<int32_t>() + static_cast<size_t>(nq) * k_query + off;
  if (blocking) {
    x.h_pack.grow(2 * n * 4);
    int32_t* h = static_cast<int32_t*>(x.h_pack.p);
    GGNN_HIP_CHECK(hipMemcpyAsync(h, m_ids, n * 4, hipMemcpyDeviceToHost, st));
    GGNN_HIP_CHECK(hipMemcpyAsync(h + n, m_dists, n * 4, hipMemcpyDeviceToHost, st));
  }
  else {
    GGNN_HIP_CHECK(hipMemcpyAsync(ids_out + off, m_ids, n * 4, hipMemcpyDefault, st));
    GGNN_HIP_CHECK(hipMemcpyAsync(dists_out + off, m_dists, n * 4, hipMemcpyDefault, st));
  }
}

void ggnn_handle::finish_slices(int lane, uint32_t nq, uint32_t k_query, size_t G_slices, int32_t* ids_out,
                   float* dists_out)
{
  for (DeviceCtx& ctx : devs) {
    ctx.activate();
    GGNN_HIP_CHECK(hipStreamSynchronize(ctx.lane_stream(lane)));
  }
  for (size_t g = 0; g < G_slices; ++g) {
    uint32_t first, count;
    slice_of(nq, G_slices, g, &first, &count);
    if (!count)
      continue;
    const size_t off = static_cast<size_t>(first) * k_query;
    const size_t n = static_cast<size_t>(count) * k_query;
    const int32_t* h = static_cast<const int32_t*>(devs[g].xb[lane].h_pack.p);
    std::memcpy(ids_out + off, h, n * 4);
    std::memcpy(dists_out + off, h + n, n * 4);
  }
}

// Several GPUs, RCCL: every GPU contributes its packed sorted rows (ids and distance bit
// patterns in one buffer) to ONE grouped all-gather over xGMI, merges a 1/G slice of the
// queries with id offset g * shards_per_gpu * N_shard (result_merger.cpp:115-116) and returns
// that slice.  The reference copies everything to the host and merges there with a heap per
// query (ggnn.cu:308-329, result_merger.cpp:51-149).
void ggnn_handle::exchange_rccl(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
                   float* dists_out, bool blocking)
{
  const Rccl& rccl = Rccl::get();
  const size_t G = devs.size();
  const size_t part = nq * row;
  for (DeviceCtx& ctx : devs) {
    ctx.activate();
    grow_lane(ctx, lane, ctx.xb[lane].g_pack, G * 2 * part * 4);
    grow_lane(ctx, lane, ctx.xb[lane].m_pack, 2 * static_cast<size_t>(nq) * k_query * 4);
  }
  // fault injection (hook RCCL_FAIL_AFTER = n): the n-th exchange of the process reports an RCCL
  // failure before anything is enqueued -- the path a real failure takes from here on
  {
    static std::atomic<int64_t> exchanges{0};  // counted while the hook is set
    const int64_t fail_at = hook(kHookRcclFailAfter);
    if (fail_at <= 0)
      exchanges.store(0);
    else if (++exchanges == fail_at)
      GGNN_RCCL_CHECK(ncclInternalError);
  }
  GGNN_RCCL_CHECK(rccl.GroupStart());
  ncclResult_t first_error = ncclSuccess;
  for (size_t g = 0; g < G && first_error == ncclSuccess; ++g) {
    DeviceCtx& ctx = devs[g];
    first_error = rccl.AllGather(ctx.xb[lane].r_pack.p, ctx.xb[lane].g_pack.p, 2 * part,
                                 ncclInt32, comms[g], ctx.lane_stream(lane));
  }
  // the group is closed whatever happened inside it
  const ncclResult_t end = rccl.GroupEnd();
  GGNN_RCCL_CHECK(first_error);
  GGNN_RCCL_CHECK(end);
  merge_slices(lane, nq, k_query, row, ids_out, dists_out, blocking);
  last_exchange = "rccl";
}

// What follows the all-gather: GPU g merges its 1/G slice of the queries out of the gathered rows
// (g_pack) and returns it; a blocking call then assembles the slices in the caller's arrays.
void ggnn_handle::merge_slices(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
                  float* dists_out, bool blocking)
{
  const size_t G = devs.size();
  const size_t part = nq * row;
  for (size_t g = 0; g < G; ++g) {
    DeviceCtx& ctx = devs[g];
    uint32_t first, count;
    slice_of(nq, G, g, &first, &count);
    if (!count)
      continue;
    ctx.activate();
    DeviceCtx::ExchangeBufs& x = ctx.xb[lane];
    int32_t* m_ids = x.m_pack.as<int32_t>();
    launch_merge_results_range(nq, k_query, static_cast<uint32_t>(G), static_cast<uint32_t>(row),
                               shards_per_gpu * cfg.N, x.g_pack.as<int32_t>(),
                               reinterpret_cast<const float*>(x.g_pack.as<int32_t>() + part),
                               m_ids, reinterpret_cast<float*>(m_ids + static_cast<size_t>(nq) * k_query),
                               nullptr, nullptr, first, count, ctx.lane_stream(lane), 2 * part);
    return_slice(ctx, lane, nq, k_query, first, count, ids_out, dists_out, blocking);
  }
  if (blocking)
    finish_slices(lane, nq, k_query, G, ids_out, dists_out);
}

// Hook EXCHANGE = 3 ("gather"): the structure of the RCCL path WITHOUT RCCL -- every GPU gathers
// the packed rows of all GPUs with peer copies (an all-gather spelled out), merges its 1/G slice
// and returns it.  Exists so that the slice arithmetic, the G merge launches, the per-GPU staging
// buffers and finish_slices of a handle with 8 contexts can run on a box whose contexts share one
// device (RCCL refuses two ranks on a device); blocking calls only.
void ggnn_handle::exchange_gather_copies(int lane, uint32_t nq, uint32_t k_query, size_t row,
                            int32_t* ids_out, float* dists_out)
{
  const size_t G = devs.size();
  const size_t part = nq * row;
  for (DeviceCtx& ctx : devs) {
    ctx.activate();
    grow_lane(ctx, lane, ctx.xb[lane].g_pack, G * 2 * part * 4);
    grow_lane(ctx, lane, ctx.xb[lane].m_pack, 2 * static_cast<size_t>(nq) * k_query * 4);
    if (!ctx.xb[lane].done)
      GGNN_HIP_CHECK(hipEventCreateWithFlags(&ctx.xb[lane].done, hipEventDisableTiming));
    GGNN_HIP_CHECK(hipEventRecord(ctx.xb[lane].done, ctx.lane_stream(lane)));
  }
  for (size_t g = 0; g < G; ++g) {
    DeviceCtx& ctx = devs[g];
    ctx.activate();
    hipStream_t st = ctx.lane_stream(lane);
    for (size_t s = 0; s < G; ++s) {
      if (s != g)
        GGNN_HIP_CHECK(hipStreamWaitEvent(st, devs[s].xb[lane].done, 0));
      GGNN_HIP_CHECK(hipMemcpyAsync(ctx.xb[lane].g_pack.as<int32_t>() + s * 2 * part,
                                    devs[s].xb[lane].r_pack.p, 2 * part * 4, hipMemcpyDefault, st));
    }
  }
  merge_slices(lane, nq, k_query, row, ids_out, dists_out, /*blocking=*/true);
  last_exchange = "gather";
}

// Several contexts without RCCL (contexts sharing one device, or no librccl): packed rows to the
// first GPU with peer copies, k-way merge there.
void ggnn_handle::exchange_peer_copies(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
                          float* dists_out, bool blocking)
{
  DeviceCtx& d0 = devs[0];
  const size_t G = devs.size();
  const size_t part = nq * row;
  // the first GPU's lane waits for the local searches of the others
  for (size_t g = 1; g < G; ++g) {
    DeviceCtx& ctx = devs[g];
    ctx.activate();
    if (!ctx.xb[lane].done)
      GGNN_HIP_CHECK(hipEventCreateWithFlags(&ctx.xb[lane].done, hipEventDisableTiming));
    GGNN_HIP_CHECK(hipEventRecord(ctx.xb[lane].done, ctx.lane_stream(lane)));
  }
  d0.activate();
  hipStream_t st = d0.lane_stream(lane);
  DeviceCtx::ExchangeBufs& x = d0.xb[lane];
  grow_lane(d0, lane, x.g_pack, G * 2 * part * 4);
  grow_lane(d0, lane, x.m_pack, 2 * static_cast<size_t>(nq) * k_query * 4);
  for (size_t g = 0; g < G; ++g) {
    if (g)
      GGNN_HIP_CHECK(hipStreamWaitEvent(st, devs[g].xb[lane].done, 0));
    GGNN_HIP_CHECK(hipMemcpyAsync(x.g_pack.as<int32_t>() + g * 2 * part,
                                  devs[g].xb[lane].r_pack.p, 2 * part * 4, hipMemcpyDefault, st));
  }
  if (!blocking) {
    // the other GPUs' next batch on this lane overwrites the rows just copied: they wait for
    // this point (query_async), the copies run on THIS GPU's stream
    if (!x.consumed)
      GGNN_HIP_CHECK(hipEventCreateWithFlags(&x.consumed, hipEventDisableTiming));
    GGNN_HIP_CHECK(hipEventRecord(x.consumed, st));
  }
  int32_t* m_ids = x.m_pack.as<int32_t>();
  launch_merge_results_range(nq, k_query, static_cast<uint32_t>(G), static_cast<uint32_t>(row),
                             shards_per_gpu * cfg.N, x.g_pack.as<int32_t>(),
                             reinterpret_cast<const float*>(x.g_pack.as<int32_t>() + part), m_ids,
                             reinterpret_cast<float*>(m_ids + static_cast<size_t>(nq) * k_query),
                             nullptr, nullptr, 0, nq, st, 2 * part);
  return_slice(d0, lane, nq, k_query, 0, nq, ids_out, dists_out, blocking);
  if (blocking)
    finish_slices(lane, nq, k_query, 1, ids_out, dists_out);
  last_exchange = "copy";
}
