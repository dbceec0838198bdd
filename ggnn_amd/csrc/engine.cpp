// Engine behind the C-ABI (include/ggnn_c.h): shard residency, build schedule, query / bf_query
// drivers.  Replaces, for the hot path only, GGNNImpl (src/ggnn/base/ggnn.cu:124-413),
// GPUInstance::build/query (src/ggnn/base/gpu_instance.cu:499-584, 626-790) and
// GraphConstructionImpl::build/refine (src/ggnn/construction/graph_construction.cu:128-147).
//
// MI355X-first choices: every shard of the base and its graph stay resident in HBM (288 GB), so
// the reference's GPU<->CPU<->disk swapping is not reproduced; one engine drives one GPU
// (multi-GPU = one process per GPU, shards exchanged with an RCCL all-gather, see
// ggnn_amd/distributed.py and DESIGN.md).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"
#include "hooks.hpp"

namespace ggnn_amd {

namespace {

struct DeviceBuffer {
  void* p{nullptr};
  size_t bytes{0};
  DeviceBuffer() = default;
  explicit DeviceBuffer(size_t n) { alloc(n); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : p(o.p), bytes(o.bytes)
  {
    o.p = nullptr;
    o.bytes = 0;
  }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept
  {
    if (this != &o) {
      release();
      p = o.p;
      bytes = o.bytes;
      o.p = nullptr;
      o.bytes = 0;
    }
    return *this;
  }
  ~DeviceBuffer() { release(); }
  void alloc(size_t n)
  {
    release();
    if (n) {
      GGNN_HIP_CHECK(hipMalloc(&p, n));
      bytes = n;
    }
  }
  void release()
  {
    if (p)
      (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <typename T>
  T* as() const
  {
    return static_cast<T*>(p);
  }
};

// page-locked host memory (slice copies of the multi-GPU exchange: a D2H copy into pageable
// memory is staged and serialised by the runtime)
struct PinnedBuffer {
  void* p{nullptr};
  size_t bytes{0};
  PinnedBuffer() = default;
  PinnedBuffer(const PinnedBuffer&) = delete;
  PinnedBuffer& operator=(const PinnedBuffer&) = delete;
  PinnedBuffer(PinnedBuffer&& o) noexcept : p(o.p), bytes(o.bytes)
  {
    o.p = nullptr;
    o.bytes = 0;
  }
  PinnedBuffer& operator=(PinnedBuffer&& o) noexcept
  {
    if (this != &o) {
      release();
      p = o.p;
      bytes = o.bytes;
      o.p = nullptr;
      o.bytes = 0;
    }
    return *this;
  }
  ~PinnedBuffer() { release(); }
  void grow(size_t n)
  {
    if (bytes >= n)
      return;
    release();
    GGNN_HIP_CHECK(hipHostMalloc(&p, n, hipHostMallocDefault));
    bytes = n;
  }
  void release()
  {
    if (p)
      (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
  }
};

// HIP-event stopwatch on one stream, on two events owned by the caller (DeviceCtx creates them
// once: creating and destroying events costs ~10 us per query() otherwise)
struct EventTimer {
  hipEvent_t a, b;
  hipStream_t s;
  EventTimer(hipStream_t stream, hipEvent_t ev_a, hipEvent_t ev_b) : a(ev_a), b(ev_b), s(stream)
  {
    GGNN_HIP_CHECK(hipEventRecord(a, s));
  }
  float stop()
  {
    float ms = 0.f;
    GGNN_HIP_CHECK(hipEventRecord(b, s));
    GGNN_HIP_CHECK(hipEventSynchronize(b));
    GGNN_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
  }
};

size_t dtype_size(ggnn_dtype t)
{
  return t == GGNN_F32 ? 4 : 1;
}

// graph pool of one shard, reference layout (src/ggnn/base/graph.cpp:48-91):
// [N_all x K int32 graph][ST_all int32 translation][ST_all int32 selection][2 float nn1_stats]
struct Shard {
  uint32_t global_id{0};
  DeviceBuffer pool;
  int32_t* graph{nullptr};
  int32_t* translation{nullptr};
  int32_t* selection{nullptr};
  float* nn1_stats{nullptr};
  bool ready{false};
  // 8-bit pre-screen copy of this shard's rows (prescreen.hip) for ps_measure, made when a
  // float32 build or query first needs it: 0 = not attempted, 1 = usable, -1 = data not codable
  // (non-finite values)
  DeviceBuffer ps_codes, ps_params;
  int ps_state{0};
  ggnn_measure ps_measure{GGNN_EUCLIDEAN};

  static size_t pool_bytes(const ggnn_graph_config& c)
  {
    return (static_cast<size_t>(c.N_all) * c.KBuild + 2 * static_cast<size_t>(c.ST_all)) * 4 +
           2 * sizeof(float);
  }
  // out-of-core shards: the pool is a slot buffer of the GPU's SwapState
  void view(const ggnn_graph_config& c, void* slot_pool)
  {
    graph = static_cast<int32_t*>(slot_pool);
    translation = graph + static_cast<size_t>(c.N_all) * c.KBuild;
    selection = translation + c.ST_all;
    nn1_stats = reinterpret_cast<float*>(selection + c.ST_all);
  }
  void allocate(const ggnn_graph_config& c)
  {
    pool.alloc(align8(pool_bytes(c)));
    graph = pool.as<int32_t>();
    translation = graph + static_cast<size_t>(c.N_all) * c.KBuild;
    selection = translation + c.ST_all;
    nn1_stats = reinterpret_cast<float*>(selection + c.ST_all);
  }
};

}  // namespace
}  // namespace ggnn_amd

using namespace ggnn_amd;

// destroying a DeviceCtx switches devices; leave the caller's current device as it was
struct DeviceRestoreGuard {
  int prev{-1};
  DeviceRestoreGuard() { (void)hipGetDevice(&prev); }
  ~DeviceRestoreGuard()
  {
    int now = -1;
    if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev)
      (void)hipSetDevice(prev);
  }
};

// RCCL entry points, resolved at run time the first time a handle that drives several GPUs
// exchanges results: single-GPU use never loads the library (torch ships its own copy of
// librccl.so.1; the loader hands back that copy when it is already in the process).
struct Rccl {
  decltype(&ncclCommInitAll) CommInitAll{nullptr};
  decltype(&ncclCommDestroy) CommDestroy{nullptr};
  decltype(&ncclAllGather) AllGather{nullptr};
  decltype(&ncclGroupStart) GroupStart{nullptr};
  decltype(&ncclGroupEnd) GroupEnd{nullptr};
  decltype(&ncclGetErrorString) GetErrorString{nullptr};
  bool ok{false};

  static const Rccl& get()
  {
    static const Rccl r = [] {
      Rccl x;
      void* lib = nullptr;
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib)
          break;
      }
      if (!lib)
        return x;
      auto sym = [&](const char* n) { return dlsym(lib, n); };
      x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
      x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
      x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
      x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
      x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
      x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
      x.ok = x.CommInitAll && x.CommDestroy && x.AllGather && x.GroupStart && x.GroupEnd &&
             x.GetErrorString;
      return x;
    }();
    return r;
  }
};
// a failed RCCL call: the only kind of error the exchange answers with its peer-copy fallback
struct RcclError : Error {
  using Error::Error;
};
#define GGNN_RCCL_CHECK(expr)                                                              \
  do {                                                                                     \
    ncclResult_t _r = (expr);                                                              \
    if (_r != ncclSuccess)                                                                 \
      throw RcclError(GGNN_DEVICE_ERROR, std::string(#expr) + ": " +           \
                                                         Rccl::get().GetErrorString(_r));  \
  } while (0)

// Out-of-core shards of one GPU (reference: GPUInstance's d_buffers / h_buffers /
// part files, gpu_instance.cu:135-227, 371-497).  Only when the shards of a GPU do not fit next
// to each other -- at every BASELINE configuration they do, and then none of this exists:
//   * `slots` GPU buffers (graph pool + base shard); local shard s lives in slot s % slots;
//   * `host.size()` page-locked host buffers for the graph pools (shard s in buffer s % host
//     buffers; ggnn_set_cpu_memory_limit bounds them), the rest as part_<shard>.ggnn files in
//     the working directory -- the reference's three tiers;
//   * one copy stream: the shard a query needs next is uploaded while the current one is
//     searched (the reference uses one io thread per buffer for the same purpose).
struct SwapState {
  int device{0};
  uint32_t slots{0};
  std::vector<DeviceBuffer> pool, base;     // [slots]
  std::vector<int> pool_shard, base_shard;  // local shard held, -1: none
  std::vector<PinnedBuffer> host;           // [host buffers]
  std::vector<int> host_shard;
  std::vector<uint8_t> on_disk;             // [shards per GPU]: part file is current
  hipStream_t io{nullptr};
  std::vector<hipEvent_t> uploaded, consumed;  // [slots]: copies done (io) / last kernel done (ctx)
  bool base_borrowed{false};                // the base slice is device memory of this GPU already
  SwapState() = default;
  SwapState(const SwapState&) = delete;
  SwapState& operator=(const SwapState&) = delete;
  ~SwapState()
  {
    if (io || !uploaded.empty()) {
      (void)hipSetDevice(device);
      for (hipEvent_t e : uploaded)
        if (e)
          (void)hipEventDestroy(e);
      for (hipEvent_t e : consumed)
        if (e)
          (void)hipEventDestroy(e);
      if (io)
        (void)hipStreamDestroy(io);
    }
  }
};

// everything one GPU owns (GPUInstance of the reference, gpu_instance.cuh:60-221, reduced to
// resident shards)
struct DeviceCtx {
  int device{0};
  hipStream_t stream{nullptr};
  hipEvent_t ev_a{nullptr}, ev_b{nullptr};  // timing events, created with the stream
  DeviceBuffer base_copy;       // this GPU's slice of the base unless it is borrowed
  const void* d_base{nullptr};  // first row of the slice
  uint32_t first_shard{0};      // global id of shards[0]
  std::vector<Shard> shards;
  float build_ms{0.f}, query_ms{0.f};
  uint64_t n_dist{0}, n_pop{0}, n_float_rows{0}, n_code_rows{0};
  // several resident shards: their query kernels run on a few extra streams so that the thin tail
  // of one launch overlaps with the head of the next (the reference also uses per-shard streams,
  // gpu_instance.cu:626-743)
  static constexpr int kShardStreams = 4;
  hipStream_t shard_stream[kShardStreams] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t shard_done[kShardStreams] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_ready{nullptr};
  DeviceBuffer bf_rescanned;    // one uint32: queries of the last bf_query answered by the scan
  std::unique_ptr<SwapState> swap;  // out-of-core shards (null: every shard resident)
  // Result staging of query() / query_async(): grown on demand, kept between calls.  One set per
  // lane: lanes [0, kShardStreams) belong to the asynchronous slots (their streams), lane
  // kBlockingLane to the blocking query(): its staging, exchange and result copies run on `stream`,
  // but with several resident shards its per-shard search launches are spread over the SAME
  // shard_stream[] the asynchronous slots use (ordered behind whatever a slot has in flight there;
  // the blocking call waits for all of them before it returns).  Ids and distance bit patterns share ONE
  // buffer so that the exchange is one collective: r_pack = [ids: Nq x row][dists: Nq x row].
  static constexpr int kBlockingLane = kShardStreams;
  struct ExchangeBufs {
    DeviceBuffer q_stage;  // the query set on this GPU when it has to be copied (async lanes)
    DeviceBuffer r_pack;   // this GPU's sorted rows, ids then distances
    DeviceBuffer g_pack;   // r_pack of all GPUs after the exchange, [G][2 * Nq * row]
    DeviceBuffer m_pack;   // merged results, [ids: Nq x K][dists: Nq x K] (this GPU's slice filled)
    PinnedBuffer h_pack;   // the merged slice on the host, [ids: count x K][dists: count x K]
    hipEvent_t done{nullptr};  // local search of this lane finished (copy exchange)
    // (first GPU, copy exchange of asynchronous batches) the rows of every GPU have been copied
    // out: their owners may overwrite them with the next batch of this lane
    hipEvent_t consumed{nullptr};
  };
  ExchangeBufs xb[kShardStreams + 1];
  hipStream_t lane_stream(int lane) const
  {
    return lane == kBlockingLane ? stream : shard_stream[lane];
  }
  static void grow(DeviceBuffer& b, size_t bytes)
  {
    if (b.bytes < bytes)
      b.alloc(bytes);
  }

  DeviceCtx() = default;
  DeviceCtx(const DeviceCtx&) = delete;
  DeviceCtx& operator=(const DeviceCtx&) = delete;
  DeviceCtx(DeviceCtx&& o) noexcept { *this = std::move(o); }
  DeviceCtx& operator=(DeviceCtx&& o) noexcept
  {
    device = o.device;
    stream = o.stream;
    ev_a = o.ev_a;
    ev_b = o.ev_b;
    ev_ready = o.ev_ready;
    for (int i = 0; i < kShardStreams; ++i) {
      shard_stream[i] = o.shard_stream[i];
      shard_done[i] = o.shard_done[i];
      o.shard_stream[i] = nullptr;
      o.shard_done[i] = nullptr;
    }
    o.stream = nullptr;
    o.ev_a = o.ev_b = o.ev_ready = nullptr;
    base_copy = std::move(o.base_copy);
    bf_rescanned = std::move(o.bf_rescanned);
    for (int i = 0; i <= kShardStreams; ++i) {
      xb[i].q_stage = std::move(o.xb[i].q_stage);
      xb[i].r_pack = std::move(o.xb[i].r_pack);
      xb[i].g_pack = std::move(o.xb[i].g_pack);
      xb[i].m_pack = std::move(o.xb[i].m_pack);
      xb[i].h_pack = std::move(o.xb[i].h_pack);
      xb[i].done = o.xb[i].done;
      o.xb[i].done = nullptr;
      xb[i].consumed = o.xb[i].consumed;
      o.xb[i].consumed = nullptr;
    }
    d_base = o.d_base;
    first_shard = o.first_shard;
    shards = std::move(o.shards);
    swap = std::move(o.swap);
    return *this;
  }
  ~DeviceCtx()
  {
    if (stream) {
      (void)hipSetDevice(device);
      (void)hipEventDestroy(ev_a);
      (void)hipEventDestroy(ev_b);
      if (ev_ready)
        (void)hipEventDestroy(ev_ready);
      for (int i = 0; i < kShardStreams; ++i)
        if (shard_stream[i]) {
          (void)hipEventDestroy(shard_done[i]);
          (void)hipStreamDestroy(shard_stream[i]);
        }
      for (int i = 0; i <= kShardStreams; ++i) {
        if (xb[i].done)
          (void)hipEventDestroy(xb[i].done);
        if (xb[i].consumed)
          (void)hipEventDestroy(xb[i].consumed);
      }
      (void)hipStreamDestroy(stream);
    }
  }
  void activate()
  {
    GGNN_HIP_CHECK(hipSetDevice(device));
    if (!stream) {
      GGNN_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      GGNN_HIP_CHECK(hipEventCreate(&ev_a));
      GGNN_HIP_CHECK(hipEventCreate(&ev_b));
    }
  }
  // (after activate(); created with the first multi-shard query)
  void ensure_shard_streams()
  {
    if (shard_stream[0])
      return;
    GGNN_HIP_CHECK(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
    for (int i = 0; i < kShardStreams; ++i) {
      GGNN_HIP_CHECK(hipStreamCreateWithFlags(&shard_stream[i], hipStreamNonBlocking));
      GGNN_HIP_CHECK(hipEventCreateWithFlags(&shard_done[i], hipEventDisableTiming));
    }
  }
};

struct ggnn_handle {
  // configuration (GGNNConfig, ggnn.cu:52-59)
  std::filesystem::path graph_dir{};
  size_t cpu_memory_limit{static_cast<size_t>(-1)};
  size_t reserved_gpu_memory{0};
  std::vector<int> gpu_ids{};
  uint32_t N_shard{0};
  bool return_results_on_gpu{false};
  bool collect_counters{false};
  bool prescreen{hook(kHookPrescreen) != 0};  // ggnn_set_prescreen() per handle

  // deterministic-build hooks (ggnn_set_build_hooks): injected selection random numbers
  // ([layers-1][N_shard], shard-local) and sym launched one point at a time in ascending order
  std::vector<float> hook_rng;
  bool hook_serial_sym{false};

  // base as handed over by the caller
  const void* base_src{nullptr};
  ggnn_location base_loc{GGNN_CPU};
  int base_gpu{0};
  std::vector<uint8_t> base_host_copy;
  bool base_host_copy_registered{false};  // page-locked (hipHostRegister) while shards swap
  // out-of-core shards re-read their rows at every swap on a copy stream: from pageable memory
  // that copy is staged and synchronous, i.e. it does not overlap the search (round-4 advisor
  // finding) -- the engine's own host copy is page-locked for as long as it is swapped from
  void pin_host_copy()
  {
    if (base_host_copy_registered || base_host_copy.empty())
      return;
    if (hipHostRegister(base_host_copy.data(), base_host_copy.size(), hipHostRegisterDefault) ==
        hipSuccess)
      base_host_copy_registered = true;
    else
      (void)hipGetLastError();  // (still correct from pageable memory, only not overlapped)
  }
  void drop_host_copy()
  {
    if (base_host_copy_registered)
      (void)hipHostUnregister(base_host_copy.data());
    base_host_copy_registered = false;
    base_host_copy.clear();
    base_host_copy.shrink_to_fit();
  }
  DeviceBuffer base_dev_copy;
  uint64_t base_N{0};
  uint32_t base_D{0};  // dimension as given by the caller
  uint32_t pad_D{0};   // row length the kernels see: rows are zero-padded to a multiple of 16 bytes
                       // (zeros change neither the L2 nor the cosine distance)
  ggnn_dtype base_dtype{GGNN_F32};
  bool base_set{false};

  // graph: one DeviceCtx per GPU, shards_per_gpu resident shards each
  bool prepared{false};
  ggnn_graph_config cfg{};
  uint32_t shards_per_gpu{0};
  std::vector<DeviceCtx> devs;

  // tracing
  float build_ms{0.f}, query_ms{0.f}, bf_ms{0.f};
  uint64_t last_n_dist{0}, last_n_pop{0}, last_float_rows{0}, last_code_rows{0};
  ggnn_build_work build_work{};  // collect_counters during build(): see ggnn_last_build_work
  std::mutex build_work_mutex;   // one host thread per GPU accounts into it
  uint32_t last_bf_rescanned{0};

  std::string last_error;

  // one RCCL communicator per GPU of a multi-GPU handle (created with the first exchange)
  std::vector<ncclComm_t> comms;
  int rccl_state{0};  // 0 = not tried, 1 = communicators ready, -1 = unavailable (peer copies)
  uint32_t rccl_fallbacks{0};  // exchanges that failed inside RCCL and were served by peer copies
  const char* last_exchange{"none"};
  uint32_t last_query_parts{1};  // half-batches the last blocking query was searched in

  ~ggnn_handle()
  {
    destroy_comms();
    if (base_host_copy_registered)
      (void)hipHostUnregister(base_host_copy.data());
  }
  void destroy_comms()
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
  // 1-rank world).
  bool ensure_comms()
  {
    if (rccl_state != 0)
      return rccl_state > 0;
    rccl_state = -1;
    if (hook(kHookExchange) == 2)
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

  size_t row_bytes() const { return static_cast<size_t>(pad_D) * dtype_size(base_dtype); }
  uint32_t num_shards() const { return shards_per_gpu * static_cast<uint32_t>(devs.size()); }
  // every shard of every GPU is built or loaded (a partly loaded handle has no graph)
  bool has_graph() const
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
  void rollback_graph()
  {
    DeviceRestoreGuard keep;
    destroy_comms();
    devs.clear();
    prepared = false;
    shards_per_gpu = 0;
    cfg = ggnn_graph_config{};
  }

  // runs f(ctx) for every GPU -- inline for one GPU, one host thread per GPU otherwise (the
  // reference does the same, ggnn.cu:218-230,308-326); the first failure is rethrown
  template <typename F>
  void for_each_device(F&& f)
  {
    if (devs.size() == 1) {
      devs[0].activate();
      f(devs[0]);
      return;
    }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errors(devs.size());
    for (size_t i = 0; i < devs.size(); ++i)
      pool.emplace_back([&, i] {
        try {
          devs[i].activate();
          f(devs[i]);
        }
        catch (...) {
          errors[i] = std::current_exception();
        }
      });
    for (auto& t : pool)
      t.join();
    for (auto& e : errors)
      if (e)
        std::rethrow_exception(e);
  }

  std::vector<int> resolve_gpus() const
  {
    if (!gpu_ids.empty())
      return gpu_ids;
    int d = 0;
    GGNN_HIP_CHECK(hipGetDevice(&d));  // ggnn.cu:172-176
    return {d};
  }

  // base.referenceOnGPU (dataset.cu:236-300): rows [row0, row0+rows) resident on ctx's GPU
  void stage_base_slice(DeviceCtx& ctx, uint64_t row0, uint64_t rows)
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

  const void* shard_base(const DeviceCtx& ctx, uint32_t local_shard) const
  {
    if (ctx.swap && !ctx.swap->base_borrowed)  // (valid after acquire_shard of this shard)
      return ctx.swap->base[local_shard % ctx.swap->slots].p;
    return static_cast<const uint8_t*>(ctx.d_base) +
           static_cast<size_t>(local_shard) * cfg.N * row_bytes();
  }

  // ---- out-of-core shards (SwapState) -----------------------------------------------------------
  bool swapping() const { return !devs.empty() && devs[0].swap != nullptr; }

  // GPU slots per device: 0 = every shard resident (the normal case).  Hook RESIDENT_SHARDS forces
  // a number (tests); otherwise the shards are counted against the free device memory minus
  // ggnn_set_reserved_gpu_memory, as GPUInstance::allocateGraph does (gpu_instance.cu:157-186).
  uint32_t plan_gpu_slots(const DeviceCtx& ctx, uint32_t spg, bool base_on_this_gpu) const
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

  void setup_swap(DeviceCtx& ctx, uint32_t slots, bool base_on_this_gpu)
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
  void upload_base_shard(DeviceCtx& ctx, uint32_t si, hipStream_t st)
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
  void* host_pool_of(DeviceCtx& ctx, uint32_t si)
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
  void acquire_shard(DeviceCtx& ctx, uint32_t si, hipStream_t st, bool with_graph = true)
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
  void shard_consumed(DeviceCtx& ctx, uint32_t si, hipStream_t st)
  {
    SwapState& sw = *ctx.swap;
    GGNN_HIP_CHECK(hipEventRecord(sw.consumed[si % sw.slots], st));
  }
  // after build: the graph pool of shard si goes to its host buffer and, when the host buffers do
  // not hold every shard, to its part file at once (swapOutPart, gpu_instance.cu:372-425)
  void retire_built_shard(DeviceCtx& ctx, uint32_t si)
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
  void write_part(uint32_t global_shard, const void* host)
  {
    const auto file = part_file(global_shard);
    std::ofstream f(file, std::ios::binary | std::ios::trunc);
    GGNN_REQUIRE(f.good(), GGNN_IO_ERROR, "cannot open " + file.string());
    f.write(static_cast<const char*>(host), static_cast<std::streamsize>(Shard::pool_bytes(cfg)));
    GGNN_REQUIRE(f.good(), GGNN_IO_ERROR, "short write to " + file.string());
  }
  void read_part(uint32_t global_shard, void* host)
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

  // GGNNImpl::prepare, ggnn.cu:154-203
  void prepare(uint32_t KBuild)
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

  // GraphConstructionImpl::build / refine, graph_construction.cu:128-147, for all shards of ctx
  void build_device(DeviceCtx& ctx, float tau_build, uint32_t refinement_iterations,
                    ggnn_measure measure)
  {
    const uint32_t N = cfg.N, K = cfg.KBuild, KF = cfg.KF;
    hipStream_t stream = ctx.stream;
    // scratch (GraphBuffer, graph_buffer.cu:38-81); not overlapped -- HBM is plentiful
    DeviceBuffer nn1_dist(static_cast<size_t>(N) * 4), graph_buffer(static_cast<size_t>(N) * K * 4),
        rng(static_cast<size_t>(N) * 4), sym_buffer(static_cast<size_t>(N) * KF * 4),
        sym_atomic(static_cast<size_t>(N) * 4), stats_scratch(3 * kStatsBlocks * 4);
    ctx.build_ms = 0.f;
    uint64_t rng_calls = static_cast<uint64_t>(ctx.first_shard) << 16;
    // diagnostic mode (collect_counters): per-point work counters and an own HIP-event time of
    // every merge / sym launch, summed into build_work (ggnn_last_build_work)
    DeviceBuffer work;
    std::vector<uint32_t> h_work;
    hipEvent_t wev_a = nullptr, wev_b = nullptr;
    if (collect_counters) {
      work.alloc(static_cast<size_t>(N) * 16);
      h_work.resize(static_cast<size_t>(N) * 4);
      GGNN_HIP_CHECK(hipEventCreate(&wev_a));
      GGNN_HIP_CHECK(hipEventCreate(&wev_b));
    }
    struct EventGuard {
      hipEvent_t &a, &b;
      ~EventGuard()
      {
        if (a)
          (void)hipEventDestroy(a);
        if (b)
          (void)hipEventDestroy(b);
      }
    } event_guard{wev_a, wev_b};
    auto account = [&](ggnn_kernel_work& kw, uint32_t points, float ms) {
      GGNN_HIP_CHECK(hipMemcpy(h_work.data(), work.p, static_cast<size_t>(points) * 16,
                               hipMemcpyDeviceToHost));
      std::lock_guard<std::mutex> lock(build_work_mutex);
      kw.launches += 1;
      kw.points += points;
      kw.ms += ms;
      for (uint32_t i = 0; i < points; ++i) {
        kw.n_dist += h_work[4 * i];
        kw.float_rows += h_work[4 * i + 1];
        kw.code_rows += h_work[4 * i + 2];
        kw.pops += h_work[4 * i + 3];
      }
    };

    for (uint32_t si = 0; si < ctx.shards.size(); ++si) {
      if (ctx.swap)  // out-of-core shards: this shard's rows into its slot, the pool is built in place
        acquire_shard(ctx, si, stream, /*with_graph=*/false);
      // the pre-screen copy serves the merge kernel too (made outside the timed region: it
      // depends on the base only and is kept for the queries)
      const bool use_ps = ensure_prescreen(ctx, si, measure);
      Shard& sh = ctx.shards[si];
      const void* base = shard_base(ctx, si);
      EventTimer timer(stream, ctx.ev_a, ctx.ev_b);

      auto layer_graph = [&](uint32_t l) {
        return sh.graph + static_cast<size_t>(cfg.Ns_offsets[l]) * K;
      };
      auto layer_tr = [&](uint32_t l) -> int32_t* {
        return l ? sh.translation + cfg.STs_offsets[l] : nullptr;
      };
      auto do_merge = [&](uint32_t top, uint32_t btm) {
        if (top == btm) {
          TopLaunch t{base,        base_dtype,          pad_D,
                      measure,     K,                   layer_tr(btm),
                      cfg.Ns[btm], btm ? cfg.S : cfg.S0, btm ? 0u : cfg.S0_off,
                      btm,         layer_graph(btm),    nn1_dist.as<float>()};
          launch_top(t, stream);
        }
        else {
          MergeLaunch m{base,
                        base_dtype,
                        measure,
                        cfg,
                        sh.graph,
                        sh.translation,
                        sh.selection,
                        sh.nn1_stats,
                        tau_build,
                        top,
                        btm,
                        graph_buffer.as<int32_t>(),
                        nn1_dist.as<float>(),
                        nullptr};
          if (use_ps) {
            m.ps_codes = sh.ps_codes.as<uint8_t>();
            m.ps_params = sh.ps_params.as<float>();
            m.ps_Dc = prescreen_code_dim(pad_D);
          }
          if (collect_counters) {
            m.n_work = work.as<uint32_t>();
            EventTimer t(stream, wev_a, wev_b);
            launch_merge(m, stream);
            account(build_work.merge, cfg.Ns[btm], t.stop());
          }
          else
            launch_merge(m, stream);
          GGNN_HIP_CHECK(hipMemcpyAsync(layer_graph(btm), graph_buffer.p,
                                        static_cast<size_t>(cfg.Ns[btm]) * K * 4,
                                        hipMemcpyDeviceToDevice, stream));
        }
        if (!btm)
          launch_nn1_stats(nn1_dist.as<float>(), N, stats_scratch.as<float>(), sh.nn1_stats,
                           stream);
      };
      auto do_select = [&](uint32_t layer) {
        if (!hook_rng.empty()) {
          GGNN_REQUIRE(hook_rng.size() >= static_cast<size_t>(layer + 1) * N, GGNN_INVALID_ARGUMENT,
                       "build hooks: rng needs (layers - 1) * N_shard numbers");
          GGNN_HIP_CHECK(hipMemcpyAsync(rng.p, hook_rng.data() + static_cast<size_t>(layer) * N,
                                        static_cast<size_t>(cfg.Ns[layer]) * 4,
                                        hipMemcpyHostToDevice, stream));
        }
        else
          launch_uniform(rng.as<float>(), cfg.Ns[layer], 1234ull, rng_calls++, stream);
        launch_select(cfg, layer, nn1_dist.as<float>(), rng.as<float>(), sh.translation,
                      sh.selection, stream);
      };
      auto do_sym = [&](uint32_t layer) {
        GGNN_HIP_CHECK(hipMemsetAsync(sym_buffer.p, 0xff,
                                      static_cast<size_t>(cfg.Ns[layer]) * KF * 4, stream));
        GGNN_HIP_CHECK(
            hipMemsetAsync(sym_atomic.p, 0, static_cast<size_t>(cfg.Ns[layer]) * 4, stream));
        SymLaunch s{base,
                    base_dtype,
                    measure,
                    pad_D,
                    K,
                    layer_graph(layer),
                    layer_tr(layer),
                    cfg.Ns[layer],
                    sh.nn1_stats,
                    tau_build,
                    sym_buffer.as<int32_t>(),
                    sym_atomic.as<uint32_t>(),
                    0,
                    cfg.Ns[layer]};
        // The short sym searches only gain from the pre-screen on wide rows (measured, 1M points:
        // D = 960 cosine 805 -> 454 ms per build, D = 128 74.6 -> 78.2 ms): used from 1 KB rows on.
        // Hook SYM_PRESCREEN = 0 | 1 forces it off / on (tuning hook).
        const int64_t sym_ps_hook = hook(kHookSymPrescreen);
        const bool sym_ps = sym_ps_hook >= 0 ? sym_ps_hook == 1 : pad_D >= 256;
        if (use_ps && sym_ps) {
          s.ps_codes = sh.ps_codes.as<uint8_t>();
          s.ps_params = sh.ps_params.as<float>();
          s.ps_Dc = prescreen_code_dim(pad_D);
        }
        if (hook_serial_sym) {
          // the reference's sym races through atomics and cross-block reads of sym_buffer
          // (sym_query_layer.cu:102-104 vs :133-136); one point per launch in ascending order is
          // the one schedule that is comparable with a CPU restatement
          for (uint32_t n = 0; n < cfg.Ns[layer]; ++n) {
            s.first_n = n;
            s.count = 1;
            launch_sym(s, stream);
          }
        }
        else if (collect_counters) {
          s.n_work = work.as<uint32_t>();
          EventTimer t(stream, wev_a, wev_b);
          launch_sym(s, stream);
          account(build_work.sym, cfg.Ns[layer], t.stop());
        }
        else
          launch_sym(s, stream);
        launch_sym_buffer_merge(K, cfg.Ns[layer], sym_buffer.as<int32_t>(),
                                sym_atomic.as<uint32_t>(), layer_graph(layer), stream);
      };

      // no selection/translation on layer 0; start from a defined state
      GGNN_HIP_CHECK(hipMemsetAsync(sh.translation, 0xff,
                                    2 * static_cast<size_t>(cfg.ST_all) * 4, stream));

      for (uint32_t top = 0; top < kLayers; ++top) {
        for (uint32_t btm = top; btm != 0xffffffffu; --btm) {
          do_merge(top, btm);
          if (top < kLayers - 1 && top == btm)
            do_select(top);
          do_sym(btm);
        }
      }
      for (uint32_t r = 0; r < refinement_iterations; ++r) {
        for (uint32_t layer = kLayers - 2; layer != 0xffffffffu; --layer) {
          do_merge(kLayers - 1, layer);
          do_sym(layer);
        }
      }
      const float ms = timer.stop();
      ctx.build_ms += ms;
      if (ctx.swap) {
        retire_built_shard(ctx, si);
        shard_consumed(ctx, si, stream);
      }
      sh.ready = true;
      GGNN_LOG(0, "[GPU: %d] build(): part %u => %.3f s [%u points -> %.3f us/point]", ctx.device,
               sh.global_id, ms / 1000.f, N, ms * 1000.f / static_cast<float>(N));
    }
    GGNN_HIP_CHECK(hipStreamSynchronize(stream));
  }

  void build(uint32_t KBuild, float tau_build, uint32_t refinement_iterations,
             ggnn_measure measure)
  {
    prepare(KBuild);
    build_work = ggnn_build_work{};
    try {
      for_each_device(
          [&](DeviceCtx& ctx) { build_device(ctx, tau_build, refinement_iterations, measure); });
    }
    catch (...) {
      rollback_graph();
      throw;
    }
    build_ms = 0.f;
    for (const DeviceCtx& ctx : devs)
      build_ms += ctx.build_ms;  // "Sum of shard build times", ggnn.cu:237
    release_caller_copy();
  }

  // the engine's own host/device copy of the caller's base is no longer needed once every GPU
  // holds its slice
  void release_caller_copy()
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

  struct Staged {
    const void* ptr{nullptr};
    DeviceBuffer owned;
  };
  void check_query(uint64_t Nq, uint32_t D, ggnn_dtype dtype, const void* q) const
  {
    GGNN_REQUIRE(dtype == base_dtype, GGNN_INVALID_ARGUMENT,
                 "query data type does not match base data type");
    GGNN_REQUIRE(D == base_D, GGNN_INVALID_ARGUMENT, "query dimension does not match the base");
    GGNN_REQUIRE(Nq < 0xffffffffull, GGNN_INVALID_ARGUMENT, "too many queries");
    GGNN_REQUIRE(!Nq || q != nullptr, GGNN_INVALID_ARGUMENT, "query pointer is null");
  }
  // query.referenceOnGPU (gpu_instance.cu:638-641): the full query set on ctx's GPU
  Staged stage_query(DeviceCtx& ctx, const void* q, uint64_t Nq, uint32_t D, ggnn_dtype dtype,
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

  // Pre-screen copy of shard si (traversal.hpp "Exact pre-screen"): pays when a float row spans
  // more cache lines than its code row, i.e. from 256 bytes per row on.
  bool ensure_prescreen(DeviceCtx& ctx, uint32_t si, ggnn_measure measure)
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

  // GPUInstance::query, gpu_instance.cu:626-743: all shards of one GPU into d_ids/d_dists
  // [Nq, K * shards_per_gpu]
  void query_device(DeviceCtx& ctx, const void* d_query, uint32_t nq, uint32_t k_query,
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
  void query(const void* q, uint64_t Nq, uint32_t D, ggnn_dtype dtype, ggnn_location loc,
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
  void query_split(const void* q, uint32_t nq, uint32_t D, ggnn_dtype dtype, ggnn_location loc,
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
  void grow_lane(DeviceCtx& owner, int lane, DeviceBuffer& b, size_t bytes)
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

  // Combines the per-GPU rows of one lane into the caller's [Nq, K] arrays.  blocking: waits and
  // copies through pinned staging; otherwise everything is only enqueued on the lane's streams
  // (the caller's arrays are written by asynchronous copies: device or page-locked memory).
  void exchange(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
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
    exchange_peer_copies(lane, nq, k_query, row, ids_out, dists_out, blocking);
  }

  // slice of the query set that GPU g merges and returns
  static void slice_of(uint32_t nq, size_t G, size_t g, uint32_t* first, uint32_t* count)
  {
    const uint32_t per = (nq + static_cast<uint32_t>(G) - 1) / static_cast<uint32_t>(G);
    *first = std::min<uint32_t>(nq, static_cast<uint32_t>(g) * per);
    *count = std::min<uint32_t>(per, nq - *first);
  }

  // merged slice [first, first + count) of ctx.m_pack -> caller's arrays
  void return_slice(DeviceCtx& ctx, int lane, uint32_t nq, uint32_t k_query, uint32_t first,
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
  void finish_slices(int lane, uint32_t nq, uint32_t k_query, size_t G_slices, int32_t* ids_out,
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
  void exchange_rccl(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
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
    last_exchange = "rccl";
  }

  // Several contexts without RCCL (contexts sharing one device, or no librccl): packed rows to the
  // first GPU with peer copies, k-way merge there.
  void exchange_peer_copies(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
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
  void query_async(const void* d_query, uint64_t Nq, uint32_t D, ggnn_dtype dtype,
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
  void enqueue_local_search(DeviceCtx& ctx, int lane, const void* d_query, uint32_t nq,
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
  void synchronize_slot(uint32_t slot)
  {
    for (DeviceCtx& ctx : devs) {
      ctx.activate();
      hipStream_t st = ctx.shard_stream[slot % DeviceCtx::kShardStreams];
      if (st)
        GGNN_HIP_CHECK(hipStreamSynchronize(st));
    }
  }
  void synchronize()
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
  void bf_query(const void* q, uint64_t Nq, uint32_t D, ggnn_dtype dtype, ggnn_location loc,
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

  std::filesystem::path part_file(uint32_t shard) const
  {
    // gpu_instance.cu:86-115 (part_<global_shard_id>.ggnn)
    return graph_dir / ("part_" + std::to_string(shard) + ".ggnn");
  }

  void store()
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

  void load(uint32_t KBuild)
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
  void load_shards()
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
};

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
