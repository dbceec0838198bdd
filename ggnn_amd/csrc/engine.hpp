// Engine behind the C-ABI (include/ggnn_c.h): shard residency, build schedule, query / bf_query
// drivers.  Replaces, for the hot path only, GGNNImpl (src/ggnn/base/ggnn.cu:124-413),
// GPUInstance::build/query (src/ggnn/base/gpu_instance.cu:499-584, 626-790) and
// GraphConstructionImpl::build/refine (src/ggnn/construction/graph_construction.cu:128-147).
//
// MI355X-first choices: every shard of the base and its graph stay resident in HBM (288 GB), so
// the reference's GPU<->CPU<->disk swapping is not reproduced; one engine drives one GPU
// (multi-GPU = one process per GPU, shards exchanged with an RCCL all-gather, see
// ggnn_amd/distributed.py and DESIGN.md).
//
// engine.hpp: the handle and its helper types; the member functions live in
//   engine_residency.cpp  base staging, prepare, pre-screen copies, store / load
//   engine_swap.cpp       out-of-core shards: GPU slots <-> pinned host buffers <-> part files
//   engine_build.cpp      build / refine schedule of one GPU
//   engine_query.cpp      shard loop, blocking / split / asynchronous query, bf_query
//   engine_exchange.cpp   candidates between the GPUs of one handle: RCCL all-gather, peer copies
//   engine_cabi.cpp       extern "C" entry points (include/ggnn_c.h)
#pragma once
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

inline size_t dtype_size(ggnn_dtype t)
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
  decltype(&ncclCommCount) CommCount{nullptr};  // optional (tracing only)
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
      x.CommCount = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
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
  void pin_host_copy();
  void drop_host_copy();
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
  void destroy_comms();
  bool ensure_comms();

  size_t row_bytes() const { return static_cast<size_t>(pad_D) * dtype_size(base_dtype); }
  uint32_t num_shards() const { return shards_per_gpu * static_cast<uint32_t>(devs.size()); }
  bool has_graph() const;
  void rollback_graph();

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

  std::vector<int> resolve_gpus() const;

  void stage_base_slice(DeviceCtx& ctx, uint64_t row0, uint64_t rows);

  const void* shard_base(const DeviceCtx& ctx, uint32_t local_shard) const;

  // ---- out-of-core shards (SwapState) -----------------------------------------------------------
  bool swapping() const { return !devs.empty() && devs[0].swap != nullptr; }

  uint32_t plan_gpu_slots(const DeviceCtx& ctx, uint32_t spg, bool base_on_this_gpu) const;

  void setup_swap(DeviceCtx& ctx, uint32_t slots, bool base_on_this_gpu);

  void upload_base_shard(DeviceCtx& ctx, uint32_t si, hipStream_t st);

  void* host_pool_of(DeviceCtx& ctx, uint32_t si);

  void acquire_shard(DeviceCtx& ctx, uint32_t si, hipStream_t st, bool with_graph = true);
  void shard_consumed(DeviceCtx& ctx, uint32_t si, hipStream_t st);
  void retire_built_shard(DeviceCtx& ctx, uint32_t si);
  void write_part(uint32_t global_shard, const void* host);
  void read_part(uint32_t global_shard, void* host);

  void prepare(uint32_t KBuild);

  void build_device(DeviceCtx& ctx, float tau_build, uint32_t refinement_iterations,
                    ggnn_measure measure);

  void build(uint32_t KBuild, float tau_build, uint32_t refinement_iterations,
             ggnn_measure measure);

  void release_caller_copy();

  struct Staged {
    const void* ptr{nullptr};
    DeviceBuffer owned;
  };
  void check_query(uint64_t Nq, uint32_t D, ggnn_dtype dtype, const void* q) const;
  Staged stage_query(DeviceCtx& ctx, const void* q, uint64_t Nq, uint32_t D, ggnn_dtype dtype,
                     ggnn_location loc, int q_gpu);

  bool ensure_prescreen(DeviceCtx& ctx, uint32_t si, ggnn_measure measure);

  void query_device(DeviceCtx& ctx, const void* d_query, uint32_t nq, uint32_t k_query,
                    float tau_query, uint32_t max_iterations, ggnn_measure measure,
                    int32_t* d_ids, float* d_dists);

  void query(const void* q, uint64_t Nq, uint32_t D, ggnn_dtype dtype, ggnn_location loc,
             int q_gpu, uint32_t k_query, float tau_query, uint32_t max_iterations,
             ggnn_measure measure, int32_t* ids_out, float* dists_out, ggnn_location out_loc);

  void query_split(const void* q, uint32_t nq, uint32_t D, ggnn_dtype dtype, ggnn_location loc,
                   int q_gpu, uint32_t k_query, float tau_query, uint32_t max_iterations,
                   ggnn_measure measure, int32_t* ids_out, float* dists_out);

  void grow_lane(DeviceCtx& owner, int lane, DeviceBuffer& b, size_t bytes);

  void exchange(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
                float* dists_out, bool blocking);

  static void slice_of(uint32_t nq, size_t G, size_t g, uint32_t* first, uint32_t* count);

  void return_slice(DeviceCtx& ctx, int lane, uint32_t nq, uint32_t k_query, uint32_t first,
                    uint32_t count, int32_t* ids_out, float* dists_out, bool blocking);
  void finish_slices(int lane, uint32_t nq, uint32_t k_query, size_t G_slices, int32_t* ids_out,
                     float* dists_out);

  void exchange_rccl(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
                     float* dists_out, bool blocking);

  void exchange_peer_copies(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
                            float* dists_out, bool blocking);
  void merge_slices(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
                    float* dists_out, bool blocking);
  void exchange_gather_copies(int lane, uint32_t nq, uint32_t k_query, size_t row, int32_t* ids_out,
                              float* dists_out);

  void query_async(const void* d_query, uint64_t Nq, uint32_t D, ggnn_dtype dtype,
                   ggnn_location loc, int q_gpu, uint32_t k_query, float tau_query,
                   uint32_t max_iterations, ggnn_measure measure, int32_t* d_ids, float* d_dists,
                   uint32_t slot);

  void enqueue_local_search(DeviceCtx& ctx, int lane, const void* d_query, uint32_t nq,
                            uint32_t k_query, float tau_query, uint32_t max_iterations,
                            ggnn_measure measure, int32_t* d_ids, float* d_dists);
  void synchronize_slot(uint32_t slot);
  void synchronize();

  void bf_query(const void* q, uint64_t Nq, uint32_t D, ggnn_dtype dtype, ggnn_location loc,
                int q_gpu, uint32_t k_gt, ggnn_measure measure, int32_t* ids_out,
                float* dists_out, ggnn_location out_loc);

  std::filesystem::path part_file(uint32_t shard) const;

  void store();

  void load(uint32_t KBuild);
  void load_shards();
};
