// ggnn C++ facade: class GGNN<KeyT, ValueT> with the reference's public signatures
// (include/ggnn/base/ggnn.cuh:42-182), header-only over the C-ABI of libggnn_amd.so.
// Link with -lggnn_amd -lamdhip64; existing callers (examples/cpp-and-cuda/ggnn_main*.cpp) compile
// unchanged against this header.
#ifndef GGNN_AMD_FACADE_GGNN_CUH
#define GGNN_AMD_FACADE_GGNN_CUH

#include <array>
#include <cstdint>
#include <filesystem>
#include <span>
#include <type_traits>
#include <vector>

#include "dataset.cuh"
#include "def.h"

namespace ggnn {

// include/ggnn/base/graph.h:38-71 -- non-owning device views of one graph shard
template <typename KeyT, typename ValueT>
struct Graph {
  std::array<Dataset<KeyT>, 4> graph{};
  std::array<Dataset<KeyT>, 4> translation{};
  std::array<Dataset<KeyT>, 4> selection{};
  Dataset<ValueT> nn1_stats{};
  ggnn_graph_config config{};
};

template <typename KeyT, typename ValueT>
class GGNN {
  static_assert(std::is_same_v<KeyT, int32_t> && std::is_same_v<ValueT, float>,
                "the engine is instantiated for <int32_t, float> (lib.h:23-31 of the reference)");

 public:
  using Results = ggnn::Results<KeyT, ValueT>;
  using Graph = ggnn::Graph<KeyT, ValueT>;

  static constexpr uint32_t MIN_D = 1, MAX_D = 4096, MIN_KBUILD = 2, MAX_KBUILD = 512;

  GGNN() { detail::check(ggnn_create(&h_), nullptr); }
  GGNN(const GGNN&) = delete;
  GGNN& operator=(const GGNN&) = delete;
  GGNN(GGNN&& o) noexcept : h_(o.h_), owned_base_(std::move(o.owned_base_)), on_gpu_(o.on_gpu_)
  {
    o.h_ = nullptr;
  }
  GGNN& operator=(GGNN&& o) noexcept
  {
    if (this != &o) {
      ggnn_destroy(h_);
      h_ = o.h_;
      o.h_ = nullptr;
      owned_base_ = std::move(o.owned_base_);
      on_gpu_ = o.on_gpu_;
    }
    return *this;
  }
  virtual ~GGNN() { ggnn_destroy(h_); }

  virtual void setWorkingDirectory(const std::filesystem::path& dir)
  {
    detail::check(ggnn_set_working_directory(h_, dir.string().c_str()), h_);
  }
  virtual void setCPUMemoryLimit(const size_t memory_limit)
  {
    detail::check(ggnn_set_cpu_memory_limit(h_, memory_limit), h_);
  }
  virtual void setReservedGPUMemory(const size_t reserved_memory)
  {
    detail::check(ggnn_set_reserved_gpu_memory(h_, reserved_memory), h_);
  }
  virtual void setGPUs(const std::span<const int>& gpu_ids)
  {
    detail::check(ggnn_set_gpus(h_, gpu_ids.data(), gpu_ids.size()), h_);
  }
  void setGPUs(const std::vector<int>& gpu_ids)
  {
    setGPUs(std::span<const int>{gpu_ids.data(), gpu_ids.size()});
  }
  virtual void setShardSize(const uint32_t N_shard)
  {
    detail::check(ggnn_set_shard_size(h_, N_shard), h_);
  }
  virtual void setReturnResultsOnGPU(const bool return_results_on_gpu = true)
  {
    detail::check(ggnn_set_return_results_on_gpu(h_, return_results_on_gpu), h_);
    on_gpu_ = return_results_on_gpu;
  }

  // takes ownership of the dataset (it lives as long as this object)
  virtual void setBase(GenericDataset&& base)
  {
    owned_base_ = std::move(base);
    set_base_impl(owned_base_);
  }
  // borrows: the caller keeps the data alive
  void setBaseReference(const GenericDataset& base) { set_base_impl(base); }
  void setBaseReference(GenericDataset&&) = delete;

  virtual void build(const uint32_t KBuild, const float tau_build,
                     const uint32_t refinement_iterations = 2,
                     const DistanceMeasure measure = DistanceMeasure::Euclidean)
  {
    detail::check(ggnn_build(h_, KBuild, tau_build, refinement_iterations,
                             static_cast<ggnn_measure>(measure)),
                  h_);
  }
  virtual void store() { detail::check(ggnn_store(h_), h_); }
  virtual void load(const uint32_t KBuild) { detail::check(ggnn_load(h_, KBuild), h_); }

  [[nodiscard]] virtual Results query(const GenericDataset& query, const uint32_t KQuery,
                                      const float tau_query, const uint32_t max_iterations = 400,
                                      const DistanceMeasure measure = DistanceMeasure::Euclidean)
  {
    const uint32_t width = on_gpu_ ? KQuery * num_shards() : KQuery;
    Results r = make_results(query, width);
    detail::check(ggnn_query(h_, query.raw(), query.N, query.D, dtype_of(query), loc_of(query),
                             query.gpu_id, KQuery, tau_query, max_iterations,
                             static_cast<ggnn_measure>(measure), r.ids.data(), r.dists.data(),
                             on_gpu_ ? GGNN_GPU : GGNN_CPU),
                  h_);
    return r;
  }
  [[nodiscard]] virtual Results bfQuery(const GenericDataset& query, const uint32_t KGT = 100,
                                        const DistanceMeasure measure = DistanceMeasure::Euclidean)
  {
    Results r = make_results(query, KGT);
    detail::check(ggnn_bf_query(h_, query.raw(), query.N, query.D, dtype_of(query), loc_of(query),
                                query.gpu_id, KGT, static_cast<ggnn_measure>(measure),
                                r.ids.data(), r.dists.data(), on_gpu_ ? GGNN_GPU : GGNN_CPU),
                  h_);
    return r;
  }

  [[nodiscard]] virtual const Graph& getGraph(const uint32_t global_shard_id = 0)
  {
    ggnn_graph_view v{};
    detail::check(ggnn_get_graph(h_, global_shard_id, &v), h_);
    const ggnn_graph_config& c = v.config;
    graph_view_.config = c;
    auto* g = const_cast<KeyT*>(v.graph);
    auto* tr = const_cast<KeyT*>(v.translation);
    auto* sel = const_cast<KeyT*>(v.selection);
    for (uint32_t l = 0; l < 4; ++l) {
      graph_view_.graph[l] = Dataset<KeyT>::referenceGPUData(
          g + static_cast<size_t>(c.Ns_offsets[l]) * c.KBuild, c.Ns[l], c.KBuild, v.gpu_id);
      if (l) {
        graph_view_.translation[l] =
            Dataset<KeyT>::referenceGPUData(tr + c.STs_offsets[l], c.Ns[l], 1, v.gpu_id);
        graph_view_.selection[l] =
            Dataset<KeyT>::referenceGPUData(sel + c.STs_offsets[l], c.Ns[l], 1, v.gpu_id);
      }
    }
    graph_view_.nn1_stats =
        Dataset<ValueT>::referenceGPUData(const_cast<ValueT*>(v.nn1_stats), 2, 1, v.gpu_id);
    return graph_view_;
  }

 protected:
  GGNN(int) {}

 private:
  static ggnn_dtype dtype_of(const GenericDataset& d)
  {
    if (d.type == DataType::FLOAT)
      return GGNN_F32;
    if (d.type == DataType::UINT8)
      return GGNN_U8;
    throw std::runtime_error("unsupported datatype (float and uint8_t are supported)");
  }
  static ggnn_location loc_of(const GenericDataset& d)
  {
    return (d.isGPUAccessible() && !d.isCPUAccessible()) ? GGNN_GPU : GGNN_CPU;
  }
  void set_base_impl(const GenericDataset& b)
  {
    detail::check(ggnn_set_base(h_, b.raw(), b.N, b.D, dtype_of(b), loc_of(b), b.gpu_id, 0), h_);
  }
  // shards per GPU as the engine laid them out (results on the GPU are [Nq, KQuery * shards],
  // ggnn.cu:299-306); asked from the engine so that nothing is cached in this movable object
  uint32_t num_shards() const
  {
    uint32_t per_gpu = 1;
    detail::check(ggnn_get_shard_layout(h_, nullptr, &per_gpu, nullptr), h_);
    return per_gpu;
  }
  Results make_results(const GenericDataset& query, uint32_t width)
  {
    if (on_gpu_) {
      int dev = query.gpu_id >= 0 ? query.gpu_id : 0;
      ggnn_graph_view v{};
      if (ggnn_get_graph(h_, 0, &v) == GGNN_OK)
        dev = v.gpu_id;
      return Results{Dataset<KeyT>::emptyOnGPU(query.N, width, dev),
                     Dataset<ValueT>::emptyOnGPU(query.N, width, dev)};
    }
    return Results{Dataset<KeyT>::empty(query.N, width, true),
                   Dataset<ValueT>::empty(query.N, width, true)};
  }

  ggnn_t* h_{nullptr};
  GenericDataset owned_base_{};
  Graph graph_view_{};
  bool on_gpu_{false};
};

}  // namespace ggnn

#endif
