// ggnn C++ facade: Dataset<T> / GenericDataset / Results with the public surface the reference's
// callers use (include/ggnn/base/dataset.cuh:38-166, data.cuh), implemented from scratch on the
// HIP runtime API.  A dataset is a move-only 2-D array [N x D] on the host or on one GPU that
// either owns its memory or references foreign memory.
#ifndef GGNN_AMD_FACADE_DATASET_CUH
#define GGNN_AMD_FACADE_DATASET_CUH

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <limits>
#include <span>
#include <stdexcept>
#include <string>
#include <utility>

#include "def.h"

namespace ggnn {

enum class DataType : uint16_t { UNKNOWN, BYTE, UINT8, INT32, UINT32, FLOAT };
enum class DataLocation : uint16_t {
  UNKNOWN, GPU, MANAGED, CPU_PINNED, CPU_MALLOC, FOREIGN_GPU, FOREIGN_CPU
};

namespace detail {
template <typename T> struct TypeTag;
template <> struct TypeTag<std::byte> { static constexpr DataType value = DataType::BYTE; };
template <> struct TypeTag<uint8_t> { static constexpr DataType value = DataType::UINT8; };
template <> struct TypeTag<int32_t> { static constexpr DataType value = DataType::INT32; };
template <> struct TypeTag<uint32_t> { static constexpr DataType value = DataType::UINT32; };
template <> struct TypeTag<float> { static constexpr DataType value = DataType::FLOAT; };
inline size_t size_of(DataType t)
{
  return (t == DataType::BYTE || t == DataType::UINT8) ? 1 : (t == DataType::UNKNOWN ? 0 : 4);
}
inline void hip_check(hipError_t e, const char* what)
{
  if (e != hipSuccess)
    throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace detail

struct GenericDataset {
  uint64_t N{0};
  uint32_t D{0};
  DataType type{DataType::UNKNOWN};
  DataLocation location{DataLocation::UNKNOWN};
  int32_t gpu_id{-1};

  GenericDataset() = default;
  GenericDataset(const GenericDataset&) = delete;
  GenericDataset& operator=(const GenericDataset&) = delete;
  GenericDataset(GenericDataset&& o) noexcept { steal(o); }
  GenericDataset& operator=(GenericDataset&& o) noexcept
  {
    if (this != &o) {
      release();
      steal(o);
    }
    return *this;
  }
  virtual ~GenericDataset() { release(); }

  void* raw() { return mem_; }
  const void* raw() const { return mem_; }
  size_t element_size() const { return detail::size_of(type); }
  size_t numel() const { return static_cast<size_t>(N) * D; }
  size_t size_bytes() const { return numel() * element_size(); }
  bool isGPUAccessible() const
  {
    return location == DataLocation::GPU || location == DataLocation::MANAGED ||
           location == DataLocation::FOREIGN_GPU;
  }
  bool isCPUAccessible() const
  {
    return location == DataLocation::CPU_MALLOC || location == DataLocation::CPU_PINNED ||
           location == DataLocation::MANAGED || location == DataLocation::FOREIGN_CPU;
  }
  // non-owning view of the same memory
  GenericDataset reference() const
  {
    GenericDataset r;
    r.N = N;
    r.D = D;
    r.type = type;
    r.gpu_id = gpu_id;
    r.mem_ = mem_;
    r.location = isGPUAccessible() && !isCPUAccessible() ? DataLocation::FOREIGN_GPU
                                                         : DataLocation::FOREIGN_CPU;
    return r;
  }

 protected:
  void* mem_{nullptr};
  void steal(GenericDataset& o)
  {
    N = o.N;
    D = o.D;
    type = o.type;
    location = o.location;
    gpu_id = o.gpu_id;
    mem_ = o.mem_;
    o.mem_ = nullptr;
    o.N = 0;
    o.location = DataLocation::UNKNOWN;
  }
  void release()
  {
    if (!mem_)
      return;
    switch (location) {
      case DataLocation::GPU:
      case DataLocation::MANAGED:
        (void)hipFree(mem_);
        break;
      case DataLocation::CPU_PINNED:
        (void)hipHostFree(mem_);
        break;
      case DataLocation::CPU_MALLOC:
        std::free(mem_);
        break;
      default:
        break;  // foreign memory is not ours
    }
    mem_ = nullptr;
  }
  friend struct DatasetAccess;
};

template <typename T>
struct Dataset : public GenericDataset {
  Dataset() { type = detail::TypeTag<T>::value; }
  Dataset(GenericDataset&& g) : GenericDataset(std::move(g))
  {
    if (type != detail::TypeTag<T>::value)
      throw std::runtime_error("dataset element type mismatch");
  }
  Dataset(Dataset&&) noexcept = default;
  Dataset& operator=(Dataset&&) noexcept = default;

  T* data() { return static_cast<T*>(mem_); }
  const T* data() const { return static_cast<const T*>(mem_); }
  size_t size() const { return numel(); }
  T& operator[](size_t i) { return data()[i]; }
  const T& operator[](size_t i) const { return data()[i]; }
  T& at(size_t i)
  {
    if (i >= size())
      throw std::out_of_range("Index " + std::to_string(i) + " is out of bounds (size " +
                              std::to_string(size()) + ").");
    return data()[i];
  }
  const T& at(size_t i) const { return const_cast<Dataset*>(this)->at(i); }
  T* begin() { return data(); }
  T* end() { return data() + size(); }
  const T* begin() const { return data(); }
  const T* end() const { return data() + size(); }
  operator T*() { return data(); }
  operator const T*() const { return data(); }

  static Dataset empty(uint64_t N, uint32_t D, bool pin_memory = false)
  {
    Dataset d;
    d.N = N;
    d.D = D;
    const size_t bytes = d.size_bytes();
    if (pin_memory) {
      detail::hip_check(hipHostMalloc(&d.mem_, bytes ? bytes : 1, hipHostMallocDefault),
                        "hipHostMalloc");
      d.location = DataLocation::CPU_PINNED;
    }
    else {
      d.mem_ = std::malloc(bytes ? bytes : 1);
      if (!d.mem_)
        throw std::bad_alloc();
      d.location = DataLocation::CPU_MALLOC;
    }
    return d;
  }
  static Dataset emptyOnGPU(uint64_t N, uint32_t D, int32_t gpu_id)
  {
    Dataset d;
    d.N = N;
    d.D = D;
    d.gpu_id = gpu_id;
    detail::hip_check(hipSetDevice(gpu_id), "hipSetDevice");
    detail::hip_check(hipMalloc(&d.mem_, d.size_bytes() ? d.size_bytes() : 1), "hipMalloc");
    d.location = DataLocation::GPU;
    return d;
  }
  static Dataset copy(const std::span<const T>& src, uint32_t D, bool pin_memory = false)
  {
    if (!D || src.size() % D)
      throw std::invalid_argument("data size is not a multiple of D");
    Dataset d = empty(src.size() / D, D, pin_memory);
    std::memcpy(d.mem_, src.data(), src.size_bytes());
    return d;
  }
  static Dataset referenceCPUData(T* data, uint64_t N, uint32_t D)
  {
    Dataset d;
    d.N = N;
    d.D = D;
    d.mem_ = data;
    d.location = DataLocation::FOREIGN_CPU;
    return d;
  }
  static Dataset referenceGPUData(T* data, uint64_t N, uint32_t D, int32_t gpu_id)
  {
    Dataset d;
    d.N = N;
    d.D = D;
    d.mem_ = data;
    d.gpu_id = gpu_id;
    d.location = DataLocation::FOREIGN_GPU;
    return d;
  }
  // XVECS files (per vector: uint32 D, then D values), dataset.cu:118-233
  static Dataset load(const std::filesystem::path& file, uint32_t from = 0,
                      uint32_t num = std::numeric_limits<uint32_t>::max(), bool pin_memory = false)
  {
    std::ifstream f(file, std::ios::binary);
    uint32_t D = 0;
    if (!f.read(reinterpret_cast<char*>(&D), 4) || !D)
      throw std::runtime_error("cannot read " + file.string());
    const size_t rec = 4 + static_cast<size_t>(D) * sizeof(T);
    const size_t total = std::filesystem::file_size(file) / rec;
    const size_t n = from < total ? std::min<size_t>(num, total - from) : 0;
    Dataset d = empty(n, D, pin_memory);
    for (size_t i = 0; i < n; ++i) {
      f.seekg(static_cast<std::streamoff>((from + i) * rec + 4));
      f.read(reinterpret_cast<char*>(d.data() + i * D), static_cast<std::streamsize>(D * sizeof(T)));
    }
    if (!f)
      throw std::runtime_error("short read from " + file.string());
    return d;
  }
  void store(const std::filesystem::path& file) const
  {
    if (!isCPUAccessible())
      throw std::runtime_error("store() needs CPU-accessible data");
    std::ofstream f(file, std::ios::binary | std::ios::trunc);
    for (uint64_t i = 0; i < N; ++i) {
      f.write(reinterpret_cast<const char*>(&D), 4);
      f.write(reinterpret_cast<const char*>(data() + i * D),
              static_cast<std::streamsize>(D * sizeof(T)));
    }
    if (!f)
      throw std::runtime_error("cannot write " + file.string());
  }
  // copies between any two locations
  void copyTo(Dataset& other, hipStream_t stream = nullptr) const
  {
    if (other.size_bytes() < size_bytes())
      throw std::out_of_range("destination dataset is too small");
    detail::hip_check(hipMemcpyAsync(other.mem_, mem_, size_bytes(), hipMemcpyDefault, stream),
                      "hipMemcpyAsync");
    detail::hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
  }
  // rows [from, from + num) -> the head of `other` (include/ggnn/base/dataset.cuh:156,
  // dataset.cu:269-301: the reference's shard upload, gpu_instance.cu:491); enqueued on `stream`
  // like the reference's cudaMemcpyAsync, host-to-host copies are immediate
  void copyRangeTo(uint64_t from, uint64_t num, Dataset& other, hipStream_t stream = nullptr) const
  {
    if (!mem_ || !other.mem_)
      throw std::runtime_error("copyRangeTo: dataset without memory");
    if (from > N || num > N - from)
      throw std::out_of_range("copyRangeTo: rows [" + std::to_string(from) + ", " +
                              std::to_string(from + num) + ") are out of bounds (N " +
                              std::to_string(N) + ").");
    const size_t bytes = static_cast<size_t>(num) * D * sizeof(T);
    if (other.size_bytes() < bytes)
      throw std::out_of_range("destination dataset is too small");
    const T* src = data() + static_cast<size_t>(from) * D;
    if (isCPUAccessible() && other.isCPUAccessible()) {
      std::memcpy(other.mem_, src, bytes);
      return;
    }
    // (GPUs of one process address each other's memory: no staging through the host as in
    // dataset.cu:284-291)
    detail::hip_check(hipMemcpyAsync(other.mem_, src, bytes, hipMemcpyDefault, stream),
                      "hipMemcpyAsync");
  }
  Dataset clone(hipStream_t stream = nullptr) const
  {
    Dataset d = (isGPUAccessible() && !isCPUAccessible())
                    ? emptyOnGPU(N, D, gpu_id)
                    : empty(N, D, location == DataLocation::CPU_PINNED);
    copyTo(d, stream);
    return d;
  }
  // the data as seen from GPU gpu_id (include/ggnn/base/dataset.cuh:159, dataset.cu:326-334): a
  // non-owning reference when it already lives there, else a copy in that GPU's memory
  Dataset referenceOnGPU(int gpu_id_, hipStream_t stream = nullptr) const
  {
    if (isGPUAccessible() && gpu_id == gpu_id_)
      return reference();
    Dataset d = emptyOnGPU(N, D, gpu_id_);
    copyTo(d, stream);
    return d;
  }
  Dataset reference() const { return Dataset(GenericDataset::reference()); }
};

// include/ggnn/base/dataset.cuh:162-166
template <typename KeyT, typename ValueT>
struct Results {
  Dataset<KeyT> ids;
  Dataset<ValueT> dists;
};

}  // namespace ggnn

#endif
