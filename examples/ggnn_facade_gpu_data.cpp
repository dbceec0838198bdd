// Data that already lives on the GPU, through the facade -- what the reference's
// examples/cpp-and-cuda/ggnn_main_gpu_data.cu does (setBase(referenceGPUData(...)), a query
// dataset that references device memory, results on the host; tests/test_cpp_facade.py compiles
// that program itself with its CUDA / cuRAND calls swapped for HIP) -- followed by the two Dataset
// members the reference's own shard upload and query staging use: copyRangeTo (dataset.cuh:156,
// gpu_instance.cu:491) and referenceOnGPU (dataset.cuh:159, gpu_instance.cu:638).
#include <ggnn/base/ggnn.cuh>

#include <cstddef>
#include <cstdint>
#include <iostream>
#include <random>
#include <vector>

#include <hip/hip_runtime_api.h>

using namespace ggnn;
int main()
{
  using GGNN = ggnn::GGNN<int32_t, float>;

  const size_t N_base = 10'000, N_query = 10'000;
  const uint32_t D = 128;
  float* base = nullptr;
  float* query = nullptr;
  if (hipMalloc(&base, N_base * D * sizeof(float)) != hipSuccess ||
      hipMalloc(&query, N_query * D * sizeof(float)) != hipSuccess)
    return 3;
  {
    // uniform [0, 1) rows, generated on the host and copied over
    std::vector<float> h((N_base + N_query) * D);
    std::mt19937 prng{7};
    std::uniform_real_distribution<float> uniform{0.f, 1.f};
    for (float& x : h)
      x = uniform(prng);
    (void)hipMemcpy(base, h.data(), N_base * D * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(query, h.data() + N_base * D, N_query * D * sizeof(float), hipMemcpyHostToDevice);
  }

  // base and queries stay where they are: datasets that REFERENCE device memory
  GGNN engine{};
  const int32_t gpu_id = 0;
  engine.setBase(Dataset<float>::referenceGPUData(base, N_base, D, gpu_id));
  Dataset<float> d_query = Dataset<float>::referenceGPUData(query, N_query, D, gpu_id);
  engine.build(/*KBuild=*/24, /*tau_build=*/0.5f);
  const int32_t KQuery = 10;
  const auto [indices, dists] = engine.query(d_query, KQuery, 0.5f);   // results on the host

  for (int32_t i = 0; i < KQuery; i++)
    std::cout << "query 0, neighbour " << i << ": base[" << indices[i] << "] at " << dists[i] << "\n";
  bool ok = indices.N == N_query && indices.D == static_cast<uint32_t>(KQuery) && indices.isCPUAccessible();
  for (uint32_t i = 0; i + 1 < KQuery; i++)
    ok = ok && dists[i] <= dists[i + 1] && indices[i] >= 0 && indices[i] < static_cast<int32_t>(N_base);

  // Dataset::copyRangeTo / referenceOnGPU as GPUInstance uses them
  {
    Dataset<float> h_base = Dataset<float>::empty(N_base, D, true);
    d_query.copyRangeTo(0, N_query, h_base);                  // GPU -> pinned host, all rows
    ok = ok && hipStreamSynchronize(nullptr) == hipSuccess;
    const uint64_t from = 1234, num = 500;                     // one "shard" of the host data
    Dataset<float> shard = Dataset<float>::emptyOnGPU(num, D, gpu_id);
    h_base.copyRangeTo(from, num, shard);                      // host -> GPU, rows [from, from + num)
    Dataset<float> back = Dataset<float>::empty(num, D);
    shard.copyRangeTo(0, num, back);                           // GPU -> host
    ok = ok && hipStreamSynchronize(nullptr) == hipSuccess;
    Dataset<float> again = Dataset<float>::empty(num, D);
    h_base.copyRangeTo(from, num, again);                      // host -> host: immediate
    for (size_t i = 0; i < num * D; ++i)
      ok = ok && back[i] == h_base[from * D + i] && again[i] == back[i];
    bool threw = false;
    try {
      h_base.copyRangeTo(N_base - 10, 11, back);
    }
    catch (const std::out_of_range&) {
      threw = true;
    }
    ok = ok && threw;
    // data that is on the GPU already is referenced, host data is copied there
    Dataset<float> same = d_query.referenceOnGPU(gpu_id);
    Dataset<float> up = back.referenceOnGPU(gpu_id);
    ok = ok && same.data() == d_query.data() && same.location == DataLocation::FOREIGN_GPU;
    ok = ok && up.location == DataLocation::GPU && up.N == num && up.data() != back.data();
    Dataset<float> down = Dataset<float>::empty(num, D);
    up.copyTo(down);
    for (size_t i = 0; i < num * D; ++i)
      ok = ok && down[i] == back[i];
  }
  std::cout << (ok ? "gpu data example ok\n" : "gpu data example FAILED\n");

  (void)hipFree(base);
  (void)hipFree(query);

  return ok ? 0 : 1;
}
