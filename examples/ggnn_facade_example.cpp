// Exercises the C++ facade the way the reference's examples/cpp-and-cuda programs do: random
// data, setBaseReference, build, query, bfQuery, structured bindings, exceptions on misuse.
#include <ggnn/base/ggnn.cuh>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

using namespace ggnn;

int main()
{
  const size_t N_base = 10'000, N_query = 1'000;
  const uint32_t dim = 128, KQuery = 10;
  std::vector<float> base_data(N_base * dim), query_data(N_query * dim);
  std::mt19937 prng{42};
  std::uniform_int_distribution<int> uniform{0, 255};
  for (float& x : base_data)
    x = static_cast<float>(uniform(prng));
  for (float& x : query_data)
    x = static_cast<float>(uniform(prng));

  using GGNN = ggnn::GGNN<int32_t, float>;
  GGNN engine{};
  bool threw = false;
  try {
    engine.build(24, 0.5f);
  }
  catch (const std::runtime_error&) {
    threw = true;  // "The base needs to be set before building a graph."
  }
  if (!threw)
    return 2;

  Dataset<float> base = Dataset<float>::copy(base_data, dim, true);
  Dataset<float> query = Dataset<float>::copy(query_data, dim, true);
  engine.setBaseReference(base);
  engine.build(24, 0.5f);
  const auto [indices, dists] = engine.query(query, KQuery, 0.64f);
  const auto [gt, gt_dists] = engine.bfQuery(query, KQuery);

  size_t hits = 0;
  for (size_t n = 0; n < N_query; ++n)
    for (uint32_t i = 0; i < KQuery; ++i)
      for (uint32_t j = 0; j < KQuery; ++j)
        hits += indices[n * KQuery + i] == gt[n * KQuery + j];
  const double recall = static_cast<double>(hits) / (N_query * KQuery);
  std::printf("recall@%u = %.4f, first neighbour of query 0: base[%d] at squared distance %.1f\n",
              KQuery, recall, indices[0], dists[0]);
  const auto& graph = engine.getGraph(0);
  std::printf("graph layers: %llu %llu %llu %llu\n", (unsigned long long)graph.graph[0].N,
              (unsigned long long)graph.graph[1].N, (unsigned long long)graph.graph[2].N,
              (unsigned long long)graph.graph[3].N);
  return recall > 0.9 ? 0 : 1;
}
