// Host layout math of one graph shard.
// Reference: GraphDerivedParameters / GraphDimensions, src/ggnn/base/graph_config.cpp:39-98 and
// include/ggnn/base/graph_config.h:31-112.  The float arithmetic (powf on float operands,
// truncation to uint32) follows the reference exactly because it decides G and S0.
#include <cmath>
#include <cstring>

#include "common.hpp"

namespace ggnn_amd {

int g_log_level = 0;

void graph_config_init(uint32_t N, uint32_t D, uint32_t KBuild, ggnn_graph_config* c)
{
  GGNN_REQUIRE(N > 0, GGNN_INVALID_ARGUMENT, "graph needs at least one point");
  GGNN_REQUIRE(KBuild >= 2 && KBuild <= 512, GGNN_INVALID_ARGUMENT,
               "KBuild must be in [2, 512]");  // ggnn.cuh:51-52, ggnn.cu:169-170
  std::memset(c, 0, sizeof(*c));
  c->N = N;
  c->D = D;
  c->KBuild = KBuild;
  c->KF = KBuild / 2;
  c->S = next_multiple32(c->KF + 1);

  const float levels = static_cast<float>(kLayers - 1);
  const float growth =
      std::pow(static_cast<float>(N) / static_cast<float>(c->S), 1.f / (kLayers - 1));
  GGNN_REQUIRE(growth >= 1.0f, GGNN_INVALID_ARGUMENT,
               "base too small for a 4-layer graph with this KBuild");
  const uint32_t g_floor = static_cast<uint32_t>(growth);
  const uint32_t g_ceil = g_floor + 1;
  const float s0_floor = static_cast<float>(N) / std::pow(static_cast<float>(g_floor), levels);
  const float s0_ceil = static_cast<float>(N) / std::pow(static_cast<float>(g_ceil), levels);
  const float s = static_cast<float>(c->S);
  // take the smaller growth factor when the larger one would leave bottom segments too small
  // for KBuild neighbours, or when its segment size is closer to S (graph_config.cpp:81-87)
  const bool use_floor = (static_cast<uint32_t>(s0_ceil) < KBuild) ||
                         (std::abs(s0_floor - s) < std::abs(s0_ceil - s));
  c->G = use_floor ? g_floor : g_ceil;
  c->S0 = static_cast<uint32_t>(use_floor ? s0_floor : s0_ceil);
  GGNN_REQUIRE(c->G >= 1 && c->S0 >= 1, GGNN_INVALID_ARGUMENT,
               "base too small for a 4-layer graph with this KBuild");
  c->S0_off = N - c->G * c->G * c->G * c->S0;
  c->SG = c->S / c->G;
  c->SG_off = c->S - c->SG * c->G;

  uint32_t blocks = 1;
  for (int l = kLayers - 1; l >= 0; --l, blocks *= c->G) {
    c->Bs[l] = blocks;
    c->Ns[l] = blocks * c->S;
  }
  c->Ns[0] = N;
  c->Ns_offsets[0] = 0;
  c->Ns_offsets[1] = N;
  c->STs_offsets[0] = 0;
  c->STs_offsets[1] = 0;
  for (uint32_t l = 2; l < kLayers; ++l) {
    c->Ns_offsets[l] = c->Ns_offsets[l - 1] + c->Ns[l - 1];
    c->STs_offsets[l] = c->STs_offsets[l - 1] + c->Ns[l - 1];
  }
  c->N_all = c->Ns_offsets[kLayers - 1] + c->Ns[kLayers - 1];
  c->ST_all = c->STs_offsets[kLayers - 1] + c->Ns[kLayers - 1];
}

}  // namespace ggnn_amd
