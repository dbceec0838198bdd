// Registry behind hooks.hpp / ggnn_set_hook().
#include "hooks.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>

namespace ggnn_amd {
namespace {

struct Entry {
  const char* name;
  int64_t def;
};
constexpr Entry kTable[kHookCount] = {
    {"PRESCREEN", 1},     {"EXCHANGE", 0},      {"SYM_PRESCREEN", -1}, {"SHARD_OVERLAP", 1},
    {"VIS_SLOTS", 8},     {"VIS_TAG_SET", 1}, {"QUERY_SPLIT", -1}, {"RESIDENT_SHARDS", 0}, {"XCD_MAP", 3}, {"BF_POOL_KEEP_MB", 1024}, {"BF_NO_I8", 0},
    {"BF_I8_V1", 0},      {"BF_SLICES", 0},     {"BF_NO_CENTER", 0},   {"BF_TILES", 3},
    {"BF_I8_NOSHARE", 0}, {"BF_I8_RANKS", -1},    {"BF_SCAN", 0},        {"RCCL_FAIL_AFTER", 0},
    {"QUERY_EARLY", 1},   {"MERGE_EARLY", 1},   {"QUERY_LDS_PAD", 0},
    {"QUERY_GLOBAL_RING", 1}, {"BF_I8_REFRESH", 64}, {"BF_I8_SEED", 0},
    {"MERGE_COUNTING", 0},
};
std::atomic<bool> g_set[kHookCount];
std::atomic<int64_t> g_value[kHookCount];

bool env_enabled()
{
  const char* e = std::getenv("GGNN_TEST_HOOKS");
  return e && e[0] == '1';
}

}  // namespace

const char* hook_name(Hook h)
{
  return kTable[h].name;
}

int hook_by_name(const char* name)
{
  if (!name)
    return -1;
  if (std::strncmp(name, "GGNN_", 5) == 0)
    name += 5;
  for (int i = 0; i < kHookCount; ++i)
    if (std::strcmp(name, kTable[i].name) == 0)
      return i;
  return -1;
}

void hook_set(Hook h, int64_t value)
{
  g_value[h].store(value);
  g_set[h].store(true);
}

void hook_reset(Hook h)
{
  g_set[h].store(false);
}

int64_t hook(Hook h)
{
  if (g_set[h].load())
    return g_value[h].load();
  if (env_enabled()) {
    const std::string var = std::string("GGNN_") + kTable[h].name;
    if (const char* e = std::getenv(var.c_str())) {
      if (h == kHookExchange) {
        if (std::strcmp(e, "rccl") == 0)
          return 1;
        if (std::strcmp(e, "copy") == 0)
          return 2;
        if (std::strcmp(e, "gather") == 0)
          return 3;
      }
      return std::atoll(e);
    }
  }
  return kTable[h].def;
}

}  // namespace ggnn_amd
