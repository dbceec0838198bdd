// Stream-ordered scratch memory of the launchers (bf_query tiles and lists, visited rings of long
// searches).
#include <algorithm>
#include <mutex>

#include "common.hpp"
#include "hooks.hpp"

namespace ggnn_amd {

// Per-call scratch comes from a PRIVATE stream-ordered pool per device (not the device's default
// pool, whose settings belong to the rest of the process): freed blocks stay in it up to a
// bounded amount, so repeated calls cost no allocation, and nothing else in the process
// is affected.  Hook BF_POOL_KEEP_MB sets the amount kept (default 1024).  Creation is serialised:
// two handles (or threads) may reach their first bf_query on one device at the same time.
static hipMemPool_t scratch_pool()
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
    return nullptr;
  static std::mutex mtx;
  static hipMemPool_t pools[64] = {};
  static bool tried[64] = {};
  std::lock_guard<std::mutex> lock(mtx);
  if (!tried[dev]) {
    tried[dev] = true;
    hipMemPoolProps props{};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = dev;
    hipMemPool_t pool = nullptr;
    if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool) {
      uint64_t keep = static_cast<uint64_t>(std::clamp<int64_t>(hook(kHookBfPoolKeepMb), 0, 1 << 20))
                      << 20;
      (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
      pools[dev] = pool;
    }
    (void)hipGetLastError();
  }
  return pools[dev];
}
void* scratch_alloc(size_t bytes, hipStream_t stream)
{
  void* p = nullptr;
  if (hipMemPool_t pool = scratch_pool())
    GGNN_HIP_CHECK(hipMallocFromPoolAsync(&p, bytes, pool, stream));
  else
    GGNN_HIP_CHECK(hipMallocAsync(&p, bytes, stream));
  return p;
}

void scratch_free(void* p, hipStream_t stream)
{
  if (p)
    (void)hipFreeAsync(p, stream);
}

}  // namespace ggnn_amd
