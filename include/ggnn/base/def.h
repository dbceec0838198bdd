// ggnn C++ facade over the C-ABI (include/ggnn_c.h) -- definitions shared by the headers.
// Mirrors the names of the reference's include/ggnn/base/def.h:27-30 so that its callers compile
// unchanged; the implementation behind is libggnn_amd.so.
#ifndef GGNN_AMD_FACADE_DEF_H
#define GGNN_AMD_FACADE_DEF_H

#include <ggnn_c.h>

#include <stdexcept>
#include <string>

namespace ggnn {

enum class DistanceMeasure : int { Euclidean = GGNN_EUCLIDEAN, Cosine = GGNN_COSINE };

namespace detail {
// status -> the exception the reference throws in the same situation (ggnn.cu:96 out_of_range,
// everything else runtime_error)
inline void check(ggnn_status s, const ggnn_t* h)
{
  if (s == GGNN_OK)
    return;
  const char* m = ggnn_last_error(h);
  const std::string msg = (m && *m) ? m : ("ggnn status " + std::to_string(static_cast<int>(s)));
  if (s == GGNN_OUT_OF_RANGE)
    throw std::out_of_range(msg);
  throw std::runtime_error(msg);
}
}  // namespace detail

}  // namespace ggnn

#endif
