// TEST INFRASTRUCTURE ONLY.  Harness around the REFERENCE's own host helpers:
// /root/reference/include/ggnn/base/def.h (bit_ceil, next_multiple, align8; lines 37-68) is
// #included from where it lies -- it needs nothing but the standard library, so plain g++ compiles
// it unchanged; nothing of it is copied into this repository.  Built by oracle/Makefile into
// oracle/_ref/ (git-ignored, ships to the GPU box), used by tests/test_oracle.py to pin the
// sizing helpers of the oracle and of the engine (query_kernels.cu:55-110 is built on them).
#include <cstddef>
#include <cstdint>

#include <ggnn/base/def.h>

extern "C" {

uint32_t ref_bit_ceil(uint32_t v)
{
  return ggnn::bit_ceil(v);
}
uint32_t ref_next_multiple32(uint32_t v)
{
  return ggnn::next_multiple<uint32_t, 32U>(v);
}
size_t ref_align8(size_t v)
{
  return ggnn::align8(v);
}
int ref_measure_euclidean(void)
{
  return static_cast<int>(ggnn::DistanceMeasure::Euclidean);
}
int ref_measure_cosine(void)
{
  return static_cast<int>(ggnn::DistanceMeasure::Cosine);
}
}
