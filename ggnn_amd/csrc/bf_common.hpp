// Shared between bf_mfma.hip (f32 / LDS-list kernels, host side) and bf_i8.hip (the register-set
// uint8 kernel, compiled with the VGPR form of the MFMA instructions).
#pragma once
#include "traversal.hpp"

namespace ggnn_amd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBfQueriesPerBlock = 128;
constexpr int kBfTileRows = 32;
constexpr uint32_t kBfMaxKP = 120;  // lists of 128 queries must fit into LDS next to the tiles

struct BfMfmaArgs {
  const void* base;
  const void* query;
  const float* query_packed;  // chunked float kernel: operand-order copy (bf_mfma.hip QueryWindow)
  const float* mean;   // [D] shift applied to base and query rows (float32 squared L2), or null
  const float* bnorm;
  const float* qnorm;
  int32_t* part_ids;   // [slices][Nq][KP]
  float* part_dists;   // [slices][Nq][KP]
  uint32_t D, Dh, DP, Nq, N_base, KP, slices, rows_per_slice;
  uint32_t DM;  // floats of the shift vector kept in LDS by the chunked kernel (D rounded up)
  uint32_t* gthr;  // i8 v2 kernel: exchange area of the slices' published set entries
                   // ([Nq][5 ranks][slices padded to 4] ints, 0x7fffffff = none), or null
  uint32_t rank_mask;  // which of the published positions are used (bit i = rank i)
  uint32_t refresh_every;  // stages between exchanges after the doubling phase
  int32_t* seed;       // [Nq] K-th best distance over the head of the base (seeding launch), or null
  uint32_t seeding;    // this launch IS the seeding launch: writes `seed`, no lists
  // float tile kernels, equal_ranges != 0: the (query block, unit) sequence is cut into equal
  // ranges, a unit being 32 rows (single chunk) or one accumulator group of T x 32 rows (chunked)
  uint32_t equal_ranges;
  uint32_t tiles_per_q;      // units per query block
  uint32_t tiles_per_block;  // range of one workgroup
  uint64_t total_tiles;      // query blocks * tiles_per_q
};

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kBfI8RowStride = 144;

// bf_i8.hip: launches bf_i8v2_kernel for KP in {4, 10, 16} (a seeding launch over the first
// seed_rows rows of the base first, when m.seed is set and there are several slices); m.gthr must
// point to bf_i8v2_exchange_ints(Nq, slices) words initialised to 0x7fffffff (or be null: no exchange)
void launch_bf_i8v2(const BfMfmaArgs& m, uint32_t qblocks, uint32_t slices, uint32_t seed_rows,
                    hipStream_t stream);
size_t bf_i8v2_lds_bytes();
size_t bf_i8v2_exchange_ints(uint32_t Nq, uint32_t slices);
uint32_t bf_i8v2_default_rank_mask(uint32_t KP, uint32_t slices);

}  // namespace ggnn_amd
