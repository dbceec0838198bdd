// Test and tuning hooks of libggnn_amd.so: process-wide named integers.
//
// Set with ggnn_set_hook(name, value) (include/ggnn_c.h documents every name and its values).
// The environment is consulted ONLY when GGNN_TEST_HOOKS=1 is set (variable GGNN_<NAME>, read at
// every use), so that a production process that merely inherits such a variable is not steered
// by it.  Precedence: ggnn_set_hook > environment (with GGNN_TEST_HOOKS=1) > built-in default.
#pragma once
#include <cstdint>

namespace ggnn_amd {

enum Hook {
  kHookPrescreen = 0,   // PRESCREEN        default of new handles: 1 = exact pre-screen on
  kHookExchange,        // EXCHANGE         0 auto | 1 "rccl" | 2 "copy" | 3 "gather"
  kHookSymPrescreen,    // SYM_PRESCREEN    -1 auto (rows >= 1 KB) | 0 | 1
  kHookShardOverlap,    // SHARD_OVERLAP    1 = resident shards searched concurrently
  kHookVisSlots,        // VIS_SLOTS        usable keys per bucket of the hashed visited set (1..8)
  kHookVisTagSet,       // VIS_TAG_SET      0 = rings of 481..2016 keys scanned instead of the tag set
  kHookQuerySplit,      // QUERY_SPLIT      -1 auto | 0 | 1: blocking multi-GPU query as two half-batches
  kHookResidentShards,  // RESIDENT_SHARDS  0 auto (by free memory) | GPU slots per device for its shards
  kHookXcdMap,          // XCD_MAP          bit 0: merge, bit 1: sym -- XCD-contiguous point ranges
  kHookBfPoolKeepMb,    // BF_POOL_KEEP_MB  release threshold of the bf scratch pool
  kHookBfNoI8,          // BF_NO_I8         1 = uint8 bf_query through the float kernels
  kHookBfI8V1,          // BF_I8_V1         1 = LDS-list i8 kernel instead of the register-set one
  kHookBfSlices,        // BF_SLICES        0 auto | base slices per query block
  kHookBfNoCenter,      // BF_NO_CENTER     1 = rows not shifted by the column mean
  kHookBfTiles,         // BF_TILES         2 | 4 base tiles per accumulator group (D > 128)
  kHookBfI8NoShare,     // BF_I8_NOSHARE    1 = slices do not share their bound
  kHookBfI8Ranks,       // BF_I8_RANKS      bit mask of the set positions the slices exchange (-1 = auto: one, 31 = all)
  kHookBfScan,          // BF_SCAN          1 = scan kernels instead of the matrix-core path
  kHookRcclFailAfter,   // RCCL_FAIL_AFTER  fault injection: the n-th exchange (1-based) reports an
                        //                  RCCL failure (0 = never); exercises the peer-copy fallback
  kHookQueryEarly,      // QUERY_EARLY      0 = the query kernel's round-1..4 order (rows requested after
                        //                  the membership test); 1 = early rows where the layout allows
  kHookMergeEarly,      // MERGE_EARLY      the same switch for the merge kernel
  kHookQueryLdsPad,     // QUERY_LDS_PAD    extra bytes of LDS per wave of the early-rows query kernels
  kHookQueryGlobalRing, // QUERY_GLOBAL_RING 0 = early-rows kernels keep a visited ring in LDS even when it cannot wrap
  kHookBfI8Refresh,     // BF_I8_REFRESH    stages between bound exchanges of the i8 kernel's slices
  kHookBfI8Seed,        // BF_I8_SEED       rows of the i8 kernel's seeding launch (0 = none)
  kHookMergeCounting,   // MERGE_COUNTING   1 = merge launches without counters use the counting kernel too
  kHookCount
};

int64_t hook(Hook h);
const char* hook_name(Hook h);
// -1: unknown name
int hook_by_name(const char* name);
void hook_set(Hook h, int64_t value);
void hook_reset(Hook h);

}  // namespace ggnn_amd
