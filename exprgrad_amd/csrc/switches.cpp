// The one table of environment switches the library honours (switches.hpp).
#include "switches.hpp"

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "eg_internal.hpp"

namespace {

struct Row {
  const char* name;
  const char* cls;      // execution | data-parallel | compiler | detector | tuning
  const char* purpose;
};

// execution: turn ONE optimisation off (bisecting a wrong result, measuring what it buys; tools/stress_suite.sh runs the
//   GPU suite under rotations of them).  detector: dumps, traces, poison.  tuning: honoured under EG_TUNING=1 only.
const Row kSwitches[] = {
    {"EG_TUNING", "execution", "honour the rows of class `tuning` (measurement aids); unset: they are ignored"},
    {"EG_NO_GRAPH", "execution", "launch one by one instead of replaying captured HIP graphs"},
    {"EG_NO_DPP_BUTTERFLY", "execution", "row groups exchange lanes through the LDS crossbar in every step of a wave reduction (no DPP moves)"},
    {"EG_NO_DEFERRED_FOLD", "execution", "a row group in front of a side-lane group folds its partial rows itself, not on the side lane"},
    {"EG_NO_OVERLAP", "execution", "no side lane: bandwidth-bound launches run in front of the long contraction, not next to it"},
    {"EG_NO_ROWFUSE", "execution", "no row / sample / map / small fusion groups: one launch per kernel"},
    {"EG_NO_EPILOGUE", "execution", "elementwise consumers of a contraction stay their own launches"},
    {"EG_NO_INLINE", "execution", "no producer / consumer inlining of elementwise kernels into generated kernels"},
    {"EG_NO_ALIAS", "execution", "whole-tensor raw copies (reshape) are copied instead of sharing storage"},
    {"EG_NO_ONES_ROW", "execution", "bias gradient as its own column sum instead of the weight gradient's virtual row of ones"},
    {"EG_NO_SMALL_GEMM", "execution", "tiny contractions on the matrix tiles instead of one wave per output element"},
    {"EG_NO_NARROW_INDEX", "execution", "64-bit index arithmetic everywhere in generated kernels"},
    {"EG_NO_PREDICATE", "execution", "pre-activations stored as values, not as predicate bits"},
    {"EG_NO_ROW_PRODUCT", "execution", "the 10-wide forward product as its own launch, not in the previous layer's epilogue"},
    {"EG_NO_ROW_DIRECT", "execution", "a one-block row group writes a partial row for row_finalize instead of its totals"},
    {"EG_NO_SMALL_PAIR", "execution", "two independent tiny contractions as two launches"},
    {"EG_NO_SKINNY_GEMM", "execution", "N <= 16 products on the matrix tiles instead of the streaming skinny kernel"},
    {"EG_NO_NARROW_K", "execution", "K <= 16 products with a generated epilogue on the matrix tile instead of the streaming kernel"},
    {"EG_NO_SAMPLE_FUSE", "execution", "no sample groups (one block per sample): the launch chain of a small-batch step"},
    {"EG_OVERLAP_SIDE_FIRST", "execution", "the side lane's launches are issued in front of the long contraction they run beside (the order of rounds 2 - 5)"},
    {"EG_SAMPLE_KEEP_BARRIERS", "execution", "sample kernels keep the barrier between independent members"},
    {"EG_SAMPLE_NO_STAGE", "execution", "a sample group's members read parameters from global memory, not from a copy in LDS"},
    {"EG_SAMPLE_NO_MFMA", "execution", "convolution members of a sample group as scalar loop nests, not on the matrix cores"},
    {"EG_NO_SLAB_FOLD", "execution", "the optimizer's map group does not add up the sample kernel's slab rows itself"},
    {"EG_NO_SLAB_SUM", "execution", "k-slices folded by the two-launch column sum instead of slab_sum"},
    {"EG_NO_ROW_TAIL", "execution", "a row group's last block neither folds the partial rows nor runs the update"},
    {"EG_PIPELINE", "execution", "the batch pipeline (two half batches, streaming launches under the other half's contraction): OFF unless 1"},
    {"EG_GEMM_NO_SKEW", "execution", "every wave of a contraction block runs the k loop in phase (the round-3 loop)"},
    {"EG_GEMM_NO_BK32", "execution", "16-deep k-tiles for long whole-tile products"},
    {"EG_GEMM_NO_PAIR", "execution", "no wave-pair / eight-wave small-tile kernels"},
    {"EG_GEMM_NO_T96", "execution", "no 96 x 96 whole-round tiles"},
    {"EG_GEMM_NO_STREAMK", "execution", "64 x 64 tiles one block per tile, not persistent stream-K blocks"},
    {"EG_GEMM_NO_XROW", "execution", "1 .. 32 rows beyond whole tiles as a ragged tile row, not a ninth accumulator block"},
    {"EG_GEMM_NO_WIDE_STORE", "execution", "tiles leave as 128-byte pieces instead of through LDS as whole rows"},
    {"EG_CONV_NO_TINY", "execution", "small convolutions on the contraction route"},
    {"EG_CONV_NO_GRADF_HALO", "execution", "filter gradient as one gathered contraction"},
    {"EG_CONV_NO_WIDE_STORE", "execution", "halo convolution stores 128-byte pieces"},
    {"EG_CONV_NO_VIRTUAL_PAD", "execution", "image gradient reads a padded copy of the output gradient"},
    {"EG_CONV_NO_HALO", "execution", "3 x 3-class convolutions on the implicit-GEMM route"},
    {"EG_CONV_NO_DIRECT", "execution", "few-channel convolutions on the implicit-GEMM route"},
    {"EG_CONV_NO_BAND", "execution", "small-channel convolutions on the routes the band kernels replaced"},
    {"EG_NO_STAGED_COPY", "execution", "downloads into pageable memory as one runtime copy"},
    {"EG_FIT_NO_DIRECT", "execution", "fit copies every batch into the inputs' staging buffers"},
    {"EG_DP_REAGREE_STEPS", "data-parallel", "steps between two negotiations of the exchange schedule (default 256; 0: only the first)"},
    {"EG_DP_INIT_TIMEOUT_S", "data-parallel", "watchdog on ncclCommInitRank, seconds (default 180)"},
    {"EG_DP_RESERVE_CUS", "data-parallel", "compute units the tail range leaves to RCCL's kernel (default 8)"},
    {"EG_DP_TEST_AS_MULTI", "data-parallel", "a one-rank group takes the N > 1 code paths (one-GPU boxes)"},
    {"EG_DP_NO_SPLIT", "data-parallel", "groups start with the early / late split of the bucket forbidden (eg_dp_set_split)"},
    {"EG_HIPRTC_LIB", "compiler", "path of the libhiprtc the library opens"},
    {"EG_KERNEL_CACHE", "compiler", "directory of the on-disk code-object cache"},
    {"EG_NO_KERNEL_CACHE", "compiler", "no on-disk code-object cache"},
    {"EG_POISON", "detector", "NaN patterns in every scratch block and every slot that is overwritten, before each run"},
    {"EG_NO_PLAN_CHECK", "detector", "skip the plan invariants (host/plan_check.cpp)"},
    {"EG_DEBUG_GRAPH", "detector", "print graph captures, replays and refusals"},
    {"EG_DEBUG_OVERLAP", "detector", "print the side-lane groups of a plan"},
    {"EG_DEBUG_TILE", "detector", "print the tile model's estimate per candidate"},
    {"EG_DEBUG_SAMPLE", "detector", "print why a sample group did or did not form"},
    {"EG_DUMP_FUSED", "detector", "directory: generated translation units of fused contractions"},
    {"EG_DUMP_CODE", "detector", "directory: hiprtc code objects and the text they were built from"},
    {"EG_DUMP_BAND", "detector", "directory: generated band-convolution sources"},
    {"EG_GRADF_TRACE", "detector", "per-wave cycle stamps of the halo filter-gradient kernel"},
    {"EG_HALO_TRACE", "detector", "per-wave cycle stamps of the halo convolution kernel"},
    {"EG_GEMM_TRACE", "detector", "per-wave cycle stamps of the fused and the extra-row contraction kernels (k loop begins / ends, epilogue done)"},
    {"EG_ROW_TRACE", "detector", "the last block of a row group with a tail prints cycle stamps of its hand-off"},
    {"EG_SAMPLE_TRACE", "detector", "block 0 of a sample kernel prints cycle stamps behind every member's barrier"},
    {"EG_GEMM_FORCE_TILE", "tuning", "bm,bn: force the contraction tile"},
    {"EG_GEMM_FORCE_SPLITS", "tuning", "n: force the k-slice count"},
    {"EG_GEMM_OLD_TILE_MODEL", "tuning", "round-1 cost model for wide outputs"},
    {"EG_STREAMK_BLOCKS_PER_CU", "tuning", "persistent blocks per CU of a stream-K launch"},
    {"EG_STREAMK_MIN_RATIO", "tuning", "microseconds the balance model must promise before a 64 x 64 launch goes stream-K (default 24)"},
    {"EG_GEMM_SMALL_BK32", "tuning", "32-deep k-tiles for every 64 x 64 launch"},
    {"EG_DGEMM_TILE", "tuning", "config[,splits]: force the float64 tile"},
    {"EG_CONV_BAND_PIXELS", "tuning", "pixels per band of the band convolutions"},
    {"EG_CONV_DIRECT_BLOCKS", "tuning", "block cap of the direct filter gradient"},
    {"EG_ROW_TAIL_BLOCKS", "tuning", "blocks of a row group that carries a tail (default: 12 KB of rows per block, at least 64)"},
    {"EG_SAMPLE_STOP", "tuning", "k: a sample kernel ends behind member k (wrong numbers: the time of its first k + 1 members)"},
    {"EG_SAMPLE_FUSE_MAX_BATCH", "tuning", "largest batch that forms a sample group (default 1280)"},
    {"EG_EPILOGUE_MIN_ELEMS", "tuning", "smallest output that gets a generated epilogue (default 2^20; tests: 0)"},
    {"EG_PIPELINE_MIN_FLOPS", "tuning", "smallest contraction the batch pipeline cuts"},
    {"EG_FIT_GROUP", "tuning", "batches per captured graph launch in fit (1: every batch its own launch)"},
    {"EG_FIT_PIECE_BYTES", "tuning", "upload piece of fit"},
    {"EG_FIT_SEGMENT_BYTES", "tuning", "device-resident segment of fit"},
};
constexpr int kCount = (int)(sizeof(kSwitches) / sizeof(kSwitches[0]));

struct Cache {
  std::mutex mu;
  bool loaded = false;
  bool tuning = false;
  std::unordered_map<std::string, int> index;
  std::vector<std::string> value;
  std::vector<char> set;
  void load() {
    if (index.empty())
      for (int i = 0; i < kCount; ++i) index[kSwitches[i].name] = i;
    value.assign(kCount, std::string());
    set.assign(kCount, 0);
    const char* t = getenv("EG_TUNING");
    tuning = t && t[0] && t[0] != '0';
    for (int i = 0; i < kCount; ++i) {
      const char* e = getenv(kSwitches[i].name);
      if (!e) continue;
      if (!tuning && strcmp(kSwitches[i].cls, "tuning") == 0) continue;
      value[i] = e;
      set[i] = 1;
    }
    loaded = true;
  }
};
Cache& cache() {
  static Cache c;
  return c;
}

}  // namespace

namespace eg {
namespace sw {

const char* raw(const char* name) {
  Cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  if (!c.loaded) c.load();
  auto it = c.index.find(name);
  if (it == c.index.end()) {
    fprintf(stderr, "[exprgrad_hip] switch %s is not in the table of csrc/switches.cpp: read as unset\n", name);
    return nullptr;
  }
  return c.set[it->second] ? c.value[it->second].c_str() : nullptr;
}

void reload() {
  Cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  c.load();
}

}  // namespace sw
}  // namespace eg

extern "C" {

int eg_switches_reload(void) {
  eg::sw::reload();
  return EG_OK;
}

// "<name>\t<class>\t<purpose>\n" per switch; returns the length needed (without the terminator), copies at most cap - 1.
int64_t eg_switch_table(char* buf, size_t cap) {
  std::string s;
  for (int i = 0; i < kCount; ++i) s += std::string(kSwitches[i].name) + "\t" + kSwitches[i].cls + "\t" + kSwitches[i].purpose + "\n";
  if (buf && cap > 0) {
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return (int64_t)s.size();
}

}  // extern "C"
