// Exchange schedule of the data-parallel step: WHICH all-reduce calls a rank issues per step, decided so that every
// rank of the group issues the same sequence whatever its own launch plan looks like.
//
// The reference has no multi-device code; north_star asks for an RCCL all-reduce of the parameter
// gradients before the gradientDescent kernels (base.nim:37-38).  Rounds 2-4 cut the bucket where THIS rank's plan
// allowed it (early gradients under the last long contraction) and compared the cut across the ranks once per PLAN.
// A plan is a per-rank object: one rank arriving with a new plan (another shard shape, one step before its peers)
// issued the comparison collective (int64 MAX) while the others replayed their agreed float SUM calls — mismatched
// collectives on one communicator, i.e. a hang.
//
// Now the schedule belongs to the (group, target), not to a plan:
//   * it is negotiated at points every rank reaches in the same step BY COUNTING STEPS — the first data-parallel step of
//     the target, the first step after eg_dp_set_split changed the setting (a call every rank makes together), and
//     every `reagree_every` steps after that — never because of something only one rank can see;
//   * between two negotiations every rank issues exactly the agreed calls.  A rank whose current plan proposes the
//     same cut runs it overlapped (early pieces on the side lane under the last contraction); a rank whose plan
//     proposes anything else — a new shard shape, a missing overlap group — runs its whole backward range first and
//     then issues the SAME calls in the same order.  Byte ranges of the bucket are a property of the model, so any
//     plan can serve any agreed cut;
//   * ranks whose proposals differ at a negotiation settle on one call over the whole bucket; ranks whose buckets differ
//     in size are different models: every rank gets the same error.
//
// Pure host logic, no HIP: tests/dp_schedule_sim.cpp drives it on the CPU with W simulated ranks and a transport that
// flags any step in which two ranks issue different collectives.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace eg {
namespace dp {

using Segs = std::vector<std::pair<long, long>>;  // (float offset, float count) runs of the gradient bucket

// What this rank's current plan could do best.
struct Proposal {
  long bucket_floats = 0;
  bool split = false;  // early / late cut available (an overlap group in front of the last long contraction)
  Segs early, late;    // split: both non-empty; otherwise late = the whole bucket
};

struct Schedule {
  int state = 0;       // 0 not negotiated, 1 split, 2 whole bucket
  int agreed_for = 0;  // split setting the negotiation was made under: 1 allowed, 2 forbidden
  long bucket_floats = 0;
  Segs early, late;
  long steps = 0;         // data-parallel steps of this target so far (the same number on every rank)
  long negotiations = 0;  // comparison rounds so far (the same number on every rank)
};

enum class How {
  Overlapped,  // agreed cut == this plan's cut: early pieces on the side lane under the last contraction
  Sequential,  // agreed cut, but this plan cannot overlap it: backward range, then early pieces, then late pieces
  Whole        // one call over the whole bucket behind the backward range
};

// Do all ranks hold the same n numbers?  One small collective + a host read.
using AgreeFn = int (*)(void* user, const int64_t* values, int n, int* same);

inline void fingerprint(const Proposal& p, bool split, int64_t out[6]) {
  uint64_t h1 = 1469598103934665603ull, h2 = 0x9e3779b97f4a7c15ull;
  auto mix = [&](int64_t v) {
    for (int b = 0; b < 8; ++b) {
      const unsigned char c = (unsigned char)((uint64_t)v >> (8 * b));
      h1 = (h1 ^ c) * 1099511628211ull;
      h2 = (h2 + c + 0x9e3779b97f4a7c15ull) * 0xbf58476d1ce4e5b9ull;
      h2 ^= h2 >> 29;
    }
  };
  if (split) {
    for (auto& s : p.early) { mix(s.first); mix(s.second); }
    mix(-1);
    for (auto& s : p.late) { mix(s.first); mix(s.second); }
  }
  out[0] = (int64_t)p.bucket_floats;
  out[1] = split ? 1 : 0;
  out[2] = split ? (int64_t)p.early.size() : 0;
  out[3] = split ? (int64_t)p.late.size() : 0;
  out[4] = (int64_t)(h1 >> 1);
  out[5] = (int64_t)(h2 >> 1);
}

// Is a negotiation due in this step?  Depends only on quantities every rank holds identically.
inline bool negotiation_due(const Schedule& s, bool split_allowed, long reagree_every) {
  const int want_for = split_allowed ? 1 : 2;
  if (s.state == 0 || s.agreed_for != want_for) return true;
  return reagree_every > 0 && s.steps > 0 && s.steps % reagree_every == 0;
}

// One step's decision.  `agree` NULL: a one-rank group, the proposal is the schedule.
// Returns 0, or -1 with *error set (bucket sizes differ across the ranks; a failing transport passes its code on).
inline int step_decision(Schedule& s, const Proposal& mine, bool split_allowed, AgreeFn agree, void* user,
                         long reagree_every, How* how, std::string* error) {
  const bool propose_split = split_allowed && mine.split && !mine.early.empty() && !mine.late.empty();
  if (!agree) {
    s.state = propose_split ? 1 : 2;
    s.agreed_for = split_allowed ? 1 : 2;
    s.bucket_floats = mine.bucket_floats;
    s.early = propose_split ? mine.early : Segs();
    s.late = propose_split ? mine.late : Segs();
    s.steps++;
    *how = propose_split ? How::Overlapped : How::Whole;
    return 0;
  }
  if (negotiation_due(s, split_allowed, reagree_every)) {
    int64_t finger[6];
    fingerprint(mine, propose_split, finger);
    int same = 0;
    int rc = agree(user, finger, 6, &same);
    if (rc) return rc;
    if (!same) {
      int same_bucket = 0;
      rc = agree(user, finger, 1, &same_bucket);
      if (rc) return rc;
      if (!same_bucket) {
        if (error) *error = "data-parallel step: the ranks hold gradient buckets of different sizes (different models or targets)";
        return -1;
      }
    }
    s.negotiations++;
    s.agreed_for = split_allowed ? 1 : 2;
    s.bucket_floats = mine.bucket_floats;
    if (same && propose_split) {
      s.state = 1;
      s.early = mine.early;
      s.late = mine.late;
    } else {
      s.state = 2;
      s.early.clear();
      s.late.clear();
    }
  }
  s.steps++;
  if (mine.bucket_floats != s.bucket_floats) {
    // (cannot happen for one model: the bucket is laid out at compile time; a host that swaps models under one group)
    if (error) *error = "data-parallel step: the gradient bucket changed size since the exchange schedule was agreed";
    return -1;
  }
  if (s.state == 2) {
    *how = How::Whole;
  } else {
    *how = (propose_split && mine.early == s.early && mine.late == s.late) ? How::Overlapped : How::Sequential;
  }
  return 0;
}

// The schedules of one target, one per data-parallel GROUP.  The key is an identity every rank of the group derives
// identically — a hash of the communicator's 128-byte unique id (group_identity) — never the address of the rank's
// group object: an allocator may hand the address of a destroyed group to its successor on SOME ranks only, and those
// ranks would find the old group's Schedule (state != 0: no negotiation due, float SUM calls) while the others start from
// state 0 with the int64 comparison collective — mismatched collectives on one communicator (ADVICE r5;
// tests/dp_schedule_sim.cpp "regroup" / "regroup_by_address_is_caught").
struct ScheduleTable {
  std::map<uint64_t, Schedule> by_group;
  Schedule& of(uint64_t group) { return by_group[group]; }
  void forget(uint64_t group) { by_group.erase(group); }
};

inline uint64_t group_identity(const void* unique_id, size_t bytes) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < bytes; ++i) h = (h ^ static_cast<const unsigned char*>(unique_id)[i]) * 1099511628211ull;
  return h ? h : 1;  // 0 is "no identity"
}

}  // namespace dp
}  // namespace eg
