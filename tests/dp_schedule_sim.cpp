// CPU simulation of the data-parallel exchange schedule (exprgrad_amd/csrc/host/dp_schedule.hpp — the header the
// library compiles, unchanged): W ranks as threads, a transport in which every collective is a rendezvous of all ranks
// that records what each rank asked for.  On a real communicator two ranks issuing different collectives in the same
// slot is a hang or silently mixed data; here the rendezvous flags it (and a rank that never arrives times out).
//
//   g++ -std=c++17 -O1 -pthread tests/dp_schedule_sim.cpp -o dp_schedule_sim && ./dp_schedule_sim
//
// Scenarios (tests/test_dp_schedule_cpu.py runs them and reads the last line):
//   steady        every rank proposes the same cut: negotiated once, overlapped afterwards
//   early_replan  rank 1 arrives with another plan (another shard shape) ONE STEP BEFORE rank 0 — the first-contact bug
//                 of VERDICT r4 weak #10: no slot may mismatch, the late rank serves the agreed cut sequentially, and at
//                 the next counted negotiation the ranks settle on the new common cut
//   lone_replan   one rank keeps another plan for good: the others keep the agreed calls, the periodic negotiation turns
//                 the cut off for everybody (whole bucket)
//   toggle        every rank switches the split setting in the same step (eg_dp_set_split): renegotiated at once
//   buckets       ranks with buckets of different sizes: every rank gets the same clean error, nobody hangs
//   legacy        the round-4 policy (agreement per plan) under early_replan: the simulation must SEE the mismatch,
//                 which shows that the transport would have caught the bug the new policy removes
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../exprgrad_amd/csrc/host/dp_schedule.hpp"

using namespace eg::dp;

struct Op {
  int kind;  // 0 float SUM all-reduce, 1 int64 MAX comparison
  long count;
};

struct Fabric {
  int world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  std::vector<Op> slot;
  std::vector<std::vector<int64_t>> payload;
  std::vector<int64_t> reduced;
  bool mismatch = false, timed_out = false;
  long collectives = 0, comparisons = 0;
  explicit Fabric(int w) : world(w), slot(w), payload(w) {}

  // every rank calls this for every collective; returns false when the slot was inconsistent or a rank never came
  bool meet(int rank, Op op, const int64_t* values, std::vector<int64_t>* out) {
    std::unique_lock<std::mutex> lock(mu);
    slot[rank] = op;
    payload[rank].assign(values ? values : nullptr, values ? values + op.count : nullptr);
    const long gen = generation;
    if (++arrived == world) {
      bool same = true;
      for (int r = 1; r < world; ++r) same = same && slot[r].kind == slot[0].kind && slot[r].count == slot[0].count;
      if (!same) mismatch = true;
      ++collectives;
      if (slot[0].kind == 1) ++comparisons;
      reduced.clear();
      if (same && slot[0].kind == 1) {
        reduced = payload[0];
        for (int r = 1; r < world; ++r)
          for (size_t i = 0; i < reduced.size(); ++i) reduced[i] = std::max(reduced[i], payload[r][i]);
      }
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else if (!cv.wait_for(lock, std::chrono::seconds(5), [&] { return generation != gen; })) {
      timed_out = true;
      return false;
    }
    if (out) *out = reduced;
    return !mismatch;
  }
};

struct RankCtx {
  Fabric* fabric;
  int rank;
};

static int agree_fn(void* user, const int64_t* values, int n, int* same) {
  RankCtx* c = static_cast<RankCtx*>(user);
  std::vector<int64_t> both(2 * n), red;
  for (int i = 0; i < n; ++i) {
    both[i] = values[i];
    both[n + i] = -values[i];
  }
  if (!c->fabric->meet(c->rank, Op{1, 2L * n}, both.data(), &red)) return 7;
  *same = 1;
  for (int i = 0; i < n; ++i)
    if (red.size() != (size_t)(2 * n) || red[i] != -red[n + i]) *same = 0;
  return 0;
}

static bool allreduce(RankCtx* c, const Segs& segs) {
  for (auto& s : segs)
    if (s.second > 0 && !c->fabric->meet(c->rank, Op{0, s.second}, nullptr, nullptr)) return false;
  return true;
}

static Proposal cut(long bucket, long early) {
  Proposal p;
  p.bucket_floats = bucket;
  if (early > 0) {
    p.split = true;
    p.early = {{bucket - early, early}};
    p.late = {{0, bucket - early}};
  } else {
    p.late = {{0, bucket}};
  }
  return p;
}

struct Result {
  bool ok = true;
  std::string note;
  std::vector<std::string> how;  // per rank: one letter per step (O overlapped, S sequential, W whole, E error)
};

// proposal(rank, step), split_allowed(step)
static uint64_t one_group(int, int) { return 1; }

// proposal(rank, step), split_allowed(step), key(rank, step) = the key under which this rank looks up the group's schedule in
// its ScheduleTable (a new key = the host destroyed the group and created another)
template <class P, class A, class K = uint64_t (*)(int, int)>
static Result run(int world, int steps, long reagree, P proposal, A allowed, bool legacy = false, K key = one_group) {
  Fabric fabric(world);
  Result res;
  res.how.assign(world, "");
  std::vector<std::thread> threads;
  std::vector<int> errors(world, 0);
  for (int r = 0; r < world; ++r)
    threads.emplace_back([&, r] {
      RankCtx ctx{&fabric, r};
      ScheduleTable table;
      // legacy: one agreement per PLAN (a plan = a distinct proposal), as rounds 2-4 did
      std::vector<std::pair<Proposal, Schedule>> per_plan;
      for (int step = 0; step < steps; ++step) {
        const Proposal mine = proposal(r, step);
        Schedule* s = &table.of(key(r, step));
        if (legacy) {
          s = nullptr;
          for (auto& pp : per_plan)
            if (pp.first.early == mine.early && pp.first.late == mine.late && pp.first.bucket_floats == mine.bucket_floats) s = &pp.second;
          if (!s) {
            per_plan.push_back({mine, Schedule()});
            s = &per_plan.back().second;
          }
        }
        How how;
        std::string why;
        int rc = step_decision(*s, mine, allowed(step), agree_fn, &ctx, legacy ? 0 : reagree, &how, &why);
        if (rc) {
          errors[r] = rc;
          res.how[r] += 'E';
          break;  // a clean error: the host stops stepping
        }
        res.how[r] += how == How::Overlapped ? 'O' : how == How::Sequential ? 'S' : 'W';
        bool fine = true;
        if (how == How::Whole) {
          fine = allreduce(&ctx, {{0, mine.bucket_floats}});
        } else {
          fine = allreduce(&ctx, s->early) && allreduce(&ctx, s->late);
        }
        if (!fine) break;
      }
    });
  for (auto& t : threads) t.join();
  res.ok = !fabric.mismatch && !fabric.timed_out;
  char buf[256];
  snprintf(buf, sizeof buf, "collectives %ld comparisons %ld mismatch %d timeout %d errors", fabric.collectives, fabric.comparisons,
           (int)fabric.mismatch, (int)fabric.timed_out);
  res.note = buf;
  for (int e : errors) res.note += " " + std::to_string(e);
  return res;
}

int main() {
  bool all = true;
  auto report = [&](const char* name, const Result& r, bool want_ok, bool extra) {
    const bool pass = r.ok == want_ok && extra;
    all = all && pass;
    printf("%s %s: %s |", pass ? "PASS" : "FAIL", name, r.note.c_str());
    for (auto& h : r.how) printf(" %s", h.c_str());
    printf("\n");
  };
  const long B = 407050;
  {  // steady
    Result r = run(4, 20, 8, [&](int, int) { return cut(B, 5130); }, [](int) { return true; });
    bool shape = true;
    for (auto& h : r.how) shape = shape && h == std::string(20, 'O');
    report("steady", r, true, shape);
  }
  {  // early_replan: rank 1 changes its shard (no overlap group any more) at step 5, the others at step 6
    Result r = run(2, 20, 8, [&](int rank, int step) { return step >= (rank == 1 ? 5 : 6) ? cut(B, 0) : cut(B, 5130); }, [](int) { return true; });
    // rank 1 serves the agreed cut sequentially from step 5, rank 0 from step 6, both go to the whole bucket at step 8
    const bool shape = r.how[0] == "OOOOOOSSWWWWWWWWWWWW" && r.how[1] == "OOOOOSSSWWWWWWWWWWWW";
    report("early_replan", r, true, shape);
  }
  {  // early_replan towards ANOTHER cut: renegotiated to the new common cut at the counted step
    Result r = run(2, 20, 8, [&](int rank, int step) { return step >= (rank == 1 ? 5 : 6) ? cut(B, 1024) : cut(B, 5130); }, [](int) { return true; });
    const bool shape = r.how[0] == "OOOOOOSSOOOOOOOOOOOO" && r.how[1] == "OOOOOSSSOOOOOOOOOOOO";
    report("early_replan_new_cut", r, true, shape);
  }
  {  // lone_replan
    Result r = run(4, 20, 8, [&](int rank, int step) { return rank == 2 && step >= 3 ? cut(B, 0) : cut(B, 5130); }, [](int) { return true; });
    const bool shape = r.how[0] == "OOOOOOOOWWWWWWWWWWWW" && r.how[2] == "OOOSSSSSWWWWWWWWWWWW";
    report("lone_replan", r, true, shape);
  }
  {  // toggle: split forbidden from step 4, allowed again from step 9
    Result r = run(2, 14, 0, [&](int, int) { return cut(B, 5130); }, [](int step) { return step < 4 || step >= 9; });
    const bool shape = r.how[0] == "OOOOWWWWWOOOOO" && r.how[1] == r.how[0];
    report("toggle", r, true, shape);
  }
  {  // buckets of different sizes: a clean error on every rank in the first step
    Result r = run(2, 5, 8, [&](int rank, int) { return cut(rank == 0 ? B : B + 4, 0); }, [](int) { return true; });
    const bool shape = r.how[0] == "E" && r.how[1] == "E";
    report("buckets", r, true, shape);
  }
  {  // the round-4 policy under early_replan: must be caught by the transport
    Result r = run(2, 20, 0, [&](int rank, int step) { return step >= (rank == 1 ? 5 : 6) ? cut(B, 0) : cut(B, 5130); }, [](int) { return true; }, true);
    report("legacy_policy_is_caught", r, false, true);
  }
  {  // regroup: at step 6 every rank destroys the group and creates a new one (another unique id); keyed by the
     // identity derived from the id, every rank starts the new group's schedule from state 0 in the same step
    Result r = run(2, 12, 0, [&](int, int) { return cut(B, 5130); }, [](int) { return true; }, false,
                   [](int, int step) { return group_identity(step < 6 ? "first unique id" : "second unique id", 15); });
    const bool shape = r.how[0] == std::string(12, 'O') && r.how[1] == r.how[0];
    report("regroup", r, true, shape);
  }
  {  // the round-5 key (the group object's ADDRESS): rank 0's allocator hands the freed address to the new group, rank
     // 1's does not — rank 0 replays the old schedule's float SUM calls while rank 1 negotiates: must be caught
    Result r = run(2, 12, 0, [&](int, int) { return cut(B, 5130); }, [](int) { return true; }, false,
                   [](int rank, int step) { return (uint64_t)(rank == 0 ? 0x7f00aa00 : (step < 6 ? 0x7f00aa00 : 0x7f00bb40)); });
    report("regroup_by_address_is_caught", r, false, true);
  }
  printf("%s\n", all ? "ALL PASS" : "SOME FAILED");
  return all ? 0 : 1;
}
