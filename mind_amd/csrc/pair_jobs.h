// Host side of the pair kernels' job lists (k_pair, k_pair_bf, k_pair_t): which wave runs which column job.
//
// The kernels walk the list with a fixed rule -- wave w of workgroup b runs jobs[w * grid + b + k * 8 * grid], k = 0, 1, ... -- so the ORDER of the
// list is the schedule.  pair_jobs_deal() orders it such that
//   * every wave slot gets the same amount of work: a job costs its tiles plus a constant (job set-up, column partial), jobs are dealt longest first
//     to the least loaded slot.  (The plain order gave slot s every 256th job of its lane: with 4 splits of a 21-tile column -- 5, 5, 5, 6 tiles --
//     a quarter of the waves ran 6-tile jobs only and the launch waited for them: profiles/r05y_pair_bench_splits.txt, 3 splits 9 % faster than 4);
//   * with `lanes` == 8 the list -- scene by scene -- is cut into eight runs of equal cost and run x goes to the workgroups b = x (mod 8), i.e. to
//     one XCD, whose L2 then keeps the T rows and folded queries of its few scenes (round 3: they missed the per-XCD L2 40 % of the time with
//     every scene's jobs spread over all XCDs).
// Slots that run out get empty jobs (t0 == t1, skipped by the kernels).  The order does not touch a job's own arithmetic: results are bit-identical
// whatever the deal.
#pragma once
#include <algorithm>
#include <cstring>
#include <queue>
#include <vector>

#define PAIR_JOB_OVERHEAD_TILES 1      // set-up + column partial of a job, in tiles (pair_bench: 4 against 3 jobs per column, ~7 us per job; a tile ~9 us)

// Jobs per column (i-tile ranges with a softmax partial each).  A function of the scene's own size only: a scene's result is then bit-identical
// whatever batch it is collated into (the AIME rounds sharded over GPUs rely on it).  Small scenes: as many as fill the device from one scene's
// columns (down to one tile per job).  From 256 tokens on: about seven tiles per job -- pair_bench, 24 x N = 321 (21 tiles per column), balanced
// deal: 2 / 3 / 4 / 5 jobs per column 0.680 / 0.655 / 0.692 / 0.689 ms (profiles/r05y_pair_bench_deal.txt); the older rule, 1024 / N, took 4 there
// and a single job per column from 1024 tokens on (half the waves idle on one such scene).
static inline int pair_column_splits(int N) {
  const int tiles = (N + 15) / 16;
  int ns;
  if (N < 256) {
    ns = (1024 + N - 1) / N;
    ns = ns > 8 ? 8 : ns;
  } else {
    ns = (tiles + 3) / 7;
    ns = ns > 8 ? 8 : ns;      // (k_token combines at most eight partials per column)
  }
  ns = ns < 1 ? 1 : ns;
  return ns > tiles ? tiles : ns;
}

// tile range [t0, t1) of a column's job s of ns
static inline void pair_job_range(int tiles, int ns, int s, int *t0, int *t1) {
  *t0 = (int)((long long)tiles * s / ns);
  *t1 = (int)((long long)tiles * (s + 1) / ns);
}

template <class Job>
static void pair_jobs_deal(std::vector<Job> &jl, int grid, int waves, int lanes) {
  const int slots = grid * waves;
  if (grid <= 0 || (int)jl.size() <= slots) return;      // at most one job per slot: nothing to balance
  if (lanes < 1 || grid % lanes != 0) lanes = 1;
  // lanes: the list (scene by scene, column by column) cut into `lanes` runs of equal cost -- an XCD then works on its share of the scenes plus
  // at most two scenes it shares with a neighbour, whatever the scenes' sizes
  long long total = 0;
  for (const Job &J : jl)
    if (J.t1 > J.t0) total += J.t1 - J.t0 + PAIR_JOB_OVERHEAD_TILES;
  std::vector<std::vector<const Job *>> lane_jobs(lanes);
  {
    long long acc = 0;
    for (const Job &J : jl) {
      if (J.t1 <= J.t0) continue;
      int x = (int)(acc * lanes / (total > 0 ? total : 1));
      lane_jobs[x < lanes ? x : lanes - 1].push_back(&J);
      acc += J.t1 - J.t0 + PAIR_JOB_OVERHEAD_TILES;
    }
  }
  std::vector<std::vector<Job>> per_slot(slots);
  for (int x = 0; x < lanes; ++x) {
    std::vector<const Job *> &mine = lane_jobs[x];
    std::stable_sort(mine.begin(), mine.end(), [](const Job *a, const Job *b) { return a->t1 - a->t0 > b->t1 - b->t0; });
    typedef std::pair<long long, int> Load;      // (cost so far, slot): the least loaded slot first, ties by slot number
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int w = 0; w < waves; ++w)
      for (int b = x; b < grid; b += lanes) heap.push(Load(0, w * grid + b));
    for (const Job *J : mine) {
      Load l = heap.top();
      heap.pop();
      per_slot[l.second].push_back(*J);
      l.first += J->t1 - J->t0 + PAIR_JOB_OVERHEAD_TILES;
      heap.push(l);
    }
  }
  size_t rounds = 0;
  for (const auto &s : per_slot) rounds = std::max(rounds, s.size());
  Job nullj;
  memset(&nullj, 0, sizeof(nullj));
  nullj.N = 1;
  std::vector<Job> out(rounds * (size_t)slots, nullj);
  for (int s = 0; s < slots; ++s)
    for (size_t k = 0; k < per_slot[s].size(); ++k) out[k * (size_t)slots + s] = per_slot[s][k];
  jl.swap(out);
}
