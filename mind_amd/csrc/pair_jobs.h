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
  const size_t n = jl.size();
  if (grid <= 0 || (int)n <= slots) return;      // at most one job per slot: nothing to balance
  if (lanes < 1 || grid % lanes != 0) lanes = 1;
  // lanes: the list (scene by scene, column by column) cut into `lanes` runs of equal cost -- an XCD then works on its share of the scenes plus
  // at most two scenes it shares with a neighbour, whatever the scenes' sizes
  long long total = 0;
  int max_cost = 0;
  for (const Job &J : jl)
    if (J.t1 > J.t0) {
      const int c = J.t1 - J.t0 + PAIR_JOB_OVERHEAD_TILES;
      total += c;
      max_cost = std::max(max_cost, c);
    }
  // One pass per lane over its run of the list: the jobs by descending cost (a counting sort -- costs are a few tiles -- that keeps the list's
  // order among equals), each to the least loaded slot of the lane.  Only (slot, position in the slot) are kept per job; every record is copied
  // once, into its place (a stress round deals twelve million jobs).
  std::vector<int> slot_of(n, -1), pos_of(n, 0), count(slots, 0);
  std::vector<size_t> lane_begin(lanes + 1, n);
  {
    long long acc = 0;
    int x_prev = -1;
    for (size_t i = 0; i < n; ++i) {
      const Job &J = jl[i];
      if (J.t1 <= J.t0) continue;
      int x = (int)(acc * lanes / (total > 0 ? total : 1));
      x = x < lanes ? x : lanes - 1;
      for (; x_prev < x; ++x_prev) lane_begin[x_prev + 1] = i;
      acc += J.t1 - J.t0 + PAIR_JOB_OVERHEAD_TILES;
    }
    for (; x_prev < lanes; ++x_prev) lane_begin[x_prev + 1] = n;
  }
  std::vector<size_t> order;
  std::vector<size_t> bucket(max_cost + 2);
  for (int x = 0; x < lanes; ++x) {
    const size_t b0 = lane_begin[x], b1 = lane_begin[x + 1];
    std::fill(bucket.begin(), bucket.end(), 0);
    for (size_t i = b0; i < b1; ++i)
      if (jl[i].t1 > jl[i].t0) ++bucket[max_cost - (jl[i].t1 - jl[i].t0 + PAIR_JOB_OVERHEAD_TILES) + 1];
    for (int c = 0; c <= max_cost; ++c) bucket[c + 1] += bucket[c];
    order.assign(bucket[max_cost + 1], 0);
    for (size_t i = b0; i < b1; ++i)
      if (jl[i].t1 > jl[i].t0) order[bucket[max_cost - (jl[i].t1 - jl[i].t0 + PAIR_JOB_OVERHEAD_TILES)]++] = i;
    // the least loaded slot first: the loads of a lane's slots never differ by more than one job's cost, so a ring of max_cost + 2 queues
    // indexed by load serves as the priority queue (first in, first out among equals)
    const int R = max_cost + 2;
    std::vector<std::vector<int>> ring(R);
    std::vector<size_t> head(R, 0);
    for (int w = 0; w < waves; ++w)
      for (int b = x; b < grid; b += lanes) ring[0].push_back(w * grid + b);
    long long lo = 0;
    for (size_t i : order) {
      while (head[lo % R] == ring[lo % R].size()) {
        ring[lo % R].clear();
        head[lo % R] = 0;
        ++lo;
      }
      const int sl = ring[lo % R][head[lo % R]++];
      slot_of[i] = sl;
      pos_of[i] = count[sl]++;
      ring[(lo + jl[i].t1 - jl[i].t0 + PAIR_JOB_OVERHEAD_TILES) % R].push_back(sl);
    }
  }
  int rounds = 0;
  for (int c : count) rounds = std::max(rounds, c);
  Job nullj;
  memset(&nullj, 0, sizeof(nullj));
  nullj.N = 1;
  std::vector<Job> out((size_t)rounds * slots, nullj);
  for (size_t i = 0; i < n; ++i)
    if (slot_of[i] >= 0) out[(size_t)pos_of[i] * slots + slot_of[i]] = jl[i];
  jl.swap(out);
}
