"""The pair kernels' job schedule (mind_amd/csrc/pair_jobs.h) checked on the host -- no GPU needed.

k_pair / k_pair_bf / k_pair_t walk the job list with a fixed rule (wave w of workgroup b runs jobs[w * grid + b + k * 8 * grid]), so the
ORDER of the list mind_predict_batch builds is the schedule.  mind_debug_pair_schedule returns that list for a batch of scene sizes:
  * every (column, tile range) exactly once, the ranges of a column a partition of its tiles, partial slots unique;
  * jobs per column a function of the scene's size only (a scene's result must not depend on the batch: reference network.py:165-232 is
    evaluated scene by scene), the rule of rounds 2-4 below 256 tokens, at most eight partials per column (k_token combines eight);
  * the wave slots carry the same number of tiles give or take one job, and from eight scenes on an XCD (the workgroups b = x mod 8) works
    on a contiguous run of the scenes: no scene on more than two XCDs unless it is bigger than an XCD's share.
"""
import ctypes as C

import numpy as np
import pytest

from mind_amd import _lib

WAVES = 8


def schedule(tokens, actors, n_cu=256, last_layer=False):
    lib = _lib.load()
    tokens = np.asarray(tokens, np.int32)
    actors = np.asarray(actors, np.int32)
    cap = int(sum(int(n) * 8 for n in tokens))
    out = np.zeros((cap, 6), np.int32)
    info = np.zeros(4, np.int32)
    n = lib.mind_debug_pair_schedule(tokens.ctypes.data_as(C.POINTER(C.c_int)), actors.ctypes.data_as(C.POINTER(C.c_int)), len(tokens), n_cu,
                                     int(last_layer), out.ctypes.data_as(C.POINTER(C.c_int)), cap, info.ctypes.data_as(C.POINTER(C.c_int)))
    assert 0 <= n <= cap, n
    return out[:n], info


def old_rule(N):
    tiles = (N + 15) // 16
    return max(1, min((1024 + N - 1) // N, 8, tiles))


def splits_of(jobs, scene, column):
    rows = jobs[(jobs[:, 0] == scene) & (jobs[:, 1] == column)]
    return rows[np.argsort(rows[:, 2])]


@pytest.mark.parametrize("tokens", [[41], [96, 33, 257], [321] * 24, [321, 96, 1500, 40, 700, 256, 255, 2048, 17]])
def test_every_tile_of_every_column_is_covered_once(tokens):
    actors = [max(1, n // 3) for n in tokens]
    jobs, info = schedule(tokens, actors)
    assert info[3] == len(jobs)
    assert len(set(map(int, jobs[:, 4]))) == len(jobs)                     # partial slots unique
    for b, N in enumerate(tokens):
        tiles = (N + 15) // 16
        per_column = None
        for j in (0, N // 2, N - 1):
            rows = splits_of(jobs, b, j)
            assert rows[0, 2] == 0 and rows[-1, 3] == tiles
            assert np.all(rows[1:, 2] == rows[:-1, 3])                     # a partition, no gaps
            assert np.all(rows[:, 3] > rows[:, 2])
            assert np.all(np.diff(rows[:, 4]) == 1)                        # the column's partials are consecutive slots (TokMeta.slot0, nsplit)
            per_column = len(rows) if per_column is None else per_column
            assert len(rows) == per_column
        assert per_column <= 8 and per_column <= tiles
        assert int((jobs[:, 0] == b).sum()) == N * per_column


def test_jobs_per_column_depend_on_the_scene_alone():
    alone = {N: len(splits_of(schedule([N], [N // 2])[0], 0, 0)) for N in (17, 40, 96, 255, 256, 321, 700, 1500, 2048)}
    mixed, _ = schedule(list(alone), [n // 2 for n in alone])
    for b, N in enumerate(alone):
        assert len(splits_of(mixed, b, 1)) == alone[N], N
        if N < 256:
            assert alone[N] == old_rule(N), N                              # the demo-size goldens rest on the rule of rounds 2-4
        else:
            tiles = (N + 15) // 16
            assert 5 <= tiles / alone[N] <= 9 or alone[N] == 8, (N, alone[N])   # about seven tiles per job, eight partials at most
    assert alone[321] == 3 and alone[2048] == 8


@pytest.mark.parametrize("tokens,last", [([321] * 24, False), ([321] * 24, True), ([300, 321, 340, 280] * 4, False), ([257] * 3, False)])
def test_wave_slots_carry_equal_work_and_scenes_stay_on_their_xcd(tokens, last):
    actors = [64] * len(tokens)
    jobs, info = schedule(tokens, actors, last_layer=last)
    grid = int(info[1])
    slots = grid * WAVES
    if len(jobs) <= slots:
        pytest.skip("at most one job per wave slot: nothing to balance")
    tiles = np.zeros(slots, np.int64)
    count = np.zeros(slots, np.int64)
    np.add.at(tiles, jobs[:, 5], jobs[:, 3] - jobs[:, 2])
    np.add.at(count, jobs[:, 5], 1)
    lanes = 8 if len(tokens) >= 8 and grid % 8 == 0 else 1
    biggest = int((jobs[:, 3] - jobs[:, 2]).max())
    load = tiles + count                                                   # the deal's cost: tiles + one per job
    assert load.max() - load.min() <= 2 * (biggest + 1), (load.min(), load.max())
    assert tiles.max() <= tiles.mean() + biggest + 1                       # (rounds 3-4: the 6-tile jobs of a 5, 5, 5, 6 split sat on a quarter of the slots)
    if lanes == 8:
        xcd = (jobs[:, 5] % grid) % 8                                      # workgroup b runs on XCD b % 8
        share = load.sum() / 8
        for b in range(len(tokens)):
            mine = jobs[:, 0] == b
            cost = int((jobs[mine, 3] - jobs[mine, 2] + 1).sum())
            assert len(set(map(int, xcd[mine]))) <= 2 + cost // int(share), b
        for x in range(8):
            scenes = sorted(set(map(int, jobs[xcd == x][:, 0])))
            assert scenes == list(range(scenes[0], scenes[-1] + 1))       # a contiguous run of scenes


def test_last_layer_list_holds_the_consumed_columns_only():
    tokens, actors = [321, 96], [40, 12]
    jobs, _ = schedule(tokens, actors, last_layer=True)
    for b, (N, a) in enumerate(zip(tokens, actors)):
        cols = set(map(int, jobs[jobs[:, 0] == b][:, 1]))
        assert cols == set(range(a)) | {N - 1}
