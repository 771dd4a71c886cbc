"""f3: AV2 scene readers and the scene I/O either side of the planner (CPU).

Two layers, two kinds of evidence:
  * mind_amd.scene_io restates the reference's OWN scene logic (semantic lanes, track selection / padding /
    resampling, target lane).  tests/golden/scene_io.npz holds what the reference's code produced on its four
    demo scenes (tests/golden/gen_golden.py scenes) -> exact comparison.
  * mind_amd.av2_lite restates the two av2 readers underneath (av2 is not importable here): UNPINNED, checked by
    geometry invariants and, in the build container, against the raw files.
"""
import glob
import json
import os

import numpy as np
import pytest

from mind_amd import av2_lite, scene_io
from mind_amd.closed_loop import ClosedLoopSim
from oracle import ref_harness as rh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "scene_io.npz"))
DEMOS = sorted(scene_io.DEMO_SCENES)


@pytest.fixture(scope="module")
def scenes():
    return {n: av2_lite.load_scene(scene_io.scene_fixture_path(n)) for n in DEMOS}


@pytest.fixture(scope="module")
def worlds(scenes):
    return {n: scene_io.ReplayWorld(scenes[n][0], scenes[n][1], json.loads(scenes[n][2]["cl_agent"])) for n in DEMOS}


# ---------------------------------------------------------------- av2 layer: invariants
def _dist_to_polyline(p, line):
    a, b = line[:-1], line[1:]
    d = b - a
    t = np.clip(((p - a) * d).sum(1) / np.maximum((d * d).sum(1), 1e-18), 0, 1)
    return np.min(np.linalg.norm(a + t[:, None] * d - p, axis=1))


def test_interp_arc_known_values():
    # an L-shaped polyline of length 3 resampled to 4 points: corners at arc 0,1,2,3
    pts = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [1.0, 2.0, 0.0]])
    out = av2_lite.interp_arc(4, pts)
    assert np.allclose(out, [[0, 0, 0], [1, 0, 0], [1, 1, 0], [1, 2, 0]], atol=1e-12)
    # straight line: exactly equidistant, endpoints preserved, z carried along
    pts = np.array([[2.0, 1.0, 5.0], [4.0, 1.0, 7.0], [12.0, 1.0, 15.0]])
    out = av2_lite.interp_arc(10, pts)
    assert np.allclose(out[0], pts[0]) and np.allclose(out[-1], pts[-1])
    assert np.allclose(np.diff(out, axis=0), np.diff(out, axis=0)[0], atol=1e-12)
    # repeated vertex does not produce NaN
    pts = np.array([[0.0, 0, 0], [1.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0]])
    assert np.all(np.isfinite(av2_lite.interp_arc(7, pts)))


@pytest.mark.parametrize("name", DEMOS)
def test_centerline_invariants(scenes, name):
    smap = scenes[name][0]
    segs = smap.vector_lane_segments
    assert len(segs) == {"demo_1": 55, "demo_2": 32, "demo_3": 18, "demo_4": 48}[name]      # data/<uuid>
    for lid, ls in segs.items():
        c = smap.get_lane_segment_centerline(lid)
        L, R = ls.left_lane_boundary.xyz, ls.right_lane_boundary.xyz
        assert c.shape == (10, 3)
        assert np.allclose(c[0], (L[0] + R[0]) / 2) and np.allclose(c[-1], (L[-1] + R[-1]) / 2)
        # every centerline point is the midpoint of two points that lie ON the boundaries at equal arc fractions
        for line, other in ((L, R), (R, L)):
            on = 2 * c - av2_lite.interp_arc(10, other)
            assert max(_dist_to_polyline(p, line) for p in on) < 1e-9
            seg = np.linalg.norm(np.diff(av2_lite.interp_arc(10, line), axis=0), axis=1)
            if len(line) == 2:
                assert np.allclose(seg, seg[0], rtol=1e-9)
        # the file's own sparse centerline runs within 0.35 m (xy) of the computed one; widths are lane-like
        rec = ls.recorded_centerline.xyz[:, :2]
        assert max(_dist_to_polyline(p, rec) for p in c[:, :2]) < 0.35
        w = av2_lite.compute_midpoint_line(L, R)[1]
        assert 1.0 < w.min() and w.max() < 12.0
        # successors start where this segment ends
        for s in ls.successors:
            if s in segs:
                assert np.linalg.norm(smap.get_lane_segment_centerline(s)[0, :2] - c[-1, :2]) < 0.5


@pytest.mark.parametrize("name", DEMOS)
def test_scenario_tables(scenes, name):
    sc = scenes[name][1]
    assert len(sc.tracks) == {"demo_1": 100, "demo_2": 110, "demo_3": 43, "demo_4": 70}[name]
    ids = [t.track_id for t in sc.tracks]
    assert ids == sorted(ids) and "AV" in ids and sc.focal_track_id in ids
    for t in sc.tracks:
        ts = np.array([s.timestep for s in t.object_states])
        assert np.all(np.diff(ts) > 0) and ts[0] >= 0 and ts[-1] <= 109
    av = sc.tracks[ids.index("AV")]
    assert len(av.object_states) == 110 and av.object_type.name == "VEHICLE"
    # recorded speed agrees with finite differences of the recorded positions (10 Hz) for the AV
    p = np.array([s.position for s in av.object_states])
    v = np.linalg.norm(np.array([s.velocity for s in av.object_states]), axis=1)
    fd = np.linalg.norm(np.diff(p, axis=0), axis=1) / 0.1
    assert np.median(np.abs(fd - v[1:])) < 0.3


@pytest.mark.reference
@pytest.mark.skipif(not rh.available(), reason="reference tree not present")
@pytest.mark.parametrize("name", DEMOS)
def test_raw_file_readers_match_fixture(scenes, name):
    """Build container only: the JSON / parquet readers reproduce the committed compact scene files."""
    seq = scene_io.DEMO_SCENES[name]
    d = os.path.join(rh.REF_ROOT, "data", seq)
    smap = av2_lite.StaticMap.from_json(os.path.join(d, f"log_map_archive_{seq}.json"))
    sc = av2_lite.load_argoverse_scenario_parquet(os.path.join(d, f"scenario_{seq}.parquet"))
    a, b = smap.to_arrays(), scenes[name][0].to_arrays()
    assert a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)
    a, b = sc.to_arrays(), scenes[name][1].to_arrays()
    assert a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)
    cfg = json.load(open(os.path.join(rh.REF_ROOT, "configs", name + ".json")))
    assert cfg["seq_id"] == seq
    assert json.loads(scenes[name][2]["cl_agent"]) == {k: cfg["cl_agents"][0][k] for k in
                                                       ("id", "enable_timestep", "semantic_lane", "target_velocity")}


# ---------------------------------------------------------------- reference scene logic: golden
@pytest.mark.parametrize("name", DEMOS)
def test_semantic_map_matches_reference(scenes, name):
    smp = scene_io.SemanticMap.from_static_map(scenes[name][0])
    g = lambda k: GOLD[f"{name}_{k}"]
    assert len(smp.semantic_lanes) == int(g("n_sem")) and list(smp.semantic_lanes) == list(range(len(smp.semantic_lanes)))
    assert np.array_equal([len(v) for v in smp.semantic_lanes.values()], g("sem_len"))
    pts = np.concatenate(list(smp.semantic_lanes.values()))
    assert pts.dtype == np.float32 and np.array_equal(pts, g("sem_pts"))
    for c, nm in enumerate(("intersect", "lane_type", "cross_left", "cross_right", "left", "right")):
        col = np.concatenate([v[c] for v in smp.semantic_lanes_infos.values()])
        assert col.dtype == np.float32 and np.array_equal(col.astype(np.int8), g("sem_" + nm))
    assert np.array_equal(np.array(smp.limits), g("limits"))
    # each semantic lane is a root-to-leaf path of the successor graph
    segs = scenes[name][0].vector_lane_segments
    for seq in smp.semantic_lane_seqs:
        assert not any(p in segs for p in segs[seq[0]].predecessors)
        assert not any(s in segs for s in segs[seq[-1]].successors)
        assert all(b in segs[a].successors for a, b in zip(seq, seq[1:]))


@pytest.mark.parametrize("name", DEMOS)
def test_track_loader_matches_reference(scenes, name):
    smp = scene_io.SemanticMap.from_static_map(scenes[name][0])
    pos, ang, vel, types, tids, cats, flags = scene_io.load_trajs_info(scenes[name][1], smp)
    g = lambda k: GOLD[f"{name}_{k}"]
    assert tids == list(g("tids")) and cats == list(g("cats")) and [t[0].name for t in types] == list(g("types"))
    assert np.array_equal(pos.shape, g("shape")) and pos.shape[1] == 546
    assert pos.dtype == np.float32 and ang.dtype == np.float32 and vel.dtype == np.float32 and flags.dtype == np.int16
    assert np.array_equal(np.packbits(flags.astype(bool), axis=1), g("flags"))
    assert np.array_equal(pos[:, ::7], g("pos7")) and np.array_equal(vel[:, ::7], g("vel7"))
    assert np.array_equal(ang[:, ::7], g("ang7"))
    sums = np.array([pos.astype(np.float64).sum(), ang.astype(np.float64).sum(), vel.astype(np.float64).sum(), flags.sum()])
    assert np.array_equal(sums, g("sums"))
    assert all(len(t) == 546 for t in types) and tids[1] == "AV" and cats[:2] == ["focal", "av"]


@pytest.mark.parametrize("name", DEMOS)
def test_target_lane_matches_reference(worlds, name):
    w = worlds[name]
    g = lambda k: GOLD[f"{name}_{k}"]
    closest = scene_io.get_closest_semantic_lane(w.smp, w.pos[0], w.ang[0])
    assert (-1 if closest is None else closest) == int(g("closest_lane"))
    assert np.array_equal(w.target_lane, g("target_lane")) and w.target_velocity == float(g("target_velocity"))
    assert np.array_equal(w.gt_tgt_lane, g("gt_tgt_lane"))
    assert np.all(np.linalg.norm(np.diff(w.gt_tgt_lane, axis=0), axis=1) > 4.0)
    assert len(w.target_lane_info) == 6 and all(len(c) == len(w.target_lane) for c in w.target_lane_info)


def test_target_lane_branches():
    """The three target-lane constructions on a two-lane toy map (agent.py:179-222)."""
    smp = scene_io.SemanticMap()
    xs = np.arange(0.0, 60.0, 2.0)
    smp.semantic_lanes = {0: np.stack([xs, np.zeros_like(xs)], 1).astype(np.float32),
                          1: np.stack([xs, np.full_like(xs, 3.5)], 1).astype(np.float32)}
    smp.semantic_lanes_infos = {k: [np.zeros(len(xs), np.float32)] * 6 for k in (0, 1)}
    t = np.linspace(0, 1, 40)[:, None]
    pos = (np.array([[1.0, 0.4]]) * (1 - t) + np.array([[21.0, 3.2]]) * t).astype(np.float32)     # lane change 0 -> 1
    ang = np.full(40, np.arctan2(2.8, 20.0), np.float32)
    # start near lane 0 AND lane 1 (both within 5 m): the lane closest at the END wins
    assert scene_io.get_closest_semantic_lane(smp, pos, ang) == 1
    lane, info = scene_io.target_lane_for(smp, pos, ang, False)
    assert lane is smp.semantic_lanes[1] and info is smp.semantic_lanes_infos[1]
    spliced, info = scene_io.target_lane_for(smp, pos, ang, True)
    assert info is None and np.array_equal(spliced[-1], smp.semantic_lanes[1][-1]) and np.array_equal(spliced[0], pos[0])
    k = int(np.argmin(np.linalg.norm(smp.semantic_lanes[1] - pos[-1], axis=1)))
    assert len(spliced) == len(scene_io.remove_close_points(pos, 0.1)) + len(xs) - k
    forced, _ = scene_io.target_lane_for(smp, pos, ang, True, 0)                                   # explicit lane id
    assert np.array_equal(forced[-1], smp.semantic_lanes[0][-1])
    with pytest.raises(ValueError):
        scene_io.target_lane_for(smp, pos, ang, True, 7)
    # heading off by 90 deg: no lane qualifies -> recorded path extended by 10 x its last step
    none_lane, info = scene_io.target_lane_for(smp, pos, ang + np.float32(np.pi / 2), True)
    path = scene_io.remove_close_points(pos, 0.1)
    assert info is None and len(none_lane) == len(path) + 1
    assert np.allclose(none_lane[-1], path[-1] + (path[-1] - path[-2]) * 10.0)


def test_pad_nearest_and_close_points():
    have = np.array([0, 0, 1, 0, 1, 0, 0], bool)
    vals = np.arange(7.0)[:, None] * np.ones((1, 2))
    assert np.array_equal(scene_io._pad_nearest(vals, have)[:, 0], [2, 2, 2, 2, 4, 4, 4])
    pts = np.array([[0.0, 0], [0.05, 0], [0.2, 0], [0.25, 0], [0.5, 0]])
    assert np.array_equal(scene_io.remove_close_points(pts, 0.1)[:, 0], [0.0, 0.2, 0.5])
    assert len(scene_io.remove_close_points(pts[:1], 0.1)) == 1


# ---------------------------------------------------------------- replay world behind the closed-loop driver
class _CountingPlanner:
    def __init__(self):
        self.lane, self.n_exo, self.ids = None, [], set()

    def update_target_lane(self, lane):
        self.lane = lane

    def update_observation(self, lcl):
        self.n_exo.append(len(lcl.exo_agents))
        self.ids.update(a.id for a in lcl.exo_agents)
        self.last = lcl

    def update_state_ctrl(self, s, c):
        pass

    def plan(self, lcl):
        return True, np.array([0.0, 0.0]), None


@pytest.mark.parametrize("name", DEMOS)
def test_replay_world_drives_closed_loop(worlds, name):
    w = worlds[name]
    assert w.agent_ids[0] == "AV" and w.cats[1] == "focal" and w.enable_time == 4.0
    p = _CountingPlanner()
    sim = ClosedLoopSim(w, p)
    assert p.lane is w.gt_tgt_lane
    for _ in range(250):
        sim.step()
    assert len(p.n_exo) == 50 and sim.n_plans == 10 and sim.enabled
    # all kept tracks are observed at frame 49 (t = 4.9 s): the loader's selection rule
    assert all(w.is_valid(i, 4.9) for i in range(w.n_agents))
    assert max(p.n_exo) <= w.n_agents - 1 and p.n_exo[-1] == w.n_agents - 1
    # replayed states hit the recording at 10 Hz frames
    k = 245
    assert np.array_equal(w.agent_state(1, 4.9), [w.pos[1, k, 0], w.pos[1, k, 1], w.vel[1, k], w.ang[1, k]])
    # the map is what the lane featuriser reads
    from mind_amd.planners.mind import utils as U
    lcl = w.local_semantic_map(4.9)
    st = lcl.ego_agent.state
    orig = st[:2].astype(np.float32)
    rot = np.array([[np.cos(st[3]), -np.sin(st[3])], [np.sin(st[3]), np.cos(st[3])]], np.float32)
    graph = U.lane_graph_from_map(lcl.map_data, orig, rot)
    assert graph["num_lanes"] >= len(w.vector_lane_segments)
    assert lcl.ego_agent.id == "AV" and len(lcl.exo_agents) == w.n_agents - 1


def test_replay_world_closed_loop_agent_options(scenes):
    """configs/demo_*.json: semantic_lane -1 = closest lane, an id = that lane; target_velocity -1 = mean recorded speed
    (loader.py:56-63, agent.py:164-166)."""
    smap, sc, meta = scenes["demo_3"]
    base = scene_io.ReplayWorld(smap, sc, dict(id="AV", enable_timestep=2.5, semantic_lane=-1, target_velocity=-1))
    assert base.enable_time == 2.5
    assert base.target_velocity == pytest.approx(float(np.mean(base.vel[0])))
    lane_id = scene_io.get_closest_semantic_lane(base.smp, base.pos[0], base.ang[0])
    forced = scene_io.ReplayWorld(smap, sc, dict(id="AV", enable_timestep=4.0, semantic_lane=int(lane_id), target_velocity=6))
    assert forced.target_velocity == 6 and np.array_equal(forced.target_lane, base.target_lane)
    other = (lane_id + 1) % len(base.smp.semantic_lanes)
    w2 = scene_io.ReplayWorld(smap, sc, dict(id="AV", enable_timestep=4.0, semantic_lane=int(other), target_velocity=6))
    assert np.array_equal(w2.target_lane, base.smp.semantic_lanes[other])
    # the spliced target lane starts on the recorded path and ends on the chosen semantic lane
    assert np.array_equal(w2.gt_tgt_lane[0], base.pos[0][0]) or np.linalg.norm(w2.gt_tgt_lane[0] - base.pos[0][0]) < 0.2
    assert np.array_equal(w2.gt_tgt_lane[-1], base.smp.semantic_lanes[other][-1]) or \
        np.linalg.norm(w2.gt_tgt_lane[-1] - base.smp.semantic_lanes[other][-1]) <= 4.0
    with pytest.raises(ValueError):
        scene_io.ReplayWorld(smap, sc, dict(id="AV", semantic_lane=10_000))
    with pytest.raises(ValueError):
        scene_io.ReplayWorld(smap, sc, dict(id="no-such-track"))


def test_track_selection_rules_on_a_toy_scenario():
    """loader.py:99-129 rule by rule: a track is dropped when it starts after frame 49, when frame 49 is missing, or when any
    of its first 50 RECORDED samples is 5 m or more from every semantic lane; everything else is kept in the order focal,
    AV, scored, unscored, fragments; missing frames are padded with the nearest earlier sample (leading gap: the first)."""
    from types import SimpleNamespace as NS
    smp = scene_io.SemanticMap()
    xs = np.arange(0.0, 200.0, 2.0)
    smp.semantic_lanes = {0: np.stack([xs, np.zeros_like(xs)], 1).astype(np.float32)}
    smp.semantic_lanes_infos = {0: [np.zeros(len(xs), np.float32)] * 6}

    def track(tid, cat, frames, y=0.5, typ="VEHICLE"):
        st = [NS(observed=True, timestep=int(t), position=(1.0 + 0.5 * t, y), heading=0.0, velocity=(5.0, 0.0)) for t in frames]
        return NS(track_id=tid, object_states=st, object_type=av2_lite.ObjectType[typ], category=av2_lite.TrackCategory[cat])

    tracks = [
        track("frag", "TRACK_FRAGMENT", range(40, 60)),
        track("late", "SCORED_TRACK", range(60, 110)),                       # starts after frame 49
        track("gap49", "SCORED_TRACK", [t for t in range(110) if t != 49]),  # not observed at frame 49
        track("far", "UNSCORED_TRACK", range(110), y=5.5),                   # 5.5 m from the only lane
        track("far_later", "UNSCORED_TRACK", range(30, 110), y=0.0),         # leaves the lane only after its first 50 samples:
        track("AV", "UNSCORED_TRACK", range(110)),
        track("focal", "FOCAL_TRACK", range(110), typ="BUS"),
        track("holes", "SCORED_TRACK", [0, 10, 49, 50, 109]),
    ]
    for s in tracks[4].object_states[55:]:
        s.position = (s.position[0], 30.0)                                   # ... samples 55.. (frames 85..) are far away
    sc = av2_lite.Scenario("toy", "focal", tracks)
    pos, ang, vel, types, tids, cats, flags = scene_io.load_trajs_info(sc, smp)
    assert tids == ["focal", "AV", "holes", "far_later", "frag"] and cats == ["focal", "av", "score", "unscore", "frag"]
    assert types[0][0].name == "BUS" and pos.shape == (5, 546, 2) and flags.dtype == np.int16
    h = tids.index("holes")
    assert flags[h, ::5].tolist()[:12] == [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0]      # 10 Hz frames 0..11: observed at 0 and 10
    assert flags[h, 49 * 5] == 1 and flags[h, 50 * 5] == 1 and flags[h, 80 * 5] == 0
    assert pos[h, 5 * 5, 0] == np.float32(1.0)                                        # frame 5: padded with frame 0's sample
    assert pos[h, 30 * 5, 0] == np.float32(1.0 + 0.5 * 10)                            # frame 30: frame 10's sample
    assert vel[h, 30 * 5] == 0.0 and vel[h, 49 * 5] == 5.0                            # speed is not padded
    f = tids.index("frag")
    assert flags[f, 39 * 5] == 0 and flags[f, 40 * 5] == 1 and pos[f, 0, 0] == np.float32(1.0 + 0.5 * 40)   # leading gap: first sample
    # between frames 49 and 50 the resampling is linear in position, observed when the blend exceeds one half
    assert np.allclose(pos[0, 49 * 5 + 2, 0], 1.0 + 0.5 * 49.4, atol=1e-5)
    assert flags[h, 10 * 5 + 2] == 1 and flags[h, 10 * 5 + 3] == 0                    # frame 10 observed, frame 11 not


@pytest.mark.reference
@pytest.mark.skipif(not rh.available(), reason="reference tree not present")
def test_toy_scenario_matches_reference_loader():
    """Build container only: the reference's own ArgoAgentLoader.get_trajs_info on random toy scenarios (tracks with holes,
    late starts, off-lane stretches) against load_trajs_info -- the golden scenes do not exercise every drop rule."""
    import importlib
    import sys
    from types import SimpleNamespace as NS
    rh.install()
    loader_mod = importlib.import_module("loader")
    ser = sys.modules["av2.datasets.motion_forecasting.scenario_serialization"]
    rng = np.random.default_rng(7)
    smp = scene_io.SemanticMap()
    xs = np.arange(0.0, 300.0, 3.0)
    smp.semantic_lanes = {0: np.stack([xs, 2.0 * np.sin(xs / 40.0)], 1).astype(np.float32),
                          1: np.stack([xs, 3.6 + 2.0 * np.sin(xs / 40.0)], 1).astype(np.float32)}
    smp.semantic_lanes_infos = {k: [np.zeros(len(xs), np.float32)] * 6 for k in (0, 1)}
    cats = ["TRACK_FRAGMENT", "UNSCORED_TRACK", "SCORED_TRACK"]
    for trial in range(4):
        tracks = []
        for i in range(14):
            t0 = int(rng.integers(0, 70)) if i > 1 else 0
            frames = [t for t in range(t0, 110) if i < 2 or rng.random() > 0.15]
            x0, v, y0 = rng.uniform(0, 120), rng.uniform(0, 9), rng.choice([0.0, 3.6, 9.0, -4.8])
            st = [rh.ObjectState(True, int(t), (float(x0 + 0.1 * v * t), float(y0 + 2.0 * np.sin((x0 + 0.1 * v * t) / 40.0) + 0.3 * np.sin(t / 7.0))),
                                 float(rng.uniform(-3.1, 3.1)), (float(v), float(rng.normal() * 0.2))) for t in frames]
            tid = ["focal", "AV"][i] if i < 2 else str(100 + i)
            cat = rh.TrackCategory.FOCAL_TRACK if i == 0 else rh.TrackCategory[cats[i % 3]]
            tracks.append(rh.Track(tid, st, list(rh.ObjectType)[i % 5], cat))
        scen = NS(tracks=tracks, focal_track_id="focal")
        orig = ser.load_argoverse_scenario_parquet
        ser.load_argoverse_scenario_parquet = lambda path: scen
        try:
            want = loader_mod.ArgoAgentLoader("unused").get_trajs_info(smp)
        finally:
            ser.load_argoverse_scenario_parquet = orig
        got = scene_io.load_trajs_info(scen, smp)
        assert got[4] == want[4] and got[5] == want[5], trial
        assert [t[0].name for t in got[3]] == [t[0].name for t in want[3]]
        for k in (0, 1, 2, 6):
            assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), (trial, k)
        assert 2 <= len(got[4]) < 14                                     # some tracks were dropped, some kept


@pytest.mark.reference
@pytest.mark.skipif(not rh.available(), reason="reference tree not present")
def test_target_lane_branches_match_reference_agent():
    """Build container only: CustomizedAgent.get_target_lane / get_closest_semantic_lane (agent.py:179-250) on toy lanes and
    random recorded paths -- closest lane found / not found, explicit lane id, with and without the recorded path."""
    import importlib
    rh.install()
    agent_mod = importlib.import_module("agent")
    rng = np.random.default_rng(3)
    xs = np.arange(0.0, 160.0, 2.5)
    smp = scene_io.SemanticMap()
    smp.semantic_lanes = {0: np.stack([xs, np.zeros_like(xs)], 1).astype(np.float32),
                          1: np.stack([xs, 3.6 + 0.02 * xs], 1).astype(np.float32),
                          2: np.stack([40.0 + 0.0 * xs, xs - 60.0], 1).astype(np.float32)}          # a crossing lane
    smp.semantic_lanes_infos = {k: [np.full(len(xs), float(k), np.float32)] * 6 for k in smp.semantic_lanes}
    seen = set()
    for trial in range(40):
        x0, y0, yaw = rng.uniform(0, 60), rng.choice([0.2, 3.9, -9.0, 1.8]), rng.choice([0.0, 0.05, 1.4, -0.6])
        v = rng.uniform(0.0, 8.0)
        t = np.arange(0, 546)[:, None] * 0.02
        pos = (np.array([[x0, y0]]) + v * t * np.array([[np.cos(yaw), np.sin(yaw)]]) + rng.normal(size=(546, 2)) * 0.01).astype(np.float32)
        ang = np.full(546, yaw, np.float32)
        ag = agent_mod.CustomizedAgent()
        ag.traj_info = [pos, ang, np.full(546, v, np.float32), np.ones(546, np.int16)]
        want_c = ag.get_closest_semantic_lane(smp, pos, ang)
        assert scene_io.get_closest_semantic_lane(smp, pos, ang) == want_c
        seen.add(want_c)
        for use_traj in (False, True):
            for lane_id in (None, 1, 2):
                if v < 0.05 and (use_traj or want_c is None):
                    continue                                             # a standing agent has a one-point path: both sides index [-2]
                want, want_info = ag.get_target_lane(smp, use_traj, lane_id)
                got, got_info = scene_io.target_lane_for(smp, pos, ang, use_traj, lane_id)
                assert got.dtype == want.dtype and np.array_equal(got, want), (trial, use_traj, lane_id)
                assert (got_info is None) == (want_info is None)
    assert None in seen and len(seen) >= 3                               # found / not found, different lanes


@pytest.mark.reference
@pytest.mark.skipif(not rh.available(), reason="reference tree not present")
def test_local_semantic_map_matches_reference(scenes):
    """Build container only: LocalSemanticMap (common/semantic_map.py:176-233) on the recorded demo_1 map: closest semantic
    lane for random poses, observation split into ego / exo."""
    import importlib
    from types import SimpleNamespace as NS
    rh.install()
    ref_mod = importlib.import_module("common.semantic_map")
    smp = scene_io.SemanticMap.from_static_map(scenes["demo_1"][0])
    ref_smp = NS(map_data=None, semantic_lanes=smp.semantic_lanes, semantic_lanes_infos=smp.semantic_lanes_infos)
    mine, ref = scene_io.LocalSemanticMap("AV", smp), ref_mod.LocalSemanticMap("AV", ref_smp)
    rng = np.random.default_rng(5)
    lo, hi = np.array(smp.limits)[:, 0], np.array(smp.limits)[:, 1]
    hits = 0
    for _ in range(200):
        p, a = rng.uniform(lo, hi), rng.uniform(-np.pi, np.pi)
        want = ref.get_closest_semantic_lane(p, a)
        assert mine.get_closest_semantic_lane(p, a) == want
        hits += want is not None
    assert 20 < hits <= 200
    agents = [NS(id=i, state=None) for i in ("7", "AV", "9")]
    mine.update_observation(agents)
    ref.update_observation(agents)
    assert mine.ego_agent is ref.ego_agent and [a.id for a in mine.exo_agents] == [a.id for a in ref.exo_agents] == ["7", "9"]
