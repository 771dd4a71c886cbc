"""f4: frame recorder + top-down renderer on a recorded scene (CPU, stub planner returning contract-shaped trees)."""
import json
import os

import numpy as np
import pytest

from mind_amd import av2_lite, scene_io
from mind_amd.closed_loop import ClosedLoopSim
from mind_amd.planners.basic.tree import Node, Tree
from mind_amd.recorder import FrameRecorder, render_frame


class _TreePlanner:
    """Returns a two-branch scenario tree and a matching ego trajectory tree with the reference's data layout."""

    def update_target_lane(self, lane):
        self.lane = lane

    def update_observation(self, lcl):
        self.lcl = lcl

    def update_state_ctrl(self, s, c):
        self.state = np.asarray(s, np.float64)

    def plan(self, lcl):
        ags = [lcl.ego_agent] + list(lcl.exo_agents)
        a = len(ags)
        t = np.arange(1, 51)[None, :, None] * 0.1
        base = np.stack([np.asarray(g.state, np.float64) for g in ags])
        vel = base[:, None, 2:3] * np.concatenate([np.cos(base[:, None, 3:4]), np.sin(base[:, None, 3:4])], axis=2)
        pos = (base[:, None, :2] + vel * t).astype(np.float32)                       # constant velocity, [a,50,2]
        cov = np.broadcast_to((0.2 + 0.3 * t[..., 0]).astype(np.float32)[..., None], (a, 50, 1)).copy()
        st = Tree()
        st.add_node(Node("0_0_0", None, [np.float32(1.0), pos[:, :20], cov[:, :20], np.zeros((11, 2))]))
        st.add_node(Node("1_0_0", "0_0_0", [np.float32(0.6), pos[:, 20:], cov[:, 20:], np.zeros((11, 2))]))
        st.add_node(Node("1_0_1", "0_0_0", [np.float32(0.4), pos[:, 20:] + np.float32(0.5), cov[:, 20:], np.zeros((11, 2))]))
        tt = Tree()
        x0 = np.concatenate([self.state, [0.0, 0.0]])
        tt.add_node(Node(-1, None, [x0, np.zeros(2)]))
        for k in range(10):
            xs = x0.copy()
            xs[:2] = pos[0, 2 * k + 1]
            tt.add_node(Node(k, k - 1, [xs, np.array([0.1, 0.0])]))
        return True, np.array([0.2, 0.0]), [[st], [tt]]


@pytest.fixture(scope="module")
def world():
    smap, sc, meta = av2_lite.load_scene(scene_io.scene_fixture_path("demo_3"))
    return scene_io.ReplayWorld(smap, sc, json.loads(meta["cl_agent"]))


def test_record_save_load_round_trip(world, tmp_path):
    rec = FrameRecorder(ClosedLoopSim(world, _TreePlanner())).run(211)
    assert len(rec.frames) == 211
    planned = [i for i, f in enumerate(rec.frames) if "scen_tree" in f]
    assert planned == [200, 205, 210]                                         # simulator.py:85-94: planning steps only
    f = rec.frames[200]
    assert f["ids"][0] == "AV" and f["states"].shape == (len(f["ids"]), 4)
    s = f["scen_tree"][0]
    assert list(s["keys"]) == ["0_0_0", "1_0_0", "1_0_1"] and list(s["parent"]) == [-1, 0, 0]
    assert list(s["dur"]) == [20, 30, 30] and s["pos"].shape == (len(f["ids"]), 80, 2) and np.allclose(s["prob"], [1, .6, .4])
    t = f["traj_tree"][0]
    assert t["xs"].shape == (11, 6) and list(t["parent"]) == [-1] + list(range(10))
    # before the ego is enabled the frame shows the recording, afterwards the propagated state
    assert np.array_equal(rec.frames[10]["states"][0], world.agent_state(0, rec.frames[10]["time"]))
    assert not np.array_equal(rec.frames[210]["states"][0], world.agent_state(0, rec.frames[210]["time"]))
    p = os.path.join(tmp_path, "run.npz")
    rec.save(p)
    back = FrameRecorder.load(p)
    assert len(back) == 211 and back[200]["ids"] == f["ids"] and np.array_equal(back[200]["states"], f["states"])
    for k in s:
        assert np.array_equal(back[200]["scen_tree"][0][k], s[k])
    for k in t:
        assert np.array_equal(back[205]["traj_tree"][0][k], rec.frames[205]["traj_tree"][0][k])
    assert "scen_tree" not in back[201]


def test_render_frame_writes_png(world, tmp_path):
    pytest.importorskip("matplotlib")
    rec = FrameRecorder(ClosedLoopSim(world, _TreePlanner())).run(204)
    p = os.path.join(tmp_path, "frame.png")
    render_frame(rec.frames, 203, static_map=world.map_data, path=p)          # frame 203 shows the plan of frame 200
    assert os.path.getsize(p) > 5000
    with open(p, "rb") as fh:
        assert fh.read(8) == b"\x89PNG\r\n\x1a\n"
