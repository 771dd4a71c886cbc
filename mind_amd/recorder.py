"""Frame recorder and a light top-down renderer for closed-loop results.

The reference keeps one dict per simulator step (``frame['agents']``, and on planning steps ``frame['scen_tree']`` /
``frame['traj_tree']``, simulator.py:56-94), renders every frame as a 3-D matplotlib figure in a process pool and
joins the PNGs with ffmpeg (simulator.py:109-219, common/visualization.py).  Here the frames are arrays that can be
saved to one .npz per run and inspected anywhere; drawing is a single 2-D matplotlib axes (lane boundaries, agents,
predicted branches with their max-sigma circles, the ego trajectory tree), no ffmpeg, no shapely.

What the drawing reads from the returned trees is exactly the data contract of SURVEY 8(b):
scenario tree node.data = [prob, trajs [a,dur,2], covs [a,dur,1], tgt_pts]; trajectory tree root key -1 = [x0, 0],
node k = [xs[6], us[2]].
"""
import numpy as np


class FrameRecorder:
    """Wraps a ClosedLoopSim: ``step()`` advances the simulation and stores what the reference's frame dict holds."""

    def __init__(self, sim):
        self.sim = sim
        self.frames = []

    def step(self):
        sim, w = self.sim, self.sim.world
        t = sim.sim_time
        valid = getattr(w, "is_valid", None)
        ids, states = ["AV"], [np.array(sim.state if sim.enabled else w.agent_state(0, t), dtype=np.float64)]
        for i in range(1, w.n_agents):
            if valid is None or valid(i, t):
                ids.append(str(w.agent_ids[i]))
                states.append(np.asarray(w.agent_state(i, t), dtype=np.float64))
        frame = dict(time=t, ids=ids, states=np.stack(states))
        planned = sim.step()
        if planned and sim.last_result is not None:
            scen, traj = sim.last_result
            frame["scen_tree"] = [scen_tree_arrays(tr) for tr in scen]
            frame["traj_tree"] = [traj_tree_arrays(tr) for tr in traj]
        self.frames.append(frame)
        return planned

    def run(self, n_steps):
        for _ in range(n_steps):
            self.step()
        return self

    # ---- one file per run
    def save(self, path):
        out = {"n_frames": np.array(len(self.frames))}
        for i, f in enumerate(self.frames):
            out[f"f{i}_time"] = np.array(f["time"])
            out[f"f{i}_ids"] = np.array(f["ids"])
            out[f"f{i}_states"] = f["states"]
            for kind in ("scen_tree", "traj_tree"):
                if kind in f:
                    out[f"f{i}_{kind}_n"] = np.array(len(f[kind]))
                    for j, tr in enumerate(f[kind]):
                        for k, v in tr.items():
                            out[f"f{i}_{kind}{j}_{k}"] = v
        np.savez_compressed(path, **out)

    @staticmethod
    def load(path):
        with np.load(path, allow_pickle=False) as z:
            a = {k: z[k] for k in z.files}
        frames = []
        for i in range(int(a["n_frames"])):
            f = dict(time=float(a[f"f{i}_time"]), ids=[str(x) for x in a[f"f{i}_ids"]], states=a[f"f{i}_states"])
            for kind in ("scen_tree", "traj_tree"):
                if f"f{i}_{kind}_n" in a:
                    pre = [f"f{i}_{kind}{j}_" for j in range(int(a[f"f{i}_{kind}_n"]))]
                    f[kind] = [{k[len(p):]: a[k] for k in a if k.startswith(p)} for p in pre]
            frames.append(f)
        return frames


def scen_tree_arrays(tree):
    """Scenario tree -> arrays: keys, parent index (-1 root), prob, window start/length into the concatenated
    trajectories [a, sum(dur), 2] and max-sigma [a, sum(dur)]."""
    keys = list(tree.nodes.keys())
    idx = {k: i for i, k in enumerate(keys)}
    durs = [tree.nodes[k].data[1].shape[1] for k in keys]
    return dict(keys=np.array([str(k) for k in keys]),
                parent=np.array([idx.get(tree.nodes[k].parent_key, -1) for k in keys], np.int32),
                prob=np.array([float(np.ravel(tree.nodes[k].data[0])[0]) for k in keys]),
                start=np.concatenate([[0], np.cumsum(durs)[:-1]]).astype(np.int32), dur=np.array(durs, np.int32),
                pos=np.concatenate([tree.nodes[k].data[1] for k in keys], axis=1).astype(np.float32),
                cov=np.concatenate([tree.nodes[k].data[2][..., 0] for k in keys], axis=1).astype(np.float32))


def traj_tree_arrays(tree):
    """Trajectory tree -> arrays in key order (root -1 first): parent index, xs [M+1,6], us [M+1,2]."""
    keys = list(tree.nodes.keys())
    idx = {k: i for i, k in enumerate(keys)}
    return dict(parent=np.array([idx.get(tree.nodes[k].parent_key, -1) for k in keys], np.int32),
                xs=np.array([np.asarray(tree.nodes[k].data[0], np.float64) for k in keys]),
                us=np.array([np.asarray(tree.nodes[k].data[1], np.float64) for k in keys]))


def latest(frames, i, kind):
    """The reference draws the most recent planning result on frames without one (simulator.py:148-166)."""
    for j in range(i, -1, -1):
        if kind in frames[j]:
            return frames[j][kind]
    return None


def render_frame(frames, i, static_map=None, path=None, view=60.0, history=100):
    """Top-down picture of frame i (matplotlib, Agg): lane boundaries, agent positions with a heading tick and their
    last `history` positions, every predicted branch of the latest scenario trees (line width ~ probability, a circle
    of radius sqrt(max-sigma) every second, visualization.py:19-21) and the latest ego trajectory trees.
    Returns the Figure (and writes a PNG when `path` is given)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from matplotlib.patches import Circle

    f = frames[i]
    fig, ax = plt.subplots(figsize=(8, 8))
    ego = f["states"][0]
    if static_map is not None:
        for ls in static_map.vector_lane_segments.values():
            for b in (ls.left_lane_boundary.xyz, ls.right_lane_boundary.xyz):
                ax.plot(b[:, 0], b[:, 1], color="0.75", lw=0.8, zorder=1)
    hist = {}
    for j in range(max(0, i - history), i + 1):
        for aid, st in zip(frames[j]["ids"], frames[j]["states"]):
            hist.setdefault(aid, []).append(st[:2])
    for aid, st in zip(f["ids"], f["states"]):
        h = np.array(hist.get(aid, [st[:2]]))
        if len(h) > 1 and np.linalg.norm(h[0] - h[-1]) >= 0.1:
            ax.plot(h[:, 0], h[:, 1], color="mediumpurple", lw=1.0, zorder=2)
        col = "tab:blue" if aid == "AV" else "indianred"
        ax.plot(st[0], st[1], "o", color=col, ms=6, zorder=5)
        ax.plot([st[0], st[0] + 2.0 * np.cos(st[3])], [st[1], st[1] + 2.0 * np.sin(st[3])], color=col, lw=1.5, zorder=5)
    scen = latest(frames, i, "scen_tree")
    for tr in scen or []:
        for n in range(len(tr["prob"])):
            s, d = int(tr["start"][n]), int(tr["dur"][n])
            p = tr["pos"][:, s:s + d]
            c = tr["cov"][:, s:s + d]
            for a in range(p.shape[0]):
                ax.plot(p[a, :, 0], p[a, :, 1], color="tab:orange" if a else "tab:green", lw=0.5 + 2.5 * float(tr["prob"][n]),
                        alpha=0.8, zorder=3)
                for q in range(9, d, 10):
                    ax.add_patch(Circle((p[a, q, 0], p[a, q, 1]), float(np.sqrt(max(c[a, q], 0.0))), fill=False,
                                        ec="tab:orange" if a else "tab:green", lw=0.4, alpha=0.6, zorder=3))
    for tr in latest(frames, i, "traj_tree") or []:
        xs, par = tr["xs"], tr["parent"]
        for k in range(1, len(par)):
            ax.plot([xs[par[k], 0], xs[k, 0]], [xs[par[k], 1], xs[k, 1]], color="tab:blue", lw=1.8, zorder=4)
    ax.set_xlim(ego[0] - view, ego[0] + view)
    ax.set_ylim(ego[1] - view, ego[1] + view)
    ax.set_aspect("equal")
    ax.set_title("t = %.2f s   ego v = %.2f m/s   %d agents" % (f["time"], ego[2], len(f["ids"])))
    if path is not None:
        fig.savefig(path, dpi=80)
    plt.close(fig)
    return fig
