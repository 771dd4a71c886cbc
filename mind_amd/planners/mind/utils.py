"""Host-side array helpers of the AIME scenario-tree generator (numpy, float32 where the reference
computes in torch float32).

Mirrors the call surface / numerics of the reference ``planners/mind/utils.py``:
  get_agent_trajectories :245-342   Track histories -> padded [a,50,*] arrays (NN pad pos/heading,
                                    zero pad velocity, agents unobserved at the last step skipped, AV first)
  lane_graph_from_map    :345-483   (``update_lane_graph_from_argo``) without shapely: arc-length
                                    resampling of 10-pt centre lines into 15 m pieces x 10 sub-segments
  get_origin_rotation    :180-190   frame at history step 49
  get_rpe                :193-242   5-channel relative pose encoding
  actor_features         :114-139   (``actor_gather``)  [a,14,48]
  lane_features          :103-110   (``graph_gather``)  [l,10,16]
  get_distance_to_polyline :502-513, get_max_covariance :536-551, get_angle :215-216
Everything is vectorised over agents (the reference loops in Python, scenario_tree.py:146-152).
"""
import numpy as np

OBS_LEN = 50
F32 = np.float32

_CROSSABLE = {"DASH_SOLID_YELLOW", "DASH_SOLID_WHITE", "DASHED_WHITE", "DASHED_YELLOW", "DOUBLE_DASH_YELLOW",
              "DOUBLE_DASH_WHITE"}
_NOT_CROSSABLE = {"DOUBLE_SOLID_YELLOW", "DOUBLE_SOLID_WHITE", "SOLID_YELLOW", "SOLID_WHITE", "SOLID_DASH_WHITE",
                  "SOLID_DASH_YELLOW", "SOLID_BLUE"}
_TYPE_SLOT = {"VEHICLE": 0, "PEDESTRIAN": 1, "MOTORCYCLIST": 2, "CYCLIST": 3, "BUS": 4, "UNKNOWN": 5}


def _name(x):
    """Enum member (av2 or look-alike) or plain string -> upper-case name."""
    n = getattr(x, "name", x)
    return str(n).upper()


def get_angle(vel):
    return np.arctan2(vel[..., 1], vel[..., 0])


def get_max_covariance(data):
    return np.maximum(data[..., 0], data[..., 1])[..., None]


def rot2(theta):
    """[[cos, -sin], [sin, cos]] in float32 (get_origin_rotation :185-186)."""
    c, s = np.cos(theta, dtype=F32), np.sin(theta, dtype=F32)
    return np.array([[c, -s], [s, c]], dtype=F32)


def get_origin_rotation(traj_pos, traj_ang):
    orig = traj_pos[OBS_LEN - 1]
    theta = traj_ang[OBS_LEN - 1]
    return orig, rot2(theta), theta


def _nn_fill(vals, have):
    """Nearest-neighbour padding along axis 0 (forward fill, then backward fill) (:38-58)."""
    n = len(have)
    idx = np.where(have, np.arange(n), -1)
    fwd = np.maximum.accumulate(idx)
    first = int(np.argmax(have))
    fwd[fwd < 0] = first
    return vals[fwd]


_FILL_LIB = []


def _fill_lib():
    """libmind_hip.so for its host-side track routine, or None (MIND_NATIVE_TRACKS=0, library not built): the numpy form below is the same arithmetic"""
    if not _FILL_LIB:
        import os
        lib = None
        if os.environ.get("MIND_NATIVE_TRACKS", "1") != "0":
            try:
                from ... import _lib
                lib = _lib.load()
            except Exception:      # noqa: BLE001
                lib = None
        _FILL_LIB.append(lib)
    return _FILL_LIB[0]


def _track_slot(tr):
    """one-hot slot of a track's object type (constant per track: kept on the Track object when it takes attributes)"""
    slot = getattr(tr, "_type_slot", None)
    if slot is None:
        slot = _TYPE_SLOT.get(_name(tr.object_type), 6)
        try:
            tr._type_slot = slot
        except AttributeError:
            pass
    return slot


def _agent_trajectories_full_windows(agent_obs, keys, order):
    """Same result as the per-agent loop of get_agent_trajectories, all agents in one set of array ops.  Needs the
    array mirror of every kept track (maintained by MINDPlanner.update_observation); tracks shorter than OBS_LEN
    (agents that appeared less than 5 s ago) are left-padded with unobserved rows, which is where the per-agent
    loop places them (`ts = arange(OBS_LEN - n, OBS_LEN)`)."""
    sel, raws = [], []
    for rank, ki in enumerate(order):
        tr = agent_obs[keys[ki]]
        if tr.object_states[-1].observed is False:
            continue
        raw = getattr(tr, "_arr", None)
        n = len(tr.object_states)
        if raw is None or len(raw) != n or n > OBS_LEN:
            return None
        if n < OBS_LEN:
            raw = np.concatenate([np.zeros((OBS_LEN - n, raw.shape[1]), raw.dtype), raw])
        sel.append((rank, keys[ki], tr))
        raws.append(raw)
    raw = np.stack(raws)                                   # [a,50,6] float64
    a = len(sel)
    lib = _fill_lib()
    if lib is not None and raw.dtype == np.float64:
        # the same arrays from the library's host routine (mind_fill_tracks: copies and casts, no device involved)
        raw = np.ascontiguousarray(raw)
        slot = np.array([_track_slot(tr) for _, _, tr in sel], np.int32)
        pos, ang, vel = np.empty((a, OBS_LEN, 2), F32), np.empty((a, OBS_LEN), F32), np.empty((a, OBS_LEN, 2), F32)
        typ, have = np.empty((a, OBS_LEN, 7), np.int16), np.empty((a, OBS_LEN), np.int16)
        if lib.mind_fill_tracks(raw.ctypes.data, a, OBS_LEN, slot.ctypes.data, pos.ctypes.data, ang.ctypes.data, vel.ctypes.data, typ.ctypes.data,
                                have.ctypes.data) == 0:
            return pos, ang, vel, typ, have, [k for _, k, _ in sel], ["av" if r == 0 else "exo" for r, _, _ in sel]
    have = raw[..., 0] != 0
    pos = np.where(have[..., None], raw[..., 1:3], 0.0)
    ang = np.where(have, raw[..., 3], 0.0)
    vel = np.where(have[..., None], raw[..., 4:6], 0.0)
    idx = np.where(have, np.arange(OBS_LEN)[None, :], -1)
    fwd = np.maximum.accumulate(idx, axis=1)
    first = np.argmax(have, axis=1)
    fwd = np.where(fwd < 0, first[:, None], fwd)
    rows = np.arange(a)[:, None]
    pos, ang = pos[rows, fwd], ang[rows, fwd]
    slot = np.array([_track_slot(tr) for _, _, tr in sel])
    typ = np.zeros((a, OBS_LEN, 7), np.int16)
    typ[rows, np.arange(OBS_LEN)[None, :], slot[:, None]] = have
    return (pos.astype(F32), ang.astype(F32), vel.astype(F32), typ, have.astype(np.int16),
            [k for _, k, _ in sel], ["av" if r == 0 else "exo" for r, _, _ in sel])


def get_agent_trajectories(agent_obs):
    """agent_obs: {id: Track(track_id, object_states, object_type, category)}; 'AV' is moved first.
    Returns pos [a,50,2] f32, ang [a,50] f32, vel [a,50,2] f32, type [a,50,7] i16, flags [a,50] i16, tids, cats."""
    keys = list(agent_obs.keys())
    order = [keys.index("AV")] + [i for i, k in enumerate(keys) if k != "AV"]
    fast = _agent_trajectories_full_windows(agent_obs, keys, order)
    if fast is not None:
        return fast
    P, A, V, T, Fl, tids, cats = [], [], [], [], [], [], []
    for rank, ki in enumerate(order):
        key = keys[ki]
        states = agent_obs[key].object_states
        if states[-1].observed is False:
            continue
        n = len(states)
        raw = getattr(agent_obs[key], "_arr", None)       # maintained incrementally by MINDPlanner.update_observation
        if raw is None or len(raw) != n:
            raw = np.array([(s.observed, s.position[0], s.position[1], s.heading, s.velocity[0], s.velocity[1])
                            for s in states], dtype=float)
        obs = raw[:, 0] != 0
        ts = np.arange(OBS_LEN - n, OBS_LEN)[obs]
        have = np.zeros(OBS_LEN, bool)
        have[ts] = True
        pos = np.zeros((OBS_LEN, 2))
        ang = np.zeros(OBS_LEN)
        vel = np.zeros((OBS_LEN, 2))
        pos[ts] = raw[obs, 1:3]
        ang[ts] = raw[obs, 3]
        vel[ts] = raw[obs, 4:6]
        pos, ang = _nn_fill(pos, have), _nn_fill(ang, have)
        slot = _TYPE_SLOT.get(_name(agent_obs[key].object_type), 6)
        typ = np.zeros((OBS_LEN, 7))
        typ[ts, slot] = 1
        P.append(pos); A.append(ang); V.append(vel); T.append(typ); Fl.append(have.astype(np.int16))
        tids.append(key)
        cats.append("av" if rank == 0 else "exo")
    return (np.array(P).astype(F32), np.array(A).astype(F32), np.array(V).astype(F32),
            np.array(T).astype(np.int16), np.array(Fl).astype(np.int16), tids, cats)


def normalize_agents(pos, ang, vel):
    """AV-centric then per-agent-centric frames (scenario_tree.py:128-158, vectorised).
    pos [a,50,2], ang [a,50], vel [a,50,2] float32 (world or previous frame).
    Returns orig [2], rot [2,2], theta, pos_n, ang_n, vel_n, ctrs [a,2], vecs [a,2]."""
    orig, rot, theta = get_origin_rotation(pos[0], ang[0])
    pos = np.matmul(pos - orig, rot)
    ang = ang - theta
    vel = np.matmul(vel, rot)
    ctrs = pos[:, OBS_LEN - 1].copy()
    th = ang[:, OBS_LEN - 1].copy()
    c, s = np.cos(th), np.sin(th)
    R = np.stack([np.stack([c, -s], -1), np.stack([s, c], -1)], -2).astype(F32)   # [a,2,2]
    pos_n = np.matmul(pos - ctrs[:, None, :], R)
    ang_n = ang - th[:, None]
    vel_n = np.matmul(vel, R)
    vecs = np.stack([c, s], -1).astype(F32)
    return orig, rot, theta, pos_n.astype(F32), ang_n.astype(F32), vel_n.astype(F32), ctrs.astype(F32), vecs


def actor_features(pos_n, ang_n, vel_n, types, pad):
    """[a,14,48]: [dx,dy, cos,sin, vx,vy, 7-way type, pad flag]; first 2 of the 50 steps dropped (Q11)."""
    disp = np.zeros_like(pos_n)
    disp[:, 1:] = pos_n[:, 1:] - pos_n[:, :-1]
    feat = np.concatenate([disp, np.stack([np.cos(ang_n), np.sin(ang_n)], -1), vel_n, types.astype(F32),
                           pad.astype(F32)[..., None]], axis=-1)
    return np.ascontiguousarray(np.transpose(feat, (0, 2, 1))[..., 2:], dtype=F32)


def get_rpe(ctrs, vecs, radius=100.0):
    """[5,n,n]: cos/sin(v_j, v_i), cos/sin(v_j, c_j - c_i), |c_j - c_i| * 2 / radius (float32)."""
    d = ctrs[None, :, :] - ctrs[:, None, :]
    dist = np.sqrt((d * d).sum(-1))
    v1 = np.broadcast_to(vecs[None, :, :], d.shape)
    v2 = np.broadcast_to(vecs[:, None, :], d.shape)

    def cs(a, b):
        den = np.sqrt((a * a).sum(-1)) * np.sqrt((b * b).sum(-1)) + F32(1e-10)
        return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) / den, (a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]) / den

    c1, s1 = cs(v1, v2)
    c2, s2 = cs(v1, d)
    return np.stack([c1, s1, c2, s2, dist * F32(2) / F32(radius)]).astype(F32)


def get_rpe_batch(ctrs, vecs, radius=100.0):
    """get_rpe for G independent point sets at once: ctrs, vecs [G,n,2] -> [G,5,n,n] (same float32 arithmetic)."""
    d = ctrs[:, None, :, :] - ctrs[:, :, None, :]
    dist = np.sqrt((d * d).sum(-1))
    v1 = np.broadcast_to(vecs[:, None, :, :], d.shape)
    v2 = np.broadcast_to(vecs[:, :, None, :], d.shape)

    def cs(a, b):
        den = np.sqrt((a * a).sum(-1)) * np.sqrt((b * b).sum(-1)) + F32(1e-10)
        return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) / den, (a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]) / den

    c1, s1 = cs(v1, v2)
    c2, s2 = cs(v1, d)
    return np.stack([c1, s1, c2, s2, dist * F32(2) / F32(radius)], axis=1).astype(F32)


def normalize_agents_batch(pos, ang, vel):
    """normalize_agents for G scenes with the same agent count at once: pos [G,a,50,2], ang [G,a,50], vel [G,a,50,2].
    Returns orig [G,2], rot [G,2,2], theta [G], pos_n, ang_n, vel_n, ctrs [G,a,2], vecs [G,a,2]."""
    orig = pos[:, 0, OBS_LEN - 1]
    theta = ang[:, 0, OBS_LEN - 1]
    c0, s0 = np.cos(theta, dtype=F32), np.sin(theta, dtype=F32)
    rot = np.stack([np.stack([c0, -s0], -1), np.stack([s0, c0], -1)], -2).astype(F32)      # [G,2,2]
    pos = np.matmul(pos - orig[:, None, None, :], rot[:, None])
    ang = ang - theta[:, None, None]
    vel = np.matmul(vel, rot[:, None])
    ctrs = pos[:, :, OBS_LEN - 1].copy()
    th = ang[:, :, OBS_LEN - 1].copy()
    c, s_ = np.cos(th), np.sin(th)
    R = np.stack([np.stack([c, -s_], -1), np.stack([s_, c], -1)], -2).astype(F32)          # [G,a,2,2]
    pos_n = np.matmul(pos - ctrs[:, :, None, :], R)
    ang_n = ang - th[..., None]
    vel_n = np.matmul(vel, R)
    vecs = np.stack([c, s_], -1).astype(F32)
    return orig, rot, theta, pos_n.astype(F32), ang_n.astype(F32), vel_n.astype(F32), ctrs.astype(F32), vecs


def actor_features_batch(pos_n, ang_n, vel_n, types, pad):
    """actor_features with a leading scene axis: -> [G,a,14,48]; types [G,a,50,7] (or broadcastable), pad [G,a,50]."""
    disp = np.zeros_like(pos_n)
    disp[:, :, 1:] = pos_n[:, :, 1:] - pos_n[:, :, :-1]
    feat = np.concatenate([disp, np.stack([np.cos(ang_n), np.sin(ang_n)], -1), vel_n, types.astype(F32),
                           pad.astype(F32)[..., None]], axis=-1)
    return np.ascontiguousarray(np.transpose(feat, (0, 1, 3, 2))[..., 2:], dtype=F32)


def get_distance_to_polyline(polyline, point):
    """min over segments of |closest point - point| (float32, :486-513), vectorised."""
    p1, p2 = polyline[:-1], polyline[1:]
    seg = p2 - p1
    t = ((point - p1) * seg).sum(-1) / (seg * seg).sum(-1)
    t = np.clip(t, 0, 1)
    closest = p1 + t[:, None] * seg
    d = closest - point
    return np.sqrt((d * d).sum(-1)).min()


def get_distances_to_polyline(polyline, points):
    """get_distance_to_polyline for several points [m,2] at once (same float32 arithmetic per point) -> [m]."""
    p1, p2 = polyline[:-1], polyline[1:]
    seg = p2 - p1
    rel = points[:, None, :] - p1[None]
    t = (rel * seg[None]).sum(-1) / (seg * seg).sum(-1)[None]
    t = np.clip(t, 0, 1)
    closest = p1[None] + t[..., None] * seg[None]
    d = closest - points[:, None, :]
    return np.sqrt((d * d).sum(-1)).min(axis=1)


# ---------------------------------------------------------------------------------------------
# lane graph featuriser (reference update_lane_graph_from_argo, no shapely)
# ---------------------------------------------------------------------------------------------
def _interp_along(cl, cum, s):
    s = np.clip(s, 0.0, cum[-1])
    k = np.clip(np.searchsorted(cum, s, side="right") - 1, 0, len(cl) - 2)
    seg = cum[k + 1] - cum[k]
    t = np.where(seg > 0, (s - cum[k]) / np.where(seg > 0, seg, 1.0), 0.0)
    return cl[k] + t[:, None] * (cl[k + 1] - cl[k])


_LANE_CACHE = {}


def _static_lane_pieces(static_map, seg_length, n_sub):
    """World-frame resampled polyline points [l, n_sub+1, 2] (float64) and per-piece flags; they do not
    depend on the ego pose, so they are computed once per map object."""
    key = (id(static_map), seg_length, n_sub)
    hit = _LANE_CACHE.get(key)
    if hit is not None and hit[0] is static_map:
        return hit[1]
    pts_all, flags = [], []
    for lane_id, lane in static_map.vector_lane_segments.items():
        cl = np.asarray(static_map.get_lane_segment_centerline(lane_id))[:, 0:2].astype(float)
        assert cl.shape[0] == n_sub, f"[Error] Wrong num of points in lane - {lane_id}:{cl.shape[0]}"
        cum = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(cl, axis=0), axis=1))])
        length = cum[-1]
        num_segs = max(int(np.floor(length / seg_length)), 1)
        ds = length / num_segs
        lt = {"VEHICLE": 0, "BIKE": 1, "BUS": 2}[_name(lane.lane_type)]

        def mark(m):
            n = _name(m)
            return 0 if n in _CROSSABLE else (1 if n in _NOT_CROSSABLE else 2)

        fl = (lt, 1 if lane.is_intersection else 0, mark(lane.left_mark_type), mark(lane.right_mark_type),
              0 if lane.left_neighbor_id is None else 1, 0 if lane.right_neighbor_id is None else 1)
        for i in range(num_segs):
            pts_all.append(_interp_along(cl, cum, np.linspace(i * ds, (i + 1) * ds, n_sub + 1)))
            flags.append(fl)
    pts_all = np.stack(pts_all)
    flags = np.asarray(flags)
    L = len(flags)
    one_hot = lambda idx: np.repeat(np.eye(3, dtype=np.int16)[idx][:, None, :], n_sub, 1)
    static = dict(
        pts=pts_all,
        lane_type=one_hot(flags[:, 0]), cross_left=one_hot(flags[:, 2]), cross_right=one_hot(flags[:, 3]),
        intersect=np.repeat(flags[:, 1:2], n_sub, 1).astype(np.int16),
        left=np.repeat(flags[:, 4:5], n_sub, 1).astype(np.int16), right=np.repeat(flags[:, 5:6], n_sub, 1).astype(np.int16),
        num_lanes=L, flags=np.ascontiguousarray(flags, np.int32))      # flags [l,6]: lane type, intersection, left / right mark class, neighbours
    if len(_LANE_CACHE) >= 16:          # a handful of maps live side by side (several scenes planned by one process: one entry each)
        _LANE_CACHE.clear()
    _LANE_CACHE[key] = (static_map, static)
    return static


def lane_graph_from_map(static_map, orig, rot, seg_length=15.0, n_sub=10):
    """Lane polylines of an AV2-style static map in the AV frame (orig [2], rot [2,2]):
    node_ctrs/node_vecs [l,10,2] f32 (instance frame), lane_ctrs/lane_vecs [l,2] f32 (AV frame), flags i16."""
    st = _static_lane_pieces(static_map, seg_length, n_sub)
    orig = np.asarray(orig)
    rot = np.asarray(rot)
    ctrln = np.matmul(st["pts"] - orig, rot)                          # [l,11,2] float64, AV frame
    anch_pos = ctrln.mean(axis=1)
    dv = ctrln[:, -1] - ctrln[:, 0]
    anch_vec = dv / np.linalg.norm(dv, axis=1, keepdims=True)
    anch_rot = np.stack([np.stack([anch_vec[:, 0], -anch_vec[:, 1]], -1),
                         np.stack([anch_vec[:, 1], anch_vec[:, 0]], -1)], -2)   # [l,2,2]
    inst = np.matmul(ctrln - anch_pos[:, None, :], anch_rot)
    g = dict()
    g["node_ctrs"] = ((inst[:, :-1] + inst[:, 1:]) / 2.0).astype(F32)
    g["node_vecs"] = (inst[:, 1:] - inst[:, :-1]).astype(F32)
    g["lane_ctrs"] = anch_pos.astype(F32)
    g["lane_vecs"] = anch_vec.astype(F32)
    for k in ("lane_type", "intersect", "cross_left", "cross_right", "left", "right"):
        g[k] = st[k]
    g["num_nodes"] = g["node_ctrs"].shape[0] * g["node_ctrs"].shape[1]
    g["num_lanes"] = st["num_lanes"]
    return g


update_lane_graph_from_argo = lane_graph_from_map   # reference name


def lane_features(g):
    """[l,10,16] = [ctr xy, vec xy, intersect, lane_type(3), cross_left(3), cross_right(3), left, right]."""
    f = np.concatenate([g["node_ctrs"], g["node_vecs"], g["intersect"][..., None].astype(F32),
                        g["lane_type"].astype(F32), g["cross_left"].astype(F32), g["cross_right"].astype(F32),
                        g["left"][..., None].astype(F32), g["right"][..., None].astype(F32)], axis=-1)
    return np.ascontiguousarray(f, dtype=F32)
