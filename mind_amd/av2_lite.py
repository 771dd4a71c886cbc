"""Dependency-free readers for the two Argoverse-2 motion-forecasting file formats the reference loads
(SURVEY 8 f3), so the demo scenes run without the `av2` package.

What is replaced (the reference pins av2==0.2.1 in requirements.txt; it is NOT importable in this image,
so the av2 layer is restated from its published algorithm -- parity for this layer is UNPINNED and is
checked by geometry invariants in tests/test_scene_io.py):
  * ``ArgoverseStaticMap.from_json`` + ``.vector_lane_segments`` + ``.get_lane_segment_centerline``
    (call sites: common/semantic_map.py:16-19,63; planners/mind/utils.py:350-353): every lane segment keeps
    id / lane_type / mark types / neighbours / predecessors / successors / is_intersection and both boundary
    polylines (xyz); the centerline is the midpoint of the two boundaries after each is resampled to 10
    points equally spaced in 3-D chord length.
  * ``scenario_serialization.load_argoverse_scenario_parquet`` (call site: loader.py:70): one Track per
    track_id, tracks sorted by track_id as strings (av2 groups the table with a sorting pandas groupby),
    object states in file order.

The enum classes can be injected (``lane_type_cls`` ...) so the same objects can be handed to code that
compares against another set of enum types; everything in this package compares enum *names* only.
"""
import enum
import json
from types import SimpleNamespace

import numpy as np

NUM_CENTERLINE_INTERP_PTS = 10


class LaneType(enum.Enum):
    VEHICLE = "VEHICLE"
    BIKE = "BIKE"
    BUS = "BUS"


class LaneMarkType(enum.Enum):
    DASH_SOLID_YELLOW = "DASH_SOLID_YELLOW"
    DASH_SOLID_WHITE = "DASH_SOLID_WHITE"
    DASHED_WHITE = "DASHED_WHITE"
    DASHED_YELLOW = "DASHED_YELLOW"
    DOUBLE_SOLID_YELLOW = "DOUBLE_SOLID_YELLOW"
    DOUBLE_SOLID_WHITE = "DOUBLE_SOLID_WHITE"
    DOUBLE_DASH_YELLOW = "DOUBLE_DASH_YELLOW"
    DOUBLE_DASH_WHITE = "DOUBLE_DASH_WHITE"
    SOLID_YELLOW = "SOLID_YELLOW"
    SOLID_WHITE = "SOLID_WHITE"
    SOLID_DASH_WHITE = "SOLID_DASH_WHITE"
    SOLID_DASH_YELLOW = "SOLID_DASH_YELLOW"
    SOLID_BLUE = "SOLID_BLUE"
    NONE = "NONE"
    UNKNOWN = "UNKNOWN"


class ObjectType(enum.Enum):
    VEHICLE = "vehicle"
    PEDESTRIAN = "pedestrian"
    MOTORCYCLIST = "motorcyclist"
    CYCLIST = "cyclist"
    BUS = "bus"
    STATIC = "static"
    BACKGROUND = "background"
    CONSTRUCTION = "construction"
    RIDERLESS_BICYCLE = "riderless_bicycle"
    UNKNOWN = "unknown"


class TrackCategory(enum.Enum):
    TRACK_FRAGMENT = 0
    UNSCORED_TRACK = 1
    SCORED_TRACK = 2
    FOCAL_TRACK = 3


def _by_name(cls, name):
    return cls[name]


def interp_arc(t, points):
    """`t` points equally spaced in normalised chord length along `points` [n,d] (piecewise linear)."""
    points = np.asarray(points, dtype=np.float64)
    if points.ndim != 2:
        raise ValueError("points must be [n, d]")
    n = points.shape[0]
    eq = np.linspace(0, 1, t)
    chord = np.linalg.norm(np.diff(points, axis=0), axis=1)
    chord = chord / np.sum(chord)
    cumarc = np.zeros(len(chord) + 1)
    cumarc[1:] = np.cumsum(chord)
    tbins = np.digitize(eq, bins=cumarc).astype(int)
    tbins[np.where((tbins <= 0) | (eq <= 0))] = 1
    tbins[np.where((tbins >= n) | (eq >= 1))] = n - 1
    s = np.divide(eq - cumarc[tbins - 1], chord[tbins - 1])
    return points[tbins - 1, :] + (points[tbins, :] - points[tbins - 1, :]) * s.reshape(-1, 1)


def compute_midpoint_line(left, right, num_interp_pts=NUM_CENTERLINE_INTERP_PTS):
    left = np.asarray(left, dtype=np.float64)
    right = np.asarray(right, dtype=np.float64)
    if left.ndim != 2 or right.ndim != 2 or left.shape[1] != right.shape[1]:
        raise ValueError("boundaries must be [n,d] polylines of the same dimension")
    le = interp_arc(num_interp_pts, left)
    ri = interp_arc(num_interp_pts, right)
    return (le + ri) / 2.0, np.linalg.norm(le - ri, axis=1)


class Polyline:
    def __init__(self, xyz):
        self.xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)

    @classmethod
    def from_json_data(cls, pts):
        return cls([[p["x"], p["y"], p["z"]] for p in pts])

    def __len__(self):
        return self.xyz.shape[0]


class LaneSegment:
    __slots__ = ("id", "is_intersection", "lane_type", "right_lane_boundary", "left_lane_boundary",
                 "right_mark_type", "left_mark_type", "predecessors", "successors", "right_neighbor_id",
                 "left_neighbor_id", "recorded_centerline")

    def __init__(self, **kw):
        # recorded_centerline: the file's own (sparser, z = 0) centerline polyline; av2 ignores it and so does
        # everything here except the invariant tests.
        kw.setdefault("recorded_centerline", None)
        for k in self.__slots__:
            setattr(self, k, kw[k])


class StaticMap:
    """The slice of av2's ArgoverseStaticMap that the reference touches."""

    def __init__(self, lane_segments):
        self.vector_lane_segments = {ls.id: ls for ls in lane_segments}

    @classmethod
    def from_json(cls, path, lane_type_cls=LaneType, lane_mark_cls=LaneMarkType):
        with open(path, "rb") as f:
            data = json.load(f)
        segs = []
        for d in data["lane_segments"].values():
            segs.append(LaneSegment(
                id=d["id"], is_intersection=d["is_intersection"], lane_type=_by_name(lane_type_cls, d["lane_type"]),
                right_lane_boundary=Polyline.from_json_data(d["right_lane_boundary"]),
                left_lane_boundary=Polyline.from_json_data(d["left_lane_boundary"]),
                right_mark_type=_by_name(lane_mark_cls, d["right_lane_mark_type"]),
                left_mark_type=_by_name(lane_mark_cls, d["left_lane_mark_type"]),
                predecessors=list(d["predecessors"]), successors=list(d["successors"]),
                right_neighbor_id=d["right_neighbor_id"], left_neighbor_id=d["left_neighbor_id"],
                recorded_centerline=(Polyline.from_json_data(d["centerline"]) if "centerline" in d else None)))
        return cls(segs)

    def get_lane_segment_centerline(self, lane_segment_id):
        ls = self.vector_lane_segments[lane_segment_id]
        ctr, _ = compute_midpoint_line(ls.left_lane_boundary.xyz, ls.right_lane_boundary.xyz)
        return ctr

    # ---- compact array form (what tests/golden/scenes/*.npz hold) ----
    def to_arrays(self):
        segs = list(self.vector_lane_segments.values())

        def ragged(get, dtype, width=None):
            rows = [np.asarray(get(s), dtype=dtype).reshape((-1,) + (() if width is None else (width,))) for s in segs]
            off = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
            flat = np.concatenate(rows) if rows else np.zeros((0,), dtype)
            return flat, off

        out = {}
        out["lane_id"] = np.array([s.id for s in segs], np.int64)
        out["lane_intersection"] = np.array([bool(s.is_intersection) for s in segs])
        out["lane_type"] = np.array([s.lane_type.name for s in segs])
        out["lane_left_mark"] = np.array([s.left_mark_type.name for s in segs])
        out["lane_right_mark"] = np.array([s.right_mark_type.name for s in segs])
        out["lane_left_nb"] = np.array([-1 if s.left_neighbor_id is None else s.left_neighbor_id for s in segs], np.int64)
        out["lane_right_nb"] = np.array([-1 if s.right_neighbor_id is None else s.right_neighbor_id for s in segs], np.int64)
        out["lane_left_xyz"], out["lane_left_off"] = ragged(lambda s: s.left_lane_boundary.xyz, np.float64, 3)
        out["lane_right_xyz"], out["lane_right_off"] = ragged(lambda s: s.right_lane_boundary.xyz, np.float64, 3)
        out["lane_pred"], out["lane_pred_off"] = ragged(lambda s: s.predecessors, np.int64)
        out["lane_succ"], out["lane_succ_off"] = ragged(lambda s: s.successors, np.int64)
        if all(s.recorded_centerline is not None for s in segs):
            out["lane_ctr_xyz"], out["lane_ctr_off"] = ragged(lambda s: s.recorded_centerline.xyz, np.float64, 3)
        return out

    @classmethod
    def from_arrays(cls, a, lane_type_cls=LaneType, lane_mark_cls=LaneMarkType):
        segs = []
        for i, lid in enumerate(a["lane_id"]):
            sl = lambda flat, off: flat[off[i]:off[i + 1]]
            segs.append(LaneSegment(
                id=int(lid), is_intersection=bool(a["lane_intersection"][i]),
                lane_type=_by_name(lane_type_cls, str(a["lane_type"][i])),
                right_lane_boundary=Polyline(sl(a["lane_right_xyz"], a["lane_right_off"])),
                left_lane_boundary=Polyline(sl(a["lane_left_xyz"], a["lane_left_off"])),
                right_mark_type=_by_name(lane_mark_cls, str(a["lane_right_mark"][i])),
                left_mark_type=_by_name(lane_mark_cls, str(a["lane_left_mark"][i])),
                predecessors=[int(x) for x in sl(a["lane_pred"], a["lane_pred_off"])],
                successors=[int(x) for x in sl(a["lane_succ"], a["lane_succ_off"])],
                right_neighbor_id=None if a["lane_right_nb"][i] < 0 else int(a["lane_right_nb"][i]),
                left_neighbor_id=None if a["lane_left_nb"][i] < 0 else int(a["lane_left_nb"][i]),
                recorded_centerline=(Polyline(sl(a["lane_ctr_xyz"], a["lane_ctr_off"])) if "lane_ctr_xyz" in a else None)))
        return cls(segs)


class Scenario:
    """tracks (list of namespaces: track_id, object_type, category, object_states[]), focal_track_id, scenario_id."""

    def __init__(self, scenario_id, focal_track_id, tracks, city_name=""):
        self.scenario_id = scenario_id
        self.focal_track_id = focal_track_id
        self.tracks = tracks
        self.city_name = city_name

    def to_arrays(self):
        rows = [len(t.object_states) for t in self.tracks]
        off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int32)
        st = [s for t in self.tracks for s in t.object_states]
        return dict(
            scenario_id=np.array(self.scenario_id), focal_track_id=np.array(self.focal_track_id),
            city=np.array(self.city_name),
            track_id=np.array([t.track_id for t in self.tracks]),
            track_type=np.array([t.object_type.name for t in self.tracks]),
            track_cat=np.array([t.category.value for t in self.tracks], np.int8),
            track_off=off,
            st_timestep=np.array([s.timestep for s in st], np.int16),
            st_observed=np.array([s.observed for s in st], bool),
            st_pos=np.array([s.position for s in st], np.float64).reshape(-1, 2),
            st_heading=np.array([s.heading for s in st], np.float64),
            st_vel=np.array([s.velocity for s in st], np.float64).reshape(-1, 2))

    @classmethod
    def from_arrays(cls, a, object_type_cls=ObjectType, category_cls=TrackCategory):
        tracks = []
        off = a["track_off"]
        for i, tid in enumerate(a["track_id"]):
            states = [SimpleNamespace(observed=bool(a["st_observed"][j]), timestep=int(a["st_timestep"][j]),
                                      position=(float(a["st_pos"][j, 0]), float(a["st_pos"][j, 1])),
                                      heading=float(a["st_heading"][j]),
                                      velocity=(float(a["st_vel"][j, 0]), float(a["st_vel"][j, 1])))
                      for j in range(off[i], off[i + 1])]
            tracks.append(SimpleNamespace(track_id=str(tid), object_states=states,
                                          object_type=_by_name(object_type_cls, str(a["track_type"][i])),
                                          category=category_cls(int(a["track_cat"][i]))))
        return cls(str(a["scenario_id"]), str(a["focal_track_id"]), tracks, str(a["city"]))


def load_argoverse_scenario_parquet(path, object_type_cls=ObjectType, category_cls=TrackCategory):
    """Scenario parquet (one row per (track, timestep)) -> Scenario; tracks sorted by track_id (string order)."""
    import pyarrow.parquet as pq

    tbl = pq.read_table(str(path)).to_pydict()
    n = len(tbl["track_id"])
    by_type = {e.value: e for e in object_type_cls}
    groups = {}
    for i in range(n):
        groups.setdefault(tbl["track_id"][i], []).append(i)
    tracks = []
    for tid in sorted(groups):
        idx = groups[tid]
        states = [SimpleNamespace(observed=bool(tbl["observed"][j]), timestep=int(tbl["timestep"][j]),
                                  position=(float(tbl["position_x"][j]), float(tbl["position_y"][j])),
                                  heading=float(tbl["heading"][j]),
                                  velocity=(float(tbl["velocity_x"][j]), float(tbl["velocity_y"][j])))
                  for j in idx]
        j0 = idx[0]
        tracks.append(SimpleNamespace(track_id=str(tid), object_states=states,
                                      object_type=by_type[tbl["object_type"][j0]],
                                      category=category_cls(int(tbl["object_category"][j0]))))
    return Scenario(str(tbl["scenario_id"][0]), str(tbl["focal_track_id"][0]), tracks,
                    str(tbl["city"][0]) if "city" in tbl else "")


def save_scene(path, static_map, scenario, meta=None):
    arrays = {}
    arrays.update(static_map.to_arrays())
    arrays.update(scenario.to_arrays())
    for k, v in (meta or {}).items():
        arrays["meta_" + k] = np.array(v)
    np.savez_compressed(path, **arrays)


def load_scene(path, lane_type_cls=LaneType, lane_mark_cls=LaneMarkType, object_type_cls=ObjectType,
               category_cls=TrackCategory):
    """-> (StaticMap, Scenario, meta dict) from a compact scene file written by save_scene."""
    with np.load(path, allow_pickle=False) as z:
        a = {k: z[k] for k in z.files}
    meta = {k[5:]: a[k].item() for k in a if k.startswith("meta_")}
    return (StaticMap.from_arrays(a, lane_type_cls, lane_mark_cls),
            Scenario.from_arrays(a, object_type_cls, category_cls), meta)
