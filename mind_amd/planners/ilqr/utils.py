"""gen_dist_field of the reference (planners/ilqr/utils.py:5-22) on the device (k_lane_field)."""
import numpy as np

from ...runtime import get_runtime


def gen_dist_field(ego_pos, polyline, discrete_size, resolution):
    """-> (field_offset [2], xx [H,W], yy [H,W], distance_field [H,W]); discrete_size = (nx, ny)."""
    W, H = int(discrete_size[0]), int(discrete_size[1])
    off, gx, gy, dist = get_runtime().lane_dist_field(ego_pos, polyline, W, H, resolution)
    xx, yy = np.meshgrid(gx, gy)
    return off, xx, yy, dist
