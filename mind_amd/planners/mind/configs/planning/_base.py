"""Planning configuration values (reference planners/mind/configs/planning/demo_1.py:3-81;
demo_3 differs only in the velocity weight, demo_3.py:23,51)."""
import numpy as np


class ScenTreeCfg:
    def __init__(self):
        self.max_depth = 5
        self.tar_dist_thres = 10.0
        self.tar_time_ahead = 5.0
        self.seg_length = 15.0
        self.seg_n_node = 10
        self.far_dist_thres = 10.0


def _opt_block(w_vel):
    w_des = np.zeros((6, 6))
    w_des[2, 2], w_des[4, 4], w_des[5, 5] = w_vel, 1.0, 10.0
    w_con = np.zeros((6, 6))
    w_con[2, 2], w_con[4, 4], w_con[5, 5] = 50.0, 50.0, 500.0
    return {
        "w_des_state": w_des, "w_state_con": w_con,
        "state_upper_bound": np.array([100000.0, 100000.0, 8.0, 10.0, 4.0, 0.2]),
        "state_lower_bound": np.array([-100000.0, -100000.0, 0.0, -10.0, -6.0, -0.2]),
        "w_ctrl": 5.0 * np.eye(2), "w_tgt": 1.0, "smooth_grid_res": 0.4, "smooth_grid_size": (256, 256),
    }


class TrajTreeCfgBase:
    W_VEL = 0.1

    def __init__(self):
        self.dt = 0.2
        self.state_size = 6
        self.action_size = 2
        self.w_opt_cfg = _opt_block(self.W_VEL)
        self.opt_cfg = _opt_block(self.W_VEL)
        self.opt_cfg.update({"w_ego": 1.0, "w_ego_cov_offset": 1.0, "w_exo": 10.0, "w_exo_cov_offset": 2.5,
                             "w_exo_cost_offset": 10.0})
