"""Predictor hyper-parameters (values of the reference planners/mind/configs/networks/net_cfg.py:1-28)."""


class NetCfg:
    GLOBAL = {"g_num_modes": 6, "g_obs_len": 50, "g_pred_len": 60}

    def __init__(self):
        self.g_cfg = dict(self.GLOBAL)

    def get_net_cfg(self):
        cfg = {
            "network": "planners.mind.networks.network:ScenePredNet",
            "in_actor": 14, "d_actor": 128, "n_fpn_scale": 4,
            "in_lane": 16, "d_lane": 128,
            "d_rpe_in": 5, "d_rpe": 128, "d_embed": 128,
            "n_scene_layer": 6, "n_scene_head": 8, "dropout": 0.1, "update_edge": True,
            "param_out": "bezier",
        }
        cfg.update(self.g_cfg)
        return cfg
