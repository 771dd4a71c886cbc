from ._base import ScenTreeCfg, TrajTreeCfgBase  # noqa: F401


class TrajTreeCfg(TrajTreeCfgBase):
    W_VEL = 0.5
