"""Process-wide handle on the HIP context (one ``mind_ctx`` per process / GPU)."""
import os

_default = None


def get_runtime(device=None):
    """The shared ``HipPredictor`` (predictor + tree-iLQR entry points).  Raises without a GPU / library:
    the product path has no CPU fallback."""
    global _default
    if _default is None:
        from .predictor import HipPredictor
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        _default = HipPredictor(device)
    return _default


def reset_runtime():
    global _default
    if _default is not None:
        _default.close()
    _default = None
