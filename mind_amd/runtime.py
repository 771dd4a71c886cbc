"""Handle on the HIP context: one ``mind_ctx`` per (process, thread).

A planner uses the runtime of the thread that built it.  One thread per scene, each under its own
``torch.cuda.stream``, gives independent contexts/streams whose kernels overlap on the device (BASELINE config 3:
several scenes planned concurrently on one MI355X); the common single-threaded case sees one runtime per process."""
import os
import threading

_local = threading.local()


def get_runtime(device=None):
    """This thread's ``HipPredictor`` (predictor + tree-iLQR entry points), bound to the thread's current torch
    stream when first requested.  Raises without a GPU / library: the product path has no CPU fallback."""
    rt = getattr(_local, "rt", None)
    if rt is None:
        from .predictor import HipPredictor
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        rt = _local.rt = HipPredictor(device)
    elif device is not None and int(device) != rt.device.index:
        # one context per thread: a second device in the same thread would silently run on the first one
        raise RuntimeError(f"this thread's HIP runtime is bound to cuda:{rt.device.index}, cuda:{int(device)} was requested; "
                           "use one host thread (or process) per device, or reset_runtime() first")
    return rt


def new_runtime(device=None):
    """A further ``HipPredictor`` of this thread: its own ``mind_ctx`` on its own (new) torch stream.  For a driver that plans several
    scenes from ONE thread and wants their kernels to overlap on the device (bench.py --concurrent P --pipelined: scene i's contingency
    solves run while scene i + 1's AIME rounds do); the caller owns it (``close()``)."""
    import torch
    from .predictor import HipPredictor
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", int(device))
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        rt = HipPredictor(int(device))
    rt.torch_stream = stream            # kept alive with the runtime; work the caller does through torch for this scene belongs on it
    return rt


def reset_runtime():
    rt = getattr(_local, "rt", None)
    if rt is not None:
        rt.close()
    _local.rt = None
