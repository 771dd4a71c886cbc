"""CPU: the sin / cos / tan routine k_ilqr and the C oracle share (mind_amd/csrc/mind_trig.h), against x87 long-double functions.

The bicycle model of the contingency planner (planners/mind/trajectory_tree.py:153-177) is the only place of the path that calls
trigonometric functions; the reference evaluates them with numpy's.  The shared routine must stay within 1 ulp of the exact value
(sin, cos: 0.75 measured, tan: 0.9) on the arguments the planner can produce (headings and steering angles, a few radians) and far
beyond, and be exact where exactness is cheap (0, signs, symmetry)."""
import ctypes as C

import numpy as np
import pytest

from oracle import ilqr as oi


def _fns():
    lib = oi.lib()
    for f in (lib.oracle_sincos, lib.oracle_tan_cos):
        f.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        f.restype = None

    def sincos(x):
        s, c = C.c_double(), C.c_double()
        lib.oracle_sincos(float(x), C.byref(s), C.byref(c))
        return s.value, c.value

    def tancos(x):
        t, c = C.c_double(), C.c_double()
        lib.oracle_tan_cos(float(x), C.byref(t), C.byref(c))
        return t.value, c.value
    return sincos, tancos


def _ulps(got, ref):
    """|got - ref| in units of the last place of the double nearest to ref (ref: long double)"""
    ref = np.asarray(ref, dtype=np.longdouble)
    _, e = np.frexp(ref)
    u = np.ldexp(np.longdouble(1.0), e - 53)
    return np.abs((np.asarray(got, dtype=np.longdouble) - ref) / u).astype(np.float64)


@pytest.mark.skipif(np.finfo(np.longdouble).nmant < 63, reason="needs an extended-precision long double as the reference")
@pytest.mark.parametrize("span", [0.8, 3.2, 100.0, 1e5])
def test_shared_trig_routine_is_within_one_ulp(span):
    sincos, tancos = _fns()
    rng = np.random.default_rng(int(span * 10))
    x = (rng.random(40000) * 2 - 1) * span
    got = np.array([sincos(v) + tancos(v) for v in x])
    xl = x.astype(np.longdouble)
    es, ec, et = _ulps(got[:, 0], np.sin(xl)), _ulps(got[:, 1], np.cos(xl)), _ulps(got[:, 2], np.tan(xl))
    assert np.array_equal(got[:, 1], got[:, 3])          # the cosine that comes with the tangent is the same number
    assert es.max() < 0.8 and ec.max() < 0.8 and et.max() < 1.0, (es.max(), ec.max(), et.max())
    # against the platform's double functions (what numpy gives the reference): the last bit differs in a few per cent at most
    assert np.mean(got[:, 0] != np.sin(x)) < 0.03 and np.mean(got[:, 1] != np.cos(x)) < 0.03 and np.mean(got[:, 2] != np.tan(x)) < 0.05


def test_shared_trig_routine_exact_cases():
    sincos, tancos = _fns()
    assert sincos(0.0) == (0.0, 1.0) and tancos(0.0) == (0.0, 1.0)
    for v in (1e-300, 1e-9, 0.3, 0.7853981633974483, 1.0, 2.5, 7.0, 1234.5):
        s, c = sincos(v)
        sm, cm = sincos(-v)
        assert sm == -s and cm == c                       # odd / even to the bit
        t, _ = tancos(v)
        assert tancos(-v)[0] == -t
        assert abs(s * s + c * c - 1.0) < 4e-16
    assert sincos(1e-9)[0] == 1e-9 and tancos(1e-9)[0] == 1e-9
    assert all(np.isnan(sincos(np.nan))) and all(np.isnan(sincos(np.inf)))
