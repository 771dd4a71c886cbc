"""Dynamics descriptor for the solver (reference planners/ilqr/dynamics.py + the model of
planners/mind/trajectory_tree.py:153-177).  The reference compiles the six update expressions with
Theano; here the same kinematic bicycle is part of the kernel (ilqr_kernels.hip: il_dyn_sc, Jacobian in
il_node_derivs), so the object only carries its parameters."""


class Dynamics:
    state_size = 6
    action_size = 2
    has_hessians = False


class BicycleDynamics(Dynamics):
    """x = [x, y, v, yaw, a, delta], u = [jerk, delta_rate]:
    x += v cos(yaw) dt; y += v sin(yaw) dt; v += a dt; yaw += v / wb * tan(delta) dt; a += u0 dt; delta += u1 dt."""

    def __init__(self, dt=0.2, wheelbase=2.5):
        self.dt = dt
        self.wheelbase = wheelbase
