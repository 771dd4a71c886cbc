"""planners/ilqr call surface of the reference (solver.iLQR, cost.TreeCost, potential.*, dynamics, utils)
backed by the HIP tree-iLQR kernels of libmind_hip.so -- no host arithmetic: every value these classes
return is computed on the MI355X (mind_ilqr_solve_fields / mind_cost_eval / mind_lane_dist_field)."""
