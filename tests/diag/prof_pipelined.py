"""GPU-box diagnostic: cProfile of the one-thread event loop over four recorded scenes (mind_amd/pipelined.py), the timed part only."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import make_closed_loop, scene_workload
from mind_amd.pipelined import PipelinedClosedLoops

loops = [make_closed_loop(scene_workload("demo_all", i), scripted=False, speculative=False, own_context=True) for i in range(4)]
pc = PipelinedClosedLoops([l[1] for l in loops])
pc.run_plans(5)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
pc.run_plans(60)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(38)
print(s.getvalue()[:9000])
