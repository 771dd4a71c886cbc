import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from test_aime_host import _full_tree_run
from oracle import ilqr as oi
from mind_amd.predictor import HipPredictor
hp = HipPredictor(0)
g,trees=_full_tree_run(True)
st=max(trees,key=lambda t:len(t.nodes)); nodes=[(k,n.parent_key,n.data) for k,n in st.nodes.items()]
flat=oi.flatten(nodes); M=len(flat["parent"])
lane=np.asarray(g.target_lane[::2],np.float64); d0=nodes[0][2][1][0,0]
x0=oi.init_state(np.array([float(d0[0]),float(d0[1]),4.0,0.0]), np.array([0.,0.]))
cfg=oi.default_cfg(max_iter=1)
xs,us,stt=hp.ilqr_solve(cfg,[flat],x0,lane,4.0,1)
rd=lambda k: hp.debug_read(k).view(np.float64)
L,Lx,Lxx,Fx=rd("il_L"),rd("il_Lx").reshape(M,6),rd("il_Lxx").reshape(M,36),rd("il_Fx").reshape(M,36)
# nominal (initial) rollout on the host: zero controls
xs0=np.zeros((M,6))
def f(x,u,dt=0.2,wb=2.5):
    return np.array([x[0]+x[2]*np.cos(x[3])*dt, x[1]+x[2]*np.sin(x[3])*dt, x[2]+x[4]*dt, x[3]+x[2]/wb*np.tan(x[5])*dt, x[4]+u[0]*dt, x[5]+u[1]*dt])
for i in range(M):
    p=flat["parent"][i]; xs0[i]=f(x0 if p<0 else xs0[p], np.zeros(2))
want=oi.node_derivs(cfg, flat, x0, lane, 4.0, 1, xs0, np.zeros((M,2)))
print("L   max diff %.3e" % np.abs(L-want["l"]).max(), "nodes differing", int((L!=want["l"]).sum()))
print("Lx  max diff %.3e" % np.abs(Lx-want["l_x"]).max(), int((Lx!=want["l_x"]).any(1).sum()))
print("Lxx max diff %.3e" % np.abs(Lxx-want["l_xx"].reshape(M,36)).max(), int((Lxx!=want["l_xx"].reshape(M,36)).any(1).sum()))
bad=np.nonzero(L!=want["l"])[0][:5]; print("first bad nodes", bad, [ (L[b], want["l"][b]) for b in bad])
