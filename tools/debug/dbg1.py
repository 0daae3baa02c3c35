import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, helpers as H, test_gpu_parity as T
from helpers import O
d=np.load(H.GOLDEN/'grip_state_2k.npz'); state=d['state']
specs,ps=T._palm_scene(state,4)
ct=int(sys.argv[1]) if len(sys.argv)>1 else 1
cfg=H.sim_cfg(len(state),n_grid=64,dt=2e-4,ptype=0,collision_type=ct,precision='float64')
P=H.oracle_params(cfg,1e-3)
orc=H.OracleRollout(P,state,specs,ps).forward(1)
N=len(state); rng=np.random.default_rng(0)
seeds={1:(rng.standard_normal((N,3)),rng.standard_normal((N,3)),None,None)}
adj,pg,_=orc.backward(seeds,None)
sim,prims=H.build_engine(cfg,1e-3,specs,ps)
sim.reset(state); sim.substep(0)
sim.clear_grads(); sim.add_grad(1,gx=seeds[1][0],gv=seeds[1][1])
sim.substep_grad(0)
gx,gv,gF,gC=sim.get_grad_full(0)
ex=np.abs(gx-adj[0][0].numpy()).max(1); ev=np.abs(gv-adj[0][1].numpy()).max(1)
x=orc.frames[0][0]; pr=orc.prims_at(0)[0]
dist=O.prim_sdf(pr,x).numpy()
hit=dist<=5e-3
print('hits',hit.sum(),'max err gx on hits',ex[hit].max(),'non-hits',ex[~hit].max(),'scale',np.abs(adj[0][0].numpy()).max())
print('gv err hits',ev[hit].max(),'non',ev[~hit].max())
print('prim grad', prims[0].get_all_states_grad(0), pg[0][0])
i=np.argmax(ex); print(i, hit[i], gx[i], adj[0][0].numpy()[i])
