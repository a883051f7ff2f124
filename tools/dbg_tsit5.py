import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'oracle')); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, mgn_oracle as orc
import test_gpu_golden_and_partition as T
g, ps, cfg, eng, onehot = T._gold_d_problem()
dt=float(g["dt"]); saves=np.arange(11)*dt; inflow=g["inflow_mask"]
def f_dev(x,t):
    k=min(int(np.floor(t/dt+1e-6)), g["gt"].shape[0]-1)
    x[inflow]=g["gt"][k][inflow]
    return eng.ode_step(x.astype(np.float32), onehot, g["ef_raw"], g["val_mask"]).astype(np.float64)
ref,rst=orc.tsit5_rollout(f_dev,g["x0"],0.0,10*dt,saves)
sol,st=eng.rollout("Tsit5",g["x0"],onehot,g["ef_raw"],0.0,10*dt,dt,11,val_mask=g["val_mask"],inflow_mask=inflow[:,0],inflow_data=g["gt"])
print(rst, st)
for i in range(11): print(i, np.abs(sol[i]-ref[i]).max())
sol2,st2=eng.rollout("Tsit5",g["x0"],onehot,g["ef_raw"],0.0,10*dt,dt,11,val_mask=g["val_mask"])
f2=lambda x,t: eng.ode_step(x.astype(np.float32), onehot, g["ef_raw"], g["val_mask"]).astype(np.float64)
ref2,_=orc.tsit5_rollout(f2,g["x0"],0.0,10*dt,saves)
print("no inflow:", [float(np.abs(sol2[i]-ref2[i]).max()) for i in (1,5,10)], st2)
