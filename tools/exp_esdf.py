"""dev: timing of 1 M uniformly random trilinear ESDF queries on a 256^3 lattice and of vigo_set_esdf (re-tiling into bricks)"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from trajectory_planner_amd import synth
from trajectory_planner_amd.vigo import Vigo
dev=torch.device("cuda",0)
v=Vigo(0)
n,res=256,0.1
dist,origin=synth.sphere_esdf(n,res,(0.3,-0.2,0.1),1.0)
v.set_esdf(torch.from_numpy(dist).to(dev),origin,res)
rng=np.random.default_rng(1)
pts=torch.from_numpy(rng.uniform(-12.7,12.7,size=(1<<20,3))).to(dev)
for name,p in (("random",pts),):
    for _ in range(5): v.esdf_query(p)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(50): v.esdf_query(p)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/50
    print(json.dumps({"case":name,"ms":dt*1e3}))
t0=time.perf_counter()
d=torch.from_numpy(dist).to(dev)
for _ in range(20): v.set_esdf(d,origin,res)
torch.cuda.synchronize(); print(json.dumps({"set_esdf_ms":(time.perf_counter()-t0)/20*1e3}))
