import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import dsp_map_amd as D
scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
w=dict(nx=66,ny=66,nz=40,res=0.15,ppv=24)
for est in (0, 1):
    m=D.DSPMap(D.make_config(**w, seed=1234)); m.L.dspmap_init_device(m.h)
    m.set_param(D.capi.P_VELOCITY_ESTIMATOR, est)
    sc=scene_mod.CorridorScene(w["nx"]*w["res"], w["ny"]*w["res"], w["nz"]*w["res"], device="cuda")
    fr=[sc.frame(f/30) for f in range(390)]
    host=[(p.cpu().numpy().copy(), pos, q) for p,pos,q in fr]
    torch.cuda.synchronize()
    for f in range(90):
        p,pos,q=host[f]; m.update(p, pos, f/30, q); m.clearOccupancyMapPrediction()
    m.sync(); t0=time.perf_counter()
    for f in range(90,390):
        p,pos,q=host[f]; m.update(p, pos, f/30, q); m.clearOccupancyMapPrediction()
    m.sync(); dt=time.perf_counter()-t0
    print("host-buffer path, estimator", est, ":", round(300/dt,1), "frames/s", round(dt/300*1e3,4), "ms; points", host[-1][0].shape[0])
    m.close()
