import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'dsp-map_amd'))
import numpy as np, torch
import dsp_map_amd as D
scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
w=dict(nx=132, ny=132, nz=60, res=0.15, ppv=24)
name=sys.argv[1] if len(sys.argv)>1 else "C_sat"
if name.startswith("B"): w=dict(nx=66,ny=66,nz=40,res=0.15,ppv=24)
m=D.DSPMap(D.make_config(**w, seed=1234)); m.L.dspmap_init_device(m.h)
sc=scene_mod.CorridorScene(w["nx"]*w["res"], w["ny"]*w["res"], w["nz"]*w["res"], device="cuda")
if name.endswith("sat"): m.seed_uniform(w["ppv"],0.01,99)
nf = 12 if name.endswith("sat") else 70
for f in range(nf):
    pts,pos,q=sc.frame(f/30); m.update_device(pts.data_ptr(), pts.shape[0], pos, f/30, q); m.clearOccupancyMapPrediction()
m.sync()
P=m.pyramid_counts(); obs,cnt,ml,lam=m.observations()
nb=np.zeros(m.NP,np.int64)
for b in range(m.NP):
    h,v=divmod(b,16)
    for dh in (-1,0,1):
        for dv in (-1,0,1):
            if 0<=h+dh<28 and 0<=v+dv<16: nb[b]+=cnt[(h+dh)*16+v+dv]
print("P: sum",P.sum(),"max",P.max(),"mean",P.mean(),"; O(nbhd): max",nb.max(),"mean",nb.mean(), "; obs per bin max", cnt.max())
print("pairs per pass:", int((P.astype(np.int64)*nb).sum()), " max per pyramid", int((P.astype(np.int64)*nb).max()))
print(m.counters())
