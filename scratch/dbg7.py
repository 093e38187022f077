import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import dsp_map_amd as D
scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
w=dict(nx=66,ny=66,nz=40,res=0.15,ppv=24)
m=D.DSPMap(D.make_config(**w, seed=1234)); m.L.dspmap_init_device(m.h)
sc=scene_mod.CorridorScene(w["nx"]*w["res"], w["ny"]*w["res"], w["nz"]*w["res"], device="cuda")
fr=[sc.frame(f/30) for f in range(80)]
torch.cuda.synchronize()
for f in range(80):
    pts,pos,q=fr[f]; m.update_device(pts.data_ptr(), pts.shape[0], pos, f/30, q); m.clearOccupancyMapPrediction()
m.sync()
r=np.asarray(m.results()).reshape(-1,4)
nt=r.shape[0]//64
t=r[:nt*64,3].reshape(nt,64)
ne=t[:,4]>0
big=ne&(t[:,5]>=40)
d=t[big]
print("big tiles", big.sum(), "T1 %.0f  B0 %.0f B1 %.0f B2 %.0f  T2(pass1 end) %.0f  T3(pass2 end) %.0f  T4 %.0f  ncp %.1f" % (d[:,1].mean(), d[:,7].mean(), d[:,8].mean(), d[:,9].mean(), d[:,2].mean(), d[:,3].mean(), d[:,4].mean(), d[:,6].mean()))
i=np.argmax(t[:,4]*ne); print("longest", t[i,:10])
