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
r=m.results()  # [V,4]
r=np.asarray(r).reshape(-1,4)
V=r.shape[0]; nt=V//64
t=r[:nt*64,3].reshape(nt,64)
t0=t[:,0]; d1=t[:,1]; d2=t[:,2]; d3=t[:,3]; d4=t[:,4]; rows=t[:,5]; ncp=t[:,6]
ne = d4>0
print("tiles",nt,"nonempty",ne.sum())
base=t0[ne].min()
start=(t0[ne]-base); end=start+d4[ne]
print("kernel span ticks(10ns):", end.max(), " start spread:", np.percentile(start,[0,50,90,100]))
print("block dur: pct", np.percentile(d4[ne],[0,50,90,99,100]))
i=np.argmax(d4*ne)
print("longest: rows",rows[i],"ncp",ncp[i],"T1",d1[i],"T2",d2[i],"T3",d3[i],"T4",d4[i], "start", t0[i]-base)
for q in (50,90,99):
    j=np.argsort(d4*ne)[int(nt*q/100)]
print("mean phases:", d1[ne].mean(), (d2-d1)[ne].mean(), (d3-d2)[ne].mean(), (d4-d3)[ne].mean(), "rows mean", rows[ne].mean(), "ncp mean", ncp[ne].mean())
big = ne & (rows>=40)
print("tiles rows>=40:", big.sum(), "phases:", d1[big].mean(), (d2-d1)[big].mean(), (d3-d2)[big].mean(), (d4-d3)[big].mean())
