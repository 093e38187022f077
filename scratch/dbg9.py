import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import dsp_map_amd as D
scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
w=dict(nx=66,ny=66,nz=40,res=0.15,ppv=24)
m=D.DSPMap(D.make_config(**w, seed=1234)); m.L.dspmap_init_device(m.h)
sc=scene_mod.CorridorScene(w["nx"]*w["res"], w["ny"]*w["res"], w["nz"]*w["res"], device="cuda")
fr=[sc.frame(f/30) for f in range(120)]
torch.cuda.synchronize()
for f in range(120):
    pts,pos,q=fr[f]; m.update_device(pts.data_ptr(), pts.shape[0], pos, f/30, q); m.clearOccupancyMapPrediction()
m.sync()
nt=(m.V+63)//64
out=np.zeros((nt,8),np.float32)
m.L.dspmap_debug_read_staging.restype=C.c_int
m.L.dspmap_debug_read_staging.argtypes=[C.c_void_p, C.c_void_p, C.c_int]
n=m.L.dspmap_debug_read_staging(m.h, out.ctypes.data_as(C.c_void_p), nt)
t=out[:n]
ne=t[:,4]>0
ne &= t[:,5]>0
print("tiles",n,"nonempty",ne.sum())
t0=t[ne,0]; base=t0.min(); st=t0-base; en=st+t[ne,4]
print("span(10ns ticks)", en.max(), "start pct", np.percentile(st,[0,50,90,100]))
print("dur pct", np.percentile(t[ne,4],[0,50,90,99,100]))
i=np.argmax(t[:,4]*ne)
print("longest tile: T1..T4", t[i,1:5], "rows",t[i,5],"nst",t[i,6],"nmv",t[i,7], "start", t[i,0]-base)
d=t[ne]
print("mean phases: pro %.0f loop %.0f tail1 %.0f tail2 %.0f epi %.0f" % (d[:,1].mean(), (d[:,2]-d[:,1]).mean(), (d[:,3]-d[:,2]).mean(), 0, (d[:,4]-d[:,3]).mean()))
big=ne & (t[:,5]>=40)
d=t[big]; print("rows>=40:", big.sum(), "pro %.0f loop %.0f tail1 %.0f tail2+epi %.0f  nst %.0f nmv %.0f" % (d[:,1].mean(), (d[:,2]-d[:,1]).mean(), (d[:,3]-d[:,2]).mean(), (d[:,4]-d[:,3]).mean(), d[:,6].mean(), d[:,7].mean()))
late=ne & (t[:,0]-base > 1500)
print("late starters (>15us):", late.sum())
