import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import dsp_map_amd as dsp
from oracle import oracle_py as orc
from tests import common
from tests.test_gpu_parity import _setup_update_scene, gpu_state
f32=np.float32
o, m, pts, q, n = _setup_update_scene(dsp, orc, 31, 0, nx=50, ny=50, nz=24, ppv=12)
o.bin_points(pts, q); m.bin_points(pts, q)
o.predict(-0.01, 0.0, 0.002, 1 / 30.0); m.predict(-0.01, 0.0, 0.002, 1 / 30.0)
# snapshot pre-update
vo, so, ro = o.export_sparse()
pl = o.pyramid_lists.copy()
o.map_update(); m.map_update()
obs, cnt, ml, lam = m.observations()
oo = o.obs
worst=None
errs=[]
for b in np.nonzero(cnt)[0]:
    for j in range(cnt[b]):
        e=abs(obs[b,j,3]-oo[b,j,3])/oo[b,j,3]
        errs.append((e,b,j))
errs.sort(reverse=True)
print("top errors", errs[:8])
print("n obs with err>1e-5:", sum(1 for e in errs if e[0]>1e-5), "of", len(errs))
e,b,j = errs[0]
print("obs", b, j, oo[b,j], obs[b,j], "lambda", lam, o.L.dspo_expected_newborn(o.h))
# recompute Ck in float32 following reference for this obs
nb = o.neighbors[b]
lut = o.pdf_lut
P=o.particles
def q_pdf(x,mu,s=f32(0.1)):
    z=f32(f32(x-mu)/s)
    z=min(max(z,f32(-9.9)),f32(9.9))
    return lut[int(f32(f32(z*f32(1000))+f32(10000)))]
ck=f32(0); terms=[]
# pre-update weights from ro snapshot
wmap={(int(v),int(s)):r for v,s,r in zip(vo,so,ro)}
for k in range(nb[0]):
    bb=nb[k+1]
    for s_ in range(o.capp):
        if pl[bb,s_,0]&1:
            r=wmap[(int(pl[bb,s_,1]),int(pl[bb,s_,2]))]
            g=f32(f32(q_pdf(r[4],oo[b,j,0])*q_pdf(r[5],oo[b,j,1]))*q_pdf(r[6],oo[b,j,2]))
            t=f32(f32(f32(0.95)*r[7])*g)
            ck=f32(ck+t); terms.append((float(t),bb,r[4:8].tolist()))
ck=f32(ck+f32(f32(lam)+f32(0.01)))
print("numpy recompute", ck, " oracle", oo[b,j,3], " gpu", obs[b,j,3], "nterms", len(terms))
terms.sort(reverse=True)
print("largest terms", terms[:5])
print("n particles in nbhd lists", len(terms), "fov count gpu", m.counters()["n_fov"], "oracle", int((pl[:,:,0]&1).sum()))
vo2, so2, ro2 = o.export_sparse()
vg, sg, rg = gpu_state(m)
print("live", len(vo2), len(vg), m.counters())
ko = {(int(v),)+tuple(r[4:7].tolist()):r for v,s,r in zip(vo2,so2,ro2)}
kg = {(int(v),)+tuple(r[4:7].tolist()):r for v,s,r in zip(vg,sg,rg)}
pre = {(int(v),)+tuple(r[4:7].tolist()):r for v,s,r in zip(vo,so,ro)}
print("keys only oracle", len(set(ko)-set(kg)), "only gpu", len(set(kg)-set(ko)))
bad=[]
for k in ko:
    if k in kg:
        a=ko[k]; b=kg[k]
        if abs(a[7]-b[7])>1e-4*abs(a[7]): bad.append((k,a[7],b[7],pre[k][7]))
print("n bad w", len(bad))
for x in bad[:10]:
    k=x[0]
    print(x, "pyr h,v", o.L.dspo_pyramid_h(o.h, k[1],k[2],k[3]), o.L.dspo_pyramid_v(o.h, k[1],k[2],k[3]), "inarea", o.L.dspo_in_pyramids_area(o.h, k[1],k[2],k[3]), "dist", np.sqrt(k[1]**2+k[2]**2+k[3]**2))
