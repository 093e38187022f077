import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import dsp_map_amd as D
m = D.DSPMap(D.make_config(nx=132, ny=132, nz=60, ppv=24, seed=1)); m.L.dspmap_init_device(m.h)
b = C.c_longlong()
for mode in (0, 1):
    for _ in range(3):
        m.L.dspmap_debug_stream(m.h, mode, C.byref(b))
    print("mode", mode, "bytes", b.value)
