"""What the memory system sustains for the prediction sweep's access pattern (dspmap_debug_sweep_probe): run on the GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_map_amd as D  # noqa: E402

m = D.DSPMap(D.make_config(nx=132, ny=132, nz=60, ppv=24, seed=1))
m.L.dspmap_init_device(m.h)
m.seed_uniform(24)
m.sync()
ms, b = C.c_float(), C.c_longlong()
for what, name in ((1, "read pos"), (7, "read pos+vel+w"), (15, "read pos+vel+w, write pos"), (11, "read pos+vel, write pos"),
                   (8, "write pos"), (4, "read w"), (2, "read vel")):
    for nb in (1, 2, 3, 6):
        m._chk(m.L.dspmap_debug_sweep_probe(m.h, what, 24, nb, 20, C.byref(ms), C.byref(b)))
        print("%-28s rows/batch %d: %.4f ms  %7.1f MB  %.2f TB/s" % (name, nb, ms.value, b.value / 1e6, b.value / ms.value / 1e9))
