"""Calibration of rocprofv3's WRITE_SIZE / FETCH_SIZE for the prediction sweep's OWN access pattern (12-byte position records, one per lane:
buffer_store_dwordx3, 768 contiguous bytes per wave) -- the calibration stream of profiles/collect.sh writes 4 bytes per lane.
Run under the profiler:  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/wc -- python tools/prof/write_calib.py
then profiles/pmc_reduce.py on the counter CSV: k_sweep_probe<2>'s WRITE_SIZE against the bytes printed here."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dsp_map_amd as D  # noqa: E402

m = D.DSPMap(D.make_config(nx=132, ny=132, nz=60, ppv=24, seed=1))
m.L.dspmap_init_device(m.h)
m.seed_uniform(24)
m.sync()
ms, b = C.c_float(), C.c_longlong()
for what, name in ((8, "write pos"), (1, "read pos")):
    m._chk(m.L.dspmap_debug_sweep_probe(m.h, what, 24, 2, 10, C.byref(ms), C.byref(b)))
    print("probe %d (%s): %d bytes per launch, %.4f ms" % (what, name, b.value, ms.value))
