"""CPU tests of the drop-in boundary: the shared library loads, exports every symbol
include/dspmap.h declares, host-only entry points work, and compute entry points FAIL LOUDLY
without a GPU (there is no CPU fallback in the product)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dspmap.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dspmap_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound(dsp):
    names = declared_symbols()
    assert len(names) >= 45
    lib = dsp.load_library()
    out = subprocess.check_output(["nm", "-D", "--defined-only", dsp.capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (dspmap_[a-z_0-9]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    assert sorted(dsp.capi.SIGNATURES) == names  # the ctypes binding covers exactly the header
    for n in names:
        assert getattr(lib, n) is not None


def test_library_has_gfx950_code_object(dsp):
    blob = open(dsp.capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for k in (b"k_predict", b"k_place", b"k_ck_partial", b"k_weight", b"k_birth_insert", b"k_resample"):
        assert k in blob, k


def test_host_only_entry_points(dsp, orc):
    m = dsp.DSPMap(dsp.make_config(ppv=24))
    o = orc.Oracle(orc.make_config(ppv=24))
    assert (m.V, m.slots, m.NP, m.capp) == (o.V, o.slots, o.NP, o.capp)
    idx = C.c_int()
    x, y, z = C.c_float(), C.c_float(), C.c_float()
    rng = np.random.default_rng(0)
    for _ in range(2000):
        p = rng.uniform(-5.2, 5.2, 3).astype(np.float32)
        ok, i = m.getPointVoxelsIndexPublic(float(p[0]), float(p[1]), float(p[2]))
        ok_o = o.L.dspo_voxel_index(o.h, float(p[0]), float(p[1]), float(p[2]), C.byref(idx))
        assert ok == ok_o and (not ok or i == idx.value)
    for v in (0, 1, 66, 4356, 174239):
        o.L.dspo_voxel_center(o.h, v, C.byref(x), C.byref(y), C.byref(z))
        assert m.getVoxelPositionFromIndexPublic(v) == (x.value, y.value, z.value)
    assert m.L.dspmap_get_param(m.h, dsp.capi.P_OBSERVATION_STDDEV) == pytest.approx(0.1)
    assert m.L.dspmap_set_param(m.h, 999, 1.0) < 0
    assert b"unknown parameter" in m.L.dspmap_last_error(m.h)
    m.close(); o.close()


def test_create_rejects_bad_config(dsp):
    L = dsp.load_library()
    bad = dsp.make_config(nx=0)
    assert not L.dspmap_create(C.byref(bad))
    bad = dsp.make_config(ppv=65)
    assert not L.dspmap_create(C.byref(bad))
    bad = dsp.make_config(z_lo=30, z_hi=50)
    assert not L.dspmap_create(C.byref(bad))


def test_compute_fails_loudly_without_gpu(dsp):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    m = dsp.DSPMap()
    pts = np.zeros((4, 3), np.float32)
    with pytest.raises(dsp.capi.DSPMapError, match="no HIP device"):
        m.update(pts, (0, 0, 0), 0.0, (1, 0, 0, 0))
    with pytest.raises(dsp.capi.DSPMapError):
        m.getOccupancyMap(0.2)
    m.close()


def test_product_never_imports_oracle():
    """the product package must not reference oracle/ (a CPU fallback would void parity claims)"""
    pkg = os.path.join(ROOT, "dsp-map_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                t = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in t and "dsp_oracle" not in t and "dspo_" not in t, f


def test_dropin_header_compiles_without_ros(dsp):
    """include/dsp_dynamic.h (ours) offers the reference's DSPMap surface; the ROS-free twin of
    src/map_sim_example.cpp must compile and link against libdspmap_hip.so"""
    exe = os.path.join(ROOT, "examples", "map_example")
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "map_example.cpp"),
                           "-L" + os.path.join(ROOT, "dsp-map_amd", "lib"), "-ldspmap_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "dsp-map_amd", "lib"), "-o", exe])
    # the reference's two other headers as macro sets in front of the same header
    for macros in (["-DMAP_LENGTH_VOXEL_NUM=50", "-DMAP_WIDTH_VOXEL_NUM=50", "-DMAP_HEIGHT_VOXEL_NUM=30", "-DVOXEL_RESOLUTION=0.2",
                    "-DANGLE_RESOLUTION=1", "-DPYRAMID_NEIGHBOR_N=2", "-DMAX_PARTICLE_NUM_VOXEL=30", "-DDSPMAP_HALF_FOV_V=27",
                    "-DDSPMAP_OCCLUSION_MARGIN=VOXEL_RESOLUTION"],
                   ["-DMAP_LENGTH_VOXEL_NUM=50", "-DMAP_WIDTH_VOXEL_NUM=50", "-DMAP_HEIGHT_VOXEL_NUM=30", "-DVOXEL_RESOLUTION=0.2",
                    "-DMAX_PARTICLE_NUM_VOXEL=10", "-DDSPMAP_STATIC_MODEL=1", "-DDSPMAP_SAFE_PARTICLE_FACTOR=5",
                    "-DPREDICTION_TIMES=1", "-DDSPMAP_HALF_FOV_V=27", "-DDSPMAP_OCCLUSION_MARGIN=VOXEL_RESOLUTION"],
                   ["-DDSPMAP_WORLD=1"]):   # the sharded build of the same class (C++ RCCL driver)
        subprocess.check_call(["g++", "-std=c++14", "-Wall", "-fsyntax-only", "-I" + os.path.join(ROOT, "include")] + macros +
                              [os.path.join(ROOT, "examples", "map_example.cpp")])
    hdr = open(os.path.join(ROOT, "include", "dsp_dynamic.h")).read()
    for member in ("int update(int point_cloud_num, int size_of_one_point, float* point_cloud_ptr",
                   "void setPredictionVariance(float p_stddev, float v_stddev)", "void setObservationStdDev(",
                   "void setNewBornParticleWeight(", "void setNewBornParticleNumberofEachPoint(",
                   "void setParticleRecordFlag(", "static void setOriginalVoxelFilterResolution(",
                   "void getOccupancyMap(int& obstacles_num", "void getOccupancyMapWithFutureStatus(int& obstacles_num",
                   "void clearOccupancyMapPrediction()", "void getKMClusterResult(",
                   "void mapAddNewBornParticlesByObservation()", "static float generateRandomFloat(",
                   "void getVoxelPositionFromIndexPublic(", "int getPointVoxelsIndexPublic(", "void getFutureStatus("):
        assert member in hdr, member


def test_parameter_ids_of_the_binding_match_the_header(dsp):
    """every DSPMAP_P_* enumerator of include/dspmap.h has a P_* constant of the same value in the ctypes binding (a new
    parameter that is added to one side only would silently address another one), and the scheduling knobs are plain
    host state: settable and readable without a device"""
    text = open(os.path.join(ROOT, "include", "dspmap.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    ids = dict((k, int(v)) for k, v in re.findall(r"\bDSPMAP_P_([A-Z_0-9]+)\s*=\s*(\d+)", text))
    assert len(ids) >= 19 and len(set(ids.values())) == len(ids)
    for name, val in ids.items():
        assert getattr(dsp.capi, "P_" + name) == val, name
    m = dsp.DSPMap(dsp.make_config(ppv=24))
    for key in (dsp.capi.P_SPARSE_SWEEP, dsp.capi.P_ROLLOUT_INLINE):
        for v in (1, 0, -1):
            assert m.L.dspmap_set_param(m.h, key, float(v)) == 1
    m.close()


def test_rendezvous_file_is_matched_by_its_nonce_string(dsp, tmp_path, monkeypatch):
    """dspmap_mgpu_comm_init_from_env (the drop-in class's sharded start-up, -DDSPMAP_WORLD): a rank != 0 accepts exactly the record
    whose nonce STRING is this launch's -- whatever follows the string's NUL in the 120-byte field (rank 0's stack residue before
    the fix; zeros; anything) does not count; another launch's nonce, a wrong magic or a short file are refused"""
    L = dsp.load_library()
    monkeypatch.setenv("DSPMAP_RDZV_NONCE", "run-42")
    path = str(tmp_path / "rdzv")
    uid = bytes(range(128))
    out = C.create_string_buffer(128)
    # what rank 0 publishes is what the others wait for
    assert L.dspmap_debug_rdzv_publish(path.encode(), uid) == 1
    assert L.dspmap_debug_rdzv_wait(path.encode(), 50, out) == 1 and out.raw == uid
    nonce = ("run-42:%d" % os.getppid()).encode()
    assert open(path, "rb").read()[8:8 + len(nonce) + 1] == nonce + b"\0"
    # a hand-made record: correct string, garbage behind its NUL
    rec = b"DSPRDZV1" + nonce + b"\0" + b"\xa5" * (120 - len(nonce) - 1) + uid
    open(path, "wb").write(rec)
    out = C.create_string_buffer(128)
    assert L.dspmap_debug_rdzv_wait(path.encode(), 50, out) == 1 and out.raw == uid
    # ... zero-padded
    open(path, "wb").write(b"DSPRDZV1" + nonce.ljust(120, b"\0") + uid)
    assert L.dspmap_debug_rdzv_wait(path.encode(), 50, out) == 1
    # refused: another launch's nonce (also one that merely starts like ours), wrong magic, truncated file, no file
    for bad in (b"DSPRDZV1" + (nonce + b"7").ljust(120, b"\0") + uid,
                b"DSPRDZV1" + b"run-41:1".ljust(120, b"\0") + uid,
                b"DSPRDZV0" + nonce.ljust(120, b"\0") + uid,
                (b"DSPRDZV1" + nonce.ljust(120, b"\0") + uid)[:200]):
        open(path, "wb").write(bad)
        assert L.dspmap_debug_rdzv_wait(path.encode(), 30, out) == 0
    os.unlink(path)
    assert L.dspmap_debug_rdzv_wait(path.encode(), 30, out) == 0
    monkeypatch.setenv("DSPMAP_RDZV_NONCE", "run-43")                      # a later launch does not take the old file
    open(path, "wb").write(b"DSPRDZV1" + nonce.ljust(120, b"\0") + uid)
    assert L.dspmap_debug_rdzv_wait(path.encode(), 30, out) == 0


def test_rebuild_decision_follows_the_source_hash_not_the_clock(dsp, tmp_path, monkeypatch):
    """build_ext.needs_build() compares a sha256 of the sources / headers / flags with the one stored next to the .so when it was
    built: a tree that arrives by a `gpurun` push (fresh mtimes everywhere) is not rebuilt needlessly and -- the dangerous
    direction -- a .so that is NEWER than edited sources is not trusted."""
    import build_ext
    assert os.path.exists(build_ext.LIB) and os.path.exists(build_ext.STAMP)
    assert open(build_ext.STAMP).read().strip() == build_ext.source_hash() and not build_ext.needs_build()
    os.utime(os.path.join(build_ext.CSRC, build_ext.SOURCES[0]))          # a newer clock on a source changes nothing
    assert not build_ext.needs_build()
    monkeypatch.setattr(build_ext, "FLAGS", build_ext.FLAGS + ["-DX=1"])   # other flags = another library
    assert build_ext.needs_build()
    monkeypatch.undo()
    monkeypatch.setattr(build_ext, "STAMP", str(tmp_path / "missing"))     # no record of what the .so was built from
    assert build_ext.needs_build()


def test_reference_caller_type_checks_against_the_drop_in_header():
    """tools/check_reference_caller.sh: g++ -fsyntax-only on the reference's own src/map_sim_example.cpp, in place, against
    include/dsp_dynamic.h (declaration-only ROS / PCL / Eigen stubs in a temp dir).  Only where the reference tree exists (the build
    container); the GPU box has none."""
    import shutil
    import subprocess
    import pytest
    if not os.path.exists("/root/reference/src/map_sim_example.cpp") or not shutil.which("g++"):
        pytest.skip("no reference tree here")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "check_reference_caller.sh")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OK: the reference's caller type-checks" in r.stdout
