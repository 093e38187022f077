"""CPU tests of bench.py's byte accounting (the numbers behind the `roofline` block): a kernel is priced with the bytes IT
has to move, the per-kernel bytes never exceed the kernel's share of SURVEY 8(d)'s B_alg, and a fraction above the HBM peak
is never printed."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)   # (main() only runs under __main__)
    return mod


# device counters of a saturated 132x132x60 frame (profiles/r02_b) and of the metric's workload
C_SAT = dict(n_live_in=16963337, n_fov=501505, n_born=65316, n_obs=4531, n_moved=1926781, n_out_of_map=17676,
             n_voxel_full=0, n_pyramid_full=406, n_live_out=16937429)
B_RUN = dict(n_live_in=118698, n_fov=28874, n_born=18467, n_obs=1655, n_moved=15490, n_out_of_map=4,
             n_voxel_full=0, n_pyramid_full=0, n_live_out=118486)


def test_kernel_bytes_follow_what_each_kernel_touches():
    b = _bench()
    V, T = 132 * 132 * 60, 6
    assert b.kernel_alg_bytes("predict", C_SAT, V, T) == 2 * 32 * C_SAT["n_live_in"]
    # k_place is priced with the particles that changed voxel, not with every live one
    assert b.kernel_alg_bytes("claim", C_SAT, V, T) == 2 * 32 * C_SAT["n_moved"]
    assert b.kernel_alg_bytes("claim", C_SAT, V, T) < 0.12 * b.kernel_alg_bytes("predict", C_SAT, V, T)
    # the resampler: weights + velocities in, kept weights out, newborn records, per-voxel words -- far below 2 * 32 per particle
    rs = b.kernel_alg_bytes("resample", C_SAT, V, T)
    assert rs < 0.4 * 2 * 32 * C_SAT["n_live_in"]
    # and all the sweeps together stay below the frame's B_alg
    tot = sum(b.kernel_alg_bytes(k, C_SAT, V, T) for k in ("predict", "claim", "resample", "ck_partial", "weight"))
    assert tot < b.b_alg(C_SAT, V, T)


def test_no_fraction_above_the_peak_is_printed():
    b = _bench()
    V, T = 132 * 132 * 60, 6
    # honest times: every fraction is printed and below 1
    ok = b.roofline_block({"setup+bin": 0.008, "predict": 0.238, "claim": 0.129, "ck_partial": 0.077, "weight": 0.045,
                           "ck_finalize": 0.005, "birth": 0.025, "resample": 0.13}, C_SAT, V, T, 1, {}, "C_sat")
    assert ok["kernel"] == "k_predict" and 0.5 < ok["frac"] < 0.65
    assert all(0 <= v["frac"] <= 1 for v in ok["per_kernel"].values())
    # a timer that claims the prediction took 0.05 ms would mean 21 TB/s: rejected, never dominant, never a frac > 1
    bad = b.roofline_block({"predict": 0.05, "claim": 0.129, "ck_partial": 0.077, "weight": 0.045, "resample": 0.13},
                           C_SAT, V, T, 1, {}, "C_sat")
    assert bad["per_kernel"]["predict"]["frac"] is None and "error" in bad["per_kernel"]["predict"]
    assert bad["kernel"] != "k_predict" and bad["frac"] <= 1
    json.dumps(bad)   # the block stays serialisable


def test_traffic_comes_from_the_committed_pmc_summary():
    b = _bench()
    tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    db = tj["workloads"]
    V, T = 66 * 66 * 40, 6
    meta = {"commit": tj.get("commit"), "csrc_sha16": tj.get("csrc_sha16")}
    r = b.roofline_block({"predict": 0.04, "claim": 0.017, "ck_partial": 0.026, "weight": 0.018, "resample": 0.045},
                         B_RUN, V, T, 1, db, "B", traffic_meta=meta)
    # the stage's kernels: the resampler of this map size + the rollout behind it
    assert r["kernel"] == "k_resample" and r["traffic"] == sum(db["B"].get(k, {}).get("hbm_bytes", 0) for k in b.STAGE_KERNELS["resample"])
    assert r["traffic_frame"] >= (r["traffic"] or 0)
    # every kernel with a PMC figure also reports the bytes it really moved against the sustained copy rate of the part
    assert "frac_of_6.3TBps_on_pmc_bytes" in r["per_kernel"]["predict"] and 0 < r["per_kernel"]["predict"]["frac_of_6.3TBps_on_pmc_bytes"] < 1
    assert str(tj.get("csrc_sha16")) in r["traffic_source"]
    # the measured traffic of the dominant saturated kernel is close to what it has to move (no wasted re-reads)
    c = db["C_sat"]["k_predict"]["hbm_bytes"]
    assert 0.8 < c / b.kernel_alg_bytes("predict", C_SAT, 132 * 132 * 60, 6) < 1.3


def test_traffic_is_withheld_when_the_kernel_sources_changed():
    """profiles/pmc_traffic.json carries the fingerprint of the kernel sources its passes ran on (collect.sh) and the commit
    (pmc_merge.py); a bench run on other sources must not print those figures as its own"""
    b = _bench()
    tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert tj.get("csrc_sha16") and tj.get("commit") and len(b.csrc_fingerprint()) == 16
    V, T = 66 * 66 * 40, 6
    stale = {"stale": "profiles/pmc_traffic.json was measured on other kernel sources"}
    r = b.roofline_block({"predict": 0.04, "claim": 0.017, "ck_partial": 0.026, "weight": 0.018, "resample": 0.045},
                         B_RUN, V, T, 1, {}, "B", traffic_meta=stale)
    assert r["traffic"] is None and "other kernel sources" in r["traffic_source"] and "traffic_frame" not in r
