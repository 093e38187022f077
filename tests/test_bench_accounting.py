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


def test_ceiling_table_prices_every_kernel_with_its_own_ruler():
    """bench.py: saturated_132x132x60.ceilings: sweeps at the measured skeleton rate on their PMC bytes, k_place at the
    scattered-store rate, pair kernels at VALU issue 1.0, latency chains at >= one dependent launch; the frame at its ceilings is
    faster than the frame measured, and no kernel beats its own ceiling."""
    b = _bench()
    tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    kus = {"k_predict": 247.5, "k_place": 94.8, "k_resample": 97.9, "k_ck_partial": 62.3, "k_weight": 53.1, "k_pyr_prepare": 36.8,
           "k_birth_insert": 15.7, "k_obs_points": 7.2, "k_spin": 21.0}
    pmc = {"k_predict": {"hbm_bytes": 966481836}, "k_resample": {"hbm_bytes": 358908746}, "k_pyr_prepare": {"hbm_bytes": 42801419},
           "k_birth_insert": {"hbm_bytes": 12578555}, "k_obs_points": {"hbm_bytes": 613662}}
    sq = {"k_ck_partial": {"valu_issue": 0.50}, "k_weight": {"valu_issue": 0.61}}
    V, T = 132 * 132 * 60, 6
    ct = b.ceiling_table(C_SAT, b.b_alg(C_SAT, V, T), 0.6584, kus, pmc, sq, 5.3)
    pk = ct["per_kernel"]
    assert "k_spin" not in pk and set(kus) - {"k_spin"} == set(pk)
    assert abs(pk["k_predict"]["ceiling_us"] - 966481836 / 5.3e12 * 1e6) < 0.1
    assert abs(pk["k_weight"]["ceiling_us"] - 53.1 * 0.61) < 0.1 and "VALU" in pk["k_weight"]["ruler"]
    assert "scattered stores" in pk["k_place"]["ruler"] and 60 < pk["k_place"]["ceiling_us"] < 90
    assert pk["k_obs_points"]["ceiling_us"] == 5.0                      # a latency chain never prices below one dependent launch
    assert all(v["achieved_over_ceiling"] >= 1.0 for v in pk.values())
    assert ct["sum_of_ceilings_us"] < ct["sum_of_kernels_us"] <= 0.6584e3
    assert ct["frac_of_8TBps_now"] < ct["frac_of_8TBps_if_every_kernel_sat_on_its_ceiling"] < 1.0
    json.dumps(ct)
    # without PMC / SQ data (a run on kernel sources nobody profiled yet) the table still comes out, on algorithmic bytes
    ct2 = b.ceiling_table(C_SAT, b.b_alg(C_SAT, V, T), 0.6584, kus, {}, {}, 5.3)
    assert "algorithmic" in ct2["per_kernel"]["k_predict"]["ruler"] and ct2["per_kernel"]["k_weight"]["achieved_over_ceiling"] == 1.0


def test_projection_helpers_balance_slabs_and_sum_the_slowest_slab_per_phase():
    """bench.py: projected_8gpu: slab boundaries from per-layer costs (contiguous, every slab at least one layer,
    equal costs -> equal heights) and the critical path = sum over phases of the slowest slab + a rank's share of the list selection"""
    b = _bench()
    assert b.balanced_ranges([1.0] * 80, 8) == [(10 * i, 10 * i + 10) for i in range(8)]
    cost = [1.0] * 30 + [2.0] * 20 + [1.0] * 30
    r = b.balanced_ranges(cost, 8)
    assert r[0][0] == 0 and r[-1][1] == 80 and all(a[1] == c[0] for a, c in zip(r, r[1:])) and all(z1 > z0 for z0, z1 in r)
    sums = [sum(cost[z0:z1]) for z0, z1 in r]
    assert max(sums) - min(sums) <= 2.0 and max(sums) < max(sum(cost[10 * i:10 * i + 10]) for i in range(8))
    assert b.balanced_ranges([1, 1, 1, 1, 1, 1, 1, 100], 8) == [(i, i + 1) for i in range(8)]
    tab = [[1.0, 0.1, 0.5, 0.0, 0.3, 0.2, 1.0], [1.2, 0.1, 0.4, 0.0, 0.6, 0.2, 0.9], [0.0, 0.05, 0.0, 0.8, 0.02, 0.01, 0.0]]
    crit, per = b.critical_path(tab, 2, True)
    # segments between collectives: begin | exchange + place + Ck | weights | finish
    assert [round(x, 6) for x in per] == [1.2, 1.1, 0.2, 1.0] and abs(crit - (sum(per) + 0.4)) < 1e-9
    assert abs(b.critical_path(tab, 2, False)[0] - sum(per)) < 1e-9
