"""bench.py's one JSON line (the driver's contract): the keys it must carry, the two extra objects, sane values -- on the real
workload with few steps, and the N = 2 form started bare on one GPU (the ranks are child processes bench.py starts itself)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline")


def run_bench(*args, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout: %r" % p.stdout[-500:]
    return json.loads(lines[0])


def check_roofline(r, blocked, lists=False):
    """`bound` names the roof that binds (instruction issue for launches of several substeps, HBM for one substep per launch) and
    equals `binding_roof`; `frac` = `achieved` / peak comes from the measured bytes of a committed PMC pass when there is one, and is
    then no larger than the engine's own byte model allows (halo lines that several tiles gather are served from HBM once)."""
    assert r["bound"] == r["binding_roof"] == ("valu_issue" if blocked else "hbm"), r
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0.0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["frac_model"] <= 1.0 and r["bytes"] in ("model", "measured (PMC)")
    if r["bytes"] == "measured (PMC)" and lists:   # (with contacts the model is a floor: it prices beams and particles, not the list walks)
        assert r["traffic"] > 0 and r["frac_model"] <= r["frac_measured"] <= 1.0 and r["frac"] == r["frac_measured"]
    elif r["bytes"] == "measured (PMC)":
        assert r["traffic"] > 0 and 0.0 < r["frac_measured"] <= r["frac_model"] * 1.05 and r["frac"] == r["frac_measured"]
    else:
        assert r["frac"] == r["frac_model"]


def test_default_line_carries_the_contract():
    d = run_bench("--steps", "60", "--warmup", "12", "--cpu-seconds", "2")
    for k in KEYS:
        assert k in d, k
    assert d["metric"] == "particle-steps/sec" and d["unit"] == "particle-steps/s" and d["n_gpus"] == 1 and d["steps"] == 60
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1_000_000 * 60 / (d["ms_per_step"] * 60 * 1e-3)) / d["value"] < 1e-6
    check_roofline(d["roofline"], blocked=True)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["value"] > 1.0e10                                   # BASELINE's target for config 2
    ex = d["extra"]
    assert ex["upload_ms"] > 0 and ex["readback_ms"] > 0 and ex["config3"]["value"] > 1.0e10 and ex["config3"]["finite"]
    # uploads that keep the plan (r04: also when beams were only removed): both kept, both far below the upload that plans
    ru = ex["reupload"]
    assert ru["plan_kept"] == [2, 1] and ru["beams_cut"] > 20000 and ru["same_topology_ms"] < 40.0 and ru["beams_cut_ms"] < 40.0 < ex["upload_ms"]
    check_roofline(ex["config3"]["roofline"], blocked=False, lists=True)
    assert ex["config3"]["hash_schedule"]["grid_helper_launches"] == 0      # every hash pushed by the substep kernels themselves
    assert "r0" in ex["config3"]["contacts"] and "config3_contacts_check.txt" in ex["config3"]["contacts"]
    assert ex["single_substep_kernel"]["frac_of_hbm_peak"] > 0.4
    # the steady-state figure DESIGN.md quotes rides in the same record (never `value`), and so do one GPU's shares of configs 4 and 5
    assert ex["steady_state"]["steps"] == 960 and ex["steady_state"]["value"] > 1.0e10
    assert ex["config3"]["steady_state"]["steps"] == 960
    # ... and the same scene with the engine's default collision mode (hash on): mostly blocked launches, far above the hash-only rate
    dm = ex["default_collision_mode"]
    assert dm["substeps"] == 960 and dm["substeps_in_blocked_launches"] > 800 and dm["value"] > 5.0e10
    for k, particles in (("config4_share", 500 * 4000), ("config5_share", 1000 * 8000)):
        assert ex[k]["particles_total"] == particles and ex[k]["value"] > 1.0e10
        check_roofline(ex[k]["roofline"], blocked=True)


def test_two_ranks_with_the_drivers_arguments():
    """`--steps 20 --warmup 5` is what the driver runs: the timed region must hold a ghost refresh (at a fixed depth of 30 it
    used to hold none), and the N > 1 line carries `roofline` and `cpu_baseline` like the N = 1 line."""
    d = run_bench("--gpus", "2", "--rehearse-one-gpu", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2", timeout=1200)
    assert d["n_gpus"] == 2 and d["value"] > 1.0e10 and "REHEARSAL" in d["data"]
    x = d["config"]["exchange"]
    assert x["exchanges_in_timed_region"] >= 1 and x["ghost_depth"] == 20 and x["exchange_us_avg"] > 0 and x["transport"] == "peer"
    check_roofline(d["roofline"], blocked=True)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and "share" in c
    for k, per_gpu in (("config4", 500 * 4000), ("config5", 1000 * 8000)):
        e = d["extra"][k]
        assert e["n_gpus"] == 2 and e["particles_total"] == 2 * per_gpu and e["exchange"]["exchanges_in_timed_region"] >= 1
        check_roofline(e["roofline"], blocked=True)
        assert e["value"] > 1.0e10
    # the main scene with the engine's default collision mode on both ranks (r04: blocked launches between the ghost refreshes)
    dm = d["extra"]["default_collision_mode"]
    assert dm["n_gpus"] == 2 and dm["value"] > 1.0e10 and dm["hybrid"]["substeps_in_blocked_launches"] > 0


def test_two_ranks_long_region_keeps_the_depth_asked_for():
    d = run_bench("--gpus", "2", "--rehearse-one-gpu", "--steps", "120", "--warmup", "30", "--no-extra", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["value"] > 1.0e10 and "REHEARSAL" in d["data"]
    assert d["config"]["exchange"]["ghost_depth"] == 24 and d["config"]["exchange"]["exchanges_in_timed_region"] == 5
