#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the softbody physics step on MI355X.

One "step" = one substep = one `compute_update` (compute.wgsl:90-203) over the whole scene.
Workload at N=1: BASELINE config 2 -- one 1000x1000 lattice blob, 1 000 000 particles /
2 996 001 beams, fp32, wide (v2) layout, collisions off, subticks 64 (dt = 1/64), scene already
resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 1000 --warmup 64
    python bench.py --gpus N ...          (starts N ranks itself, as child processes, and relays rank 0's line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0) with the driver's fields plus `roofline`, `cpu_baseline` (at every N) and `extra`:
at N=1 the single-substep kernel, BASELINE config 3, a steady-state figure (960 substeps after 64) and one GPU's share of
configs 4 and 5; at N>1 configs 4 (N slabs of 500 x 4000) and 5 (N slabs of 1000 x 8000, mixed stiffness, dt 1/128)
measured in the same run.  `--config4` / `--config5` make those shapes the main workload.

N>1: every rank times its own substep launches and ghost refreshes with HIP events on its engine's stream (sb_mark), the
timed region starts on a refresh boundary and holds at least one refresh whatever --steps is (the ghost depth is lowered to
a divisor of --steps when the region is shorter than four refresh periods), and the line says how many there were.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--width", type=int, default=None, help="lattice columns per GPU (default 1000)")
    ap.add_argument("--height", type=int, default=None, help="lattice rows (default 1000)")
    ap.add_argument("--collisions", choices=["off", "grid"], default="off")
    ap.add_argument("--spacing", type=float, default=30.0, help="lattice spacing d (particle radius is 10)")
    ap.add_argument("--origin-y", type=float, default=1000.0, help="y of the lattice's bottom row (10 = resting on the floor)")
    ap.add_argument("--config3", action="store_true",
                    help="BASELINE config 3 as the main workload: the blob pile of scenes.config3_buffers (lattice blobs "
                         "resting on the floor and on each other, ~1 M particles, spatial-hash collisions); without this "
                         "flag the same scene is measured after the main workload and reported under `extra`")
    ap.add_argument("--lattice-on-floor", action="store_true",
                    help="round 1's config-3 stand-in: a 4000x250 lattice at spacing 22 resting on the floor (floor "
                         "response and 8 collision candidates per particle, but no pair closer than 2r in the window)")
    ap.add_argument("--path", choices=["auto", "atomic", "tiled"], default="auto")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--block-substeps", type=int, default=0,
                    help="collisions off: substeps per launch of the temporally blocked kernel (0 = engine default, 1 = off)")
    ap.add_argument("--ghost-depth", type=int, default=24,
                    help="N>1: ghost-zone depth in lattice columns = substeps between halo exchanges (24: four launches of six, and "
                         "an interior slab of 1000 + 2 x 24 columns still fits two rounds of the blocked kernel's 512 resident tiles "
                         "of at most 1024 particles -- at 30 it takes three: 16.2 instead of 13.4 us per substep)")
    ap.add_argument("--subticks", type=int, default=None, help="default 64 (128 with --config5)")
    ap.add_argument("--soup", action="store_true",
                    help="config 3 with EVERY mechanism acting: width x height FREE particles (no beams) on a grid of "
                         "--spacing (default here 40) jittered by +-10, thrown around at up to --soup-speed units/s: "
                         "they fall, bounce off the floor and the walls and hit each other all the time "
                         "(single GPU, spatial-hash collisions)")
    ap.add_argument("--soup-speed", type=float, default=60.0)
    ap.add_argument("--mixed-stiffness", action="store_true",
                    help="BASELINE config 5: springs drawn from {1,3,50,500}, use with --subticks 128")
    ap.add_argument("--rest-lengths", choices=["lattice", "current"], default="lattice",
                    help="current: every beam rests at the present distance of its (jittered) endpoints, as beams made in the "
                         "reference's editor do -- no two rest lengths alike, so the engine runs its material mode 1")
    ap.add_argument("--exchange", choices=["peer", "stream", "sync"], default="peer",
                    help="N>1: ghost refresh by direct stores into the neighbours' IPC-mapped mailboxes (default; "
                         "falls back to 'stream' if the mappings cannot be set up or fail their check), by RCCL "
                         "send/recv ordered on the engine stream, or by host-synchronised RCCL")
    ap.add_argument("--grid-skin", type=float, default=0.0, help="spatial-hash skin (0 = engine default)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N>1 on a single-GPU box: every rank uses cuda:0 and the process group is gloo (control "
                         "plane only), so the multi-rank code path and the peer exchange can be exercised; not a measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the records of `extra`")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE config 4 as the main workload: one 500 x 4000 slab per GPU (8 GPUs: the 4000 x 4000 lattice, "
                         "16 M particles), ghost-halo exchange between neighbours")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE config 5 as the main workload: one 1000 x 8000 slab per GPU (8 GPUs: 64 M particles), springs "
                         "drawn from {1,3,50,500}, subticks 128")
    ap.add_argument("--steady-steps", type=int, default=960, help="N=1: substeps of extra.steady_state (after 64 of warm-up)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (this process has not
    touched the GPU and never will), relay rank 0's JSON line, fail if any rank fails."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = 0
    for line in child.stdout:
        if line.startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(line)
    rc = child.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write("bench.py: the ranks printed %d result lines, expected 1\n" % lines)
        rc = 1
    return rc


def oracle_rate(orc, buf, bounds, mode, threads, budget_s, subticks):
    """particle-steps/s of the oracle on `buf` for about budget_s seconds (at least 2 substeps)."""
    ref = orc.OracleEngine(bounds, 10.0, subticks, buf.layout, mode, threads=threads)
    ref.write_buffers(buf)
    t0 = time.perf_counter()
    ref.step(2)
    per = (time.perf_counter() - t0) / 2
    n = max(2, min(2000, int(budget_s / max(per, 1e-6)) // 2 * 2))
    t0 = time.perf_counter()
    ref.step(n)
    dt = time.perf_counter() - t0
    return dict(value=buf.particle_count * n / dt, substeps=n, seconds=dt, threads=threads)


def cpu_baseline(buf, bounds, mode, budget_s, subticks=64):
    """Times the oracle (our C restatement of compute.wgsl; the reference has no CPU path) on
    the same scene for a bounded number of substeps.  Reported beside the GPU, never the target."""
    import __graft_entry__ as ge
    orc = ge.load_oracle()
    orc.build()
    # the GPU box grants about 16 cores per GPU; more OpenMP threads than that only thrash
    cores = min(16, len(os.sched_getaffinity(0)))
    one = oracle_rate(orc, buf, bounds, mode, 1, budget_s * 0.35, subticks)
    al = oracle_rate(orc, buf, bounds, mode, cores, budget_s * 0.65, subticks)
    return {
        "value": al["value"], "unit": "particle-steps/s", "cores": al["threads"], "kind": "port",
        "sample": "oracle/sb_oracle.c (C restatement of compute.wgsl; the reference has no CPU path), same "
                  "scene, %d substeps in %.1f s with %d OpenMP threads; 1 thread: %.3g particle-steps/s over %d substeps"
                  % (al["substeps"], al["seconds"], al["threads"], one["value"], one["substeps"]),
        "single_thread_value": one["value"],
    }


def cpu_baseline_config3(sb, buf, bounds, budget_s):
    """BASELINE.md 3, rows "cfg 3 CPU": the oracle's spatial-hash mode on the config-3 scene (1 thread, all
    cores) and the reference's own all-pairs scan (compute.wgsl:142-170) at its own limit of 65 536 particles."""
    import __graft_entry__ as ge
    orc = ge.load_oracle()
    orc.build()
    cores = min(16, len(os.sched_getaffinity(0)))
    g1 = oracle_rate(orc, buf, bounds, orc.COLLIDE_GRID, 1, budget_s * 0.25, 64)
    gn = oracle_rate(orc, buf, bounds, orc.COLLIDE_GRID, cores, budget_s * 0.4, 64)
    small, sbounds = sb.scenes.config3_buffers(65536)
    ap = oracle_rate(orc, small, sbounds, orc.COLLIDE_ALLPAIRS, cores, budget_s * 0.35, 64)
    return {"kind": "port", "unit": "particle-steps/s", "cores": cores,
            "grid_1M_all_cores": gn["value"], "grid_1M_one_thread": g1["value"],
            "allpairs_65536_all_cores": ap["value"],
            "sample": "oracle grid mode on the same %d-particle pile: %d substeps / %.1f s on %d threads, %d substeps / %.1f s "
                      "on 1 thread; oracle all-pairs (the reference's O(P^2) scan) on a %d-particle pile: %d substeps / %.1f s "
                      "on %d threads" % (buf.particle_count, gn["substeps"], gn["seconds"], cores, g1["substeps"], g1["seconds"],
                                         small.particle_count, ap["substeps"], ap["seconds"], cores)}


def committed_traffic(workload, kernel, key="bench"):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/rNN_summary.json, made by tools/gpu_profile.sh + tools/summarize_profiles.py from
    separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, FETCH_SIZE x2 per the
    gfx950 correction and our own calibration run).  None unless it was taken on this workload."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            s = json.load(open(f))
            if s.get("bench", {}).get("config", {}).get("workload") != workload:
                continue
            table = s.get("traffic", {}) if key == "bench" else s.get("traffic_" + key, {})
            cand = [(max(v["launches"] for v in t.values() if isinstance(v, dict)), t["hbm_bytes_per_launch"])
                    for name, t in table.items() if name.startswith(kernel + "<")]
            if cand:  # the variant launched most often (the lean substep; the last one of a call also stores strain/stress)
                best = (max(cand)[1], os.path.basename(f))
        except Exception:
            continue
    return best


def newest_profile(suffix):
    """Name of the newest committed profiles/rNN_<suffix> (the records are named per round)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + suffix)))
    return os.path.basename(found[-1]) if found else "(none committed)"


def committed_traffic_any(table_key, kernel):
    """HBM bytes per launch of `kernel` from table `table_key` of the newest committed summary (whatever workload that
    table was taken on: its name says), as (bytes, file) or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            table = json.load(open(f)).get(table_key, {})
            cand = [(max(v["launches"] for v in t.values() if isinstance(v, dict)), t["hbm_bytes_per_launch"])
                    for name, t in table.items() if name.split("<")[0] == kernel]
            if cand:
                best = (max(cand)[1], os.path.basename(f))
        except Exception:
            continue
    return best


def committed_valu(workload, kernel):
    """VALU instructions per launch of `kernel` (whole waves, SQ_INSTS_VALU) from the newest committed SQ-counter pass of this
    command (profiles/rNN_summary.json `sq_counters`), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
        try:
            s = json.load(open(f))
            if s.get("bench", {}).get("config", {}).get("workload") != workload:
                continue
            cand = [(t["SQ_INSTS_VALU"]["launches"], t["SQ_INSTS_VALU"]["per_launch"]) for name, t in s.get("sq_counters", {}).items()
                    if name.startswith(kernel + "<") and "SQ_INSTS_VALU" in t]
            if cand:
                best = (max(cand)[1], os.path.basename(f))
        except Exception:
            continue
    return best


def peer_exchanger(halo, eng, plan, buf, dist, torch, ctl):
    """Direct neighbour exchange (sb_peer_*), agreed on by ALL ranks or by none: every rank sets its
    mailbox up, maps its neighbours', runs one refresh on the freshly uploaded state (where it must
    reproduce the ghost zone bit for bit) and the ranks vote.  None = use the RCCL transport."""
    import numpy as np
    world = dist.get_world_size()

    def vote(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    ex, why = None, None
    try:
        ex = halo.PeerExchanger(eng, plan, timeout_ms=20000)
    except Exception as exc:  # noqa: BLE001 -- any failure means "use RCCL"
        why = "mailbox: %r" % (exc,)
    cards = [None] * world
    dist.all_gather_object(cards, ex.card if ex is not None else None)
    if ex is not None and all(c is not None for c in cards):
        try:
            ex.connect(cards)
        except Exception as exc:  # noqa: BLE001
            why = "connect: %r" % (exc,)
    ok = vote(ex is not None and ex.connected)
    if ok:
        try:
            ex.exchange()
            eng.sync()
            out = eng.load_buffers(buf.copy())
            gp, _, gb, _ = plan.lists()
            same = np.array_equal(out.particles[gp].view("u4"), buf.particles[gp].view("u4"))
            for f in ("target_length", "last_length"):
                same = same and np.array_equal(out.beams[f][gb].view("u4"), buf.beams[f][gb].view("u4"))
            if not same:
                why = "check: the refreshed ghost zone differs from the owners' state"
            ok = same
        except Exception as exc:  # noqa: BLE001
            why, ok = "check: %r" % (exc,), False
        ok = vote(ok)
    if ok:
        # ... and then in motion: three refresh periods (the third reuses the first receive buffer), after which
        # every ghost record must equal its owner's, compared over torch.distributed (not over the mailboxes)
        try:
            ex.step(3 * plan.depth)
            eng.sync()
            bad = halo.ghost_mismatches(plan, eng.load_buffers(buf.copy()), dist, torch, torch.device(ctl))
            if bad:
                why = "check: %d ghost records differ from their owners after three refreshes" % bad
            ok = bad == 0
        except Exception as exc:  # noqa: BLE001
            why, ok = "check: %r" % (exc,), False
        ok = vote(ok)
    if why:
        print("[bench rank %d] direct peer exchange unavailable (%s)" % (dist.get_rank(), why), file=sys.stderr, flush=True)
    dist.barrier()
    return ex if ok else None


def roofline(eng, kernel_ms, steps, P_local, B_local, workload):
    """`achieved` = the ALGORITHMIC bytes of the launched kernel per substep -- its own compulsory HBM traffic, every array
    element the substep has to read or write with the data layout the engine holds (sb_get_info "substep_hbm_bytes") -- / the
    HIP-event time per substep, as a fraction of the 8 TB/s peak.  `traffic` is what the PMC counters of the committed profile
    measured per launch: equal to the model for the single-substep kernel; BELOW it for the blocked kernel (halo lines that
    several tiles gather are charged to each of them by the model and served once from HBM, then by L2), so the object also
    carries `achieved_measured` / `frac_measured`, priced with the measured bytes.  The blocked kernel is not bound by HBM
    but by instruction issue: `binding_roof` says so and `valu_issue` holds that fraction.  The reference-layout figure of
    SURVEY.md 8(d) (52 B per beam + 48 B per particle) is an equivalent rate, not a fraction: no kernel here moves those bytes."""
    per_substep_s = kernel_ms * 1e-3 / steps
    own = float(eng.info("substep_hbm_bytes"))
    k = max(1, eng.info("substeps_per_launch"))
    tiled = eng.info("path") == 2
    hybrid_bytes = float(eng.info("hybrid_substep_hbm_bytes"))
    if hybrid_bytes:
        # SB_COLLIDE_GRID engine that ran blocked launches while nothing was about to touch: most substeps moved
        # the blocked kernel's bytes, not the single-substep kernel's.  The shares are those of the engine's whole life since
        # the upload (warm-up included); the launches are k_substep_blocked<..., TRACK> + k_hybrid_validate.
        done, blocked = float(eng.info("substeps_done")), float(eng.info("hybrid_substeps"))
        share = blocked / done if done else 0.0
        own = share * hybrid_bytes + (1.0 - share) * own
    ach = own / per_substep_s / 1e9
    binding = "valu_issue" if (k > 1 or hybrid_bytes) else "hbm"
    # `achieved` / `frac`: HBM bytes per second against the 8 TB/s peak -- from the MEASURED bytes of the committed PMC passes
    # when there are any (then `achieved_model` / `frac_model` keep the engine's own byte model), else from the model.
    # `bound` names the roof that binds this kernel: for the blocked kernel that is instruction issue
    # (`valu_issue` below holds that fraction), and the HBM fraction says how far from the OTHER roof it runs.
    roof = {"bound": binding, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "achieved_model": ach, "frac_model": ach / HBM_PEAK_GBS, "bytes": "model",
            "traffic": None,
            "kernel": eng.kernel_name() if tiled else "k_beams_atomic+k_particles",
            "substeps_per_launch": k, "avg_launch_us": per_substep_s * k * 1e6,
            "compulsory_bytes_per_launch": own * k,
            "binding_roof": binding,
            "reference_layout_bytes_per_substep": 52.0 * B_local + 48.0 * P_local,
            "reference_layout_equiv_GBps": (52.0 * B_local + 48.0 * P_local) / per_substep_s / 1e9,
            "note": "achieved = compulsory bytes of the launched kernel (engine's own data layout: %d beam copies for %d "
                    "beams, material mode %d, %d substeps per launch of a long call, plan depth %d) / HIP-event time of the "
                    "substep launches of the timed region (a short call is cut into balanced launches, which may be deeper "
                    "than a long call's); reference_layout_equiv_GBps = (52*B + 48*P) / the same time is what a kernel "
                    "streaming the reference's records would need to sustain for this step rate -- an equivalence, not a "
                    "fraction of the roof"
                    % (eng.info("beam_copies"), B_local, eng.info("material_mode"), k, eng.info("plan_depth"))}
    if hybrid_bytes:
        roof["kernel"] = "k_substep_blocked<TRACK> + k_hybrid_validate (%.0f %% of the substeps), %s (the rest)" % (100.0 * share, roof["kernel"])
        roof["substeps_per_launch"] = eng.info("hybrid_substeps_per_launch")
        roof["avg_launch_us"] = None
        roof["compulsory_bytes_per_launch"] = None
        roof["note"] = ("achieved = compulsory bytes per substep, weighted by the share of substeps that ran in blocked launches "
                        "(%d per launch) and in single substeps, / HIP-event time per substep of the timed region" % roof["substeps_per_launch"])
        return roof
    tr = committed_traffic(workload, roof["kernel"].split("<")[0])
    if tr:
        roof["traffic"] = tr[0]
        roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc passes of this command)" % tr[1]
        roof["traffic_over_compulsory"] = tr[0] / (own * k)
        roof["achieved_measured"] = tr[0] / (per_substep_s * k) / 1e9
        roof["frac_measured"] = roof["achieved_measured"] / HBM_PEAK_GBS
        roof["achieved"], roof["frac"], roof["bytes"] = roof["achieved_measured"], roof["frac_measured"], "measured (PMC)"
    if k > 1:
        # the temporally blocked kernel is bound by instruction issue, not by HBM: the second roof it is
        # measured against.  1024 SIMDs issue one wave64 fp32 instruction per 2.8 cycles with four waves each (tools/valu_rate.hip)
        va = committed_valu(workload, roof["kernel"].split("<")[0])
        if va:
            simds, clock_hz, cycles_per_inst = 1024, 2.4e9, 2.8
            roof["valu_issue"] = {"insts_per_launch": va[0], "source": "profiles/%s sq_counters (SQ_INSTS_VALU)" % va[1],
                                  "frac_of_issue_peak": va[0] * cycles_per_inst / (simds * clock_hz * per_substep_s * k),
                                  "note": "wave instructions x 2.8 cycles / (1024 SIMDs x 2.4 GHz x launch time); the launch's "
                                          "load/store phase issues almost nothing"}
    return roof


def measure_single_substep(sb, a, buf, bounds, workload):
    """The same workload with ONE substep per launch (k_substep_tiled): the HBM-bound form of the
    force-accumulate kernel, whose compulsory bytes the PMC counters reproduce; reported beside the blocked kernel
    of the headline, which trades HBM traffic for redundant ring work and is not HBM-bound."""
    eng = sb.Engine(bounds_size=bounds, particle_radius=10.0, subticks=a.subticks, layout=2, max_particles=buf.max_particles,
                    max_beams=buf.max_beams, collision_mode=0, block_substeps=1)
    eng.write_buffers(buf)
    eng.step(a.warmup)
    eng.sync()
    ms = eng.step_timed(a.steps)
    eng.sync()
    own = float(eng.info("substep_hbm_bytes"))
    per = ms * 1e-3 / a.steps
    rec = {"kernel": eng.kernel_name(), "us_per_substep": per * 1e6, "value": buf.particle_count * a.steps / (ms * 1e-3),
           "unit": "particle-steps/s", "compulsory_bytes_per_launch": own, "achieved_GBps": own / per / 1e9,
           "frac_of_hbm_peak": own / per / 1e9 / HBM_PEAK_GBS}
    tr = committed_traffic(workload, "k_substep_tiled", key="bench_single_substep")
    if tr:
        rec["traffic"] = tr[0]
        rec["traffic_source"] = "profiles/%s" % tr[1]
    eng.destroy()
    return rec


def measure_steady_state(sb, a, buf, bounds, mode):
    """The main workload once more at the step counts DESIGN.md quotes (default 960 substeps after 64 of warm-up), so that
    the driver's own record holds the steady-state rate beside the one its short protocol gives.  Never `value`."""
    eng = sb.Engine(bounds_size=bounds, particle_radius=10.0, subticks=a.subticks, layout=2, max_particles=buf.max_particles,
                    max_beams=buf.max_beams, collision_mode=mode, block_substeps=a.block_substeps, grid_skin=a.grid_skin)
    eng.write_buffers(buf)
    eng.step(64)
    eng.sync()
    t0 = time.perf_counter()
    ms = eng.step_timed(a.steady_steps)
    eng.sync()
    wall = time.perf_counter() - t0
    own = float(eng.info("substep_hbm_bytes"))
    per = ms * 1e-3 / a.steady_steps
    rec = {"steps": a.steady_steps, "warmup": 64, "kernel": eng.kernel_name(),
           "value": buf.particle_count * a.steady_steps / wall, "value_by_device_time": buf.particle_count / per,
           "unit": "particle-steps/s", "us_per_substep": per * 1e6, "substeps_per_launch": eng.info("substeps_per_launch"),
           "compulsory_GBps": own / per / 1e9, "frac_of_hbm_peak": own / per / 1e9 / HBM_PEAK_GBS,
           "note": "same scene and engine options as the driver line, %d substeps after 64: the rate a long run sustains "
                   "(the driver's protocol times %d substeps after %d)" % (a.steady_steps, a.steps, a.warmup)}
    eng.destroy()
    return rec


def measure_default_mode(sb, a, buf, bounds):
    """The main scene once more with the engine's DEFAULT collision mode (spatial hash on -- the reference always collides,
    compute.wgsl:142-170): nothing touches in config 2's lattice, so the engine runs blocked launches while the closest listed
    pair stays clear of 2r.  Same step counts as `steady_state`.  Never `value`."""
    eng = sb.Engine(bounds_size=bounds, particle_radius=10.0, subticks=a.subticks, layout=2, max_particles=buf.max_particles,
                    max_beams=buf.max_beams, collision_mode=2, block_substeps=a.block_substeps, grid_skin=a.grid_skin)
    t0 = time.perf_counter()
    eng.write_buffers(buf)
    upload_ms = (time.perf_counter() - t0) * 1e3
    eng.step(64)
    eng.sync()
    done0, blocked0 = eng.info("substeps_done"), eng.info("hybrid_substeps")
    t0 = time.perf_counter()
    ms = eng.step_timed(a.steady_steps)
    eng.sync()
    wall = time.perf_counter() - t0
    rec = {"steps": a.steady_steps, "warmup": 64, "value": buf.particle_count * a.steady_steps / wall, "unit": "particle-steps/s",
           "us_per_substep": ms * 1e3 / a.steady_steps, "upload_ms": upload_ms,
           "substeps_in_blocked_launches": eng.info("hybrid_substeps") - blocked0, "substeps": eng.info("substeps_done") - done0,
           "launches_refused": eng.info("hybrid_failed"), "grid_builds": eng.info("grid_builds"),
           "note": "collision_mode = SB_COLLIDE_GRID (sb_default_options); the timed region includes the looks, conversions, "
                   "validations and the single substeps around every hash rebuild"}
    eng.destroy()
    return rec


def measure_config3(sb, a):
    """BASELINE config 3 in the same run (rank 0, N=1): the blob pile, spatial-hash collisions, same step counts."""
    import numpy as np
    buf, bounds = sb.scenes.config3_buffers()
    eng = sb.Engine(bounds_size=bounds, particle_radius=10.0, subticks=64, layout=2, max_particles=buf.max_particles,
                    max_beams=buf.max_beams, collision_mode=2, grid_skin=a.grid_skin)
    t0 = time.perf_counter()
    eng.write_buffers(buf)
    upload_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(sb.scenes.CONFIG3_SETTLE_FRAMES):   # the pile comes to rest on itself (untimed; delete passes included)
        eng.frame()
    eng.sync()
    settled = eng.load_buffers(buf.copy())              # what the CPU baseline below starts from, too
    builds0 = eng.info("grid_builds")
    model_bytes = float(eng.info("substep_hbm_bytes"))
    eng.step(a.warmup)
    eng.sync()
    sched0 = {k: eng.info(k) for k in ("grid_aborts", "grid_helper_launches", "grid_classic_substeps")}
    ms = eng.step_timed(a.steps)
    eng.sync()
    builds = eng.info("grid_builds") - builds0
    schedule = {k: eng.info(k) - v for k, v in sched0.items()}
    steady = None
    if a.steady_steps > a.steps:                        # ... and at the step counts DESIGN.md quotes (never `value`)
        b0 = eng.info("grid_builds")
        ms2 = eng.step_timed(a.steady_steps)
        steady = {"steps": a.steady_steps, "value": buf.particle_count * a.steady_steps / (ms2 * 1e-3),
                  "us_per_substep": ms2 * 1e3 / a.steady_steps, "grid_builds": eng.info("grid_builds") - b0}
    out = eng.load_buffers(buf.copy())
    eng.destroy()
    P = buf.particle_count
    v = np.hypot(out.particles[:P, 2], out.particles[:P, 3])
    rec = {"workload": sb.scenes.CONFIG3_TEXT % (P, buf.beam_count, bounds),
           "value": P * a.steps / (ms * 1e-3), "unit": "particle-steps/s", "us_per_substep": ms * 1e3 / a.steps,
           "steps": a.steps, "warmup": a.warmup, "grid_builds": builds, "upload_ms": upload_ms,
           "finite": bool(np.isfinite(out.particles[:P]).all()), "max_speed": float(v.max()),
           "beams_left": out.beam_count, "steady_state": steady,
           "hash_schedule": dict(schedule, note="helper launches (k_grid_build) and substeps of the classic schedule inside the timed region: "
                                                 "0 / 0 = every hash was pushed by the substep kernels themselves (lagged schedule, DESIGN 4.4)"),
           "contacts": "tools/config3_contacts_check.py measures the share of particles the collision loop changes "
                       "(profiles/%s)" % newest_profile("config3_contacts_check.txt")}
    # measured HBM bytes per substep of the two kernels of this scene, from the committed PMC passes of `bench.py --config3`
    # its own roofline object (HBM-bound: one substep per launch).  Bytes: the PMC passes of `bench.py --config3` committed under
    # profiles/ (since r04 the substep kernel is the only launch of a substep: the hash's helper launch is gone, sb_physics.h
    # SbGridCtl); the engine's byte model (beams, particles, halo: no neighbour lists) beside it.
    per = ms * 1e-3 / a.steps
    tr = committed_traffic_any("traffic_config3", "k_substep_tiled_grid")
    roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "k_substep_tiled_grid", "substeps_per_launch": 1,
            "avg_launch_us": per * 1e6, "achieved_model": model_bytes / per / 1e9, "frac_model": model_bytes / per / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "binding_roof": "hbm",
            "model_note": "the engine's byte model prices beam copies, particles and halo entries; the neighbour-list walks of this scene "
                          "(list lengths, entries, the positions and velocities they gather) come on top: the measured bytes are the larger"}
    if tr:
        roof.update({"traffic": tr[0], "achieved": tr[0] / per / 1e9, "frac": tr[0] / per / 1e9 / HBM_PEAK_GBS, "bytes": "measured (PMC)",
                     "achieved_measured": tr[0] / per / 1e9, "frac_measured": tr[0] / per / 1e9 / HBM_PEAK_GBS,
                     "traffic_source": "profiles/%s traffic_config3 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --config3; "
                                       "neighbour lists and position gathers included, average over the launches of the profiled run)" % tr[1]})
    else:
        roof.update({"achieved": roof["achieved_model"], "frac": roof["frac_model"], "bytes": "model"})
    rec["roofline"] = roof
    if not a.no_cpu_baseline:
        rec["cpu_baseline"] = cpu_baseline_config3(sb, settled, bounds, a.cpu_seconds)
    return rec


def effective_depth(want, steps, width):
    """Substeps between ghost refreshes = ghost-zone depth in columns.  A timed region of at least four refresh periods takes
    the depth asked for; a shorter one (the driver times 20 substeps) takes the largest depth <= the one asked for that
    DIVIDES --steps, so that the region -- which starts on a refresh boundary -- holds whole refresh periods and ends on a
    refresh: steps / depth exchanges are inside it, never zero."""
    d = max(1, min(want, width))
    if steps >= 4 * d:
        return d
    for c in range(min(d, max(steps, 1)), 0, -1):
        if steps % c == 0:
            return c
    return 1


class Ctx:
    pass


def run_workload(ctx, a, W, H, subticks, mixed, mode, steps, warmup, named=None, want_cpu=True, cpu_seconds=None,
                 readback=False):
    """One lattice workload on every rank: scene -> engine -> exchanger -> warm-up -> timed region -> verify.  All ranks
    call it together; returns the record (rank 0 fills it in, the others get the shared numbers too)."""
    sb, halo, torch, dist = ctx.sb, ctx.halo, ctx.torch, ctx.dist
    rank, world, local, ctl = ctx.rank, ctx.world, ctx.local, ctx.ctl
    d = a.spacing
    bounds = float(max(W * world, H) * d + 2000.0)   # global scene: N slabs of W columns side by side (weak scaling)
    depth = effective_depth(a.ghost_depth, steps, W) if world > 1 else 0
    if world == 1:
        buf = sb.scenes.lattice_buffers(W, H, d=d, origin=(1000.0, a.origin_y), jitter=1.0, layout=2)
        plan = None
    else:
        buf, plan = halo.slab_scene(sb, rank, world, W, H, d=d, origin=(1000.0, a.origin_y), jitter=1.0, depth=depth)
    if mixed:
        if plan is not None:
            halo.mix_stiffness(buf, plan, subticks=subticks)   # keyed by global beam id: ghosts match their owners
        else:
            sb.scenes.mix_stiffness(buf, subticks=subticks)
    if a.rest_lengths == "current":
        sb.scenes.rest_at_current_length(buf)    # (slab scenes: positions are those of the whole lattice, so ghosts agree)
    P_local = buf.particle_count if plan is None else plan.n_owned
    B_local = buf.beam_count if plan is None else int(plan.owned_beams.size)
    eng = sb.Engine(bounds_size=bounds, particle_radius=10.0, subticks=subticks, layout=2,
                    max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=mode,
                    path={"auto": 0, "atomic": 1, "tiled": 2}[a.path], tile_particles=a.tile, device=local,
                    grid_skin=a.grid_skin, block_substeps=a.block_substeps)
    t0 = time.perf_counter()
    eng.write_buffers(buf)
    upload_ms = (time.perf_counter() - t0) * 1e3
    exchange_mode, transport, ex = None, None, None
    if plan is not None:
        ex = peer_exchanger(halo, eng, plan, buf, dist, torch, ctl) if a.exchange == "peer" else None
        if ex is not None:
            exchange_mode = "direct stores into the neighbours' IPC-mapped mailboxes, flag handshake on the engine stream"
        else:
            if a.rehearse_one_gpu:
                sys.exit("--rehearse-one-gpu: the peer exchange could not be set up and RCCL cannot share one GPU")
            if a.exchange == "peer":
                eng.sync_quiet()
                eng.write_buffers(buf)          # back to the uploaded state, mailboxes released
            transport = halo.TorchTransport(torch, dist, torch.device("cuda", local), eng.stream(),
                                            ordered=a.exchange != "sync")
            ex = halo.Exchanger(eng, plan, transport)

    def barrier():
        eng.sync()                 # the engine runs on its own HIP stream
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    timer = halo.StepTimer(eng)
    timer.prepare(1 if ex is None else 2 * (steps // max(depth, 1) + 2))   # (event creation stays out of the timed region)
    if ex is None:
        eng.step(warmup)
    else:
        ex.step(warmup)
        ex.refresh_now()           # untimed: the timed region starts on a refresh boundary, the next refresh is `depth` substeps in
        ex.exchanges = 0
        ex.timer = timer
    barrier()
    t0 = time.perf_counter()
    if ex is None:
        timer.run("step", eng.step, steps)   # HIP events on the engine's own stream, no wait
    else:
        ex.step(steps)
    barrier()
    wall = time.perf_counter() - t0
    tot = timer.totals()
    kernel_ms = tot.get("step", (0, 0.0))[1]
    n_exch, exch_ms = tot.get("exchange", (0, 0.0))
    if ex is not None:
        ex.timer = None
        ex.verify()       # refuses a halo run in which beams were removed outside Exchanger.frame()
    if dist is not None:
        t = torch.tensor([wall, kernel_ms, exch_ms], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kernel_ms, exch_ms = (float(x) for x in t.tolist())
        cnt = torch.tensor([P_local], dtype=torch.float64, device=ctl)
        dist.all_reduce(cnt)
        P_total = int(cnt.item())
    else:
        P_total = P_local
    placed = "" if (d == 30.0 and a.origin_y == 1000.0) else ", spacing %g, bottom row at y=%g" % (d, a.origin_y)
    if a.lattice_on_floor:
        placed += " (resting on the floor: floor response active, 8 collision candidates per particle tested " \
                  "every substep, none closer than 2r in the timed window)"
    cfg = named or ("3" if mode == 2 else "2")
    workload = ("BASELINE config %s: %dx%d lattice blob per GPU, %d particles / %d beams per GPU, "
                "%s, jitter 1.0, subticks %d, collisions %s, v2 (u32) layout%s"
                % (cfg, W, H, P_local, B_local,
                   ("springs {1,3,50,500} (config 5 mix)" if mixed else "spring 50 damp 700") +
                   (", rest length = current distance (editor-style: one length per beam)" if a.rest_lengths == "current" else ""),
                   subticks, {0: "off", 2: "grid"}[mode], placed))
    rec = {"value": P_total * steps / wall, "unit": "particle-steps/s", "ms_per_step": wall * 1e3 / steps,
           "steps": steps, "warmup": warmup, "n_gpus": world, "workload": workload, "particles_total": P_total,
           "path": {1: "atomic", 2: "tiled"}[eng.info("path")], "tiles": eng.info("tiles"),
           "grid_builds": eng.info("grid_builds") if mode == 2 else None, "upload_ms": upload_ms,
           "kernel_us_per_substep": kernel_ms * 1e3 / steps,
           "roofline": roofline(eng, kernel_ms, steps, P_local, B_local, workload) if kernel_ms > 0 else None}
    if mode == 2:
        rec["hybrid"] = {"substeps_in_blocked_launches": eng.info("hybrid_substeps"), "substeps": eng.info("substeps_done"),
                         "launches_refused": eng.info("hybrid_failed"), "grid_aborts": eng.info("grid_aborts"),
                         "helper_launches": eng.info("grid_helper_launches"),
                         "note": "rank 0, since the upload (warm-up included): substeps that ran in tracked blocked launches beside the tiled "
                                 "layout (DESIGN 4.1b; engines with ghost zones too since r04)"}
    if world == 1:
        rec["parallelism"] = "single GPU"
    else:
        rec["parallelism"] = ("%d x-slabs of %d columns, ghost zones %d columns deep stepped redundantly, ghost p,v,a + "
                              "beam target/last refreshed every %d substeps (%s)"
                              % (world, W, depth, depth, exchange_mode or "RCCL neighbour send/recv, " + transport.mode))
        rec["exchange"] = {"ghost_depth": depth, "ghost_depth_asked": a.ghost_depth,
                           "exchanges_in_timed_region": ex.exchanges,
                           "exchange_us_avg": (exch_ms * 1e3 / n_exch) if n_exch else None,
                           "exchange_share_of_device_time": exch_ms / (exch_ms + kernel_ms) if (exch_ms + kernel_ms) > 0 else None,
                           "transport": "peer" if exchange_mode else "rccl (" + transport.mode + ")",
                           "note": "device time of sb_peer_exchange (pack into the neighbours' mailboxes, flag handshake, unpack) "
                                   "or of pack + RCCL send/recv + unpack, HIP events on each rank's engine stream, max over ranks; "
                                   "the timed region starts on a refresh boundary"}
    if readback and world == 1:
        out = buf.copy()                       # (the destination; its allocation is not the engine's time)
        t0 = time.perf_counter()
        eng.load_buffers(out)
        rec["readback_ms"] = (time.perf_counter() - t0) * 1e3
        if not a.no_extra:   # uploads that keep the plan: the state just read back (same topology), and that state with 1 % of the beams cut
            import numpy as np
            t0 = time.perf_counter()
            eng.write_buffers(out)
            same_ms = (time.perf_counter() - t0) * 1e3
            B, maxP = out.beam_count, out.max_particles
            keep = np.random.default_rng(1).random(B) >= 0.01
            cut = out.copy()
            recs = out.beams[out.mapping[maxP:maxP + B].astype(np.int64)][keep]
            cut.beams[:len(recs)] = recs
            cut.mapping[maxP:maxP + len(recs)] = np.arange(len(recs))
            cut.beam_count = len(recs)
            t0 = time.perf_counter()
            eng.write_buffers(cut)
            cut_ms = (time.perf_counter() - t0) * 1e3
            rec["reupload"] = {"same_topology_ms": same_ms, "beams_cut_ms": cut_ms, "beams_cut": int(B - len(recs)),
                               "plan_kept": [int(eng.info("uploads_kept")), int(eng.info("uploads_edited"))],
                               "note": "sb_write_buffers on the engine that holds the scene: the state just read back (same topology: only "
                                       "state travels), then that state with 1 % of its beams removed (the plan is kept, the removed beams "
                                       "die on the device); plan_kept = [uploads that kept the plan, of them uploads that removed beams]"}
    eng.destroy()
    rec["cpu_baseline"] = None
    if want_cpu and rank == 0 and not a.no_cpu_baseline:
        cb = cpu_baseline(buf, bounds, mode, cpu_seconds or a.cpu_seconds, subticks)
        if world > 1:
            cb["sample"] = ("rank 0's slab only -- ONE GPU's share of the scene, its %d ghost columns included (%d particles); "
                            % (depth, buf.particle_count)) + cb["sample"]
            cb["share"] = "per-GPU share (1 of %d slabs)" % world
        rec["cpu_baseline"] = cb
    rec["_buf"], rec["_bounds"] = buf, bounds
    if dist is not None:
        dist.barrier()
    return rec


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))
    ctx = Ctx()
    ctx.rank = rank = int(os.environ.get("RANK", "0"))
    ctx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    import torch
    import __graft_entry__ as ge
    ctx.torch = torch
    ctx.sb = sb = ge.load_package()
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU fallback)")
    if a.rehearse_one_gpu:
        local = 0
    ctx.local = local
    torch.cuda.set_device(local)
    ctx.dist = dist = None
    ctx.ctl = "cpu" if a.rehearse_one_gpu else "cuda"      # where the few control-plane tensors live
    if world > 1:
        import torch.distributed as dist
        ctx.dist = dist
        if a.rehearse_one_gpu:
            dist.init_process_group("gloo")
            a.exchange = "peer"
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from importlib import import_module
    ctx.halo = import_module("softbody_webgpu_amd.halo")

    shaped = a.width is not None or a.height is not None or a.subticks is not None   # an explicit shape: no config-4/5 extras
    named = None
    if a.config4:
        a.width, a.height, named = a.width or 500, a.height or 4000, "4 (one slab of the 4000-row lattice per GPU)"
    if a.config5:
        a.width, a.height, a.subticks, a.mixed_stiffness = a.width or 1000, a.height or 8000, a.subticks or 128, True
        named = "5 (one slab of the 8000-row lattice per GPU)"
    a.width, a.height, a.subticks = a.width or 1000, a.height or 1000, a.subticks or 64
    if a.lattice_on_floor:
        a.width, a.height, a.spacing, a.origin_y, a.collisions = 4000, 250, 22.0, 10.0, "grid"
    if a.soup:
        a.collisions = "grid"
        if a.spacing == 30.0:
            a.spacing = 40.0
        if a.origin_y == 1000.0:
            a.origin_y = 30.0       # bottom rows within reach of the floor
    if a.config3:
        a.collisions = "grid"
    mode = {"off": 0, "grid": 2}[a.collisions]

    if a.soup or a.config3:
        line = special_scene(ctx, a, mode)
    else:
        rec = run_workload(ctx, a, a.width, a.height, a.subticks, a.mixed_stiffness, mode, a.steps, a.warmup, named=named,
                           readback=True)
        buf, bounds = rec.pop("_buf"), rec.pop("_bounds")
        line = {
            "metric": "particle-steps/sec", "value": rec["value"], "unit": "particle-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": rec["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not a.rehearse_one_gpu else "synthetic; REHEARSAL: all ranks share one GPU, not a measurement",
            "config": {"workload": rec["workload"], "particles_total": rec["particles_total"], "path": rec["path"],
                       "tiles": rec["tiles"], "grid_builds": rec["grid_builds"], "parallelism": rec["parallelism"],
                       "kernel_us_per_substep": rec["kernel_us_per_substep"]},
            "roofline": rec["roofline"], "cpu_baseline": rec["cpu_baseline"],
        }
        if "exchange" in rec:
            line["config"]["exchange"] = rec["exchange"]
        extra = {}
        if world == 1:
            if rec.get("reupload"):
                extra["reupload"] = rec["reupload"]
            extra.update({"upload_ms": rec["upload_ms"], "readback_ms": rec.get("readback_ms"),
                          "upload_note": "sb_write_buffers / sb_load_buffers of the whole scene, host buffers <-> HBM, "
                                         "tiling and AoS<->SoA transposes included; never part of `value`"})
        if not a.no_extra:
            plain = not (named or shaped or a.mixed_stiffness or a.lattice_on_floor or a.rest_lengths != "lattice")
            if world == 1 and rank == 0:
                if mode == 0 and line["roofline"] and line["roofline"]["substeps_per_launch"] > 1:
                    extra["single_substep_kernel"] = measure_single_substep(sb, a, buf, bounds, rec["workload"])
                extra["steady_state"] = measure_steady_state(sb, a, buf, bounds, mode)
                if plain and mode == 0:
                    extra["default_collision_mode"] = measure_default_mode(sb, a, buf, bounds)
                if plain:
                    extra["config3"] = measure_config3(sb, a)
            del buf
            if plain and mode == 0 and world > 1:
                # the main scene once more with the engine's DEFAULT collision mode on every rank (the reference always collides):
                # what extra.default_collision_mode is at N = 1
                r = run_workload(ctx, a, a.width, a.height, a.subticks, False, 2, a.steps, a.warmup, want_cpu=False)
                r.pop("_buf"), r.pop("_bounds")
                extra["default_collision_mode"] = r
            if plain and mode == 0:
                # BASELINE configs 4 and 5 in the same run: at N GPUs, N slabs of each (at 8 GPUs these ARE configs 4 and 5;
                # below that, N GPUs' share of them), same step counts, their own exchanges, a short CPU baseline
                for key, (W, H, st, mix, nm) in (("config4", (500, 4000, 64, False, "4 (one slab of the 4000-row lattice per GPU)")),
                                                 ("config5", (1000, 8000, 128, True, "5 (one slab of the 8000-row lattice per GPU)"))):
                    r = run_workload(ctx, a, W, H, st, mix, 0, a.steps, a.warmup, named=nm, cpu_seconds=4.0)
                    r.pop("_buf"), r.pop("_bounds")
                    r["share_of_config"] = "%d of 8 slabs" % world
                    extra[key if world > 1 else key + "_share"] = r
        if extra:
            line["extra"] = extra
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def special_scene(ctx, a, mode):
    """--soup / --config3 as the main workload (single GPU): scenes that are not one lattice per GPU."""
    sb = ctx.sb
    if ctx.world != 1:
        sys.exit("--soup / --config3 are single-GPU scenes")
    W, H, d = a.width, a.height, a.spacing
    bounds = float(max(W, H) * d + 2000.0)
    if a.soup:
        buf = sb.scenes.soup_buffers(W, H, d=d, origin=(1000.0, a.origin_y), jitter=10.0, speed=a.soup_speed)
    else:
        buf, bounds = sb.scenes.config3_buffers()
    P, B = buf.particle_count, buf.beam_count
    eng = sb.Engine(bounds_size=bounds, particle_radius=10.0, subticks=a.subticks, layout=2,
                    max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=mode,
                    path={"auto": 0, "atomic": 1, "tiled": 2}[a.path], tile_particles=a.tile, device=ctx.local,
                    grid_skin=a.grid_skin, block_substeps=a.block_substeps)
    t0 = time.perf_counter()
    eng.write_buffers(buf)
    upload_ms = (time.perf_counter() - t0) * 1e3
    if a.config3:
        for _ in range(sb.scenes.CONFIG3_SETTLE_FRAMES):   # the pile comes to rest on itself (untimed)
            eng.frame()
        eng.sync()
    eng.step(a.warmup)
    eng.sync()
    ctx.torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = eng.step_timed(a.steps)
    eng.sync()
    wall = time.perf_counter() - t0
    if a.soup:
        workload = ("BASELINE config 3 as a particle soup: %dx%d FREE particles (no beams) on a grid of %g jittered "
                    "by +-10, velocities up to %g units/s, gravity, floor and walls, spatial-hash collisions, "
                    "subticks %d, v2 (u32) layout" % (W, H, d, a.soup_speed, a.subticks))
    else:
        workload = sb.scenes.CONFIG3_TEXT % (P, B, bounds)
    line = {
        "metric": "particle-steps/sec", "value": P * a.steps / wall, "unit": "particle-steps/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall * 1e3 / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "particles_total": P, "path": {1: "atomic", 2: "tiled"}[eng.info("path")],
                   "tiles": eng.info("tiles"), "grid_builds": eng.info("grid_builds"), "parallelism": "single GPU"},
        "roofline": roofline(eng, kernel_ms, a.steps, P, B, workload),
    }
    out = buf.copy()
    t0 = time.perf_counter()
    eng.load_buffers(out)
    line["extra"] = {"upload_ms": upload_ms, "readback_ms": (time.perf_counter() - t0) * 1e3}
    eng.destroy()
    line["cpu_baseline"] = None if a.no_cpu_baseline else cpu_baseline(buf, bounds, mode, a.cpu_seconds, a.subticks)
    return line


if __name__ == "__main__":
    main()
