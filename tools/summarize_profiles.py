#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (written by tools/gpu_profile.sh on the GPU box) into tracked files
under profiles/: the rocprofv3 --stats kernel table, the PMC traffic per launch corrected as
MI355X_MICROARCH.md "HBM" prescribes, and the counter calibration that justifies the correction."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join("gpurun_out", tag)
dst = "profiles"
if not os.path.isdir(src) or not glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    sys.exit("%s holds no profile run (tools/gpu_profile.sh %s did not run or did not finish): nothing summarised, "
             "profiles/ left as it is" % (src, tag))
os.makedirs(dst, exist_ok=True)


def one(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)  # gpurun merges runs: newest wins
    return g[-1] if g else None


def counters(path):
    d = collections.defaultdict(list)
    if not path:
        return {}
    for r in csv.DictReader(open(path)):
        d[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in d.items()}


summary = {"tag": tag}
ks = one("trace/*/*_kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, "%s_kernel_stats.csv" % tag))
    rows = list(csv.DictReader(open(ks)))
    summary["kernel_stats"] = [{"name": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                                "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows]
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(dst, "%s_bench.json" % tag))
    summary["bench"] = json.loads(open(bj).read().strip().splitlines()[-1])

kg = one("trace_grid/*/*_kernel_stats.csv")
if kg:
    shutil.copy(kg, os.path.join(dst, "%s_grid_kernel_stats.csv" % tag))
    summary["grid_kernel_stats"] = [{"name": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                                     "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
                                    for r in csv.DictReader(open(kg))]
bg = os.path.join(src, "bench_grid.json")
if os.path.exists(bg):
    shutil.copy(bg, os.path.join(dst, "%s_bench_grid.json" % tag))
    summary["bench_grid"] = json.loads(open(bg).read().strip().splitlines()[-1])

kc = one("trace_cfg3/*/*_kernel_stats.csv")
if kc:
    shutil.copy(kc, os.path.join(dst, "%s_config3_kernel_stats.csv" % tag))
    summary["config3_kernel_stats"] = [{"name": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                                        "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
                                       for r in csv.DictReader(open(kc))]
for name, out in (("bench_config3.json", "%s_bench_config3.json"), ("bench_single_substep.json", "%s_bench_single_substep.json"),
                  ("bench_4M.json", "%s_bench_4M.json"), ("bench_16M.json", "%s_bench_16M.json"),
                  ("upload_timing.txt", "%s_upload_timing.txt"),
                  ("exchange_cost.txt", "%s_exchange_cost.txt"), ("rehearse_2ranks.json", "%s_rehearse_2ranks_one_gpu.json"),
                  ("node_bench.json", "%s_node_bench.json"), ("cfg4_share.json", "%s_cfg4_share.json"),
                  ("cfg5_share.json", "%s_cfg5_share.json"),
                  ("bench_config3_contacts.json", "%s_bench_config3_lattice_on_floor.json"),
                  ("bench_soup.json", "%s_bench_soup.json"), ("config3_contacts_check.txt", "%s_config3_contacts_check.txt"),
                  ("bench_grid_no_hybrid.json", "%s_bench_grid_no_hybrid.json"), ("bench_driver_protocol.json", "%s_bench_driver_protocol.json"),
                  ("bench_rest_lengths.json", "%s_bench_rest_lengths.json"),
                  ("rehearse_2ranks_driver_protocol.json", "%s_rehearse_2ranks_one_gpu_driver_protocol.json"),
                  ("rehearse_2ranks_grid.json", "%s_rehearse_2ranks_one_gpu_grid.json"), ("grid_schedule_probe.txt", "%s_grid_schedule_probe.txt")):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, out % tag))

ke = one("trace_exchange/*/*_kernel_stats.csv")
if ke:
    shutil.copy(ke, os.path.join(dst, "%s_exchange_kernel_stats.csv" % tag))

calib = {}
for which, ctr in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
    for (k, c), (n, avg) in counters(one(which + "/*/*_counter_collection.csv")).items():
        if c == ctr and "calib_" in k:
            calib["%s %s" % (k.split("(")[0].replace("void ", ""), ctr)] = {"launches": n, "counter_KiB": avg,
                                                                          "known_KiB": 1048576.0,
                                                                          "bytes_per_count_KiB": 1048576.0 / avg if avg else None}
summary["calibration"] = calib
rd = [v["bytes_per_count_KiB"] for k, v in calib.items() if "calib_read" in k and "FETCH" in k]
wr = [v["bytes_per_count_KiB"] for k, v in calib.items() if "calib_write" in k and "WRITE" in k]
f_corr = sum(rd) / len(rd) if rd else 2.0
w_corr = sum(wr) / len(wr) if wr else 1.0
summary["fetch_correction"] = f_corr
summary["write_correction"] = w_corr
def traffic_of(suffix):
    traffic = {}
    for which, ctr, corr in (("pmc_fetch" + suffix, "FETCH_SIZE", f_corr), ("pmc_write" + suffix, "WRITE_SIZE", w_corr)):
        for (k, c), (n, avg) in counters(one(which + "/*/*_counter_collection.csv")).items():
            name = k.split("(")[0].replace("void ", "")
            if c == ctr and name.startswith("k_"):
                traffic.setdefault(name, {})[ctr] = {"launches": n, "counter_KiB": avg, "bytes": avg * 1024.0 * corr}
    for name, t in traffic.items():
        t["hbm_bytes_per_launch"] = sum(x["bytes"] for x in t.values() if isinstance(x, dict))
    return traffic


traffic = traffic_of("")
summary["traffic"] = traffic
summary["traffic_bench_single_substep"] = traffic_of("_k1")   # k_substep_tiled, --block-substeps 1
summary["traffic_config3"] = traffic_of("_cfg3")   # k_substep_tiled_grid on the settled blob pile (until r03 also the helper launch k_grid_maintain)
sq = {}
for which in ("pmc_sq1", "pmc_sq2"):
    for (k, c), (n, avg) in counters(one(which + "/*/*_counter_collection.csv")).items():
        if "k_substep" in k:
            sq.setdefault(k.split("(")[0].replace("void ", ""), {})[c] = {"launches": n, "per_launch": avg}
summary["sq_counters"] = sq
json.dump(summary, open(os.path.join(dst, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("kernel_stats", "traffic", "traffic_config3", "fetch_correction", "write_correction") if k in summary}, indent=1))
