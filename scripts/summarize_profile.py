#!/usr/bin/env python3
"""Collect the rocprofv3 outputs of scripts/profile_bench.sh <tag> (under gpurun_out/prof_<tag>/) into the
small files kept under profiles/: kernel stats CSV and the per-launch PMC totals of fill_kernel (JSON).  The JSON
also records what bench.py needs to decide whether the summary still describes the library it runs: the workload
(reads, ref_len, r, max_b_rows, base_seed, mixed), the digest of the kernel sources (csrc_sha) and the wave-steps of
one launch (waves per chunk x anti-diagonals of the batch), all taken from the bench line of the traced run.
usage: python scripts/summarize_profile.py <tag> [name under profiles/ = tag]"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
ks = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(dst, name + "_kernel_stats.csv"))
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*counter_collection.csv"))):
    part = collections.defaultdict(float)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = set()
    for r in csv.DictReader(open(f)):
        if "fill_kernel" in r["Kernel_Name"]:
            part[r["Counter_Name"]] += float(r["Counter_Value"])
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            launches.add(r["Dispatch_Id"])
    for k in sorted(part):          # per launch: a profiled bench run may hold several
        agg.setdefault(k, part[k] / max(1, len(launches)))
    if "GRBM_GUI_ACTIVE" in per:    # busy cycles: the shortest launch (a first launch also spans buffer allocations beside it)
        agg["GRBM_GUI_ACTIVE"] = min(per["GRBM_GUI_ACTIVE"].values())
line = None
for l in open(os.path.join(src, "bench_trace.log")):
    if l.startswith("{"):
        line = json.loads(l)
if line:
    import bench
    c, pr = line["config"], line["roofline"]["practical"]
    agg.update({"reads": c["reads_per_gpu"], "ref_len": c["ref_len"], "r": c["r"], "max_b_rows": c.get("max_b_rows", 20000),
                "base_seed": c.get("base_seed", 2), "mixed": bool(c.get("mixed", False)), "csrc_sha": bench.csrc_sha(),
                "wave_steps": pr["waves_per_chunk"] * pr["rows_total"], "kernel_ms_traced_run": line["roofline"]["kernel_ms"]})
# the kernel's duration in the traced run: rocprofv3's own average over the fill launches (the bench line's event times
# there also span the profiler's serialisation of the neighbouring kernels)
if ks:
    for r in csv.DictReader(open(ks[0])):
        if "fill_kernel" in r["Name"]:
            agg["kernel_ms_traced_run"] = round(float(r["AverageNs"]) * 1e-6, 3)
            agg["kernel_ms_traced_run_min_max"] = [round(float(r["MinNs"]) * 1e-6, 3), round(float(r["MaxNs"]) * 1e-6, 3)]
            break
json.dump(agg, open(os.path.join(dst, name + "_fill_pmc_summary.json"), "w"), indent=1)
print(json.dumps(agg, indent=1))
