#!/usr/bin/env python3
"""Collect the rocprofv3 outputs of scripts/profile_bench.sh <tag> (under gpurun_out/prof_<tag>/) into the
small files kept under profiles/: kernel stats CSV and the per-launch PMC totals of fill_kernel (JSON)."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
ks = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(dst, name + "_kernel_stats.csv"))
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*counter_collection.csv"))):
    part = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "fill_kernel" in r["Kernel_Name"]:
            part[r["Counter_Name"]] += float(r["Counter_Value"])
    for k in sorted(part):
        agg.setdefault(k, part[k])
json.dump(agg, open(os.path.join(dst, name + "_fill_pmc_summary.json"), "w"), indent=1)
print(json.dumps(agg, indent=1))
