#!/usr/bin/env python3
"""Copy the summaries of tools/r02_profiles.sh (gpurun_out/r02prof/<name>/...) into profiles/ (tracked):
profiles/<tag>_<name>_kernel_stats.csv, profiles/<tag>_<name>_pmc_traffic.json, profiles/<tag>_bench.json, and
profiles/pmc_traffic.json (= the metric's configuration, what bench.py quotes as roofline.traffic).

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE counts 64 B per 128-B request, so the read side is doubled; WRITE_SIZE is taken as is.
(The guide calibrated this for 16-B-per-lane streams; these kernels read 4-16 B per lane: an upper estimate.)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "r02prof")      # r03: gpurun_out/r03prof (tools/r03_profiles.sh)
dst = os.path.join(ROOT, "profiles")


def newest(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    return f[-1] if f else None


def short(name):
    return name.replace("void ", "").split("(")[0].split("<")[0]


for name in sorted(os.listdir(src)):
    d = os.path.join(src, name)
    if not os.path.isdir(d):
        continue
    ks = newest(os.path.join(d, "stats", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks, os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
    traffic = collections.defaultdict(dict)
    for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = newest(os.path.join(d, kind, "*", "*_counter_collection.csv"))
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            traffic[k][ctr + "_KiB_per_launch"] = sum(v) / len(v)
            traffic[k][ctr + "_launches"] = len(v)
    for k, t in traffic.items():
        rd = t.get("FETCH_SIZE_KiB_per_launch", 0.0) * 2 * 1024
        wr = t.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
        t["hbm_read_bytes_per_launch"], t["hbm_write_bytes_per_launch"], t["hbm_bytes_per_launch"] = rd, wr, rd + wr
    if traffic:
        traffic["_source"] = {"tag": f"{tag}_{name}", "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of "
                              "bench.py (see tools/r02_profiles.sh / r03_profiles.sh for the arguments); read side doubled per MI355X_MICROARCH.md"}
        json.dump(traffic, open(os.path.join(dst, f"{tag}_{name}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
        if name == "cfg2":
            json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print(name, "->", {k: round(v.get("hbm_bytes_per_launch", 0) / 1e6, 1) for k, v in traffic.items() if not k.startswith("_")})
b = os.path.join(src, "bench.json")
if os.path.exists(b):
    shutil.copy(b, os.path.join(dst, f"{tag}_bench.json"))
