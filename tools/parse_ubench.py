#!/usr/bin/env python3
"""Copy the raw micro-benchmark logs of a gpurun call into profiles/ and derive the issue-floor
constants bench.py uses (profiles/ubench_constants.json).

    python tools/parse_ubench.py gpurun_out/r02p1 r02

Reads <dir>/ubench_mfma.txt, ubench_valu.txt, ubench_lds.txt (outputs of tools/ubench_*)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02p1")
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
dst = os.path.join(ROOT, "profiles")
out = {"tag": tag, "raw_logs": []}
for name in ("ubench_mfma", "ubench_valu", "ubench_lds"):
    f = os.path.join(src, name + ".txt")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_{name}.txt"))
        out["raw_logs"].append(f"profiles/{tag}_{name}.txt")
txt = open(os.path.join(src, "ubench_mfma.txt")).read()
m = re.search(r"all MFMA\s*:\s*([\d.]+) ms\s*->\s*([\d.]+) ns per MFMA", txt)
f = re.search(r"all FP64\s*:\s*([\d.]+) ms\s*->\s*([\d.]+) ns per FP64", txt)
h = re.search(r"half/half\s*:\s*([\d.]+) ms\s*serialised would be ([\d.]+), overlapped ([\d.]+)", txt)
out["mfma_i8_32x32x32_ns"] = float(m.group(2))
out["fp64_op_ns"] = float(f.group(2))
out["mfma_fp64_overlap"] = {"half_half_ms": float(h.group(1)), "serialised_ms": float(h.group(2)),
                            "overlapped_ms": float(h.group(3)),
                            "frac_of_serial_sum": round(float(h.group(1)) / float(h.group(2)), 3)}
out["note"] = ("ns per wave64 instruction per SIMD with 8 waves per SIMD on all 1,024 SIMDs (tools/ubench_mfma.hip); "
               "the half/half mix at ~the serial sum means the int8 MFMA does not hide behind FP64 VALU work, so both add in the floor")
json.dump(out, open(os.path.join(dst, "ubench_constants.json"), "w"), indent=1)
print(json.dumps(out))
