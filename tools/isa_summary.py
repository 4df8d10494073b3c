#!/usr/bin/env python3
"""ISA summary of the device code: per kernel its registers, LDS, private segment (scratch) and spills, and for every
loop that contains matrix instructions the number of scratch instructions inside it.

    python tools/isa_summary.py [extra hipcc flags] > profiles/rNN_isa_summary.txt

Compiles every .hip of hibag_amd/csrc to assembly for gfx950 (device side only) with the Makefile's flags."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hibag_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 --cuda-device-only -S".split()
# the kernels' extra LLVM option: from the Makefile's probe, the one place that names it
KFLAGS = subprocess.run(["make", "-s", "-C", SRC, "print-kflags"], capture_output=True, text=True).stdout.split()
def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        return n
for f in sorted(os.listdir(SRC)):
    if not f.endswith(".hip"):
        continue
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        extra = KFLAGS if f == "hibag_kernels.hip" else []
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + sys.argv[1:] + [os.path.join(SRC, f), "-o", tmp.name], check=True, stderr=subprocess.DEVNULL)
        lines = open(tmp.name).read().split("\n")
    meta = {}
    blk = None                                   # one "- .agpr_count: ..." entry of amdhsa.kernels (keys in alphabetical order)
    for l in lines:
        if re.match(r"\s+- \.\w+:", l) and not re.match(r"\s+- \.(address_space|actual_access|offset|name|size|value_kind)", l):
            blk = {}
        m = re.match(r"\s+(?:- )?\.(private_segment_fixed_size|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|agpr_count):\s+(\d+)", l)
        if m and blk is not None:
            blk[m.group(1)] = int(m.group(2))
        m = re.match(r"    \.name:\s+(\S+)", l)
        if m and blk is not None:
            meta[m.group(1)] = blk
    starts = [(i, re.match(r"^(_Z\w+|\w+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^(_Z\w+):", l)]
    print(f"== {f}")
    for k, (i0, name) in enumerate(starts):
        if name not in meta:
            continue
        i1 = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = lines[i0:i1]
        lab = {m.group(1): j for j, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for j, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in lab and lab[m.group(1)] < j:
                loops.append((lab[m.group(1)], j))
        mm = meta[name]
        hot = []
        for a, b in loops:
            nm = sum("v_mfma" in l for l in body[a:b])
            if nm:
                hot.append((nm, sum("scratch_" in l for l in body[a:b]), b - a))
        print(f"{demangle(name):<58} vgpr {mm.get('vgpr_count', 0):>3} (spilled {mm.get('vgpr_spill_count', 0)})  sgpr {mm.get('sgpr_count', 0):>3}  "
              f"lds {mm.get('group_segment_fixed_size', 0):>6} B  .private_segment_fixed_size: {mm.get('private_segment_fixed_size', 0)}"
              + ("   loops with MFMA [mfma, scratch ops, lines]: " + ", ".join(f"[{a}, {b}, {c}]" for a, b, c in hot) if hot else ""))
