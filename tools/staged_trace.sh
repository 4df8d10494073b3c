#!/bin/bash
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/staged_trace; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -- python3 $R/tools/staged_trace.py > $out/log.txt 2>&1
tail -4 $out/log.txt
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:28]))
for f in glob.glob(out + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?"))))
ev.sort()
# the last call: events after the last long gap
t_end = ev[-1][1]
sel = [e for e in ev if e[0] > t_end - 22_000_000]
t0 = sel[0][0]
for a, b, n in sel:
    if n.startswith("C ") or "k_total<" in n or "k_accum" in n or "k_codes" in n:
        print("%9.3f .. %9.3f ms  %s" % ((a - t0) / 1e6, (b - t0) / 1e6, n))
PY
