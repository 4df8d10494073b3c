#!/bin/bash
# FETCH_SIZE of k_accum for every gpurun_var_*.so (XCD-mapping experiments); one counter pass each
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/tv; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for so in $R/gpurun_var_*.so; do
  n=$(basename $so .so)
  HIBAG_HIP_LIBRARY=$so timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$n -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/$n.log 2>&1
done
cd $R && python3 - "$out" <<'PY'
import csv,glob,collections,sys,os
out=sys.argv[1]
for f in sorted(glob.glob(out+"/*/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE": agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    print(f.split("/")[-3], {k: round(sum(v)/len(v)*2*1024/1e6,1) for k,v in agg.items() if k in ("k_accum","k_total")}, "MB read per launch")
PY
