#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof_counters.sh <outdir> "<counter list 1>" "<counter list 2>" ...
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS > $out/p$i.log 2>&1
done
cd $R && python3 - "$out" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
for f in sorted(glob.glob(out+"/p*/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].replace("void ","").split("(")[0][:16], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        if k[0].startswith(("k_total","k_accum")): print("%-10s %-32s %.5g"%(k[0],k[1],sum(v)/len(v)))
PY
