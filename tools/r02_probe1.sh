#!/bin/bash
# Round-2 probe 1 (GPU box, via gpurun): raw instruction-cost logs, model block statistics,
# pass 1 (k_total) at reduced occupancy (extra dynamic LDS), LDS bank-conflict counters.
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r02p1; mkdir -p $out
cd $R
HIBAG_DEBUG_MODEL=1 timeout 300 python bench.py --steps 20 --warmup 5 > $out/bench_base.json 2> $out/bench_base.err
rm -f $out/occupancy.txt
for lds in 0 36000 60000; do
  HIBAG_DEBUG_LDS=$lds timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('k_total dyn LDS $lds', round(d['value']), d['roofline']['kernels_ms_per_step'])" >> $out/occupancy.txt
done
for n in 20000 40000; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --samples $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('samples $n', round(d['value']), d['roofline']['kernels_ms_per_step'])" >> $out/occupancy.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc_lds -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc_lds.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $out/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc_sq.log 2>&1
cd $R && python3 - "$out" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
with open(out+"/pmc_summary.txt","w") as fo:
  for f in sorted(glob.glob(out+"/pmc_*/*/*_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].replace("void ","").split("(")[0][:16], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        if k[0].startswith(("k_total","k_accum")): print("%-10s %-32s %.6g"%(k[0],k[1],sum(v)/len(v)), file=fo)
PY
cat $out/occupancy.txt $out/pmc_summary.txt; tail -c 1500 $out/bench_base.json; tail -3 $out/bench_base.err
