#!/bin/bash
# Round-2 probe 1 (GPU box, via gpurun): raw instruction-cost logs, model block statistics,
# and pass 1 (k_total) at reduced occupancy (extra dynamic LDS).
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r02p1; rm -rf $out; mkdir -p $out
cd $R
for u in ubench_valu ubench_mfma ubench_lds; do timeout 120 tools/$u > $out/$u.txt 2>&1; done
HIBAG_DEBUG_MODEL=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_base.json 2> $out/bench_base.err
for lds in 0 36000 60000; do
  HIBAG_DEBUG_LDS=$lds timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('k_total dyn LDS $lds', round(d['value']), d['roofline']['kernels_ms_per_step'])" >> $out/occupancy.txt
done
for n in 20000 40000; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --samples $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('samples $n', round(d['value']), d['roofline']['kernels_ms_per_step'])" >> $out/occupancy.txt
done
HIBAG_DEBUG_MODEL=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --shape hla-drb1 --samples 4096 > $out/bench_drb1.json 2> $out/bench_drb1.err
cat $out/*.txt; tail -c 600 $out/bench_base.json; cat $out/bench_base.err | tail -3; tail -c 600 $out/bench_drb1.json; tail -3 $out/bench_drb1.err
