#!/bin/bash
# round 3: rocprofv3 kernel trace of the benchmark step (per-kernel durations) -> gpurun_out/$1_kernel_stats.csv
R=$GRAFT_REPO_ROOT; tag=${1:-r03}; out=$R/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $BENCH_ARGS > $out/bench.log 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/${tag}_kernel_stats.csv; head -14 $f | cut -c1-150
tail -1 $out/bench.log | cut -c1-300
