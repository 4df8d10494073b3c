#!/bin/bash
# round 3: training (BASELINE config 5 shape) -- wall-clock split of the driver and rocprofv3 kernel stats of its kernels
R=$GRAFT_REPO_ROOT; tag=${1:-r03_cfg5}; out=$R/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
cd $R
HIBAG_TRAIN_PROFILE=1 python3 tools/train_bench.py 8 1000 300 2>&1 | tail -12 | tee $out/split.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/tools/train_bench.py 4 1000 300 > $out/bench.log 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/${tag}_kernel_stats.csv; head -8 $f | cut -c1-160
