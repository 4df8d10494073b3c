#!/bin/bash
# rocprofv3 kernel stats of a model whose classifiers all have 33..112 SNPs (tools/wide_bench.py)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_wide; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/tools/wide_bench.py > $out/log.txt 2>&1 < /dev/null
f=$(ls $out/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r04_wide_kernel_stats.csv && cut -d, -f1-4 "$f" | head -12
tail -1 $out/log.txt
