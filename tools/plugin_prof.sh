#!/bin/bash
# rocprofv3 kernel stats of the per-sample plugin route (hibag_sample.hip) on the benchmark model
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_plugin; rm -rf $out; mkdir -p $out
cat > /tmp/plugin_run.py <<PY
import sys, time
sys.path.insert(0, "$R")
import numpy as np
import hibag_amd
from hibag_amd import synth
from hibag_amd.plugin import PluginHost
hibag_amd.hlaSetKernelTarget("hip")
obj, founders, af = synth.make_model("hla-b")
G, truth = synth.make_samples(founders, af, 400, seed=synth.DEFAULT_SEED + 1)
host = PluginHost(obj)
geno, wt = host.pack(G)
prob = np.zeros(obj.n_cell); match = np.zeros(1)
for i in range(20): host.avg_prob(geno[i], wt[i], prob, match)
t = time.perf_counter()
for i in range(400): host.avg_prob(geno[i], wt[i], prob, match)
print("us per call", (time.perf_counter() - t) / 400 * 1e6)
host.close()
PY
python3 /tmp/plugin_run.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /tmp/plugin_run.py > $out/log.txt 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r04_plugin_kernel_stats.csv && cut -d, -f1-4 "$f" | head -6
