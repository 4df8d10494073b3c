#!/bin/bash
# One profiling pass on the GPU box (via gpurun): rocprofv3 kernel stats, the two HBM-traffic
# counter passes (separate runs, no tracing mixed in), the PCIe-inclusive rates and a full
# bench line.  Outputs under gpurun_out/prof; tools/collect_profiles.py <tag> copies the
# summaries into profiles/.
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --no-cpu-baseline > $out/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
cd $R
timeout 600 python3 tools/host_inclusive.py > $out/host_inclusive.json 2> $out/host_inclusive.log
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.log
tail -1 $out/bench.json | cut -c1-400; cat $out/host_inclusive.json
