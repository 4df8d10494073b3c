#!/bin/bash
# tools/profiles.sh TAG -- one round's profile set on the GPU box (via gpurun): rocprofv3 kernel stats + separate
# FETCH_SIZE / WRITE_SIZE passes for cfg2 (the metric's configuration), cfg2 with type="response+prob", cfg2 with
# vote="majority", cfg4 (HLA-DRB1 shape); a full bench line; kernel stats of the per-sample route, a wide-classifier
# model, the training driver and the bundled real model; the SQ / LDS / TCP counters of both passes (tools/pmc_counters.sh).
# Outputs under gpurun_out/${TAG}prof; `python tools/collect_profiles.py TAG ${TAG}prof` copies the summaries into profiles/.
# (The program follows `--` directly: no env / bash -c hop under rocprofv3.)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
TAG=${1:?usage: tools/profiles.sh TAG}
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${TAG}prof; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args
  name=$1; shift
  mkdir -p $out/$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/stats -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $out/$name/stats.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$name/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $out/$name/fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$name/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $out/$name/write.log 2>&1
  tail -1 $out/$name/stats.log | cut -c1-300
}
run cfg2
if [ -z "${PROFILES_ONLY_CFG2:-}" ]; then
run cfg2_prob --prob
run cfg2_vote2 --vote majority
run cfg4 --shape hla-drb1 --samples 4096 --steps 10 --warmup 2
fi
cd $R
timeout 1200 python3 bench.py > $out/bench.json 2> $out/bench.log
tail -c 1200 $out/bench.json
if [ -z "${PROFILES_ONLY_CFG2:-}" ]; then
cd /tmp
for what in plugin_ab wide_bench train_threads real_model_bench; do
  mkdir -p $out/$what
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$what/stats -- python3 $R/tools/$what.py > $out/$what/stats.log 2>&1 < /dev/null
  tail -2 $out/$what/stats.log | cut -c1-300
done
fi
cd $R
bash tools/pmc_counters.sh ${TAG}ctr > /dev/null 2>&1; cp gpurun_out/${TAG}ctr/pmc_summary.txt $out/sq_counters.txt 2>/dev/null
PMC_PROG="$R/tools/wide_bench.py" bash tools/pmc_counters.sh ${TAG}ctrw > /dev/null 2>&1; cp gpurun_out/${TAG}ctrw/pmc_summary.txt $out/sq_counters_wide.txt 2>/dev/null
[ -x ./tools/ubench_lds ] && ./tools/ubench_lds > $out/ubench_lds.txt 2>&1
true
