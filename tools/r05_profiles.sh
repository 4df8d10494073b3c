#!/bin/bash
# Round-5 profiles (GPU box, via gpurun): rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE passes for
#   cfg2 (the metric's configuration), cfg2 with type="response+prob", cfg2 with vote="majority", cfg4 (HLA-DRB1 shape)
# Outputs under gpurun_out/r05prof/<name>/{stats,fetch,write}; tools/collect_profiles2.py copies the summaries into profiles/.
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r05prof; rm -rf $out; mkdir -p $out
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
run cfg2_prob --prob
run cfg2_vote2 --vote majority
run cfg4 --shape hla-drb1 --samples 4096 --steps 10 --warmup 2
cd $R
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.log
tail -c 1200 $out/bench.json
# round 4 extras: the per-sample plugin route, a model of wide classifiers only, the training driver (kernel stats each)
cd /tmp
for what in plugin_ab wide_bench train_threads real_model_bench; do
  mkdir -p $out/$what
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$what/stats -- python3 $R/tools/$what.py > $out/$what/stats.log 2>&1 < /dev/null
  tail -2 $out/$what/stats.log | cut -c1-300
done
cd $R
bash tools/r02_pmc.sh r05ctr > /dev/null 2>&1; cp gpurun_out/r05ctr/pmc_summary.txt $out/sq_counters.txt 2>/dev/null
./tools/ubench_lds > $out/ubench_lds.txt 2>&1
