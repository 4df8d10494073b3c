cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05end
{
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
} > gpurun_out/r05end/log.txt 2>&1
cat gpurun_out/r05end/log.txt
timeout 900 python3 bench.py > gpurun_out/r05end/bench.json 2> gpurun_out/r05end/bench.log
tail -c 600 gpurun_out/r05end/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05end/prob_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --prob > $GRAFT_REPO_ROOT/gpurun_out/r05end/prob_stats.log 2>&1
