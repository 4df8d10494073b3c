cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05l
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"])'
{
echo "== parity quick"; timeout 600 python tools/parity_quick.py 2>&1 | tail -1
echo "== tests"; timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for rep in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "$P" new
  HIBAG_FINISH_LEGACY=1 timeout 300 $B 2>/dev/null | python -c "$P" finish_legacy
  HIBAG_HIP_LIBRARY=$PWD/gpurun_var_r04.so timeout 300 $B 2>/dev/null | python -c "$P" r04
done
timeout 300 $B --prob 2>/dev/null | python -c "$P" prob
timeout 300 $B --shape hla-drb1 --samples 4096 --steps 10 --warmup 2 2>/dev/null | python -c "$P" drb1
} > gpurun_out/r05l/log.txt 2>&1
cat gpurun_out/r05l/log.txt
