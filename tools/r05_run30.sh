cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05ah
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"])'
{
echo "== parity"; timeout 600 python tools/parity_quick.py 2>&1 | tail -1
for rep in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "$P" new
  HIBAG_HIP_LIBRARY=$PWD/gpurun_var_prev.so timeout 300 $B 2>/dev/null | python -c "$P" prev
done
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
} > gpurun_out/r05ah/log.txt 2>&1
cat gpurun_out/r05ah/log.txt
