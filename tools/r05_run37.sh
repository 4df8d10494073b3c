cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05own2
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"])'
{
echo "== parity"; timeout 600 python tools/parity_quick.py 2>&1 | tail -3
for rep in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "$P" own
  HIBAG_HIP_LIBRARY=$PWD/gpurun_var_swaps.so timeout 300 $B 2>/dev/null | python -c "$P" swaps
done

} > gpurun_out/r05own2/log.txt 2>&1
cat gpurun_out/r05own2/log.txt
