cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05nt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"])'
{
for rep in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "$P" base
  HIBAG_HIP_LIBRARY=$PWD/gpurun_var_finnt.so timeout 300 $B 2>/dev/null | python -c "$P" finnt
  timeout 300 $B --prob 2>/dev/null | python -c "$P" base_prob
  for v in finnt probnt bothnt; do
    HIBAG_HIP_LIBRARY=$PWD/gpurun_var_$v.so timeout 300 $B --prob 2>/dev/null | python -c "$P" ${v}_prob
  done
done
} > gpurun_out/r05nt/log.txt 2>&1
cat gpurun_out/r05nt/log.txt
