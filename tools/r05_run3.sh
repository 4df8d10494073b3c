cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"])'
{
for rep in 1 2 3; do
  HIBAG_HIP_LIBRARY=$PWD/gpurun_var_r04.so timeout 300 $B 2>/dev/null | python -c "$P" r04
  timeout 300 $B 2>/dev/null | python -c "$P" new
  HIBAG_PREBUILT_MB=0 timeout 300 $B 2>/dev/null | python -c "$P" new_p1generated
  HIBAG_HIP_LIBRARY=$PWD/gpurun_var_occ7.so timeout 300 $B 2>/dev/null | python -c "$P" new_occ7
done
echo "== stamps (new)"; HIBAG_HIP_LIBRARY=$PWD/gpurun_var_stamps.so timeout 300 python tools/accum_stamps.py 2>&1 | tail -9
} > gpurun_out/r05c/log.txt 2>&1
cat gpurun_out/r05c/log.txt
