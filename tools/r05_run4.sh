cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"])'
{
for so in gpurun_var_sve.so gpurun_var_a2.so gpurun_var_sve_a2.so gpurun_var_g8.so; do
  echo "== $so parity"; HIBAG_HIP_LIBRARY=$PWD/$so timeout 300 python tools/parity_quick.py 2>&1 | tail -1
done
for rep in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "$P" base
  for v in sve a2 sve_a2 g8 a1; do
    HIBAG_HIP_LIBRARY=$PWD/gpurun_var_$v.so timeout 300 $B 2>/dev/null | python -c "$P" $v
  done
  HIBAG_TOT_OCC=5 HIBAG_HIP_LIBRARY=$PWD/gpurun_var_a1.so timeout 300 $B 2>/dev/null | python -c "$P" a1_occ5
  HIBAG_TOT_OCC=5 timeout 300 $B 2>/dev/null | python -c "$P" base_occ5
done
} > gpurun_out/r05d/log.txt 2>&1
cat gpurun_out/r05d/log.txt
