cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05m
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"])'
{
for so in gpurun_var_aw2.so gpurun_var_bw2.so; do
  echo "== $so parity"; HIBAG_HIP_LIBRARY=$PWD/$so timeout 300 python tools/parity_quick.py 2>&1 | tail -1
done
for rep in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "$P" base
  for v in aw2 aw1 aw8 bw2; do
    HIBAG_HIP_LIBRARY=$PWD/gpurun_var_$v.so timeout 300 $B 2>/dev/null | python -c "$P" $v
  done
done
echo "== host path"; timeout 300 python tools/host_path_probe.py 10000 2>&1 | grep -v amdgpu.ids | tail -2
} > gpurun_out/r05m/log.txt 2>&1
cat gpurun_out/r05m/log.txt
