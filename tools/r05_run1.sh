cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
{
echo "== parity waitb"; HIBAG_HIP_LIBRARY=$PWD/gpurun_var_waitb.so timeout 300 python tools/parity_quick.py 2>&1 | tail -1
for rep in 1 2 3; do
for so in hibag_amd/csrc/libhibag_hip.so gpurun_var_waitb.so; do
  HIBAG_HIP_LIBRARY=$PWD/$so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$so', round(d['value']), d['roofline']['kernels_ms_per_step'])"
done
done
echo "== stamps (base)"; HIBAG_HIP_LIBRARY=$PWD/gpurun_var_stamps.so timeout 300 python tools/accum_stamps.py 2>&1 | tail -12
echo "== stamps (waitb)"; HIBAG_HIP_LIBRARY=$PWD/gpurun_var_stampsw.so timeout 300 python tools/accum_stamps.py 2>&1 | tail -12
} > gpurun_out/r05a/log.txt 2>&1
cat gpurun_out/r05a/log.txt
