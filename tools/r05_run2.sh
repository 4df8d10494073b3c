cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
{
echo "== parity quick"; timeout 600 python tools/parity_quick.py 2>&1 | tail -3
echo "== tests"; timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_fuzz.py tests/test_hip_status.py -x -q -m gpu 2>&1 | tail -8
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('new', round(d['value']), d['roofline']['kernels_ms_per_step'])"
done
} > gpurun_out/r05b/log.txt 2>&1
cat gpurun_out/r05b/log.txt
