cd $GRAFT_REPO_ROOT/hibag_amd/csrc
for bw in 4 8 16; do cp libhibag_hip_bw$bw.so libhibag_hip.so; cd $GRAFT_REPO_ROOT; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bw=$bw', round(d['value']), d['roofline']['kernels_ms_per_step'])"; cd hibag_amd/csrc; done
