cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i
{
echo "== two-half probe"; timeout 600 python tools/two_half_probe.py 2>&1 | grep -v amdgpu.ids | tail -25
echo "== host path"; HIBAG_STAGED_TRACE=1 timeout 300 python tools/host_path_probe.py 10000 2>&1 | grep -v amdgpu.ids | tail -8
} > gpurun_out/r05i/log.txt 2>&1
cat gpurun_out/r05i/log.txt
