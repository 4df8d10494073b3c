cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
{
echo "== tests (all gpu)"; timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -15
} > gpurun_out/r05f/log.txt 2>&1
cat gpurun_out/r05f/log.txt
