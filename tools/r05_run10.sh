cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05j
{
echo "== tests (all gpu)"; timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -8
} > gpurun_out/r05j/log.txt 2>&1
bash tools/r05_profiles.sh > gpurun_out/r05j/profiles.log 2>&1
tail -5 gpurun_out/r05j/profiles.log >> gpurun_out/r05j/log.txt
cat gpurun_out/r05j/log.txt
