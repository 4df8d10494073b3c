cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05fuzz3
rm -f gpurun_out/r05fuzz3/report.txt
export HIBAG_FUZZ_REPORT=$PWD/gpurun_out/r05fuzz3/report.txt
{
HIBAG_FUZZ_SECONDS=360 HIBAG_FUZZ_SEED=700000 timeout 1200 python -m pytest tests/test_hip_fuzz.py::test_wide_campaign -x -q -m gpu 2>&1 | tail -3
HIBAG_FUZZ_SECONDS=180 HIBAG_FUZZ_SEED=720000 HIBAG_FUZZ_BIG_EVERY=2 timeout 900 python -m pytest tests/test_hip_fuzz.py::test_wide_campaign -x -q -m gpu 2>&1 | tail -3
HIBAG_FUZZ_SECONDS=180 HIBAG_FUZZ_SEED=740000 timeout 900 python -m pytest tests/test_hip_fuzz.py::test_entry_points_campaign -x -q -m gpu 2>&1 | tail -3
HIBAG_FUZZ_SECONDS=120 HIBAG_FUZZ_SEED=760000 timeout 900 python -m pytest tests/test_hip_fuzz.py::test_plugin_campaign -x -q -m gpu 2>&1 | tail -3
cat gpurun_out/r05fuzz3/report.txt
} > gpurun_out/r05fuzz3/log.txt 2>&1
cat gpurun_out/r05fuzz3/log.txt
