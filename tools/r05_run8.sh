cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05h
{
for cap in 15 8 6 4 3 2; do echo "== tile cap $cap"; HIBAG_TILE_CAP=$cap timeout 300 python tools/real_model_bench.py 2>/dev/null | tail -1; done
echo "== default"; timeout 300 python tools/real_model_bench.py 2>/dev/null | tail -1
echo "== train tests"; timeout 1500 python -m pytest tests/test_hip_train_driver.py tests/test_hip_training.py tests/test_hip_multi_device.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -5
} > gpurun_out/r05h/log.txt 2>&1
cat gpurun_out/r05h/log.txt
