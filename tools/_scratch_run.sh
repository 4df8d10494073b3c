cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06j
{
echo "== training tests"; timeout 1200 python -m pytest tests/test_hip_training.py tests/test_hip_train_driver.py -m gpu -x -q 2>&1 | tail -4
echo "== concurrency"; timeout 900 python tools/train_concurrency.py "16x1:device 16x1:device:4 24x1:device:4 16x1:device:2 24x1:device:2" 192 2>&1 | tail -12
echo "== CPU split"; HIBAG_TRAIN_PROFILE=1 timeout 300 python tools/train_concurrency.py "16x1:device:4" 64 2>&1 | grep "hibag train" | tail -3
} > gpurun_out/r06j/log.txt 2>&1
cat gpurun_out/r06j/log.txt
