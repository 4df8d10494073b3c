#!/bin/bash
# tools/fuzz_campaign.sh NAME [SECONDS] [SEED] -- the random campaigns of tests/ run long on the GPU box (via gpurun):
# test_wide_campaign (random models of every engine, both votes, chunked work items), the same with large cohorts, and
# test_entry_points_campaign (mapped / SNP-major / BED / device / replicas / RCCL shards against the plain entry), and the two
# training campaigns (shared trainers on different cohorts through the combiner; single trainers).
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
NAME=${1:?usage: tools/fuzz_campaign.sh NAME [SECONDS] [SEED]}; SECS=${2:-300}; SEED=${3:-700000}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/$NAME; rm -f gpurun_out/$NAME/report.txt
export HIBAG_FUZZ_REPORT=$PWD/gpurun_out/$NAME/report.txt
{
HIBAG_FUZZ_SECONDS=$SECS HIBAG_FUZZ_SEED=$SEED timeout $((SECS + 600)) python -m pytest tests/test_hip_fuzz.py::test_wide_campaign -x -q -m gpu 2>&1 | tail -3
HIBAG_FUZZ_SECONDS=$((SECS / 2)) HIBAG_FUZZ_SEED=$((SEED + 20000)) HIBAG_FUZZ_BIG_EVERY=2 timeout $((SECS + 600)) python -m pytest tests/test_hip_fuzz.py::test_wide_campaign -x -q -m gpu 2>&1 | tail -3
HIBAG_FUZZ_SECONDS=$((SECS / 2)) HIBAG_FUZZ_SEED=$((SEED + 40000)) timeout $((SECS + 600)) python -m pytest tests/test_hip_fuzz.py::test_entry_points_campaign -x -q -m gpu 2>&1 | tail -3
HIBAG_FUZZ_SECONDS=$((SECS / 2)) HIBAG_FUZZ_SEED=$((SEED + 60000)) timeout $((SECS + 600)) python -m pytest tests/test_hip_train_driver.py::test_combined_trainers_on_different_cohorts_campaign -x -q -m gpu 2>&1 | tail -3
HIBAG_FUZZ_SECONDS=$((SECS / 2)) HIBAG_FUZZ_SEED=$((SEED + 80000)) timeout $((SECS + 600)) python -m pytest tests/test_hip_train_driver.py::test_training_campaign -x -q -m gpu 2>&1 | tail -3
cat gpurun_out/$NAME/report.txt
} > gpurun_out/$NAME/log.txt 2>&1
cat gpurun_out/$NAME/log.txt
