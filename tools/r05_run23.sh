cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05w
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"])'
{
for rep in 1 2; do
  timeout 300 $B 2>/dev/null | python -c "$P" default
  for sp in 9 16 24 40; do HIBAG_STORE_PAIRS=$sp timeout 300 $B 2>/dev/null | python -c "$P" pairs$sp; done
  for sf in 3 8; do HIBAG_STORE_FIT=$sf timeout 300 $B 2>/dev/null | python -c "$P" fit$sf; done
  HIBAG_STORE_PAIRS=16 HIBAG_STORE_FIT=8 timeout 300 $B 2>/dev/null | python -c "$P" pairs16fit8
done
} > gpurun_out/r05w/log.txt 2>&1
cat gpurun_out/r05w/log.txt
