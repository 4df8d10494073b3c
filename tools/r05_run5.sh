cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]["issue"]; print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"], "stored", r["cell_sums_stored_per_sample"], "pairs2", r["pairs_evaluated_per_sample"]["pass2"])'
{
echo "== tests (all gpu)"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
for rep in 1 2; do
  timeout 300 $B 2>/dev/null | python -c "$P" base
  for sp in 6 9 16 24; do HIBAG_STORE_PAIRS=$sp timeout 300 $B 2>/dev/null | python -c "$P" store_pairs_$sp; done
  for sf in 0 3 8; do HIBAG_STORE_FIT=$sf timeout 300 $B 2>/dev/null | python -c "$P" store_fit_$sf; done
  HIBAG_TAIL_K=1 timeout 300 $B 2>/dev/null | python -c "$P" tail_k1
  HIBAG_TAIL_K=4 timeout 300 $B 2>/dev/null | python -c "$P" tail_k4
done
} > gpurun_out/r05e/log.txt 2>&1
cat gpurun_out/r05e/log.txt
