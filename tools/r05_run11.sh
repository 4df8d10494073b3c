cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]["issue"]; print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"], "stored", r["cell_sums_stored_per_sample"], "pairs2", r["pairs_evaluated_per_sample"]["pass2"])'
{
for so in gpurun_var_p1pre7.so gpurun_var_t12.so; do
  echo "== $so parity"; HIBAG_HIP_LIBRARY=$PWD/$so timeout 300 python tools/parity_quick.py 2>&1 | tail -1
done
for rep in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "$P" base
  for v in p1pre p1pre7 t12 t12p7; do
    HIBAG_HIP_LIBRARY=$PWD/gpurun_var_$v.so timeout 300 $B 2>/dev/null | python -c "$P" $v
  done
done
} > gpurun_out/r05k/log.txt 2>&1
cat gpurun_out/r05k/log.txt
