cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05v
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["kernels_ms_per_step"])'
{
echo "== parity hw"; HIBAG_HIP_LIBRARY=$PWD/gpurun_var_hw.so timeout 600 python tools/parity_quick.py 2>&1 | tail -1
for rep in 1 2; do
  for v in prev prev_l2 prev_hot prev_nosv deep deep_l2 deep_hot hw nostore; do
    HIBAG_HIP_LIBRARY=$PWD/gpurun_var_$v.so timeout 300 $B 2>/dev/null | python -c "$P" $v
  done
done
} > gpurun_out/r05v/log.txt 2>&1
cat gpurun_out/r05v/log.txt
