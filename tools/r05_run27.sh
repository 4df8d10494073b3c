cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05dry
{
for n in 2 4; do
  echo "== dry --gpus $n (driver's launch line)"
  HIBAG_BENCH_DRY_RANKS=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-1500
done
} > gpurun_out/r05dry/log.txt 2>&1
cat gpurun_out/r05dry/log.txt
