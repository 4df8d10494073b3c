cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05last
{
echo "== parity"; timeout 600 python tools/parity_quick.py 2>&1 | tail -1
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-400
} > gpurun_out/r05last/log.txt 2>&1
cat gpurun_out/r05last/log.txt
