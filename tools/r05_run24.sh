cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05x
{
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
} > gpurun_out/r05x/log.txt 2>&1
cat gpurun_out/r05x/log.txt
bash tools/r05_profiles.sh > gpurun_out/r05x/profiles.log 2>&1
tail -5 gpurun_out/r05x/profiles.log
