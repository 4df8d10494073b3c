#!/bin/bash
# tools/gpu_validate.sh NAME [pytest args...] -- the round's validation run on the GPU box (via gpurun): quick parity of two
# models, the -m gpu suite (or the tests given), smoke(), a short bench line.  Log: gpurun_out/NAME/log.txt.
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
NAME=${1:?usage: tools/gpu_validate.sh NAME [pytest args]}; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/$NAME
{
echo "== parity"; timeout 600 python tools/parity_quick.py 2>&1 | tail -1
echo "== tests"; if [ $# -gt 0 ]; then timeout 2400 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -15; else timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5; fi
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:---no-cpu-baseline --no-extras} 2>gpurun_out/$NAME/bench.err | tail -1 > gpurun_out/$NAME/bench.json; cut -c1-600 gpurun_out/$NAME/bench.json
} > gpurun_out/$NAME/log.txt 2>&1
cat gpurun_out/$NAME/log.txt
