#!/bin/bash
# like run_variants.sh, with bench.py's other configurations (cfg4 = the DRB1 shape, vote, prob, ...)
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd $GRAFT_REPO_ROOT
for so in gpurun_var_*.so; do
  HIBAG_HIP_LIBRARY=$PWD/$so timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); o=d.get('other_configs',{})
print('$so', round(d['value']), d['roofline']['kernels_ms_per_step'], {k:(round(v.get('samples_per_s',v.get('value',0))) if isinstance(v,dict) else v) for k,v in o.items()})"
done
