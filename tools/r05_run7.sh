cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
{
echo "== vote tests"; timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_configs.py tests/test_hip_status.py tests/test_hip_shard.py -x -q -m gpu 2>&1 | tail -6
echo "== bench with extras"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05g/bench.json 2> gpurun_out/r05g/bench.err; tail -3 gpurun_out/r05g/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05g/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), d["roofline"]["kernels_ms_per_step"])
print("host_inclusive", d.get("host_inclusive",{}).get("value"), d.get("host_inclusive",{}).get("frac_of_device_resident"))
oc=d.get("other_configs",{})
for k,v in oc.items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ("samples_per_s","ms_per_step","error","kernels_ms_per_step","pass2","issue","oracle_check","speedup_vs_cpu_baseline","classifiers_per_s","s_per_classifier")})
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("one_thread",{}).get("value"))
PY
} > gpurun_out/r05g/log.txt 2>&1
cat gpurun_out/r05g/log.txt
