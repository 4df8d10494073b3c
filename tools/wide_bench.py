import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import hibag_amd
from hibag_amd import synth
hibag_amd.hlaSetKernelTarget("hip")
dev = torch.device("cuda", 0)
ks = [33, 40, 56, 57, 84, 85, 100, 112] * 4
obj, founders, af = synth.make_model("hla-b", seed=31, n_classifier=len(ks), n_snp=150, snp_counts=ks, wide_classifier=False)
nw = 10000
G, truth = synth.make_samples(founders, af, nw, seed=32)
m = hibag_amd.hlaModelFromObj(obj)
dg = torch.from_numpy(G).to(dev)
o = [torch.empty(nw, dtype=torch.int32, device=dev), torch.empty(nw, dtype=torch.int32, device=dev), torch.empty(nw, dtype=torch.float64, device=dev), torch.empty(nw, dtype=torch.float64, device=dev), torch.empty((nw, obj.n_hla), dtype=torch.float64, device=dev)]
st = torch.cuda.current_stream(dev).cuda_stream
run = lambda: m.predict_device(dg.data_ptr(), nw, 1, *[x.data_ptr() for x in o], None, stream=st)
run(); torch.cuda.synchronize(dev)
m.set_timing(True); m.reset_timing()
t = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize(dev)
dt = (time.perf_counter() - t) / 5
tm = m.get_timing()
print("wide model: ms/step", round(dt * 1e3, 3), {k: round(v[0] / 5, 3) for k, v in tm.items()}, "faults", m.handover_faults())
