import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import hibag_amd
from hibag_amd import synth
from hibag_amd.plugin import PluginHost
hibag_amd.hlaSetKernelTarget("hip")
obj, founders, af = synth.make_model("hla-b")
G, truth = synth.make_samples(founders, af, 400, seed=synth.DEFAULT_SEED + 1)
host = PluginHost(obj)
geno, wt = host.pack(G)
prob = np.zeros(obj.n_cell); match = np.zeros(1)
for i in range(20): host.avg_prob(geno[i], wt[i], prob, match)
ts = []
for r in range(3):
    t = time.perf_counter()
    for i in range(400): host.avg_prob(geno[i], wt[i], prob, match)
    ts.append((time.perf_counter() - t) / 400 * 1e6)
print("us per call", min(ts), ts)
best, match, sec = host.avg_prob_loop(geno, wt)
best, match, sec = host.avg_prob_loop(geno, wt)
nh = obj.n_hla
cell = {(a, b): b + a * (2 * nh - a - 1) // 2 for a in range(nh) for b in range(a, nh)}
want = np.array([cell[(min(a, b), max(a, b))] for a, b in truth])
print("compiled loop: us per call", sec / len(geno) * 1e6, "calls right", float(np.mean(best == want)))
