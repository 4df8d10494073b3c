"""Would config 4 (every cell stored: pass 1 issue-bound, pass 2 HBM-bound) gain from running pass 2 of one half of the batch beside
pass 1 of the other?  Emulated with two replicas of the model on one device (a workspace each), each predicting half of the samples
on a stream of its own, the first at higher priority -- against one call on the whole batch.

    python tools/split_probe.py [samples]
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import hibag_amd
from hibag_amd import synth
hibag_amd.hlaSetKernelTarget("hip")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mdl, founders, af = synth.make_model("hla-drb1", seed=synth.DEFAULT_SEED)
G, _ = synth.make_samples(founders, af, n, seed=5)
dev = torch.device("cuda", 0)
m = hibag_amd.hlaModelFromObj(mdl)
r = m.replicate(0)
dg = torch.from_numpy(G).to(dev)

def outs(k):
    return [torch.empty(k, dtype=torch.int32, device=dev), torch.empty(k, dtype=torch.int32, device=dev), torch.empty(k, dtype=torch.float64, device=dev),
            torch.empty(k, dtype=torch.float64, device=dev), torch.empty((k, mdl.n_hla), dtype=torch.float64, device=dev)]
whole = outs(n)
def one():
    m.predict_device(dg.data_ptr(), n, 1, *[t.data_ptr() for t in whole], None, stream=torch.cuda.current_stream(dev).cuda_stream)
def timed(f, reps=8):
    f(); torch.cuda.synchronize(dev)
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); torch.cuda.synchronize(dev); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
t_one = timed(one)
print(f"one call, {n} samples: {t_one:.3f} ms = {n / t_one * 1e3:,.0f} samples/s")
for frac in (0.5, 0.4, 0.6):
    for pri in ((-1, 0), (0, 0)):
        na = int(n * frac) // 256 * 256
        nb = n - na
        sa, sb = torch.cuda.Stream(dev, priority=pri[0]), torch.cuda.Stream(dev, priority=pri[1])
        oa, ob = outs(na), outs(nb)
        ga, gb = dg[:na].contiguous(), dg[na:].contiguous()
        def two():
            m.predict_device(ga.data_ptr(), na, 1, *[t.data_ptr() for t in oa], None, stream=sa.cuda_stream)
            r.predict_device(gb.data_ptr(), nb, 1, *[t.data_ptr() for t in ob], None, stream=sb.cuda_stream)
        t_two = timed(two)
        same = all(torch.equal(torch.cat([a, b]), w) for a, b, w in zip(oa, ob, whole))
        print(f"two halves ({na} + {nb}), stream priorities {pri}: {t_two:.3f} ms = {n / t_two * 1e3:,.0f} samples/s ({100 * (t_one / t_two - 1):+.1f} %), outputs identical: {same}; "
              f"hand-over faults so far {m.handover_faults()} + {r.handover_faults()}, status {m.status()} {r.status()}")
        # each half alone, for scale
        if frac == 0.5 and pri == (-1, 0):
            ta = timed(lambda: m.predict_device(ga.data_ptr(), na, 1, *[t.data_ptr() for t in oa], None, stream=sa.cuda_stream))
            print(f"   the first half alone: {ta:.3f} ms")
