"""Would cutting ONE 10,000-sample call into two halves on two streams of different priority cost the kernels anything?
(host-inclusive rate, DESIGN.md section 5: with the halves' kernels side by side, the upload of the second half and the download
of the first could run beside kernels -- if the halves together take no longer than the whole.)  Device-resident data, two
replicas of the model on ONE device; the second stream starts `delay` us late (its upload).  Python only.

    python tools/two_half_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hibag_amd
from hibag_amd import synth

dev = torch.device("cuda", 0)
hibag_amd.hlaSetKernelTarget("hip")
obj, founders, afreq = synth.make_model("hla-b")
n = 10_000
geno, truth = synth.make_samples(founders, afreq, n, seed=synth.DEFAULT_SEED + 1)
S, n_hla = obj.n_snp, obj.n_hla
m = [hibag_amd.hlaModelFromObj(obj, device=0)]
m.append(m[0].replicate(0))
g = torch.from_numpy(geno).to(dev)
h1 = torch.empty(n, dtype=torch.int32, device=dev); h2 = torch.empty_like(h1)
pr = torch.empty(n, dtype=torch.float64, device=dev); mt = torch.empty_like(pr)
ds = torch.empty((n, n_hla), dtype=torch.float64, device=dev)


def run(mi, stream, lo, hi):
    m[mi].predict_device(g.data_ptr() + 4 * lo * S, hi - lo, 1, h1.data_ptr() + 4 * lo, h2.data_ptr() + 4 * lo,
                         pr.data_ptr() + 8 * lo, mt.data_ptr() + 8 * lo, ds.data_ptr() + 8 * lo * n_hla, None, stream=stream.cuda_stream)


torch.cuda.synchronize()
s_probe = torch.cuda.Stream(dev)
with torch.cuda.stream(s_probe):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(10_000_000); e1.record()
torch.cuda.synchronize()
cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)


def med(fn, reps=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3


one = torch.cuda.Stream(dev)
print(f"one batch of {n}: {med(lambda: run(0, one, 0, n)):.3f} ms")
want1 = h1.clone()
for pa, pb, label in ((0, 0, "equal priority"), (-1, 0, "first half high priority")):
    sa, sb = torch.cuda.Stream(dev, priority=pa), torch.cuda.Stream(dev, priority=pb)
    for frac in (0.5, 0.33, 0.25):
        cut = (int(n * frac) + 63) // 64 * 64
        for delay in (0, 60, 120):
            def two():
                if delay:
                    with torch.cuda.stream(sb):
                        torch.cuda._sleep(int(delay * cyc_per_us))
                run(0, sa, 0, cut)
                run(1, sb, cut, n)
            h1.zero_()
            ms = med(two)
            ok = bool(torch.equal(h1, want1))
            print(f"  {label:26s} first part {cut:5d}, second stream {delay:3d} us late: {ms:.3f} ms{'' if ok else '  [CALLS DIFFER]'}")
print("faults", [int(x.handover_faults()) for x in m])
