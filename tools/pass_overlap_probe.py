"""Does running pass 1 of one half-batch beside pass 2 of the other pay?  (round 4)

Pass 1 (k_total) keeps the vector ALU ~90 % busy; pass 2 (k_accum) waits on LDS and memory about half the time.  Two replicas
of the model on ONE device, each fed its share of the 10,000 samples on its own stream, free-running for K steps with the second
stream started `delay` microseconds late, so that the second stream's pass 1 falls beside the first stream's pass 2.
Prints the aggregate samples/s per arrangement next to the one-stream baseline.  Python only: no library change.

    python tools/pass_overlap_probe.py [--steps 20]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import hibag_amd
from hibag_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--samples", type=int, default=10_000)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    hibag_amd.hlaSetKernelTarget("hip")
    obj, founders, afreq = synth.make_model("hla-b")
    n = a.samples
    geno, truth = synth.make_samples(founders, afreq, n, seed=synth.DEFAULT_SEED + 1)
    S, n_hla = obj.n_snp, obj.n_hla
    m = [hibag_amd.hlaModelFromObj(obj, device=0)]
    m.append(m[0].replicate(0))
    g = torch.from_numpy(geno).to(dev)
    h1 = torch.empty(n, dtype=torch.int32, device=dev); h2 = torch.empty_like(h1)
    pr = torch.empty(n, dtype=torch.float64, device=dev); mt = torch.empty_like(pr)
    ds = torch.empty((n, n_hla), dtype=torch.float64, device=dev)
    st = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

    def run(mi, si, lo, hi):
        m[mi].predict_device(g.data_ptr() + 4 * lo * S, hi - lo, 1, h1.data_ptr() + 4 * lo, h2.data_ptr() + 4 * lo,
                             pr.data_ptr() + 8 * lo, mt.data_ptr() + 8 * lo, ds.data_ptr() + 8 * lo * n_hla, None,
                             stream=st[si].cuda_stream)

    # the spin kernel's clock: how many `cycles` make a microsecond
    torch.cuda.synchronize()
    with torch.cuda.stream(st[1]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(10_000_000); e1.record()
    torch.cuda.synchronize()
    cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)

    def timed(plan, delay_us, steps):
        """plan: list of (model, stream, lo, hi) per step"""
        for _ in range(3):
            for p in plan:
                run(*p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if delay_us:
            with torch.cuda.stream(st[1]):
                torch.cuda._sleep(int(delay_us * cyc_per_us))
        for _ in range(steps):
            for p in plan:
                run(*p)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    want1, want2 = None, None
    res = []
    base = [(0, 0, 0, n)]
    for rep in range(2):
        dt = timed(base, 0, a.steps)
        res.append(("one stream, one batch", 0, n * a.steps / dt, dt / a.steps * 1e3))
        if want1 is None:
            want1, want2 = h1.clone(), h2.clone()
    half = (n // 2 + 63) // 64 * 64
    q = (n // 4 + 63) // 64 * 64
    plans = {
        "two streams, halves": [(0, 0, 0, half), (1, 1, half, n)],
        "two streams, quarters": [(0, 0, 0, q), (1, 1, q, 2 * q), (0, 0, 2 * q, 3 * q), (1, 1, 3 * q, n)],
        "one stream, halves (control)": [(0, 0, 0, half), (0, 0, half, n)],
    }
    for name, plan in plans.items():
        for delay in ((0,) if "control" in name else (0, 100, 200, 300, 400, 600)):
            h1.zero_(); h2.zero_()
            dt = timed(plan, delay, a.steps)
            ok = bool(torch.equal(h1, want1) and torch.equal(h2, want2))
            res.append((name + ("" if ok else "  [CALLS DIFFER]"), delay, n * a.steps / dt, dt / a.steps * 1e3))
    dt = timed(base, 0, a.steps)
    res.append(("one stream, one batch (again)", 0, n * a.steps / dt, dt / a.steps * 1e3))
    # what the per-kernel HIP events cost the step
    for label, arg in (("events: all four kernels", True), ("events: total + accum", ("total", "accum")), ("events: total", ("total",)),
                       ("events: none", False), ("events: all four kernels", True), ("events: none", False)):
        m[0].set_timing(arg)
        m[0].reset_timing()
        dt = timed(base, 0, a.steps)
        tm = m[0].get_timing()
        res.append((label + "  " + " ".join(f"{k}={v[0] / max(v[1], 1):.4f}" for k, v in tm.items() if v[1]), 0, n * a.steps / dt, dt / a.steps * 1e3))
    m[0].set_timing(False)
    print(f"{n} samples per step, {a.steps} steps free-running; sleep clock {cyc_per_us:.1f} cycles/us")
    for name, delay, v, ms in res:
        print(f"  {name:44s} delay {delay:4d} us   {v / 1e6:6.3f} M samples/s   {ms:7.3f} ms/step")
    print("faults", [int(x.handover_faults()) for x in m])


if __name__ == "__main__":
    main()
