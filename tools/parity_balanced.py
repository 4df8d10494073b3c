"""Balanced launches (more work items than resident workgroups) against the oracle: every output bit-equal.
Run on the GPU box: python tools/parity_balanced.py [n_samples ...]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hibag_amd
from hibag_amd import synth
from oracle import oracle as O
O.build()
hibag_amd.hlaSetKernelTarget("hip")
ok = True
sizes = [int(a) for a in sys.argv[1:]] or [2200, 10000]
model, founders, af = synth.make_model("hla-b")
flat = O.flatten(model)
m = hibag_amd.hlaModelFromObj(model)
for n in sizes:
    G, _ = synth.make_samples(founders, af, n, seed=n)
    G[3, :] = hibag_amd.NA_INTEGER
    t0 = time.time()
    got = m.predict_raw(G, 1, want_dosage=True, want_prob=True)
    t1 = time.time()
    sub = np.arange(n) if n <= 3000 else np.sort(np.random.default_rng(1).choice(n, 1500, replace=False))
    want = O.predict(flat, G[sub], vote_method=1, avx2=True, n_threads=8)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        e = np.array_equal(got[k][sub], want[k], equal_nan=True)
        ok &= e
        print(n, k, e, flush=True)
    print(n, "gpu %.2fs oracle %.2fs" % (t1 - t0, time.time() - t1))
print("ALL EQUAL" if ok else "MISMATCH")
