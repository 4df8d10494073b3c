#!/usr/bin/env python3
"""Per-width parity diagnostic: one small model per SNP count k, HIP library against the oracle (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hibag_amd
from hibag_amd import synth
from oracle import oracle as O

O.build()
hibag_amd.hlaSetKernelTarget("hip")
bad = []
for k in list(range(1, 34)) + [40]:
    model, founders, af = synth.make_model("hla-a-small", n_classifier=3, snp_counts=[k, k, k], wide_classifier=False, seed=100 + k)
    G, _ = synth.make_samples(founders, af, 96, seed=k, miss=0.05)
    got = hibag_amd.hlaModelFromObj(model).predict_raw(G, 1, want_dosage=True, want_prob=True)
    want = O.predict(O.flatten(model), G, vote_method=1)
    eq = {key: bool(np.array_equal(got[key], want[key], equal_nan=True)) for key in ("h1", "prob", "matching", "postprob")}
    if not all(eq.values()):
        i = int(np.argmax(np.any(got["postprob"] != want["postprob"], axis=1)))
        bad.append(k)
        print("k", k, eq, "first bad sample", i, "matching", got["matching"][i], want["matching"][i])
print("bad widths:", bad)
