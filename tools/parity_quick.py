import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hibag_amd
from hibag_amd import synth
from oracle import oracle as O
O.build()
hibag_amd.hlaSetKernelTarget("hip")
ok = True
for shape, n in (("hla-b", 256), ("hla-a-small", 300), ("hla-drb1", 128)):      # (DRB1: every cell stored, pass 2 = k_accum_cells)
    model, founders, af = synth.make_model(shape)
    G, _ = synth.make_samples(founders, af, n)
    got = hibag_amd.hlaModelFromObj(model).predict_raw(G, 1, want_dosage=True, want_prob=True)
    want = O.predict(O.flatten(model), G, vote_method=1, avx2=True, n_threads=8)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        e = np.array_equal(got[k], want[k], equal_nan=True)
        ok &= e
        print(shape, k, e)
print("ALL EQUAL" if ok else "MISMATCH")
