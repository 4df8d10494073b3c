"""Randomised models and genotypes through every form of pass 2 (HIBAG_PASS2 = stream / hybrid / recompute) against the
oracle: every output bit-equal.  A wider net than tests/test_hip_parity.py::test_random_models (run by hand on the GPU box):
python tools/fuzz_modes.py [n_seeds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hibag_amd as hib
from oracle import oracle as O
O.build()
hib.hlaSetKernelTarget("hip")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = []
for mode in ("stream", "hybrid", "recompute"):
    os.environ["HIBAG_PASS2"] = mode
    for seed in range(n_seeds):
        rng = np.random.default_rng(7000 + seed)
        os.environ["HIBAG_STORE_PAIRS"] = str(int(rng.integers(0, 6)))
        n_hla = int(rng.integers(1, 40))
        n_snp = int(rng.integers(1, 80))
        cls = []
        for _ in range(int(rng.integers(1, 12))):
            k = int(rng.integers(1, min(n_snp, 45) + 1))
            H = int(rng.integers(1, 120))
            hla = np.sort(rng.integers(0, n_hla, H)).astype(np.int32)
            freq = 10.0 ** rng.uniform(-300 if seed % 5 == 0 else -5, 0, H)
            haplo = ["".join(rng.choice(["0", "1"], k)) for _ in range(H)]
            cls.append(hib.Classifier(snpidx=rng.choice(n_snp, k, replace=False), freq=freq, hla=hla, haplo=haplo))
        model = hib.HlaAttrBagObj(n_samp=0, n_snp=n_snp, hla_allele=[f"{i:02d}" for i in range(n_hla)], classifiers=cls)
        n = int(rng.integers(1, 700))
        if seed % 2:
            G = rng.choice(np.array([0, 1, 2, hib.NA_INTEGER, -1, 3], np.int64), size=(n, n_snp), p=[.3, .3, .3, .04, .03, .03]).astype(np.int32)
        else:
            c0 = cls[0]
            H0 = np.array([[int(ch) for ch in h] for h in c0.haplo])
            G = rng.integers(0, 3, size=(n, n_snp)).astype(np.int32)
            a, b = rng.integers(0, len(c0.haplo), n), rng.integers(0, len(c0.haplo), n)
            G[:, np.asarray(c0.snpidx)] = H0[a] + H0[b]
            G[rng.random(G.shape) < 0.02] = hib.NA_INTEGER
        for vote in (1, 2):
            m = hib.hlaModelFromObj(model)
            got = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
            m.close()
            want = O.predict(O.flatten(model), G, vote_method=vote)
            for key in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
                if not np.array_equal(got[key], want[key], equal_nan=True):
                    bad.append((mode, seed, vote, key)); break
    print(mode, "done;", len(bad), "mismatches so far", flush=True)
print("MISMATCHES:", bad if bad else "none")
