"""Randomised models and genotypes through every form of pass 2 (HIBAG_PASS2 = stream / hybrid / recompute, read when a
model is finalized) and random store thresholds, both vote methods, against the oracle: every output bit-equal.  The
loop being split three ways is CAlg_Prediction::_PostProb2 + the ensemble step, src/LibHLA.cpp:1776-1829, :2448-2480."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_SEEDS = 20


@pytest.mark.parametrize("mode", ["stream", "hybrid", "recompute"])
def test_random_models_in_every_pass2_mode(mode, oracle, monkeypatch):
    import hibag_amd as hib
    hib.hlaSetKernelTarget("hip")
    monkeypatch.setenv("HIBAG_PASS2", mode)
    bad = []
    for seed in range(N_SEEDS):
        rng = np.random.default_rng(7000 + seed)
        monkeypatch.setenv("HIBAG_STORE_PAIRS", str(int(rng.integers(0, 6))))
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
            G = rng.choice(np.array([0, 1, 2, hib.NA_INTEGER, -1, 3], np.int64), size=(n, n_snp),
                           p=[.3, .3, .3, .04, .03, .03]).astype(np.int32)
        else:
            c0 = cls[0]
            H0 = np.array([[int(ch) for ch in h] for h in c0.haplo])
            G = rng.integers(0, 3, size=(n, n_snp)).astype(np.int32)
            a, b = rng.integers(0, len(c0.haplo), n), rng.integers(0, len(c0.haplo), n)
            G[:, np.asarray(c0.snpidx)] = H0[a] + H0[b]
            G[rng.random(G.shape) < 0.02] = hib.NA_INTEGER
        flat = oracle.flatten(model)
        for vote in (1, 2):
            m = hib.hlaModelFromObj(model)
            got = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
            m.close()
            want = oracle.predict(flat, G, vote_method=vote)
            for key in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
                if not np.array_equal(got[key], want[key], equal_nan=True):
                    bad.append((seed, vote, key))
                    break
    assert not bad, bad
