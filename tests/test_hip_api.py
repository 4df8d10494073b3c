"""GPU tests of the host API's input routes: ``hlaPredict`` / the C ABI on a genotype matrix in EITHER memory order.

R hands ``HIBAG_Predict_*`` the memory of its SNP x sample matrix, i.e. sample-major (``R/HIBAG.R:715-725``); a numpy
[SNP, sample] array is the transpose of that.  ``hibag_hip_predict_snp_major`` takes the latter as it is (rows picked and
flipped on the device, only the model's rows uploaded), ``hibag_hip_predict`` / ``_mapped`` the former; every output of
the two must agree bit for bit, and with the oracle.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("h1", "h2", "prob", "matching", "dosage", "postprob")
NA = -2147483648


@pytest.fixture(scope="module")
def hib():
    import hibag_amd
    hibag_amd.hlaSetKernelTarget("hip")
    return hibag_amd


def same(a, b, keys=KEYS):
    for k in keys:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize("vote", [1, 2])
@pytest.mark.parametrize("n", [1, 63, 64, 333, 4100])
def test_snp_major_equals_sample_major_and_the_oracle(hib, oracle, n, vote):
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-a-small", seed=21)
    G, _ = synth.make_samples(founders, af, n, seed=22)               # [n, S] sample-major
    G[n // 2] = NA
    m = hib.hlaModelFromObj(model)
    want = oracle.predict(oracle.flatten(model), G, vote_method=vote)
    plain = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
    same(plain, want)
    rows = np.ascontiguousarray(G.T)                                   # [S, n] SNP-major
    same(m.predict_snp_major(rows, None, None, vote, want_dosage=True, want_prob=True), want)
    # rows of a wider array: ld > n_samp
    wide = np.full((model.n_snp, n + 37), 7, np.int32)
    wide[:, 5:5 + n] = rows
    same(m.predict_snp_major(wide[:, 5:5 + n], None, None, vote, want_dosage=True, want_prob=True), want)
    # only some outputs
    part = m.predict_snp_major(rows, None, None, vote, want_dosage=False, want_prob=False)
    same(part, want, ("h1", "h2", "prob", "matching"))


@pytest.mark.parametrize("slice_env", [None, "128"])
def test_snp_major_scattered_rows_flips_and_slices(hib, oracle, monkeypatch, slice_env):
    """The cohort has its own SNPs in its own order: the model's rows are gathered (scattered -> pinned staging;
    consecutive -> one strided block), flipped on the device, absent ones are missing; with a small slice the cohort goes
    through the three-stream pipeline."""
    from hibag_amd import synth
    if slice_env:
        monkeypatch.setenv("HIBAG_STAGED_SLICE", slice_env)
    model, founders, af = synth.make_model("hla-a-small", seed=31)
    n = 1000
    G, _ = synth.make_samples(founders, af, n, seed=32)
    S = model.n_snp
    rng = np.random.default_rng(33)
    m = hib.hlaModelFromObj(model)
    for case in ("scattered", "consecutive"):
        n_geno = 3 * S
        cohort = rng.integers(0, 3, (n_geno, n)).astype(np.int32)
        col = np.full(S, -1, np.int32)
        have = rng.random(S) < 0.9
        if case == "scattered":
            col[have] = rng.choice(n_geno, int(have.sum()), replace=False)
        else:
            col[have] = 11 + np.arange(int(have.sum()))
        flip = (rng.random(S) < 0.3).astype(np.int32)
        for k in np.where(have)[0]:
            g = G[:, k]
            cohort[col[k]] = np.where((g >= 0) & (g <= 2) & (flip[k] != 0), 2 - g, g)
        Gm = G.copy()
        Gm[:, ~have] = NA
        want = oracle.predict(oracle.flatten(model), Gm, vote_method=1)
        got = m.predict_snp_major(cohort, col, flip, 1, want_dosage=True, want_prob=True)
        same(got, want)
        # the sample-major entry on the transposed cohort: same bits
        same(m.predict_mapped(np.ascontiguousarray(cohort.T), col, flip, 1, want_dosage=True, want_prob=True), want)
    assert m.handover_faults() == 0


def test_snp_major_device_entry_and_errors(hib, oracle):
    import ctypes as C
    import torch
    from hibag_amd import _lib, synth
    model, founders, af = synth.make_model("hla-a-small", seed=41)
    n = 700
    G, _ = synth.make_samples(founders, af, n, seed=42)
    m = hib.hlaModelFromObj(model)
    want = oracle.predict(oracle.flatten(model), G, vote_method=1)
    dev = torch.device("cuda", m.device())
    ld = n + 24
    rows = torch.zeros((model.n_snp, ld), dtype=torch.int32, device=dev)
    rows[:, :n] = torch.from_numpy(np.ascontiguousarray(G.T)).to(dev)
    o = dict(h1=torch.empty(n, dtype=torch.int32, device=dev), h2=torch.empty(n, dtype=torch.int32, device=dev),
             prob=torch.empty(n, dtype=torch.float64, device=dev), matching=torch.empty(n, dtype=torch.float64, device=dev),
             dosage=torch.empty((n, model.n_hla), dtype=torch.float64, device=dev),
             postprob=torch.empty((n, model.n_cell), dtype=torch.float64, device=dev))
    p = lambda t: C.c_void_p(t.data_ptr())
    st = torch.cuda.current_stream(dev)
    _lib.check(_lib.lib().hibag_hip_predict_snp_major_device(
        m.handle, p(rows), ld, n, model.n_snp, None, None, 1, p(o["h1"]), p(o["h2"]), p(o["prob"]), p(o["matching"]),
        p(o["dosage"]), p(o["postprob"]), C.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize(dev)
    assert m.status() == 0
    same({k: v.cpu().numpy() for k, v in o.items()}, want)
    # argument errors of the host entry
    g = np.zeros((model.n_snp, 4), np.int32)
    bad = np.full(model.n_snp, model.n_snp, np.int32)
    with pytest.raises(_lib.HibagHipError, match="outside the"):
        m.predict_snp_major(g, bad)
    with pytest.raises(ValueError):
        m.predict_snp_major(np.zeros((model.n_snp - 1, 4), np.int32))
    h = np.zeros(4, np.int32); d = np.zeros(4)
    q = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = _lib.lib().hibag_hip_predict_snp_major(m.handle, q(g), 3, 4, model.n_snp, None, None, 1, q(h), q(h), q(d), q(d), None, None)
    assert rc == -1 and b"smaller than n_samp" in _lib.lib().hibag_hip_last_error()
    # zero samples: nothing to do, no error
    assert len(m.predict_snp_major(np.zeros((model.n_snp, 0), np.int32))["h1"]) == 0
    # a cohort that has NONE of the model's SNPs: no row travels at all, every genotype is missing, every call NA -- as the
    # plain entry says for an all-missing matrix
    none = m.predict_snp_major(np.ones((5, 9), np.int32), np.full(model.n_snp, -1, np.int32), None, 1, want_dosage=True, want_prob=True)
    ref = m.predict_raw(np.full((9, model.n_snp), NA, np.int32), 1, want_dosage=True, want_prob=True)
    same(none, ref)
    assert np.all(none["h1"] == NA)
    # narrow integer types are widened once (uint8 genotypes with 255 = missing)
    g8 = np.where((G.T >= 0) & (G.T <= 2), G.T, 255).astype(np.uint8)
    r8 = hib.hlaPredict(m, np.ascontiguousarray(g8), type="response+prob", verbose=False)
    assert np.array_equal(r8.h1, want["h1"]) and np.array_equal(r8.postprob, want["postprob"].T, equal_nan=True)


def test_hlaPredict_takes_either_memory_order_without_copies(hib, oracle, hapmap_geno, model_a):
    """``hlaPredict`` on the same cohort as an R-ordered (column-major) matrix, a numpy-ordered one, a strided view and a
    double matrix with NaN: identical results; the outputs are views of the sample-major arrays the library filled."""
    from conftest import align_geno
    from hibag_amd import HlaSNPGeno
    assert hapmap_geno.genotype.flags.f_contiguous            # .RData genotypes keep R's memory order
    dev = hib.hlaModelFromObj(model_a)
    G = align_geno(model_a, hapmap_geno, hapmap_geno.sample_id)
    want = oracle.predict(oracle.flatten(model_a), G)

    def variant(mat):
        return HlaSNPGeno(genotype=mat, sample_id=hapmap_geno.sample_id, snp_id=hapmap_geno.snp_id,
                          snp_position=hapmap_geno.snp_position, snp_allele=hapmap_geno.snp_allele, assembly=hapmap_geno.assembly)
    g = np.asarray(hapmap_geno.genotype)
    wide = np.zeros((g.shape[0], 2 * g.shape[1]), np.int32)
    wide[:, ::2] = g
    dbl = np.where(g == NA, np.nan, g.astype(np.float64))
    variants = {"R order": variant(np.asfortranarray(g)), "numpy order": variant(np.ascontiguousarray(g)),
                "strided": variant(wide[:, ::2]), "double + NaN": variant(dbl), "double, R order": variant(np.asfortranarray(dbl))}
    for name, s in variants.items():
        res = hib.hlaPredict(dev, s, type="response+prob", match_type="RefSNP+Position", verbose=False)
        assert np.array_equal(res.h1, want["h1"]) and np.array_equal(res.h2, want["h2"]), name
        assert np.array_equal(res.prob, want["prob"]) and np.array_equal(res.matching, want["matching"]), name
        assert np.array_equal(res.dosage, want["dosage"].T) and np.array_equal(res.postprob, want["postprob"].T), name
        assert res.dosage.flags.f_contiguous and res.postprob.flags.f_contiguous, name       # views, R's memory order
        assert res.allele1 == [model_a.hla_allele[i] for i in want["h1"]], name
        assert res.allele2 == [model_a.hla_allele[i] for i in want["h2"]], name
    # the plain-matrix branch in both orders, and over two replicas on the one device
    cols = np.ascontiguousarray(G.T)
    for mat in (cols, np.asfortranarray(cols), cols.astype(np.float64)):
        r = hib.hlaPredict(dev, mat, type="response+dosage", verbose=False)
        assert np.array_equal(r.h1, want["h1"]) and np.array_equal(r.dosage, want["dosage"].T)
        r2 = hib.hlaPredict(dev, mat, cl=[0, 0], type="response+dosage", verbose=False)
        assert np.array_equal(r2.h1, want["h1"]) and np.array_equal(r2.dosage, want["dosage"].T)
    # NA calls are counted without a Python loop and named None
    none = np.full((model_a.n_snp, 3), NA, np.int32)
    with pytest.warns(UserWarning, match="No prediction outputs for 3 individuals"):
        r = hib.hlaPredict(dev, none, type="response", verbose=False)
    assert r.allele1 == [None] * 3 and r.allele2 == [None] * 3
