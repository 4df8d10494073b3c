"""GPU parity: the HIP path (through the C ABI of libhibag_hip.so) against the
CPU oracle on the same inputs, against values stored by the reference in its
own fixtures, and -- at BASELINE.json's full size -- through size-independent
properties.  Integer outputs (allele calls) and every double are required to be
BIT-IDENTICAL; the north star only asks for 1e-10 relative on posteriors, the
kernels are exact by construction so the tests hold them to that.
"""

import ctypes as C

import numpy as np
import pytest

from conftest import align_geno

pytestmark = pytest.mark.gpu

KEYS = ("h1", "h2", "prob", "matching", "dosage", "postprob")


@pytest.fixture(scope="module")
def hib():
    import hibag_amd
    info = hibag_amd.hlaSetKernelTarget("hip")
    assert "gfx950" in info[0]
    return hibag_amd


def assert_same(got, want, keys=KEYS):
    for k in keys:
        a, b = got[k], want[k]
        assert a.shape == b.shape, k
        if not np.array_equal(a, b, equal_nan=True):
            bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b)))) if a.dtype.kind == "f" else np.argwhere(a != b)
            i = tuple(bad[0])
            raise AssertionError(f"{k}: {len(bad)} entries differ, first at {i}: hip={a[i]!r} oracle={b[i]!r}")


def test_kernel_target_selection(hib):
    # the reference errors for a target the build lacks (src/LibHLA.cpp:1374-1375 etc.)
    with pytest.raises(hib.HibagHipError):
        hib.hlaSetKernelTarget("avx2")
    with pytest.raises(ValueError):
        hib.hlaSetKernelTarget("cuda")


@pytest.mark.parametrize("vote", [1, 2])
@pytest.mark.parametrize("which", ["a", "oob"])
def test_bundled_hla_a_model_hapmap(hib, oracle, hapmap_geno, model_a, model_oob, which, vote):
    """BASELINE config 1: bundled HLA-A model x the 60 HapMap CEU samples."""
    model = model_a if which == "a" else model_oob
    G = align_geno(model, hapmap_geno, hapmap_geno.sample_id)
    want = oracle.predict(oracle.flatten(model), G, vote_method=vote)
    got = hib.hlaModelFromObj(model).predict_raw(G, vote, want_dosage=True, want_prob=True)
    assert_same(got, want)


def test_matching_stored_by_the_reference(hib, hapmap_geno, model_oob):
    """`matching` written by the reference's own hlaPredict() into OutOfBag.RData."""
    G = align_geno(model_oob, hapmap_geno)
    got = hib.hlaModelFromObj(model_oob).predict_raw(G, 1, want_dosage=False)
    complete = np.array([np.all((g >= 0) & (g <= 2)) for g in G])
    assert complete.sum() == 27
    assert np.array_equal(got["matching"][complete], model_oob.matching[complete])


def test_knocked_out_snps_change_classifier_weights(hib, oracle, hapmap_geno, model_a):
    G = align_geno(model_a, hapmap_geno, hapmap_geno.sample_id).copy()
    rng = np.random.default_rng(7)
    G[:, rng.choice(G.shape[1], 120, replace=False)] = hib.NA_INTEGER     # whole SNPs gone
    G[rng.random(G.shape) < 0.05] = -1                                      # scattered, other missing code
    G[3, :] = 3                                                             # out-of-range value = missing
    want = oracle.predict(oracle.flatten(model_a), G)
    got = hib.hlaModelFromObj(model_a).predict_raw(G, 1, want_dosage=True, want_prob=True)
    assert_same(got, want)
    assert got["h1"][3] == hib.NA_INTEGER and got["prob"][3] == 0 and np.isnan(got["matching"][3])


@pytest.mark.parametrize("shape,n", [("hla-a-small", 200), ("hla-b", 192), ("hla-drb1", 64)])
@pytest.mark.parametrize("vote", [1, 2])
def test_synthetic_models(hib, oracle, shape, n, vote):
    from hibag_amd import synth
    model, founders, af = synth.make_model(shape)
    G, _ = synth.make_samples(founders, af, n)
    G[1, :] = hib.NA_INTEGER
    want = oracle.predict(oracle.flatten(model), G, vote_method=vote, avx2=True, n_threads=8)
    got = hib.hlaModelFromObj(model).predict_raw(G, vote, want_dosage=True, want_prob=True)
    assert_same(got, want)


@pytest.mark.parametrize("vote", [1, 2])
def test_every_classifier_width(hib, oracle, vote):
    """One classifier per SNP count 1..40 and 63..66, 96, 127, 128: every engine variant (FP4 in one K step up to 28 SNPs,
    int8 with the distance offset inside the dot product / in the accumulators for 29..31 / 32, FP4 in two to four
    chained K steps up to 112, the VALU engine's word counts beyond) and their boundaries."""
    from hibag_amd import synth
    ks = list(range(1, 41)) + [63, 64, 65, 66, 96, 127, 128]
    model, founders, af = synth.make_model("hla-a-small", seed=77, n_classifier=len(ks), n_snp=160,
                                           snp_counts=ks, wide_classifier=False)
    assert sorted(len(c.snpidx) for c in model.classifiers) == sorted(ks)
    G, _ = synth.make_samples(founders, af, 150, seed=78, miss=0.05)
    G[3, :] = hib.NA_INTEGER
    want = oracle.predict(oracle.flatten(model), G, vote_method=vote, avx2=True, n_threads=8)
    got = hib.hlaModelFromObj(model).predict_raw(G, vote, want_dosage=True, want_prob=True)
    assert_same(got, want)


def test_only_wide_classifiers(hib, oracle):
    """Every classifier between 33 and 112 SNPs: the FP4 engine in two to four K steps chained through the accumulator
    (pass 1 in k_total_wide beside an empty k_total, every cell stored), both vote methods, a batch of more than one
    group quad."""
    from hibag_amd import synth
    ks = [33, 40, 56, 57, 84, 85, 100, 112]
    model, founders, af = synth.make_model("hla-a-small", seed=31, n_classifier=len(ks), n_snp=130, snp_counts=ks, wide_classifier=False)
    G, _ = synth.make_samples(founders, af, 700, seed=32, miss=0.03)
    G[5, :] = hib.NA_INTEGER
    G[128:192, :] = hib.NA_INTEGER
    m = hib.hlaModelFromObj(model)
    assert m.stored_cells() > 0 and m.second_pass_pairs() == 0
    for vote in (1, 2):
        assert_same(m.predict_raw(G, vote, want_dosage=True, want_prob=True),
                    oracle.predict(oracle.flatten(model), G, vote_method=vote, avx2=True, n_threads=8))
    m.close()


def test_many_alleles(hib, oracle):
    """A locus with 180 alleles (16,290 allele pairs; HLA-B 4-digit reference panels are of this
    size) and 400 haplotypes per classifier: tiles, finish kernels and batching at a large P."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b", seed=21, n_hla=180, n_haplo=400, n_classifier=4, n_snp=120)
    G, _ = synth.make_samples(founders, af, 70, seed=22)
    G[2, :] = hib.NA_INTEGER
    for vote in (1, 2):
        want = oracle.predict(oracle.flatten(model), G, vote_method=vote, avx2=True, n_threads=8)
        got = hib.hlaModelFromObj(model).predict_raw(G, vote, want_dosage=True, want_prob=True)
        assert_same(got, want)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 130])
def test_ragged_batch_sizes(hib, oracle, n):
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-a-small", n_classifier=6)
    G, _ = synth.make_samples(founders, af, max(n, 1))
    G = G[:n]
    got = hib.hlaModelFromObj(model).predict_raw(G, 1, want_dosage=True, want_prob=True)
    want = oracle.predict(oracle.flatten(model), G)
    assert_same(got, want)


@pytest.mark.parametrize("vote", [1, 2])
def test_classifiers_unused_by_whole_sample_groups(hib, oracle, vote):
    """Classifiers whose SNPs are all missing for every sample of one 64-sample group (weight 0: src/LibHLA.cpp:2451
    passes them over) while other groups use them: pass 2 skips their blocks for that group only -- the running cell sum must
    come out of the skipped blocks as if they had been walked (the E-stream's 'starts a sum' header bit describes the walked
    stream) -- including first and last classifiers, runs of consecutive ones, and a group that uses none at all."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b", n_classifier=24, n_samp=400)
    G, _ = synth.make_samples(founders, af, 64 * 5 + 17, seed=synth.DEFAULT_SEED + 11)
    G = G.copy()
    snps = [np.asarray(c.snpidx) for c in model.classifiers]
    def knock(rows, classifiers):
        for k in classifiers:
            G[np.ix_(rows, snps[k])] = hib.NA_INTEGER
    knock(range(0, 64), [0, 1, 2, 9, 23])                   # group 0: the first ones, one in the middle, the last
    knock(range(64, 128), range(5, 15))                     # group 1: a run
    knock(range(128, 192), range(24))                       # group 2: uses nothing
    knock(range(192, 250), [3])                             # group 3: most of the group only -- the classifier stays in use
    knock(range(320, 337), [7, 8])                          # the ragged last group: all its real samples
    got = hib.hlaModelFromObj(model).predict_raw(G, vote, want_dosage=True, want_prob=True)
    want = oracle.predict(oracle.flatten(model), G, vote_method=vote)
    assert_same(got, want)


def test_underflow_gives_nan_like_the_reference(hib, oracle):
    """A classifier whose every pair is >= 65 mismatches away has total 0, so
    1/total = inf and 0*inf = NaN poisons the whole sample (src/LibHLA.cpp:1826-1828)."""
    from hibag_amd.model import Classifier, HlaAttrBagObj
    k = 100
    far = Classifier(np.arange(k), [0.5, 0.5], [0, 1], ["1" * k, "1" * k])
    near = Classifier(np.arange(4), [0.3, 0.3, 0.4], [0, 1, 2], ["0000", "0101", "1111"])
    model = HlaAttrBagObj(0, k, ["a", "b", "c"], [near, far])
    G = np.zeros((3, k), np.int32)        # all homozygous B: 2 mismatches per SNP against "111..."
    G[1, 40:] = hib.NA_INTEGER            # 40 typed SNPs -> 80 mismatches: exact zero; sample 2 sees neither
    G[2, :] = hib.NA_INTEGER
    want = oracle.predict(oracle.flatten(model), G)
    assert np.isnan(want["postprob"][0]).all() and want["h1"][0] == hib.NA_INTEGER
    got = hib.hlaModelFromObj(model).predict_raw(G, 1, want_dosage=True, want_prob=True)
    assert_same(got, want)
    got2 = hib.hlaModelFromObj(model).predict_raw(G, 2, want_dosage=True, want_prob=True)
    assert_same(got2, oracle.predict(oracle.flatten(model), G, vote_method=2))


def test_empty_alleles_and_single_haplotype(hib, oracle):
    from hibag_amd.model import Classifier, HlaAttrBagObj
    c1 = Classifier([2, 0], [1.0], [3], ["10"])                        # one haplotype, alleles 0-2,4 empty
    c2 = Classifier([1], [0.25, 0.75], [0, 4], ["0", "1"])
    model = HlaAttrBagObj(0, 3, list("abcde"), [c1, c2])
    G = np.array([[0, 1, 2], [2, 2, 0], [1, 1, 1], [-1, 0, -1]], np.int32)
    assert_same(hib.hlaModelFromObj(model).predict_raw(G, 1, True, True), oracle.predict(oracle.flatten(model), G))


def test_invalid_vote_method_message(hib, model_a):
    dev = hib.hlaModelFromObj(model_a)
    with pytest.raises(hib.HibagHipError, match="Invalid 'vote_method'."):
        dev.predict_raw(np.zeros((1, model_a.n_snp), np.int32), vote_method=3)


def test_hlaPredict_interface(hib, oracle, hapmap_geno, model_a):
    """The R-level call: SNPs matched by position, strands checked, R-shaped result."""
    dev = hib.hlaModelFromObj(model_a)
    res = hib.hlaPredict(dev, hapmap_geno, type="response+prob", match_type="RefSNP+Position", verbose=False)
    G = align_geno(model_a, hapmap_geno, hapmap_geno.sample_id)
    want = oracle.predict(oracle.flatten(model_a), G)
    assert res.sample_id == hapmap_geno.sample_id
    assert res.allele1 == [model_a.hla_allele[i] for i in want["h1"]]
    assert res.allele2 == [model_a.hla_allele[i] for i in want["h2"]]
    assert np.array_equal(res.prob, want["prob"]) and np.array_equal(res.matching, want["matching"])
    assert np.array_equal(res.dosage, want["dosage"].T) and np.array_equal(res.postprob, want["postprob"].T)
    assert res.pair_names[:3] == ["01:01/01:01", "02:01/01:01", "02:06/01:01"]
    # a plain matrix (n.snp x n.samp) takes the other branch of hlaPredict
    res2 = hib.hlaPredict(dev, G.T, type="response", verbose=False)
    assert res2.allele1 == res.allele1 and res2.dosage is None
    p = hib.hlaPredict(dev, G.T, type="prob", verbose=False)
    assert np.array_equal(p, want["postprob"].T)


# --- the HIBAG plugin table (TypeGPUExtProc), driven the way the host does --------------

class _THaplotype(C.Structure):          # inst/include/LibHLA_ext.h:261-299
    _fields_ = [("packed", C.c_int64 * 2), ("freq", C.c_double), ("freq_f32", C.c_float), ("hla", C.c_int)]


class _TGenotype(C.Structure):           # inst/include/LibHLA_ext.h:311-352
    _fields_ = [("s1", C.c_int64 * 2), ("s2", C.c_int64 * 2), ("boot", C.c_int), ("a1", C.c_int),
                ("a2", C.c_int), ("pad", C.c_int)]


class _Table(C.Structure):               # inst/include/LibHLA_ext.h:358-388
    _fields_ = [(n, C.c_void_p) for n in ("build_init", "build_done", "build_set_bootstrap", "build_haplomatch",
                                           "build_set_haplo_geno", "build_acc_oob", "build_acc_ib")] + [
        ("predict_init", C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(C.POINTER(_THaplotype)),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int))),
        ("predict_done", C.CFUNCTYPE(None)),
        ("predict_avg_prob", C.CFUNCTYPE(None, C.POINTER(_TGenotype), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)))]


def test_plugin_table_per_sample_path(hib, oracle, hapmap_geno, model_a):
    from hibag_amd import _lib
    assert C.sizeof(_THaplotype) == 32 and C.sizeof(_TGenotype) == 48
    tab = _Table.from_address(_lib.lib().hibag_hip_gpu_ext_proc())
    assert all(getattr(tab, n) is not None for n, _ in _Table._fields_[:7])   # build_* are implemented too
    fm = oracle.flatten(model_a)
    G = align_geno(model_a, hapmap_geno, hapmap_geno.sample_id)[:8].copy()
    G[2, ::3] = hib.NA_INTEGER
    nC = fm.n_classifier
    lists = []
    for c in range(nC):                                  # what _Init_GPU_PredHLA hands over (src/LibHLA.cpp:2498-2523)
        _, lens, bits, freq, _ = fm.classifier(c)
        arr = (_THaplotype * len(freq))()
        hla = np.repeat(np.arange(fm.n_hla), lens)
        for i in range(len(freq)):
            arr[i].packed[0] = int(bits[i, 0]) - (1 << 64 if bits[i, 0] >> 63 else 0)
            arr[i].packed[1] = -1                        # garbage above n_snp, as the reference leaves it
            arr[i].freq = freq[i]; arr[i].freq_f32 = freq[i]; arr[i].hla = int(hla[i])
        lists.append(arr)
    ptrs = (C.POINTER(_THaplotype) * nC)(*[C.cast(a, C.POINTER(_THaplotype)) for a in lists])
    n_hap = (C.c_int * nC)(*[len(a) for a in lists])
    n_snp = (C.c_int * nC)(*[int(v) for v in fm.n_snp_c])
    tab.predict_init(fm.n_hla, nC, ptrs, n_hap, n_snp)
    want = oracle.predict(fm, G)
    sw = np.zeros(fm.n_snp_total, np.int32)
    for c in model_a.classifiers:
        sw[c.snpidx] += 1
    P = fm.n_hla * (fm.n_hla + 1) // 2
    for i in range(len(G)):                              # _PredictHLA's GPU branch (src/LibHLA.cpp:2418-2441)
        geno = (_TGenotype * nC)()
        wt = (C.c_double * nC)()
        for c, cl in enumerate(model_a.classifiers):
            s1, s2 = oracle.int_to_snp(G[i], cl.snpidx)
            for w in range(2):
                geno[c].s1[w] = int(s1[w]) - (1 << 64 if s1[w] >> 63 else 0)
                geno[c].s2[w] = int(s2[w]) - (1 << 64 if s2[w] >> 63 else 0)
            g = G[i][cl.snpidx]
            ok = (g >= 0) & (g <= 2)
            tot = int(sw[cl.snpidx].sum())
            wt[c] = float(int(sw[cl.snpidx][ok].sum())) / tot if tot > 0 else 0.0
        prob = (C.c_double * P)()
        match = (C.c_double * 1)()
        tab.predict_avg_prob(geno, wt, prob, match)
        assert np.array_equal(np.frombuffer(prob, np.float64), want["postprob"][i], equal_nan=True), i
        assert match[0] == want["matching"][i]
    tab.predict_done()


# --- BASELINE config 2 at full size: properties that do not need the oracle ------------

def test_full_size_hla_b_properties(hib, oracle):
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b")
    N = 10_000
    G, truth = synth.make_samples(founders, af, N)
    dev = hib.hlaModelFromObj(model)
    out = dev.predict_raw(G, 1, want_dosage=True, want_prob=True)
    # (a classifier whose total underflows turns cells into inf/NaN like the reference; keep finite rows)
    ok = (out["h1"] != hib.NA_INTEGER) & np.isfinite(out["postprob"]).all(axis=1)
    assert ok.mean() > 0.99
    # a posterior is a distribution; dosages of a diploid sum to 2
    np.testing.assert_allclose(out["postprob"][ok].sum(axis=1), 1.0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(out["dosage"][ok].sum(axis=1), 2.0, rtol=0, atol=1e-12)
    # the call is the first maximum of the posterior and `prob` is its value
    p = out["postprob"][ok]
    assert np.array_equal(p.max(axis=1), out["prob"][ok])
    h1 = np.repeat(np.arange(model.n_hla), np.arange(model.n_hla, 0, -1))
    h2 = np.concatenate([np.arange(i, model.n_hla) for i in range(model.n_hla)])
    am = p.argmax(axis=1)
    assert np.array_equal(h1[am], out["h1"][ok]) and np.array_equal(h2[am], out["h2"][ok])
    # samples are independent: any permutation / any batch split gives the same bits per sample
    perm = np.random.default_rng(3).permutation(N)
    out_p = dev.predict_raw(G[perm], 1, want_dosage=True, want_prob=False)
    for k in ("h1", "h2", "prob", "matching", "dosage"):
        assert np.array_equal(out_p[k], out[k][perm], equal_nan=True), k
    part = dev.predict_raw(G[1234:1301], 1, want_dosage=False, want_prob=True)
    assert np.array_equal(part["postprob"], out["postprob"][1234:1301], equal_nan=True)
    # samples drawn from the model are called correctly almost always
    assert np.mean((out["h1"] == truth[:, 0]) & (out["h2"] == truth[:, 1])) > 0.97
    # and a random subset agrees with the oracle to the last bit
    sub = np.sort(np.random.default_rng(5).choice(N, 160, replace=False))
    want = oracle.predict(oracle.flatten(model), G[sub], avx2=True, n_threads=8)
    assert_same({k: v[sub] for k, v in out.items()}, want)


def test_both_engines_give_the_same_bits(hib, oracle, monkeypatch):
    """The matrix-core engine (int8 MFMA distances) and the vector-ALU engine (bit logic +
    popcount) compute the same integers; everything downstream is shared."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b", n_classifier=30)
    G, _ = synth.make_samples(founders, af, 700)
    G[9, :] = hib.NA_INTEGER
    monkeypatch.setenv("HIBAG_ENGINE", "mfma")
    a = hib.hlaModelFromObj(model).predict_raw(G, 1, want_dosage=True, want_prob=True)
    monkeypatch.setenv("HIBAG_ENGINE", "valu")
    b = hib.hlaModelFromObj(model).predict_raw(G, 1, want_dosage=True, want_prob=True)
    assert_same(a, b)
    assert_same({k: v[:96] for k, v in a.items()}, oracle.predict(oracle.flatten(model), G[:96], avx2=True, n_threads=8))


@pytest.mark.parametrize("mode", ["stream", "hybrid", "recompute"])
def test_all_forms_of_pass_two(hib, oracle, monkeypatch, mode):
    """Pass 2 reads back the cell sums pass 1 stored (all of them, or those of the cells with many pairs) and evaluates
    the haplotype pairs of the other cells again; the model picks by pairs per cell (hibag_hip_model_stored_cells,
    hibag_hip_model_second_pass_pairs).  Every form forced on models of every engine and width, on
    underflowing totals (NaN propagation through empty cells) and on unused classifiers: the same bits as the oracle."""
    from hibag_amd import synth
    from hibag_amd.model import Classifier, HlaAttrBagObj
    monkeypatch.setenv("HIBAG_PASS2", mode)
    if mode == "hybrid":
        monkeypatch.setenv("HIBAG_STORE_PAIRS", "3")      # cells with more than 3 haplotype pairs are stored
    ks = list(range(1, 41)) + [63, 64, 65, 66, 96, 127, 128]
    model, founders, af = synth.make_model("hla-a-small", seed=77, n_classifier=len(ks), n_snp=160,
                                           snp_counts=ks, wide_classifier=False)
    G, _ = synth.make_samples(founders, af, 200, seed=78, miss=0.05)
    G[3, :] = hib.NA_INTEGER
    G[64:128, :] = hib.NA_INTEGER                        # a whole wavefront that uses no classifier
    m = hib.hlaModelFromObj(model)
    # ("recompute": nothing pass 2 can evaluate again is stored -- the cells of the classifiers it cannot evaluate,
    # everything but one-step FP4, still are)
    assert m.stored_cells() > 0
    assert (m.second_pass_pairs() == 0) == (mode == "stream")
    assert_same(m.predict_raw(G, 1, want_dosage=True, want_prob=True),
                oracle.predict(oracle.flatten(model), G, avx2=True, n_threads=8))
    # the benchmark shape incl. its 100-SNP classifier, more work items than resident workgroups (chunked items)
    model, founders, af = synth.make_model("hla-b")
    G, _ = synth.make_samples(founders, af, 3400, seed=5)
    G[7, ::3] = hib.NA_INTEGER
    assert_same(hib.hlaModelFromObj(model).predict_raw(G, 1, want_dosage=True, want_prob=True),
                oracle.predict(oracle.flatten(model), G, avx2=True, n_threads=8))
    # total == 0 -> 1/total = inf -> NaN through the empty cells as well
    k = 100
    far = Classifier(np.arange(k), [0.5, 0.5], [0, 1], ["1" * k, "1" * k])
    near = Classifier(np.arange(4), [0.3, 0.3, 0.4], [0, 1, 2], ["0000", "0101", "1111"])
    model = HlaAttrBagObj(0, k, ["a", "b", "c"], [near, far])
    G = np.zeros((3, k), np.int32)
    G[1, 40:] = hib.NA_INTEGER
    G[2, :] = hib.NA_INTEGER
    assert_same(hib.hlaModelFromObj(model).predict_raw(G, 1, want_dosage=True, want_prob=True), oracle.predict(oracle.flatten(model), G))


def test_more_samples_than_one_batch(hib, oracle):
    """The driver cuts the cohort into batches (<= 131,072 samples); results must not depend on
    where the cuts fall."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-a-small", n_classifier=8)
    N = 140_000
    G, _ = synth.make_samples(founders, af, N)
    dev = hib.hlaModelFromObj(model)
    out = dev.predict_raw(G, 1, want_dosage=True, want_prob=False)
    lo, hi = 131_072 - 100, 131_072 + 100            # straddles the batch boundary
    part = dev.predict_raw(G[lo:hi], 1, want_dosage=True, want_prob=False)
    for k in part:
        assert np.array_equal(part[k], out[k][lo:hi], equal_nan=True), k
    sub = np.sort(np.random.default_rng(11).choice(N, 200, replace=False))
    want = oracle.predict(oracle.flatten(model), G[sub], want_prob=False, avx2=True, n_threads=8)
    for k in want:
        assert np.array_equal(out[k][sub], want[k], equal_nan=True), k


@pytest.mark.parametrize("seed", range(24))
def test_random_models(hib, oracle, seed):
    """Randomised structure: 1-12 alleles (some without haplotypes), 1-6 classifiers of 1-40 SNPs and 1-60
    haplotypes (duplicates allowed), frequencies over 300 orders of magnitude, genotypes uniform over
    {0, 1, 2, NA, -1, 3} -- mostly underflowing posteriors, NA calls and NaN propagation."""
    rng = np.random.default_rng(1000 + seed)
    n_hla = int(rng.integers(1, 13))
    n_snp = int(rng.integers(1, 60))
    classifiers = []
    for _ in range(int(rng.integers(1, 7))):
        k = int(rng.integers(1, min(n_snp, 40) + 1))
        H = int(rng.integers(1, 61))
        hla = np.sort(rng.integers(0, n_hla, H)).astype(np.int32)
        freq = 10.0 ** rng.uniform(-300 if seed % 3 == 0 else -6, 0, H)
        haplo = ["".join(rng.choice(["0", "1"], k)) for _ in range(H)]
        classifiers.append(hib.Classifier(snpidx=rng.choice(n_snp, k, replace=False), freq=freq, hla=hla, haplo=haplo))
    model = hib.HlaAttrBagObj(n_samp=0, n_snp=n_snp, hla_allele=[f"{i:02d}" for i in range(n_hla)], classifiers=classifiers)
    n = int(rng.integers(1, 131))
    if seed % 2:
        G = rng.choice(np.array([0, 1, 2, hib.NA_INTEGER, -1, 3], np.int64), size=(n, n_snp), p=[.3, .3, .3, .04, .03, .03]).astype(np.int32)
    else:                                   # genotypes compatible with the haplotypes of the first classifier
        c0 = classifiers[0]
        G = np.full((n, n_snp), hib.NA_INTEGER, np.int32)
        for i in range(n):
            a, b = rng.integers(0, len(c0.haplo), 2)
            G[i, c0.snpidx] = np.array([int(x) + int(y) for x, y in zip(c0.haplo[a], c0.haplo[b])])
        G[rng.random(G.shape) < 0.02] = hib.NA_INTEGER
    m = hib.hlaModelFromObj(model)
    for vote in (1, 2):
        want = oracle.predict(oracle.flatten(model), G, vote_method=vote)
        got = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
        assert_same(got, want)



@pytest.mark.parametrize("k_empty", [5, 30, 40, 120])
def test_classifier_without_haplotypes(hib, oracle, k_empty):
    """A classifier whose haplotype list is empty (every engine: FP4, int8, FP4 in several K steps -- which has a kernel
    of its own and no list segment to launch it on --, VALU): its total is 0, 1/total infinite, and the reference's
    0 * inf = NaN reaches every cell of every sample that uses it (src/LibHLA.cpp:1826-1828)."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-a-small", seed=5, n_classifier=3, n_snp=130, wide_classifier=False)
    empty = hib.Classifier(snpidx=np.arange(k_empty), freq=np.zeros(0), hla=np.zeros(0, np.int32), haplo=[])
    model.classifiers.insert(1, empty)
    G, _ = synth.make_samples(founders, af, 100, seed=6)
    G[7, :] = hib.NA_INTEGER
    G[9, :k_empty] = hib.NA_INTEGER                   # a sample that does not use the empty classifier at all
    for vote in (1, 2):
        want = oracle.predict(oracle.flatten(model), G, vote_method=vote)
        got = hib.hlaModelFromObj(model).predict_raw(G, vote, want_dosage=True, want_prob=True)
        assert_same(got, want)


def test_plugin_per_sample_route_every_width_and_edge(hib, oracle):
    """The per-sample kernels behind predict_avg_prob (hibag_sample.hip: thread = allele-pair cell) through
    hibag_amd.plugin.PluginHost, which plays the host's part (src/LibHLA.cpp:2418-2441): classifiers of 1 .. 128 SNPs (the
    32-bit and the two-word distance), a classifier without haplotypes, samples with missing SNPs, an all-missing sample,
    an underflowing total -- every posterior and matching value bit-equal to the oracle."""
    from hibag_amd import synth
    from hibag_amd.plugin import PluginHost
    ks = [1, 5, 24, 31, 32, 33, 40, 63, 64, 65, 100, 127, 128]
    model, founders, af = synth.make_model("hla-a-small", seed=41, n_classifier=len(ks), n_snp=160, snp_counts=ks, wide_classifier=False)
    model.classifiers.insert(2, hib.Classifier(snpidx=np.arange(7), freq=np.zeros(0), hla=np.zeros(0, np.int32), haplo=[]))
    G, _ = synth.make_samples(founders, af, 40, seed=42, miss=0.05)
    G[3, :] = hib.NA_INTEGER
    G[4, :7] = hib.NA_INTEGER                      # does not use the empty classifier
    want = oracle.predict(oracle.flatten(model), G, vote_method=1)
    host = PluginHost(model)
    geno, wt = host.pack(G)
    prob = np.zeros(model.n_cell); match = np.zeros(1)
    for i in range(len(G)):
        host.avg_prob(geno[i], wt[i], prob, match)
        assert np.array_equal(prob, want["postprob"][i], equal_nan=True), i
        assert match[0] == want["matching"][i] or (np.isnan(match[0]) and np.isnan(want["matching"][i])), i
    host.close()
    # total == 0 -> 1/total = inf -> NaN through the empty cells as well
    k = 100
    far = hib.Classifier(np.arange(k), [0.5, 0.5], [0, 1], ["1" * k, "1" * k])
    near = hib.Classifier(np.arange(4), [0.3, 0.3, 0.4], [0, 1, 2], ["0000", "0101", "1111"])
    model = hib.HlaAttrBagObj(0, k, ["a", "b", "c"], [near, far])
    G = np.zeros((3, k), np.int32)
    G[1, 40:] = hib.NA_INTEGER
    G[2, :] = hib.NA_INTEGER
    want = oracle.predict(oracle.flatten(model), G)
    host = PluginHost(model)
    geno, wt = host.pack(G)
    prob = np.zeros(model.n_cell); match = np.zeros(1)
    for i in range(len(G)):
        host.avg_prob(geno[i], wt[i], prob, match)
        assert np.array_equal(prob, want["postprob"][i], equal_nan=True), i
        assert match[0] == want["matching"][i] or (np.isnan(match[0]) and np.isnan(want["matching"][i])), i
    host.close()


def test_plugin_per_sample_route_beyond_one_round_of_workgroups_and_pairs(hib, oracle):
    """k_one's corners (hibag_sample.hip): more classifiers than the device holds workgroups of it at once (every workgroup then
    walks several, 300 of them), many alleles with few classifiers (more posterior cells per workgroup than one tile of the
    ensemble phase takes), and classifiers whose pair lists need several rounds (150 haplotypes: 11,325 pairs > 8,192) -- the
    compiled host loop (hibag_hip_test_time_avg_prob) and the per-call entry against the oracle, bit for bit."""
    from hibag_amd import synth
    from hibag_amd.plugin import PluginHost
    cases = [dict(shape="hla-a-small", seed=51, n_classifier=300, n_snp=60),
             dict(shape="hla-b", seed=52, n_classifier=3, n_snp=90, n_haplo=150, wide_classifier=False),
             dict(shape="hla-drb1", seed=53, n_classifier=2, n_snp=64, n_haplo=260, wide_classifier=False)]
    for kw in cases:
        shape = kw.pop("shape")
        model, founders, af = synth.make_model(shape, **kw)
        G, _ = synth.make_samples(founders, af, 12, seed=kw["seed"] + 1, miss=0.04)
        G[5, :] = hib.NA_INTEGER
        want = oracle.predict(oracle.flatten(model), G, vote_method=1, avx2=True, n_threads=8)
        host = PluginHost(model)
        geno, wt = host.pack(G)
        prob = np.zeros(model.n_cell); match = np.zeros(1)
        for i in range(len(G)):
            host.avg_prob(geno[i], wt[i], prob, match)
            assert np.array_equal(prob, want["postprob"][i], equal_nan=True), (shape, i)
            assert match[0] == want["matching"][i] or (np.isnan(match[0]) and np.isnan(want["matching"][i])), (shape, i)
        best, mt, _ = host.avg_prob_loop(geno, wt)
        nh = model.n_hla
        cell = lambda a, b: b + a * (2 * nh - a - 1) // 2
        for i in range(len(G)):
            exp = -1 if want["h1"][i] == hib.NA_INTEGER else cell(int(want["h1"][i]), int(want["h2"][i]))
            assert best[i] == exp, (shape, i)
            assert mt[i] == want["matching"][i] or (np.isnan(mt[i]) and np.isnan(want["matching"][i])), (shape, i)
        host.close()


def test_tied_cells_of_mirrored_alleles_first_one_wins(hib, oracle):
    """Two alleles with the SAME haplotypes and frequencies make bit-equal cell sums (both are summed in the same order), so
    the call -- a first strict maximum, src/LibHLA.cpp:1549-1566 -- must go to the earlier cell, for the averaged posterior
    and for the majority vote, whose choice per classifier comes from the records pass 1 logs (k_vote_pick) and is settled on
    the products cell * (1/total) themselves.  Samples are built as sums of one haplotype of the mirrored pair and one of a
    third allele, so that the tie is at the maximum: cells (0, 2) and (1, 2)."""
    rng = np.random.default_rng(77)
    n_snp, k = 40, 18
    cls = []
    pats_all = []
    for j in range(6):
        idx = np.sort(rng.choice(n_snp, k, replace=False))
        a = ["".join(rng.choice(["0", "1"], k)) for _ in range(3)]
        cpat = ["".join(rng.choice(["0", "1"], k)) for _ in range(2)]
        fa = rng.uniform(0.05, 0.3, 3)
        fc = rng.uniform(0.05, 0.3, 2)
        haplo = a + a + cpat                                           # allele 0, its mirror allele 1, allele 2
        freq = np.concatenate([fa, fa, fc])
        hla = np.array([0] * 3 + [1] * 3 + [2] * 2, np.int32)
        cls.append(hib.Classifier(snpidx=idx, freq=freq, hla=hla, haplo=haplo))
        pats_all.append((idx, a, cpat))
    model = hib.HlaAttrBagObj(n_samp=0, n_snp=n_snp, hla_allele=["a", "a'", "c"], classifiers=cls)
    rows = []
    for t in range(96):
        idx, a, cpat = pats_all[t % 6]
        g = rng.choice(np.array([0, 1, 2], np.int32), n_snp)
        h1, h2 = a[t % 3], cpat[t % 2]
        g[idx] = np.array([int(x) + int(y) for x, y in zip(h1, h2)], np.int32)
        if t % 7 == 0:
            g[rng.choice(n_snp, 3, replace=False)] = hib.NA_INTEGER
        rows.append(g)
    G = np.stack(rows).astype(np.int32)
    m = hib.hlaModelFromObj(model)
    for vote in (1, 2):
        want = oracle.predict(oracle.flatten(model), G, vote_method=vote)
        got = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
        assert_same(got, want)
        called = want["h1"] != hib.NA_INTEGER
        assert called.any() and not np.any((want["h1"][called] == 1) & (want["h2"][called] == 2))    # never the later twin
    m.close()


def test_plugin_falls_back_to_one_workgroup_when_the_barrier_times_out(hib, oracle, monkeypatch):
    """predict_avg_prob is one kernel whose workgroups meet at a barrier, i.e. must all be resident at once -- which nobody can
    promise on a shared device.  With the barrier's poll budget at zero (HIBAG_ONE_SPIN=0, read at predict_init) the first
    workgroup to arrive gives up at once: the call must then be repeated on a single workgroup and still return the
    oracle's numbers, and so must the calls after it (`hibag_hip_plugin_degraded_calls` counts them); a fresh predict_init
    without the variable runs at full width again."""
    from hibag_amd import synth, _lib
    from hibag_amd.plugin import PluginHost
    model, founders, af = synth.make_model("hla-a-small", seed=61, n_classifier=20, n_snp=80)
    G, _ = synth.make_samples(founders, af, 10, seed=62, miss=0.05)
    G[3, :] = hib.NA_INTEGER
    want = oracle.predict(oracle.flatten(model), G, vote_method=1)
    L = _lib.lib()

    def run_all(host):
        geno, wt = host.pack(G)
        prob = np.zeros(model.n_cell); match = np.zeros(1)
        for i in range(len(G)):
            host.avg_prob(geno[i], wt[i], prob, match)
            assert np.array_equal(prob, want["postprob"][i], equal_nan=True), i
            assert match[0] == want["matching"][i] or (np.isnan(match[0]) and np.isnan(want["matching"][i])), i

    monkeypatch.setenv("HIBAG_ONE_SPIN", "0")
    host = PluginHost(model)
    run_all(host)
    assert L.hibag_hip_plugin_degraded_calls() == len(G)          # every call took the one-workgroup route
    host.close()
    monkeypatch.delenv("HIBAG_ONE_SPIN")
    host = PluginHost(model)
    run_all(host)
    assert L.hibag_hip_plugin_degraded_calls() == 0
    host.close()


@pytest.mark.parametrize("k", [1, 2, 14, 15, 29, 30, 31, 32])
def test_extreme_genotypes_and_zero_frequencies(hib, oracle, k):
    """The corners of the matrix engine's K layout (hibag_device.h): every SNP heterozygous with both haplotypes carrying
    the allele (the largest w = 4 terms against the -4 terms of the lower K half), every SNP homozygous for the other
    allele (offset 2k: all four offset digits in use at k = 30), all missing; and haplotypes of frequency exactly zero,
    whose pairs have the factor 0 the host writes into the per-slot factor array (the block headers' count of slots
    worth evaluating ends at the last slot with a non-zero factor or a closing cell)."""
    rng = np.random.default_rng(500 + k)
    pats = ["1" * k, "0" * k, ("10" * k)[:k], ("01" * k)[:k]] + ["".join(rng.choice(["0", "1"], k)) for _ in range(9)]
    H = len(pats)
    hla = np.sort(rng.integers(0, 4, H)).astype(np.int32)
    freq = rng.uniform(0.01, 1.0, H)
    freq[[2, 5, H - 1]] = 0.0                                              # (incl. the last haplotype: trailing zero factors)
    n_snp = k + 3
    cl = [hib.Classifier(snpidx=np.arange(k) + j, freq=np.roll(freq, j), hla=hla, haplo=pats) for j in range(3)]
    model = hib.HlaAttrBagObj(n_samp=0, n_snp=n_snp, hla_allele=["a", "b", "c", "d"], classifiers=cl)
    rows = [np.full(n_snp, v, np.int32) for v in (0, 1, 2, hib.NA_INTEGER)]
    rows += [rng.choice(np.array([0, 1, 2], np.int32), n_snp) for _ in range(60)]
    G = np.stack(rows).astype(np.int32)
    m = hib.hlaModelFromObj(model)
    for vote in (1, 2):
        want = oracle.predict(oracle.flatten(model), G, vote_method=vote)
        got = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
        assert_same(got, want)
