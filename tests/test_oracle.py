"""CPU-only checks of the oracle itself (scalar restatement vs its AVX2/threads
variant, hand-checkable known answers for the packed types)."""

import numpy as np
import pytest


def test_mutation_table_formula(oracle):
    t = oracle.mutation_table()
    assert t[0] == 1.0 and t[1] == 1e-5 * (1 + 0) or abs(t[1] / 1e-5 - 1) < 1e-15
    assert np.all(t[65:] == 0.0) and t[64] > 0 and t[64] < 2.3e-308      # denormal tail, src/LibHLA.cpp:176-183
    ref = np.exp(np.arange(257) * np.log(1e-5))
    assert np.allclose(t[:65], ref[:65], rtol=4e-16, atol=0)


def test_hamming_truth_table(oracle):
    """Per SNP: |g - h1 - h2| for typed genotypes, 0 for missing (src/LibHLA.cpp:747-819)."""
    for g in (0, 1, 2, -1, 3, -2147483648):
        for h1 in (0, 1):
            for h2 in (0, 1):
                for pos in (0, 31, 32, 63, 64, 100, 127):
                    n = pos + 1
                    geno = np.full(n, -1, np.int32)
                    geno[pos] = g
                    s1, s2 = oracle.int_to_snp(geno, np.arange(n, dtype=np.int32))
                    a = oracle.haplo_bits("0" * pos + str(h1))
                    b = oracle.haplo_bits("0" * pos + str(h2))
                    want = abs(g - h1 - h2) if 0 <= g <= 2 else 0
                    assert oracle.hamm_d(n, s1, s2, a, b) == want


@pytest.mark.parametrize("k", [1, 7, 8, 9, 31, 32, 33, 63, 64, 65, 127, 128])
def test_int_to_snp_planes(oracle, k):
    rng = np.random.default_rng(k)
    base = rng.choice(np.array([0, 1, 2, -1, 3, -2147483648], np.int32), size=300)
    idx = rng.choice(300, size=k, replace=False).astype(np.int32)
    s1, s2 = oracle.int_to_snp(base, idx)
    for i in range(128):
        b1 = (int(s1[i >> 6]) >> (i & 63)) & 1
        b2 = (int(s2[i >> 6]) >> (i & 63)) & 1
        g = int(base[idx[i]]) if i < k else -1
        want = {0: (0, 0), 1: (1, 0), 2: (1, 1)}.get(g, (0, 1))      # inst/include/LibHLA_ext.h:251-255
        assert (b1, b2) == want


def test_avx2_threads_equal_scalar(oracle):
    from hibag_amd import synth
    for shape, n in (("hla-a-small", 150), ("hla-b", 40)):
        model, founders, af = synth.make_model(shape)
        G, _ = synth.make_samples(founders, af, n)
        G[0, :] = -2147483648
        fm = oracle.flatten(model)
        for vote in (1, 2):
            a = oracle.predict(fm, G, vote_method=vote)
            b = oracle.predict(fm, G, vote_method=vote, avx2=True, n_threads=3)
            for k in a:
                assert np.array_equal(a[k], b[k], equal_nan=True), (shape, vote, k)


def test_training_side_kernels_agree_with_postprob2(oracle, hapmap_geno, model_a):
    """_BestGuess / _PostProb / _PrepHaploMatch restatements vs _PostProb2 on the same cells."""
    from conftest import align_geno
    fm = oracle.flatten(model_a)
    G = align_geno(model_a, hapmap_geno)
    nh = fm.n_hla
    for c in (0, 17, 99):
        cl = model_a.classifiers[c]
        for s in (0, 5, 33):
            s1, s2 = oracle.int_to_snp(G[s], cl.snpidx)
            prob, total = oracle.post_prob2(fm, c, s1, s2)
            cells = prob * total
            best = oracle.best_guess(fm, c, s1, s2)
            k = int(np.argmax(cells))
            h1 = np.repeat(np.arange(nh), np.arange(nh, 0, -1))[k]
            h2 = np.concatenate([np.arange(i, nh) for i in range(nh)])[k]
            assert best == (h1, h2)
            assert abs(oracle.post_prob(fm, c, s1, s2, h2, h1) - prob[k]) <= 2e-16 * prob[k] + 1e-300
            pairs = oracle.prep_haplo_match(fm, c, s1, s2, h1, h2)
            assert len(pairs) >= 1
            _, lens, bits, _, _ = fm.classifier(c)
            d = [oracle.hamm_d(len(cl.snpidx), s1, s2, bits[i], bits[j]) for i, j in pairs]
            assert len(set(d)) == 1


def test_invalid_vote_method(oracle, model_a):
    fm = oracle.flatten(model_a)
    with pytest.raises(ValueError, match="Invalid 'vote_method'."):
        oracle.predict(fm, np.zeros((1, model_a.n_snp), np.int32), vote_method=0)
