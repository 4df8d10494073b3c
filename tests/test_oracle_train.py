"""Pin of the oracle's training driver (oracle/hibag_oracle_train.c) on the reference's
own fixture: inst/extdata/OutOfBag.RData is what the vignette's
    set.seed(100); model <- hlaAttrBagging(hlatab$training, train.geno, nclassifier=100)
produced (vignettes/HIBAG.Rmd:218-220).  Re-running that call -- R's Mersenne-Twister
stream restated from its published algorithm, bootstrap, greedy SNP selection, EM,
out-of-bag / in-bag scoring -- must give back every stored classifier bit for bit:
bootstrap counts, selected SNPs, haplotypes, frequencies (doubles compared with ==)
and out-of-bag accuracies."""

import math

import numpy as np

from conftest import align_geno


def training_inputs(model, geno, table, locus="A"):
    ti = {s: i for i, s in enumerate(table["sample.id"])}
    lut = {a: i for i, a in enumerate(model.hla_allele)}
    a1 = [lut[table[f"{locus}.1"][ti[s]]] for s in model.sample_id]
    a2 = [lut[table[f"{locus}.2"][ti[s]]] for s in model.sample_id]
    return align_geno(model, geno), np.array(a1, np.int32), np.array(a2, np.int32)


def assert_same_classifier(got, want, i):
    assert np.array_equal(got["samp_num"], want.samp_num), f"classifier {i}: bootstrap counts"
    assert np.array_equal(got["snpidx"], want.snpidx), f"classifier {i}: selected SNPs"
    assert got["haplo"] == want.haplo, f"classifier {i}: haplotypes"
    assert np.array_equal(got["hla"], want.hla), f"classifier {i}: alleles of the haplotypes"
    assert np.array_equal(got["freq"], want.freq), f"classifier {i}: frequencies"
    assert got["acc"] == want.outofbag_acc, f"classifier {i}: out-of-bag accuracy"


def test_r_random_stream(oracle):
    """set.seed(100); runif(3) in any R >= 1.7 (Mersenne-Twister, inversion)."""
    import ctypes as C
    L = oracle.lib()
    state = (C.c_uint32 * 626)()
    L.oracle_rng_set_seed(state, 100)
    L.oracle_rng_unif.restype = C.c_double
    got = [L.oracle_rng_unif(state) for _ in range(3)]
    assert [round(v, 7) for v in got] == [0.3077661, 0.2576725, 0.5523224]


def test_training_reproduces_the_stored_model(oracle, hapmap_geno, hla_type_table, model_oob):
    G, a1, a2 = training_inputs(model_oob, hapmap_geno, hla_type_table)
    assert G.shape == (34, 266)
    mtry = math.ceil(math.sqrt(model_oob.n_snp))           # mtry="sqrt", R/HIBAG.R:183-185
    out = oracle.train(G, a1, a2, model_oob.n_hla, nclassifier=100, mtry=mtry, prune=True, seed=100)
    assert len(out) == 100
    for i, (got, want) in enumerate(zip(out, model_oob.classifiers)):
        assert_same_classifier(got, want, i)


def test_training_reproduces_the_second_stored_model(oracle, hapmap_geno, hla_type_table, model_a):
    """inst/extdata/ModelList.RData, modellist$A: all 60 typed HapMap CEU samples, same seed."""
    G, a1, a2 = training_inputs(model_a, hapmap_geno, hla_type_table)
    assert G.shape == (60, 266)
    out = oracle.train(G, a1, a2, model_a.n_hla, nclassifier=100, mtry=17, prune=True, seed=100)
    for i, (got, want) in enumerate(zip(out, model_a.classifiers)):
        assert_same_classifier(got, want, i)
