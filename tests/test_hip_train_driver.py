"""GPU parity of the training driver (hibag_hip_trainer_* / hlaAttrBagging): bootstrap,
greedy SNP selection and EM on the host, haplotype-pair scoring on the device.

Known answer: the reference's own fixture inst/extdata/OutOfBag.RData is what
    set.seed(100); hlaAttrBagging(hlatab$training, train.geno, nclassifier=100)
gave in R (vignettes/HIBAG.Rmd:218-220); re-running that call here must return every
stored classifier and the stored `matching` vector bit for bit.  Further cases compare the
device-scored driver with the CPU oracle's restatement on data with missing genotypes."""

import math

import numpy as np
import pytest

from test_oracle_train import assert_same_classifier, training_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hib():
    import hibag_amd
    hibag_amd.hlaSetKernelTarget("hip")
    return hibag_amd


def _as_dict(c):
    return dict(samp_num=c.samp_num, snpidx=c.snpidx, haplo=c.haplo, hla=c.hla, freq=c.freq, acc=c.outofbag_acc)


def test_hlaAttrBagging_reproduces_the_reference_model(hib, hapmap_geno, hla_type_table, model_oob, capsys):
    """The vignette's call: 34 training samples, the 275 SNPs within 500 kb of HLA-A (9 of them
    monomorphic in the training set and dropped by mono.rm), mtry = ceil(sqrt(266)), prune."""
    ti = {s: i for i, s in enumerate(hla_type_table["sample.id"])}
    ids = list(model_oob.sample_id)
    hla = hib.hlaAllele(ids, [hla_type_table["A.1"][ti[s]] for s in ids], [hla_type_table["A.2"][ti[s]] for s in ids],
                        locus="A", assembly="hg19")
    start, end = hib.hlaLociInfo("hg19")["A"][1:]
    pos = np.asarray(hapmap_geno.snp_position)
    flank = np.where((pos >= start - 500000) & (pos <= end + 500000))[0]          # hlaFlankingSNP, R/DataUtilities.R:1732-1780
    assert len(flank) == 275                                                      # man/hlaAttrBagging.Rd:108
    cols = [hapmap_geno.sample_id.index(s) for s in ids]
    train_geno = hib.HlaSNPGeno(genotype=hapmap_geno.genotype[np.ix_(flank, cols)], sample_id=ids,
                                snp_id=[hapmap_geno.snp_id[i] for i in flank], snp_position=pos[flank],
                                snp_allele=[hapmap_geno.snp_allele[i] for i in flank], assembly="hg19")
    hib.set_seed(100)
    model = hib.hlaAttrBagging(hla, train_geno, nclassifier=100, verbose=True)
    text = capsys.readouterr().out
    assert "excluding 9 monomorphic SNPs" in text and "# of SNPs: 266" in text
    obj = model.obj
    assert obj.snp_id == list(model_oob.snp_id) and obj.hla_allele == list(model_oob.hla_allele)
    assert obj.sample_id == ids and obj.n_snp == 266 and obj.n_samp == 34
    assert np.array_equal(obj.snp_allele_freq, model_oob.snp_allele_freq)
    assert np.allclose(obj.hla_freq, model_oob.hla_freq, rtol=0, atol=1e-15)
    assert len(obj.classifiers) == 100
    for i, (got, want) in enumerate(zip(obj.classifiers, model_oob.classifiers)):
        assert_same_classifier(_as_dict(got), want, i)
    # hlaPredict() on the new model: `matching` of the samples without missing SNPs (the stored values
    # of the other 7 come from an older release's ensemble formula, tests/test_oracle_pin.py)
    complete = np.array([np.all((g >= 0) & (g <= 2)) for g in train_geno.genotype[[train_geno.snp_id.index(s) for s in obj.snp_id]].T])
    assert complete.sum() == 27 and np.array_equal(obj.matching[complete], model_oob.matching[complete])


@pytest.mark.parametrize("prune", [True, False])
def test_driver_equals_oracle_with_missing_genotypes(hib, oracle, prune):
    """Synthetic cohort drawn from a model (with missing genotypes), both random-stream routes
    (library's set_seed, host callback) against the oracle's CPU restatement."""
    from hibag_amd import synth, train
    model, founders, af = synth.make_model("hla-a-small", seed=5, n_snp=60)
    G, truth = synth.make_samples(founders, af, 150, seed=6, miss=0.03)
    n_hla = model.n_hla
    want = oracle.train(G, truth[:, 0], truth[:, 1], n_hla, nclassifier=4, mtry=8, prune=prune, seed=42)
    for route in ("seed", "callback"):
        tr = train._Trainer(G, truth[:, 0], truth[:, 1], n_hla)
        if route == "seed":
            tr.set_seed(42)
        else:
            tr.set_rng(hib.RRandom(42))
        tr.new_classifiers(2, 8, prune, False, False)
        tr.new_classifiers(2, 8, prune, False, False)        # the stream continues across calls
        got = tr.classifiers()
        tr.close()
        assert len(got) == 4
        for i, (g, w) in enumerate(zip(got, want)):
            c = hib.Classifier(snpidx=w["snpidx"], freq=w["freq"], hla=w["hla"], haplo=w["haplo"], samp_num=w["samp_num"],
                               outofbag_acc=w["acc"])
            assert_same_classifier(_as_dict(g), c, i)
            assert len(g.snpidx) > 0


@pytest.mark.timeout(300)
def test_fused_scoring_variant_equals_the_oracle(hib, oracle, monkeypatch):
    """``HIBAG_BATCH_FUSED=1``: a growth step's scoring as ONE kernel (``k_batch_score``: producer wavefronts hand the cell sums
    to a scanning wavefront through a ring in LDS) instead of two -- the same classifiers, alone and with shared trainers."""
    from hibag_amd import synth, train
    monkeypatch.setenv("HIBAG_BATCH_FUSED", "1")
    model, founders, af = synth.make_model("hla-a-small", seed=15, n_snp=70)
    G, truth = synth.make_samples(founders, af, 200, seed=16, miss=0.02)
    want = oracle.train(G, truth[:, 0], truth[:, 1], model.n_hla, nclassifier=3, mtry=9, prune=True, seed=77)
    tr = train._Trainer(G, truth[:, 0], truth[:, 1], model.n_hla)
    tr.set_seed(77)
    tr.new_classifiers(3, 9, True, False, False)
    got = tr.classifiers()
    tr.close()
    for i, (g, w) in enumerate(zip(got, want)):
        c = hib.Classifier(snpidx=w["snpidx"], freq=w["freq"], hla=w["hla"], haplo=w["haplo"], samp_num=w["samp_num"], outofbag_acc=w["acc"])
        assert_same_classifier(_as_dict(g), c, i)
    # four shared trainers (fused launches carrying several growth steps), stream r seeded with 300 + r
    both = train.grow_concurrently(G, truth[:, 0], truth[:, 1], model.n_hla, 8, 9, True, 4, 1, 300, em="device", combine=True)
    for r in range(4):
        w = oracle.train(G, truth[:, 0], truth[:, 1], model.n_hla, nclassifier=2, mtry=9, prune=True, seed=300 + r)
        for j in range(2):
            g = both[2 * r + j]
            assert np.array_equal(g.snpidx, w[j]["snpidx"]) and np.array_equal(g.freq, w[j]["freq"]) and g.haplo == w[j]["haplo"], (r, j)


def test_trainer_argument_errors(hib):
    import ctypes as C
    from hibag_amd import _lib
    L = _lib.lib()
    g = np.zeros((4, 3), np.int32)
    h = np.zeros(4, np.int32)
    def new(n_snp, n_samp, n_hla, h1=h):
        return L.hibag_hip_trainer_new(n_snp, n_samp, g.ctypes.data, n_hla, h1.ctypes.data, h.ctypes.data)
    assert not new(3, 0, 2) and L.hibag_hip_last_error() == b"Invalid number of samples: 0."          # src/HIBAG.cpp:520-521
    assert not new(0, 4, 2) and L.hibag_hip_last_error() == b"Invalid number of SNPs: 0."
    assert not new(3, 4, 0) and L.hibag_hip_last_error() == b"Invalid number of unique HLA alleles: 0."
    bad = np.array([0, 5, 0, 0], np.int32)
    assert not new(3, 4, 2, bad) and L.hibag_hip_last_error() == b"CAttrBag_Model::InitTraining, H1 error."
    t = C.c_void_p(new(3, 4, 2))
    assert t and L.hibag_hip_trainer_n_classifier(t) == 0
    assert L.hibag_hip_trainer_new_classifiers(t, 1, 0, 1, 0, 0) == -1
    L.hibag_hip_trainer_free(t)


def test_parallel_entry_single_rank(hib, oracle):
    """hlaParallelAttrBagging without a process group = one rank: stream seed + 0."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-a-small", seed=5, n_snp=40)
    G, truth = synth.make_samples(founders, af, 80, seed=6)
    ids = [f"S{i}" for i in range(80)]
    hla = hib.hlaAllele(ids, [model.hla_allele[a] for a in truth[:, 0]], [model.hla_allele[a] for a in truth[:, 1]], locus="A",
                        assembly="hg19")
    snp = hib.HlaSNPGeno(genotype=np.ascontiguousarray(G.T), sample_id=ids, snp_id=list(model.snp_id),
                         snp_position=model.snp_position, snp_allele=list(model.snp_allele), assembly="hg19")
    mod = hib.hlaParallelAttrBagging(None, hla, snp, nclassifier=3, mtry=6, mono_rm=False, verbose=False, seed=300)
    used = sorted(set(truth.ravel().tolist()))
    remap = {a: i for i, a in enumerate(used)}            # the model only knows the alleles that occur
    assert mod.obj.hla_allele == hib.hlaUniqueAllele([model.hla_allele[a] for a in used])
    order = {a: i for i, a in enumerate(mod.obj.hla_allele)}
    h1 = np.array([order[model.hla_allele[a]] for a in truth[:, 0]], np.int32)
    h2 = np.array([order[model.hla_allele[a]] for a in truth[:, 1]], np.int32)
    want = oracle.train(G, h1, h2, len(used), 3, 6, True, 300)
    assert len(mod.obj.classifiers) == 3 and mod.obj.matching is not None and len(mod.obj.matching) == 80
    for i, (g, w) in enumerate(zip(mod.obj.classifiers, want)):
        assert np.array_equal(g.snpidx, w["snpidx"]) and np.array_equal(g.freq, w["freq"]) and g.haplo == w["haplo"], i
    del remap


@pytest.mark.parametrize("combine,budget", [(True, 0), (True, 2), (False, 0)])
@pytest.mark.parametrize("em", ["host", "device"])
def test_concurrent_trainers_equal_a_serial_trainer_per_stream(hib, oracle, em, combine, budget):
    """Several trainers of one process side by side on one device (train.grow_concurrently: one host thread, one default
    stream and one training state each): trainer r draws from R's Mersenne-Twister seeded with seed + r, like the workers of
    hlaParallelAttrBagging (R/HIBAG.R:329-390), so its classifiers must equal the oracle's serial run from that seed --
    field by field, with the EM fits on the trainers' host threads and on the device; with the trainers' device work fused
    into one launch per kind of operation (csrc/hibag_combine.h), also under a budget of two runnable host threads, and with a
    stream per trainer as in round 5."""
    from hibag_amd import synth, train
    from hibag_amd.dist import shard_bounds
    model, founders, af = synth.make_model("hla-a-small", seed=15, n_snp=70)
    G, truth = synth.make_samples(founders, af, 180, seed=16, miss=0.02)
    n_hla, ncl, k, mtry = model.n_hla, 10, 4, 9
    got = train.grow_concurrently(G, truth[:, 0], truth[:, 1], n_hla, ncl, mtry, True, n_trainers=k, threads_per_trainer=2,
                                  seed=300, em=em, combine=combine, thread_budget=budget)
    assert len(got) == ncl
    at = 0
    for r in range(k):
        lo, hi = shard_bounds(ncl, k, r)
        want = oracle.train(G, truth[:, 0], truth[:, 1], n_hla, nclassifier=hi - lo, mtry=mtry, prune=True, seed=300 + r)
        for w in want:
            c = hib.Classifier(snpidx=w["snpidx"], freq=w["freq"], hla=w["hla"], haplo=w["haplo"], samp_num=w["samp_num"],
                               outofbag_acc=w["acc"])
            assert_same_classifier(_as_dict(got[at]), c, at)
            at += 1
    assert at == ncl
    # the user-level entry: a model of the same classifiers, usable at once
    ids = [f"s{i}" for i in range(len(G))]
    hla = hib.hlaAllele(ids, [model.hla_allele[a] for a in truth[:, 0]], [model.hla_allele[a] for a in truth[:, 1]], locus="A",
                        assembly="hg19")
    snp = hib.HlaSNPGeno(genotype=np.ascontiguousarray(G.T), sample_id=ids, snp_id=list(model.snp_id),
                         snp_position=model.snp_position, snp_allele=list(model.snp_allele), assembly="hg19")
    m = hib.hlaConcurrentAttrBagging(hla, snp, nclassifier=6, mtry=mtry, n_trainers=3, nthread=6, seed=300, verbose=False,
                                     mono_rm=False)
    assert len(m.obj.classifiers) == 6 and m.obj.matching is not None
    hib.hlaClose(m)


def test_more_trainers_than_one_fused_launch_holds(hib, oracle):
    """Twenty trainers, one classifier each: more operations pending at a time than one fused launch takes (sixteen views
    travel as kernel arguments), so the combiner cuts a batch into several launches per kind.  Every classifier still equals
    the oracle's serial run from its stream, and the fused launches did carry several operations."""
    import ctypes as C
    from hibag_amd import _lib, synth, train
    model, founders, af = synth.make_model("hla-a-small", seed=25, n_snp=60)
    G, truth = synth.make_samples(founders, af, 150, seed=26, miss=0.02)
    k, mtry = 20, 8
    _lib.lib().hibag_hip_train_combine_stats(None, None, 1)
    got = train.grow_concurrently(G, truth[:, 0], truth[:, 1], model.n_hla, k, mtry, True, n_trainers=k, threads_per_trainer=1,
                                  seed=700, em="device", combine=True, thread_budget=3)
    assert len(got) == k
    for r in range(k):
        w = oracle.train(G, truth[:, 0], truth[:, 1], model.n_hla, nclassifier=1, mtry=mtry, prune=True, seed=700 + r)[0]
        c = hib.Classifier(snpidx=w["snpidx"], freq=w["freq"], hla=w["hla"], haplo=w["haplo"], samp_num=w["samp_num"], outofbag_acc=w["acc"])
        assert_same_classifier(_as_dict(got[r]), c, r)
    nl, no = (C.c_longlong * 8)(), (C.c_longlong * 8)()
    _lib.lib().hibag_hip_train_combine_stats(nl, no, 0)
    assert sum(no) > sum(nl) > 0, (list(nl), list(no))        # some launches carried more than one trainer's operation
    # the device-LIST form (trainer r on device[r % len]) rehearsed on the one device: tests/test_hip_multi_device.py runs it on two
    again = train.grow_concurrently(G, truth[:, 0], truth[:, 1], model.n_hla, 6, mtry, True, n_trainers=6, threads_per_trainer=1,
                                    seed=700, device=[0, 0, 0], em="device", combine=True, thread_budget=2)
    for r in range(6):
        assert_same_classifier(_as_dict(again[r]), got[r], r)


@pytest.mark.parametrize("seed", range(8))
def test_random_training_problems(hib, oracle, seed):
    """Randomised cohorts (size, alleles, missingness, mtry, prune): the device-scored driver and the
    oracle's CPU restatement must grow identical classifiers from the same stream."""
    from hibag_amd import train
    rng = np.random.default_rng(500 + seed)
    n_hla = int(rng.integers(2, 11))
    n_snp = int(rng.integers(8, 61))
    n_samp = int(rng.integers(20, 121))
    founders = (rng.random((n_hla, n_snp)) < rng.uniform(0.1, 0.9, n_snp)).astype(np.int32)
    a = rng.integers(0, n_hla, (n_samp, 2))
    G = (founders[a[:, 0]] + founders[a[:, 1]]).astype(np.int32)
    G = np.where(rng.random(G.shape) < 0.03, (G + 1) % 3, G).astype(np.int32)
    G[rng.random(G.shape) < rng.uniform(0, 0.15)] = hib.NA_INTEGER
    mtry = int(rng.integers(1, n_snp + 1))
    prune = bool(rng.integers(0, 2))
    want = oracle.train(G, a[:, 0], a[:, 1], n_hla, nclassifier=3, mtry=mtry, prune=prune, seed=7 + seed)
    tr = train._Trainer(G, a[:, 0], a[:, 1], n_hla)
    tr.set_seed(7 + seed)
    tr.new_classifiers(3, mtry, prune, False, False)
    got = tr.classifiers()
    tr.close()
    for i, (g, w) in enumerate(zip(got, want)):
        c = hib.Classifier(snpidx=w["snpidx"], freq=w["freq"], hla=w["hla"], haplo=w["haplo"], samp_num=w["samp_num"],
                           outofbag_acc=w["acc"])
        assert_same_classifier(_as_dict(g), c, i)


def test_training_runs_on_the_selected_device(hib, oracle):
    """One process per GPU: the trainer and the model it returns live on the device chosen with
    hibag_hip_set_device / device= (LOCAL_RANK under torchrun), not on device 0.  Needs two GPUs to see
    a difference; on a one-GPU box it checks the selection plumbing and the error for a missing device."""
    import torch
    from hibag_amd import _lib, synth
    L = _lib.lib()
    n_dev = L.hibag_hip_device_count()
    assert L.hibag_hip_set_device(n_dev) != 0 and b"not available" in L.hibag_hip_last_error()
    model, founders, af = synth.make_model("hla-a-small", seed=5, n_snp=40)
    G, truth = synth.make_samples(founders, af, 80, seed=6)
    ids = [f"S{i}" for i in range(80)]
    hla = hib.hlaAllele(ids, [model.hla_allele[a] for a in truth[:, 0]], [model.hla_allele[a] for a in truth[:, 1]], locus="A",
                        assembly="hg19")
    snp = hib.HlaSNPGeno(genotype=np.ascontiguousarray(G.T), sample_id=ids, snp_id=list(model.snp_id),
                         snp_position=model.snp_position, snp_allele=list(model.snp_allele), assembly="hg19")
    dev = n_dev - 1
    free_other = torch.cuda.mem_get_info(0)[0] if n_dev > 1 else None
    free_before = torch.cuda.mem_get_info(dev)[0]
    hib.set_seed(11)
    mod = hib.hlaAttrBagging(hla, snp, nclassifier=2, mtry=6, mono_rm=False, verbose=False, device=dev)
    assert len(mod.obj.classifiers) == 2
    assert mod.device() == dev                                    # the model (and the build state) sit on `dev`
    # (free memory on `dev` is not a reliable witness: the runtime serves small allocations from blocks earlier tests freed)
    assert torch.cuda.mem_get_info(dev)[0] <= free_before + (64 << 20)
    if n_dev > 1:
        assert abs(torch.cuda.mem_get_info(0)[0] - free_other) < (8 << 20)      # nothing landed on device 0
    assert L.hibag_hip_set_device(0) == 0


@pytest.mark.parametrize("em", ["device", "host"])
def test_both_em_routes_reproduce_the_stored_model(hib, hapmap_geno, hla_type_table, model_a, em):
    """inst/extdata/ModelList.RData (set.seed(100), all 60 samples, 100 classifiers) re-trained with the EM fits forced onto the
    device (hibag_amd/csrc/hibag_em.hip: every sum in the host's order, the stopping test decided with a margin for the
    device's log() or handed back) and onto the host threads: every stored classifier bit for bit either way
    (CAlg_EM::ExpectationMaximization, src/LibHLA.cpp:1185-1255)."""
    from hibag_amd import train
    G, h1, h2 = training_inputs(model_a, hapmap_geno, hla_type_table)
    mtry = int(math.ceil(math.sqrt(G.shape[1])))
    tr = train._Trainer(G, h1, h2, len(model_a.hla_allele))
    tr.set_em_mode(em)
    tr.set_seed(100)
    tr.new_classifiers(len(model_a.classifiers), mtry, True, False, False)
    got = tr.classifiers()
    tr.close()
    assert len(got) == len(model_a.classifiers)
    for i, (g, w) in enumerate(zip(got, model_a.classifiers)):
        assert_same_classifier(_as_dict(g), w, i)


@pytest.mark.parametrize("seed", range(4))
def test_device_em_on_random_problems(hib, oracle, seed):
    """The device EM on cohorts with missing genotypes, few samples, one to many alleles -- against the oracle."""
    from hibag_amd import train
    rng = np.random.default_rng(900 + seed)
    n_hla = int(rng.integers(1, 12))
    n_snp = int(rng.integers(6, 50))
    n_samp = int(rng.integers(12, 160))
    founders = (rng.random((n_hla, n_snp)) < rng.uniform(0.1, 0.9, n_snp)).astype(np.int32)
    a = rng.integers(0, n_hla, (n_samp, 2))
    G = (founders[a[:, 0]] + founders[a[:, 1]]).astype(np.int32)
    G = np.where(rng.random(G.shape) < 0.05, (G + 1) % 3, G).astype(np.int32)
    G[rng.random(G.shape) < rng.uniform(0, 0.2)] = hib.NA_INTEGER
    mtry = int(rng.integers(1, n_snp + 1))
    want = oracle.train(G, a[:, 0], a[:, 1], n_hla, nclassifier=3, mtry=mtry, prune=True, seed=70 + seed)
    tr = train._Trainer(G, a[:, 0], a[:, 1], n_hla)
    tr.set_em_mode("device")
    tr.set_threads(1)
    tr.set_seed(70 + seed)
    tr.new_classifiers(3, mtry, True, False, False)
    got = tr.classifiers()
    tr.close()
    for i, (g, w) in enumerate(zip(got, want)):
        c = hib.Classifier(snpidx=w["snpidx"], freq=w["freq"], hla=w["hla"], haplo=w["haplo"], samp_num=w["samp_num"],
                           outofbag_acc=w["acc"])
        assert_same_classifier(_as_dict(g), c, i)


def test_training_campaign(hib, oracle):
    """Time-boxed random campaign of the training driver against the oracle's trainer (HIBAG_FUZZ_SECONDS, default 15):
    cohorts of 10..400 samples, 4..120 SNPs, 1..20 alleles, genotyping errors and up to 25 % missing, mtry and prune at
    random, EM fits on the host threads (1..8) or on the device -- every classifier (SNPs, haplotypes, frequencies,
    out-of-bag accuracy) bit for bit.  A long run's summary: profiles/r04_fuzz_campaign.txt."""
    import os
    import time
    from hibag_amd import train
    budget = float(os.environ.get("HIBAG_FUZZ_SECONDS", "15"))
    seed0 = int(os.environ.get("HIBAG_FUZZ_SEED", "40000"))
    t_end = time.time() + budget
    seed, done, bad = seed0, 0, []
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        n_hla = int(rng.integers(1, 21))
        n_snp = int(rng.integers(4, 121))
        n_samp = int(rng.choice([rng.integers(10, 60), rng.integers(10, 200), rng.integers(10, 401)]))
        founders = (rng.random((n_hla, n_snp)) < rng.uniform(0.05, 0.95, n_snp)).astype(np.int32)
        a = rng.integers(0, n_hla, (n_samp, 2))
        G = (founders[a[:, 0]] + founders[a[:, 1]]).astype(np.int32)
        G = np.where(rng.random(G.shape) < rng.uniform(0, 0.08), (G + 1) % 3, G).astype(np.int32)
        G[rng.random(G.shape) < rng.uniform(0, 0.25)] = hib.NA_INTEGER
        mtry = int(rng.integers(1, n_snp + 1))
        prune = bool(rng.integers(0, 2))
        em = ["host", "device"][int(rng.integers(0, 2))]
        threads = int(rng.integers(1, 9))
        want = oracle.train(G, a[:, 0], a[:, 1], n_hla, nclassifier=2, mtry=mtry, prune=prune, seed=seed)
        tr = train._Trainer(G, a[:, 0], a[:, 1], n_hla)
        tr.set_em_mode(em)
        tr.set_threads(threads)
        tr.set_seed(seed)
        tr.new_classifiers(2, mtry, prune, False, False)
        got = tr.classifiers()
        tr.close()
        try:
            assert len(got) == len(want)
            for i, (g, w) in enumerate(zip(got, want)):
                c = hib.Classifier(snpidx=w["snpidx"], freq=w["freq"], hla=w["hla"], haplo=w["haplo"], samp_num=w["samp_num"],
                                   outofbag_acc=w["acc"])
                assert_same_classifier(_as_dict(g), c, i)
        except AssertionError as e:
            bad.append((seed, em, threads, str(e)[:120]))
        done += 1
        seed += 1
    print(f"training campaign: {done} cohorts (seeds {seed0}..{seed - 1}), {len(bad)} mismatches")
    if os.environ.get("HIBAG_FUZZ_REPORT"):
        with open(os.environ["HIBAG_FUZZ_REPORT"], "a") as f:
            f.write(f"training, seeds {seed0}..{seed - 1}: {done} cohorts x 2 classifiers, mismatches: {bad}\n")
    assert not bad, bad


def test_combined_trainers_on_different_cohorts_campaign(hib, oracle):
    """Time-boxed random campaign of the combiner (HIBAG_FUZZ_SECONDS, default 15) with what the other tests never give it:
    trainers that share the device but work on DIFFERENT problems -- cohorts of different sizes (other sample paddings, grids,
    LDS layouts of the EM kernel, 1..4 genotype words), allele counts, mtry -- so that one fused launch carries views that
    have nothing in common.  Each of 3..8 trainers is a thread with its own hibag_hip_trainer in shared mode, EM on the device,
    under a budget of 2..4 runnable threads; every classifier must equal the oracle's serial run on that trainer's cohort."""
    import os
    import threading
    import time
    from hibag_amd import _lib, train
    budget_s = float(os.environ.get("HIBAG_FUZZ_SECONDS", "15"))
    seed0 = int(os.environ.get("HIBAG_FUZZ_SEED", "90000"))
    t_end = time.time() + budget_s
    seed, done, bad = seed0, 0, []
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        k = int(rng.integers(3, 9))
        probs = []
        for r in range(k):
            n_hla = int(rng.integers(2, 16))
            n_snp = int(rng.integers(6, 100))
            n_samp = int(rng.choice([rng.integers(12, 60), rng.integers(60, 200), rng.integers(200, 500)]))
            founders = (rng.random((n_hla, n_snp)) < rng.uniform(0.1, 0.9, n_snp)).astype(np.int32)
            a = rng.integers(0, n_hla, (n_samp, 2))
            G = (founders[a[:, 0]] + founders[a[:, 1]]).astype(np.int32)
            G = np.where(rng.random(G.shape) < rng.uniform(0, 0.06), (G + 1) % 3, G).astype(np.int32)
            G[rng.random(G.shape) < rng.uniform(0, 0.2)] = hib.NA_INTEGER
            probs.append(dict(G=G, a=a, n_hla=n_hla, mtry=int(rng.integers(1, n_snp + 1)), prune=bool(rng.integers(0, 2)), seed=seed * 16 + r))
        got, errs = [None] * k, [None] * k

        def work(r):
            try:
                p = probs[r]
                tr = train._Trainer(p["G"], p["a"][:, 0], p["a"][:, 1], p["n_hla"])
                try:
                    tr.set_em_mode("device"); tr.set_threads(1); tr.set_shared(True); tr.set_seed(p["seed"])
                    tr.new_classifiers(2, p["mtry"], p["prune"], False, False)
                    got[r] = tr.classifiers()
                finally:
                    tr.close()
            except BaseException as e:              # noqa: BLE001 -- reported below
                errs[r] = e
        _lib.lib().hibag_hip_train_set_thread_budget(int(rng.integers(2, 5)))
        ths = [threading.Thread(target=work, args=(r,)) for r in range(k)]
        [t.start() for t in ths]; [t.join() for t in ths]
        _lib.lib().hibag_hip_train_set_thread_budget(0)
        for r in range(k):
            p = probs[r]
            try:
                assert errs[r] is None, repr(errs[r])
                want = oracle.train(p["G"], p["a"][:, 0], p["a"][:, 1], p["n_hla"], nclassifier=2, mtry=p["mtry"], prune=p["prune"], seed=p["seed"])
                assert len(got[r]) == len(want)
                for i, (g, w) in enumerate(zip(got[r], want)):
                    c = hib.Classifier(snpidx=w["snpidx"], freq=w["freq"], hla=w["hla"], haplo=w["haplo"], samp_num=w["samp_num"], outofbag_acc=w["acc"])
                    assert_same_classifier(_as_dict(g), c, i)
            except AssertionError as e:
                bad.append((seed, r, str(e)[:120]))
        done += k
        seed += 1
    print(f"combined-trainers campaign: {done} trainers in {seed - seed0} rounds (seeds {seed0}..{seed - 1}), {len(bad)} mismatches")
    if os.environ.get("HIBAG_FUZZ_REPORT"):
        with open(os.environ["HIBAG_FUZZ_REPORT"], "a") as f:
            f.write(f"combined trainers on different cohorts, seeds {seed0}..{seed - 1}: {done} trainers x 2 classifiers, mismatches: {bad}\n")
    assert not bad, bad
