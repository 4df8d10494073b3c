"""BASELINE.json's configs 3, 4 and 5 at their named shapes, on the GPU, through the C ABI.

cfg3  "Same HLA-B model, 100k synthetic samples sharded across 8 MI355X, RCCL posterior merge":
      the full 100-classifier HLA-B model, 100,000 samples through the host-pointer entry (crosses
      the library's batch cut), and the classifier-sharded entry points with EIGHT shards emulated on
      one device, their partial sums added in rank order like the all-reduce would
      (src/LibHLA.cpp:2448-2476 is the sum being split; R/HIBAG.R:764-808 the reference's cluster branch).
cfg4  "HLA-DRB1 4-digit model (~500 haplotypes/classifier)": full 100-classifier model against the oracle.
cfg5  "hlaAttrBagging() training, 1k samples x 300 SNPs": the device-scored driver against the oracle's
      CPU restatement, every field of every classifier equal (src/LibHLA.cpp:2268-2305).
"""

import numpy as np
import pytest

from test_oracle_train import assert_same_classifier

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hib():
    import hibag_amd
    hibag_amd.hlaSetKernelTarget("hip")
    return hibag_amd


def test_cfg3_100k_samples_eight_classifier_shards(hib, oracle):
    import torch
    from hibag_amd import dist as hd, synth
    model, founders, af = synth.make_model("hla-b")
    n = 100_000
    G, truth = synth.make_samples(founders, af, n)
    G[7, :] = hib.NA_INTEGER                              # one sample with every SNP missing -> NA call
    dev_model = hib.hlaModelFromObj(model)
    full = dev_model.predict_raw(G, 1, want_dosage=True, want_prob=True)     # more than one batch inside the library
    # a failed hand-over would have been repaired silently by the host entry (at twice the time): it must not have happened
    assert dev_model.handover_faults() == 0 and dev_model.status() == 0
    dev_model.close()
    ok = np.ones(n, bool); ok[7] = False
    assert np.mean((full["h1"][ok] == truth[ok, 0]) & (full["h2"][ok] == truth[ok, 1])) > 0.9
    assert full["h1"][7] == hib.NA_INTEGER and full["prob"][7] == 0.0

    # a 200-sample subset against the CPU oracle: every output bit-identical
    rng = np.random.default_rng(3)
    sub = np.sort(np.concatenate([[0, 7, n - 1], rng.choice(n, 197, replace=False)]))
    want = oracle.predict(oracle.flatten(model), G[sub], vote_method=1)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        assert np.array_equal(full[k][sub], want[k], equal_nan=True), k

    # eight classifier shards (13,13,13,13,12,12,12,12 classifiers) on the same samples; the partial entry
    # takes one batch at a time, so the cohort goes through in slices
    world = 8
    shards = [hd.hip_classifier_sharded_fns(model, 0, world, r) for r in range(world)]
    worst = 0.0
    for lo in range(0, n, 25_000):
        g = G[lo:lo + 25_000]
        merged = None
        for pf, ff, m in shards:                           # rank order, as a ring all-reduce(SUM) would add them
            part = pf(g)
            merged = part if merged is None else merged + part
        got = shards[0][1](merged)
        torch.cuda.synchronize()
        sl = slice(lo, lo + len(g))
        assert np.array_equal(got["h1"], full["h1"][sl]) and np.array_equal(got["h2"], full["h2"][sl])
        fin = np.isfinite(full["postprob"][sl]).all(axis=1)
        for k in ("prob", "matching", "dosage", "postprob"):
            a, b = got[k][fin], full[k][sl][fin]
            np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-300, err_msg=k)
        live = full["postprob"][sl][fin] > 1e-200
        worst = max(worst, float(np.max(np.abs(got["postprob"][fin] - full["postprob"][sl][fin])[live] /
                                        full["postprob"][sl][fin][live])))
    assert worst <= 1e-10
    for _, _, m in shards:
        assert m.handover_faults() == 0 and m.status() == 0
        m.close()


def test_chunked_work_items_equal_the_oracle(hib, oracle):
    """3,400 samples of the HLA-B model: more work items than resident workgroups in both passes, so the last rounds'
    items run as chunks that continue each other's sums through the output rows (hibag_kernels.hip "hand-overs").
    Every output of every sample bit-equal to the oracle, samples without genotypes and unused classifiers included."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b")
    n = 3400
    G, truth = synth.make_samples(founders, af, n, seed=77)
    probe = hib.hlaModelFromObj(model)                     # 8.5 pairs per cell: the cells with many pairs are stored, the others evaluated again
    assert 0 < probe.stored_cells() < 20_000 and 0 < probe.second_pass_pairs() < probe.pair_evals()
    probe.close()
    G[5, :] = hib.NA_INTEGER
    G[64:128, :] = hib.NA_INTEGER                         # a whole wavefront without any genotype
    G[200:264, ::2] = hib.NA_INTEGER
    m = hib.hlaModelFromObj(model)
    got = m.predict_raw(G, 1, want_dosage=True, want_prob=True)
    again = m.predict_raw(G, 1, want_dosage=True, want_prob=True)     # next batch: a new epoch of hand-over flags
    m.close()
    want = oracle.predict(oracle.flatten(model), G, vote_method=1, avx2=True, n_threads=8)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k
        assert np.array_equal(again[k], want[k], equal_nan=True), k


def test_device_entry_on_a_stream_of_the_callers(hib, oracle):
    """hibag_hip_predict_device on a non-default stream, back to back without synchronising in between, with the
    benchmark model (whose 100-SNP classifier runs pass 1 on the library's second stream, forked from and joined to
    the caller's): every call's outputs equal the oracle's."""
    import torch
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b")
    n = 2500
    dev = torch.device("cuda", 0)
    m = hib.hlaModelFromObj(model)
    st = torch.cuda.Stream(dev)
    outs, genos = [], []
    with torch.cuda.stream(st):
        for rep in range(3):
            G, _ = synth.make_samples(founders, af, n, seed=200 + rep)
            dg = torch.from_numpy(G).to(dev, non_blocking=False)
            h1 = torch.empty(n, dtype=torch.int32, device=dev); h2 = torch.empty_like(h1)
            pr = torch.empty(n, dtype=torch.float64, device=dev); mt = torch.empty_like(pr)
            ds = torch.empty((n, model.n_hla), dtype=torch.float64, device=dev)
            m.predict_device(dg.data_ptr(), n, 1, h1.data_ptr(), h2.data_ptr(), pr.data_ptr(), mt.data_ptr(), ds.data_ptr(), None,
                             stream=st.cuda_stream)
            outs.append((dg, h1, h2, pr, mt, ds)); genos.append(G)
    st.synchronize()
    assert m.status() == 0 and m.handover_faults() == 0
    flat = oracle.flatten(model)
    for G, (dg, h1, h2, pr, mt, ds) in zip(genos, outs):
        sub = np.arange(0, n, 25)
        want = oracle.predict(flat, G[sub], vote_method=1, want_prob=False, avx2=True, n_threads=8)
        assert np.array_equal(h1.cpu().numpy()[sub], want["h1"]) and np.array_equal(h2.cpu().numpy()[sub], want["h2"])
        assert np.array_equal(pr.cpu().numpy()[sub], want["prob"], equal_nan=True)
        assert np.array_equal(mt.cpu().numpy()[sub], want["matching"], equal_nan=True)
        assert np.array_equal(ds.cpu().numpy()[sub], want["dosage"], equal_nan=True)
    m.close()


def test_cfg4_hla_drb1_full_model_against_oracle(hib, oracle):
    """The DRB1 shape at full size: 100 classifiers x 500 haplotypes (12.5 M haplotype pairs per sample)."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-drb1")
    G, truth = synth.make_samples(founders, af, 2048)
    m = hib.hlaModelFromObj(model)
    assert m.stored_cells() > 0 and m.second_pass_pairs() == 0   # 73 pairs per cell: pass 1 stores every cell sum, pass 2 reads them back
    got = m.predict_raw(G, 1, want_dosage=True, want_prob=True)
    m.close()
    assert np.mean((got["h1"] == truth[:, 0]) & (got["h2"] == truth[:, 1])) > 0.9
    sub = np.arange(0, 2048, 86)[:24]                      # the oracle needs ~0.1 s per sample at this size
    want = oracle.predict(oracle.flatten(model), G[sub], vote_method=1, avx2=True, n_threads=8)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        assert np.array_equal(got[k][sub], want[k], equal_nan=True), k


def test_cfg5_training_1k_samples_300_snps(hib, oracle):
    from hibag_amd import synth, train
    model, founders, af = synth.make_model("hla-b", seed=9, n_snp=300, n_classifier=1, wide_classifier=False)
    G, truth = synth.make_samples(founders, af, 1000, seed=10)
    mtry = int(np.ceil(np.sqrt(300)))
    tr = train._Trainer(G, truth[:, 0], truth[:, 1], model.n_hla)
    tr.set_seed(100)
    n_cls = 10                                            # (~4 s of oracle time each: ten classifiers at the named shape, field by field)
    tr.new_classifiers(n_cls, mtry, True, False, False)
    got = tr.classifiers()
    tr.close()
    want = oracle.train(G, truth[:, 0], truth[:, 1], model.n_hla, n_cls, mtry, True, 100)
    assert len(got) == n_cls
    for i, (g, w) in enumerate(zip(got, want)):
        c = hib.Classifier(snpidx=w["snpidx"], freq=w["freq"], hla=w["hla"], haplo=w["haplo"], samp_num=w["samp_num"],
                           outofbag_acc=w["acc"])
        assert_same_classifier(dict(samp_num=g.samp_num, snpidx=g.snpidx, haplo=g.haplo, hla=g.hla, freq=g.freq,
                                    acc=g.outofbag_acc), c, i)
        assert len(g.snpidx) >= 5 and len(g.freq) >= 20


def test_large_model_1500_haplotypes_per_classifier(hib, oracle):
    """100 classifiers x 1,500 haplotypes = 112.6 M haplotype pairs per sample: with 20-byte pair records the
    round-1 library refused this model (2 GB stream cap); the matrix engine now keeps an O(H) haplotype table
    and 4-byte index pairs (0.9 GB for both passes' lists).  Finalizes, predicts, equals the oracle bit for bit."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-drb1", n_haplo=1500, wide_classifier=False)
    assert model.pair_evals_per_sample() == 100 * 1500 * 1501 // 2
    G, truth = synth.make_samples(founders, af, 192)
    m = hib.hlaModelFromObj(model)
    got = m.predict_raw(G, 1, want_dosage=True, want_prob=True)
    m.close()
    assert np.mean((got["h1"] == truth[:, 0]) & (got["h2"] == truth[:, 1])) > 0.9
    sub = np.array([0, 1, 63, 64, 100, 191])
    want = oracle.predict(oracle.flatten(model), G[sub], vote_method=1, avx2=True, n_threads=8)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        assert np.array_equal(got[k][sub], want[k], equal_nan=True), k


def test_cfg4_at_the_benchmarks_size_chunked_items_against_oracle(hib, oracle):
    """The DRB1 shape at the 4,096 samples bench.py times: 1,600 pass-1 items on 1,280 resident workgroups, so the last
    rounds run as chunks (12 per item) that resume at HibagModelView::blk_close rows with every cell stored -- the split of
    the loop at src/LibHLA.cpp:1776-1829.  (The 2,048-sample test above has 800 items: no chunks.)"""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-drb1")
    n = 4096
    G, truth = synth.make_samples(founders, af, n)
    m = hib.hlaModelFromObj(model)
    assert m.stored_cells() > 0 and m.second_pass_pairs() == 0
    got = m.predict_raw(G, 1, want_dosage=True, want_prob=True)
    assert m.handover_faults() == 0
    m.close()
    assert np.mean((got["h1"] == truth[:, 0]) & (got["h2"] == truth[:, 1])) > 0.9
    sub = np.concatenate([np.arange(0, n, 171)[:22], [n - 65, n - 1]])     # 24 samples, the last sample groups included
    want = oracle.predict(oracle.flatten(model), G[sub], vote_method=1, avx2=True, n_threads=8)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        assert np.array_equal(got[k][sub], want[k], equal_nan=True), k


@pytest.mark.parametrize("n", [700, 10_000])
def test_generated_rows_and_prebuilt_rows_give_the_same_bits(hib, oracle, monkeypatch, n):
    """The one-step FP4 walk of pass 1 has two sources for its A-operand rows: prebuilt (HibagModelView::parow: the model is
    small enough, the default for every shape of the suite) and generated from the O(H) haplotype table (larger models,
    HIBAG_PREBUILT_MB=0 here).  Both vote methods (the vote has builds of its own) at a batch of one round -- the general
    build, five workgroups per CU -- and of several -- the FP4-only builds at six, with chunked items: bit-equal to each
    other and, on a subset, to the oracle."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b")
    G, _ = synth.make_samples(founders, af, n, seed=77)
    G[5, :] = hib.NA_INTEGER
    pre = hib.hlaModelFromObj(model)
    monkeypatch.setenv("HIBAG_PREBUILT_MB", "0")
    gen = hib.hlaModelFromObj(model)
    monkeypatch.delenv("HIBAG_PREBUILT_MB")
    sub = np.arange(0, n, max(1, n // 30))[:30]
    for vote in (1, 2):
        a = pre.predict_raw(G, vote, want_dosage=True, want_prob=True)
        b = gen.predict_raw(G, vote, want_dosage=True, want_prob=True)
        for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
            assert np.array_equal(a[k], b[k], equal_nan=True), (vote, k)
        want = oracle.predict(oracle.flatten(model), G[sub], vote_method=vote, avx2=True, n_threads=8)
        for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
            assert np.array_equal(b[k][sub], want[k], equal_nan=True), (vote, k)
    assert pre.handover_faults() == 0 and gen.handover_faults() == 0
    pre.close(); gen.close()


_TAIL_K_SCRIPT = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["HIBAG_REPO"])
import hibag_amd as hib
from hibag_amd import synth
from oracle import oracle as O
O.build()
hib.hlaSetKernelTarget("hip")
shape, n = sys.argv[1], int(sys.argv[2])
model, founders, af = synth.make_model(shape)
G, _ = synth.make_samples(founders, af, n)
m = hib.hlaModelFromObj(model)
got = m.predict_raw(G, 1, want_dosage=True, want_prob=True)
faults = m.handover_faults()
m.close()
sub = np.arange(0, n, max(1, n // 40))[:40]
want = O.predict(O.flatten(model), G[sub], vote_method=1, avx2=True, n_threads=8)
bad = [k for k in ("h1", "h2", "prob", "matching", "dosage", "postprob") if not np.array_equal(got[k][sub], want[k], equal_nan=True)]
print(json.dumps({"bad": bad, "faults": int(faults)}))
"""


@pytest.mark.parametrize("tail_k", [1, 2, 8, 12])
def test_cfg2_every_number_of_chunks_per_item_equals_the_oracle(tail_k, tmp_path):
    """HIBAG_TAIL_K = chunks per work item of the last rounds of both passes (read once per process, hence a child
    process per value): 1 = undivided items, 2 / 8 / 12 around the default of 4.  The benchmark configuration
    (10,000 samples, HLA-B shape), 40 samples against the oracle, every output bit for bit."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "tail_k.py"
    script.write_text(_TAIL_K_SCRIPT)
    env = dict(os.environ, HIBAG_TAIL_K=str(tail_k), HIBAG_REPO=root)
    p = subprocess.run([sys.executable, str(script), "hla-b", "10000"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res == {"bad": [], "faults": 0}


def test_predict_multi_two_replicas_on_one_device_equal_the_single_call(hib, oracle):
    """hibag_hip_predict_multi with the device list [0, 0]: two replicas, two host threads, contiguous slices written in
    place -- bit-equal to the single call (samples are independent, src/LibHLA.cpp:2362-2411; the reference's
    hlaPredict(cl=) does the same over cluster workers, R/HIBAG.R:764-808)."""
    from hibag_amd import synth
    from hibag_amd.hibag import predict_multi
    model, founders, af = synth.make_model("hla-b")
    n = 5003                                       # ragged: the second slice ends inside a group of 64
    G, _ = synth.make_samples(founders, af, n)
    m = hib.hlaModelFromObj(model)
    single = m.predict_raw(G, 1, want_dosage=True, want_prob=True)
    r0, r1 = m.replicate(0), m.replicate(0)
    both = predict_multi([r0, r1], G, 1, want_dosage=True, want_prob=True)
    three = predict_multi([m, r0, r1], G[:130], 2, want_dosage=True, want_prob=False)   # more replicas than it is worth: still right
    single2 = m.predict_raw(G[:130], 2, want_dosage=True)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        assert np.array_equal(both[k], single[k], equal_nan=True), k
    for k in ("h1", "h2", "prob", "matching", "dosage"):
        assert np.array_equal(three[k], single2[k], equal_nan=True), k
    # through the front end: hlaPredict(cl = [devices])
    res1 = hib.hlaPredict(m, synth.as_snp_geno(model, G[:700]), type="response+dosage", verbose=False)
    res2 = hib.hlaPredict(m, synth.as_snp_geno(model, G[:700]), cl=[0, 0], type="response+dosage", verbose=False)
    assert res1.allele1 == res2.allele1 and res1.allele2 == res2.allele2
    assert np.array_equal(res1.prob, res2.prob, equal_nan=True) and np.array_equal(res1.dosage, res2.dosage, equal_nan=True)
    for bad in ([99], [], [0, -1], [0.5]):                     # device lists are validated up front
        with pytest.raises(ValueError):
            hib.hlaPredict(m, synth.as_snp_geno(model, G[:64]), cl=bad, verbose=False)
    with pytest.raises(hib.HibagHipError):
        m.replicate(99)
    # the host entries repair a failed hand-over themselves and return 0: only the counter would show it (as twice the time)
    for r in (m, r0, r1):
        assert r.handover_faults() == 0 and r.status() == 0
    for r in (r0, r1):
        r.close()
    m.close()


@pytest.mark.parametrize("n_haplo", [16383, 16384])
def test_haplotype_count_at_the_matrix_engines_index_limit(hib, oracle, n_haplo):
    """A classifier of 16,383 haplotypes is the largest the matrix engines take (a block's slot word holds the second
    haplotype's table index in 14 bits, the first's in 16: hibag_model.hip finalize_model); one of 16,384 goes to the vector
    engine whatever its SNP count.  134 million haplotype pairs per sample either way; beside it an ordinary classifier."""
    rng = np.random.default_rng(n_haplo)
    n_hla, k = 40, 20
    founders = rng.integers(0, 2, (n_hla, k))
    hla = np.sort(rng.integers(0, n_hla, n_haplo)).astype(np.int32)
    bits = founders[hla] ^ (rng.random((n_haplo, k)) < 0.08)
    haplo = ["".join("01"[b] for b in row) for row in bits]
    freq = rng.dirichlet(np.full(n_haplo, 0.3)) + 1e-9
    big = hib.Classifier(snpidx=np.arange(k), freq=freq, hla=hla, haplo=haplo)
    small = hib.Classifier(snpidx=np.arange(3, 15), freq=[0.2, 0.3, 0.5], hla=[0, 5, 39], haplo=["0" * 12, "01" * 6, "1" * 12])
    model = hib.HlaAttrBagObj(0, k, [f"{i:02d}" for i in range(n_hla)], [small, big])
    n = 70
    a, b = rng.integers(0, n_haplo, n), rng.integers(0, n_haplo, n)
    G = (bits[a].astype(np.int32) + bits[b].astype(np.int32)).astype(np.int32)
    G[rng.random(G.shape) < 0.03] = hib.NA_INTEGER
    G[1, :] = hib.NA_INTEGER
    m = hib.hlaModelFromObj(model)
    kind = m.engine(1)
    assert (kind[0] == "valu") == (n_haplo >= 16384), kind
    sub = np.array([0, 1, 2, 63, 64, 69])
    flat = oracle.flatten(model)
    for vote in (1, 2):
        got = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
        want = oracle.predict(flat, G[sub], vote_method=vote, avx2=True, n_threads=8)
        for key in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
            assert np.array_equal(got[key][sub], want[key], equal_nan=True), (vote, key)
    assert m.handover_faults() == 0
    m.close()
