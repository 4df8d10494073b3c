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


def _campaign_case(hib, rng, big):
    """One random model + cohort of the wide campaign: classifiers of 1..128 SNPs (one-step FP4, int8, several-step FP4 and the
    vector engine mixed in one model), up to 40 classifiers, up to 200 haplotypes each, frequencies over 300 decades in some,
    cohorts up to 6,000 samples (`big`: 12,000+, where pass 1 cuts work items into chunks that hand over)."""
    n_hla = int(rng.integers(1, 60))
    n_snp = int(rng.integers(1, 200))
    kmax = [8, 30, 32, 64, 112, 128][int(rng.integers(0, 6))]
    n_cls = int(rng.integers(1, 41)) if not big else int(rng.integers(20, 60))
    tiny = rng.random() < 0.2
    cls = []
    for _ in range(n_cls):
        k = int(rng.integers(1, min(n_snp, kmax) + 1))
        H = int(rng.integers(1, 200)) if not big else int(rng.integers(40, 120))
        hla = np.sort(rng.integers(0, n_hla, H)).astype(np.int32)
        freq = 10.0 ** rng.uniform(-300 if tiny else -5, 0, H)
        if rng.random() < 0.1:
            freq[rng.random(H) < 0.3] = 0.0
        haplo = ["".join(rng.choice(["0", "1"], k)) for _ in range(H)]
        cls.append(hib.Classifier(snpidx=rng.choice(n_snp, k, replace=False), freq=freq, hla=hla, haplo=haplo))
    model = hib.HlaAttrBagObj(n_samp=0, n_snp=n_snp, hla_allele=[f"{i:02d}" for i in range(n_hla)], classifiers=cls)
    n = int(rng.integers(12_000, 16_000)) if big else int(rng.choice([rng.integers(1, 70), rng.integers(1, 1500), rng.integers(1, 6000)]))
    style = int(rng.integers(0, 3))
    if style == 0:
        G = rng.choice(np.array([0, 1, 2, hib.NA_INTEGER, -1, 3], np.int64), size=(n, n_snp), p=[.3, .3, .3, .04, .03, .03]).astype(np.int32)
    else:
        # samples made of two haplotypes of a random classifier each (so that somebody matches), 2 % missing
        G = rng.integers(0, 3, size=(n, n_snp)).astype(np.int32)
        for c0 in ([cls[int(rng.integers(0, n_cls))]] if style == 1 else cls[:: max(1, n_cls // 4)]):
            H0 = np.array([[int(ch) for ch in h] for h in c0.haplo])
            a, b = rng.integers(0, len(c0.haplo), n), rng.integers(0, len(c0.haplo), n)
            G[:, np.asarray(c0.snpidx)] = H0[a] + H0[b]
        G[rng.random(G.shape) < 0.02] = hib.NA_INTEGER
    if rng.random() < 0.3:
        G[int(rng.integers(0, n)), :] = hib.NA_INTEGER
    return model, G


def test_wide_campaign(oracle, monkeypatch):
    """Time-boxed random campaign over a wider space than the test above (every engine, up to 40 classifiers, cohorts of
    thousands, both votes, the per-sample plugin route on a few samples): HIBAG_FUZZ_SECONDS (default 20) of cases, each
    bit-equal to the oracle.  HIBAG_FUZZ_SEED moves the window; a long run's summary is profiles/r04_fuzz_campaign.txt."""
    import os
    import time
    import hibag_amd as hib
    from hibag_amd import plugin
    hib.hlaSetKernelTarget("hip")
    budget = float(os.environ.get("HIBAG_FUZZ_SECONDS", "20"))
    seed0 = int(os.environ.get("HIBAG_FUZZ_SEED", "90000"))
    big_every = max(1, int(os.environ.get("HIBAG_FUZZ_BIG_EVERY", "25")))      # every n-th case is a cohort of 12,000-16,000 samples
    t_end = time.time() + budget
    bad, done, samples, seed = [], 0, 0, seed0
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        if os.environ.get("HIBAG_FUZZ_LOG"):            # (the seed in flight, should the process die)
            with open(os.environ["HIBAG_FUZZ_LOG"], "w") as f:
                f.write(f"{seed}\n")
        monkeypatch.setenv("HIBAG_STORE_PAIRS", str(int(rng.integers(0, 14))) if rng.random() < 0.5 else "")
        if not os.environ.get("HIBAG_STORE_PAIRS"):
            monkeypatch.delenv("HIBAG_STORE_PAIRS", raising=False)
        model, G = _campaign_case(hib, rng, big=(seed % big_every == big_every - 1))
        # (drawn after the case, so that a seed names the same model whatever is varied here) which cell sums pass 1 stores for
        # pass 2 -- all / those of the big cells / none -- and, now and then, every classifier on the vector engine or on int8 MFMA
        mode = ["", "", "stream", "hybrid", "recompute"][int(rng.integers(0, 5))]
        monkeypatch.setenv("HIBAG_PASS2", mode) if mode else monkeypatch.delenv("HIBAG_PASS2", raising=False)
        eng = ["valu", "i8", ""][min(int(rng.random() / 0.08), 2)]     # 8 % vector engine, 8 % int8 MFMA instead of FP4
        monkeypatch.setenv("HIBAG_ENGINE", eng) if eng else monkeypatch.delenv("HIBAG_ENGINE", raising=False)
        flat = oracle.flatten(model)
        m = hib.hlaModelFromObj(model)
        for vote in (1, 2):
            got = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
            want = oracle.predict(flat, G, vote_method=vote, avx2=True, n_threads=8)
            for key in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
                if not np.array_equal(got[key], want[key], equal_nan=True):
                    bad.append((seed, vote, key, G.shape, len(model.classifiers)))
                    break
        if m.handover_faults():
            bad.append((seed, "handover_faults", int(m.handover_faults())))
        if seed % 5 == 0:
            # the reference's per-sample hook on the first samples: the averaged posterior of vote = prob, un-normalised
            sub = G[: min(len(G), 6)]
            want = oracle.predict(flat, sub, vote_method=1, avx2=False)
            host = plugin.PluginHost(model)
            geno, wt = host.pack(sub)
            prob = np.zeros(model.n_cell); match = np.zeros(1)
            for i in range(len(sub)):
                host.avg_prob(geno[i], wt[i], prob, match)
                if not (np.array_equal(prob, want["postprob"][i], equal_nan=True) and
                        (match[0] == want["matching"][i] or (np.isnan(match[0]) and np.isnan(want["matching"][i])))):
                    bad.append((seed, "plugin", i))
                    break
            host.close()
        m.close()
        done += 1
        samples += len(G)
        seed += 1
    monkeypatch.delenv("HIBAG_PASS2", raising=False)
    monkeypatch.delenv("HIBAG_ENGINE", raising=False)
    print(f"wide campaign: {done} models (seeds {seed0}..{seed - 1}), {samples} samples, {len(bad)} mismatches")
    if os.environ.get("HIBAG_FUZZ_REPORT"):
        with open(os.environ["HIBAG_FUZZ_REPORT"], "a") as f:
            f.write(f"seeds {seed0}..{seed - 1}: {done} models, {samples} samples, both votes, mismatches: {bad}\n")
    assert not bad, bad


def test_campaign_case_with_vector_engine_classifiers_last(oracle, monkeypatch):
    """Seed 202699 of the campaign above, found by it: 59 classifiers of up to 128 SNPs whose last ones run on the vector engine
    (no B-operand rows), every non-empty cell stored (HIBAG_STORE_PAIRS=0), 12,251 samples = 192 sample groups.  Pass 2 requests
    the operand rows named by every block header it passes, stored-sums-only blocks included; for those classifiers the header named
    the row one past the end of the batch's operand array -- harmless while the over-read stayed inside the allocation's slack, a
    device memory fault at this size.  (The headers now name row 0 and the array has two rows to spare.)"""
    import hibag_amd as hib
    hib.hlaSetKernelTarget("hip")
    seed = 202699
    rng = np.random.default_rng(seed)
    sp = str(int(rng.integers(0, 14))) if rng.random() < 0.5 else ""
    assert sp == "0"
    monkeypatch.setenv("HIBAG_STORE_PAIRS", sp)
    model, G = _campaign_case(hib, rng, big=True)
    assert len(model.classifiers) == 59 and len(G) == 12251 and len(model.classifiers[-1].snpidx) == 120
    m = hib.hlaModelFromObj(model)
    assert m.stored_cells() > 0 and m.second_pass_pairs() > 0          # pass 2 = k_accum with stored sums
    sub = np.concatenate([np.arange(0, len(G), 97), np.arange(len(G) - 70, len(G))])
    flat = oracle.flatten(model)
    for vote in (1, 2):
        got = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
        want = oracle.predict(flat, G[sub], vote_method=vote, avx2=True, n_threads=8)
        for key in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
            assert np.array_equal(got[key][sub], want[key], equal_nan=True), (vote, key)
    assert m.handover_faults() == 0
    m.close()


def test_entry_points_campaign(monkeypatch, tmp_path):
    """Time-boxed random campaign over the OTHER ways into the same kernels (HIBAG_FUZZ_SECONDS, default 15), each against
    hibag_hip_predict on the model-order matrix (which the campaign above pins to the oracle): the cohort's own matrix with a
    column map and flips (hibag_hip_predict_mapped; the same SNP-major through hibag_hip_predict_snp_major), a PLINK BED file in either storage mode (hibag_hip_predict_bed), the device
    entry on a caller's stream with device-resident data, replicas on the one device (hibag_hip_predict_multi), and classifier
    shards merged by the library's RCCL all-reduce (posteriors to 1e-10 and identical calls up to rounding-level ties: the summation order differs)."""
    import os
    import time
    import torch
    import hibag_amd as hib
    from conftest import write_bed
    hib.hlaSetKernelTarget("hip")
    NA = hib.NA_INTEGER
    budget = float(os.environ.get("HIBAG_FUZZ_SECONDS", "15"))
    seed0 = int(os.environ.get("HIBAG_FUZZ_SEED", "60000"))
    t_end = time.time() + budget
    seed, done, bad = seed0, 0, []
    dev = torch.device("cuda", 0)
    keys = ("h1", "h2", "prob", "matching", "dosage", "postprob")

    def same(a, b):
        return all(np.array_equal(a[k], b[k], equal_nan=True) for k in keys)

    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        if os.environ.get("HIBAG_FUZZ_LOG"):
            with open(os.environ["HIBAG_FUZZ_LOG"], "w") as f:
                f.write(f"{seed}\n")
        monkeypatch.delenv("HIBAG_STORE_PAIRS", raising=False)
        model, G = _campaign_case(hib, rng, big=False)
        n, S = G.shape
        vote = int(rng.integers(1, 3))
        m = hib.hlaModelFromObj(model)
        ref = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
        # -- the cohort's own matrix: model SNP k in column col[k] (or absent), reversed where flip[k]
        n_extra = int(rng.integers(0, 12))
        absent = rng.random(S) < 0.1
        flip = rng.random(S) < 0.3
        cols = rng.permutation(S + n_extra)[:S].astype(np.int32)
        cohort = rng.integers(0, 3, size=(n, S + n_extra)).astype(np.int32)
        valid = (G >= 0) & (G <= 2)
        cohort[:, cols] = np.where(flip[None, :] & valid, 2 - G, G)
        col = np.where(absent, -1, cols).astype(np.int32)
        Gm = G.copy()
        Gm[:, absent] = NA
        want = m.predict_raw(Gm, vote, want_dosage=True, want_prob=True)
        got = m.predict_mapped(cohort, col, flip, vote, want_dosage=True, want_prob=True)
        if not same(got, want):
            bad.append((seed, "mapped"))
        # -- the same cohort SNP-major ([SNP][sample], rows `ld` apart): hibag_hip_predict_snp_major gathers the model's rows
        ld = n + int(rng.integers(0, 9))
        wide = np.full((S + n_extra, ld), 9, np.int32)
        wide[:, :n] = cohort.T
        got = m.predict_snp_major(wide[:, :n], col, flip, vote, want_dosage=True, want_prob=True)
        if not same(got, want):
            bad.append((seed, "snp-major"))
        # -- the same cohort as a BED file (values outside 0..2 are missing there as here)
        if seed % 3 == 0:
            mode = int(rng.integers(0, 2))
            path = write_bed(str(tmp_path / f"c{seed}.bed"), cohort.T, mode)
            got = m.predict_bed(path, n, S + n_extra, col, flip, vote, want_dosage=True, want_prob=True)
            os.remove(path)
            if not same(got, want):
                bad.append((seed, "bed", mode))
        # -- device entry on a stream of the caller's
        if seed % 2 == 0:
            st = torch.cuda.Stream(dev)
            dg = torch.from_numpy(G).to(dev)
            o = dict(h1=torch.zeros(n, dtype=torch.int32, device=dev), h2=torch.zeros(n, dtype=torch.int32, device=dev),
                     prob=torch.zeros(n, dtype=torch.float64, device=dev), matching=torch.zeros(n, dtype=torch.float64, device=dev),
                     dosage=torch.zeros((n, model.n_hla), dtype=torch.float64, device=dev),
                     postprob=torch.zeros((n, model.n_cell), dtype=torch.float64, device=dev))
            torch.cuda.synchronize(dev)
            m.predict_device(dg.data_ptr(), n, vote, *[o[k].data_ptr() for k in keys], stream=st.cuda_stream)
            st.synchronize()
            if not same({k: o[k].cpu().numpy() for k in keys}, ref):
                bad.append((seed, "device"))
        # -- replicas on the one device
        if seed % 4 == 1:
            reps = [m] + [m.replicate(0) for _ in range(int(rng.integers(1, 4)))]
            got = hib.hibag.predict_multi(reps, G, vote, want_dosage=True, want_prob=True)
            for r in reps[1:]:
                r.close()
            if not same(got, ref):
                bad.append((seed, "multi"))
        # -- classifier shards merged by the library's all-reduce (vote = prob only)
        if seed % 4 == 3 and len(model.classifiers) >= 2:
            want1 = ref if vote == 1 else m.predict_raw(G, 1, want_dosage=True, want_prob=True)
            grp = hib.hibag.ShardGroup(m, [0] * int(rng.integers(2, min(len(model.classifiers), 6) + 1)))
            got = grp.predict_raw(G, want_dosage=True, want_prob=True)
            grp.close()
            # identical calls -- except where the two best cells of a sample tie to within rounding: the shards add the
            # classifiers' terms in another order, and a call is the FIRST strict maximum (seed 950299: two cells one ulp apart
            # in the one-model sums, equal in the shards'; tools/fuzz_repro_shards.py).  There the calls' probabilities must agree.
            differ = (got["h1"] != want1["h1"]) | (got["h2"] != want1["h2"])
            ok = bool(np.all(np.abs(got["prob"][differ] - want1["prob"][differ]) <= 1e-12 * np.abs(want1["prob"][differ])))
            with np.errstate(invalid="ignore", divide="ignore"):
                fin = np.isfinite(want1["postprob"]) & (want1["postprob"] > 1e-200)
                rel = np.abs(got["postprob"] - want1["postprob"])[fin] / want1["postprob"][fin]
            ok = ok and (rel.size == 0 or float(rel.max()) < 1e-10)
            ok = ok and np.array_equal(np.isnan(got["postprob"]), np.isnan(want1["postprob"]))
            if not ok:
                bad.append((seed, "shards"))
        if m.handover_faults():
            bad.append((seed, "handover_faults"))
        m.close()
        done += 1
        seed += 1
    print(f"entry-point campaign: {done} models (seeds {seed0}..{seed - 1}), {len(bad)} mismatches")
    if os.environ.get("HIBAG_FUZZ_REPORT"):
        with open(os.environ["HIBAG_FUZZ_REPORT"], "a") as f:
            f.write(f"entry points, seeds {seed0}..{seed - 1}: {done} models, mismatches: {bad}\n")
    assert not bad, bad


def test_plugin_campaign(oracle):
    """Time-boxed random campaign of the reference's per-sample GPU hook (TypeGPUExtProc.predict_init / predict_avg_prob /
    predict_done through hibag_amd.plugin.PluginHost; hibag_sample.hip's one kernel per call): up to 150 classifiers (workgroup =
    classifier), up to 110 alleles (6,105 cells: more than one round of 4,096 cells per workgroup), up to 250 haplotypes (31,375
    pairs: several rounds of 8,192 pairs), 1..128 SNPs -- posterior and matching of every sample bit-equal to the oracle.
    HIBAG_FUZZ_SECONDS (default 10)."""
    import os
    import time
    import hibag_amd as hib
    from hibag_amd import plugin
    hib.hlaSetKernelTarget("hip")
    budget = float(os.environ.get("HIBAG_FUZZ_SECONDS", "10"))
    seed0 = int(os.environ.get("HIBAG_FUZZ_SEED", "80000"))
    t_end = time.time() + budget
    seed, done, bad = seed0, 0, []
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        if os.environ.get("HIBAG_FUZZ_LOG"):
            with open(os.environ["HIBAG_FUZZ_LOG"], "w") as f:
                f.write(f"{seed}\n")
        n_hla = int(rng.choice([rng.integers(1, 20), rng.integers(1, 60), rng.integers(60, 111)]))
        n_snp = int(rng.integers(1, 200))
        n_cls = int(rng.choice([rng.integers(1, 6), rng.integers(1, 40), rng.integers(40, 151)]))
        hmax = int(rng.choice([8, 60, 250]))
        kmax = [8, 30, 32, 64, 128][int(rng.integers(0, 5))]
        tiny = rng.random() < 0.2
        cls = []
        for _ in range(n_cls):
            k = int(rng.integers(1, min(n_snp, kmax) + 1))
            H = int(rng.integers(0 if rng.random() < 0.03 else 1, hmax + 1))
            hla = np.sort(rng.integers(0, n_hla, H)).astype(np.int32)
            freq = 10.0 ** rng.uniform(-300 if tiny else -5, 0, H)
            haplo = ["".join(rng.choice(["0", "1"], k)) for _ in range(H)]
            cls.append(hib.Classifier(snpidx=rng.choice(n_snp, k, replace=False), freq=freq, hla=hla, haplo=haplo))
        model = hib.HlaAttrBagObj(n_samp=0, n_snp=n_snp, hla_allele=[f"{i:03d}" for i in range(n_hla)], classifiers=cls)
        n = 5
        G = rng.integers(0, 3, size=(n, n_snp)).astype(np.int32)
        live = [c for c in cls if len(c.freq)]
        for i in range(n):
            for c0 in live[:: max(1, len(live) // 3)]:
                H0 = np.array([[int(ch) for ch in h] for h in c0.haplo])
                a, b = rng.integers(0, len(c0.haplo), 2)
                G[i, np.asarray(c0.snpidx)] = H0[a] + H0[b]
        G[rng.random(G.shape) < 0.03] = hib.NA_INTEGER
        if rng.random() < 0.3:
            G[0, :] = hib.NA_INTEGER
        want = oracle.predict(oracle.flatten(model), G, vote_method=1, avx2=True, n_threads=4)
        host = plugin.PluginHost(model)
        geno, wt = host.pack(G)
        prob = np.zeros(model.n_cell); match = np.zeros(1)
        for i in range(n):
            host.avg_prob(geno[i], wt[i], prob, match)
            if not (np.array_equal(prob, want["postprob"][i], equal_nan=True) and
                    (match[0] == want["matching"][i] or (np.isnan(match[0]) and np.isnan(want["matching"][i])))):
                bad.append((seed, i, n_hla, n_cls, hmax, kmax))
                break
        host.close()
        done += 1
        seed += 1
    print(f"plugin campaign: {done} models (seeds {seed0}..{seed - 1}), {len(bad)} mismatches")
    if os.environ.get("HIBAG_FUZZ_REPORT"):
        with open(os.environ["HIBAG_FUZZ_REPORT"], "a") as f:
            f.write(f"per-sample hook, seeds {seed0}..{seed - 1}: {done} models x 5 samples, mismatches: {bad}\n")
    assert not bad, bad
