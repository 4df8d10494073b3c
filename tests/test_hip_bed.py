"""GPU parity for the PLINK BED input step:
``hibag_hip_conv_bed`` (HIBAG_ConvBED on the device) against the oracle and the
reference's own fixture pair, and ``hibag_hip_predict_bed`` (BED decoded
straight into the kernels' packed form) against the ordinary predict entry and
the oracle -- bit-identical."""

import os

import numpy as np
import pytest

from conftest import REFDATA, write_bed, write_fam_bim

pytestmark = pytest.mark.gpu

BED = os.path.join(REFDATA, "HapMap_CEU.bed")
BIM = os.path.join(REFDATA, "HapMap_CEU.bim")
FAM = os.path.join(REFDATA, "HapMap_CEU.fam")
NA = -2147483648


@pytest.fixture(scope="module")
def hib():
    import hibag_amd
    hibag_amd.hlaSetKernelTarget("hip")
    return hibag_amd


def test_bed2geno_reproduces_the_reference_fixture(hib, oracle, hapmap_geno):
    """man/hlaBED2Geno.Rd's example: HapMap_CEU.bed/.fam/.bim -> the genotypes the
    reference ships as data/HapMap_CEU_Geno.rdata (on the fixture's SNPs/samples)."""
    g = hib.hlaBED2Geno(BED, FAM, BIM, import_chr="", assembly="hg19", verbose=False)
    assert g.genotype.shape == (5316, 90) and g.assembly == "hg19"
    flag = np.ones(5316, np.int32)
    assert np.array_equal(g.genotype.T, oracle.conv_bed(open(BED, "rb").read(), 90, 5316, flag))
    cj = [g.snp_id.index(s) for s in hapmap_geno.snp_id]
    ri = [g.sample_id.index(s) for s in hapmap_geno.sample_id]
    assert np.array_equal(g.genotype[np.ix_(cj, ri)], hapmap_geno.genotype)
    assert [g.snp_allele[j] for j in cj] == list(hapmap_geno.snp_allele)
    # the default xMHC import keeps a subset, decoded identically
    x = hib.hlaBED2Geno(BED, FAM, BIM, assembly="hg19", verbose=False)
    keep = [g.snp_id.index(s) for s in x.snp_id]
    assert 0 < len(keep) < 5316 and np.array_equal(x.genotype, g.genotype[keep])


@pytest.mark.parametrize("n_snp,n_samp", [(77, 131), (1, 1), (64, 64), (4, 257), (203, 3)])
def test_conv_bed_both_modes_vs_oracle(hib, oracle, tmp_path, n_snp, n_samp):
    rng = np.random.default_rng(n_snp * 1000 + n_samp)
    g = rng.integers(0, 4, size=(n_snp, n_samp)).astype(np.int32)
    g[g == 3] = NA
    pre = str(tmp_path / "c")
    write_fam_bim(pre, [f"s{i}" for i in range(n_samp)], [f"rs{i}" for i in range(n_snp)], ["6"] * n_snp,
                  30_000_000 + np.arange(n_snp), ["A/G"] * n_snp)
    for mode in (0, 1):
        write_bed(pre + ".bed", g, mode)
        got = hib.hlaBED2Geno(pre + ".bed", pre + ".fam", pre + ".bim", import_chr="", assembly="hg19", verbose=False)
        assert np.array_equal(got.genotype, g)
        flag = rng.random(n_snp) < 0.5
        flag[0] = True
        lazy = hib.hlaBED2Geno(pre + ".bed", pre + ".fam", pre + ".bim", import_chr="", assembly="hg19",
                               verbose=False, lazy=True).subset_snps(flag)
        want = oracle.conv_bed(open(pre + ".bed", "rb").read(), n_samp, n_snp, flag)
        assert np.array_equal(lazy.load().genotype.T, want)
        assert np.allclose(lazy.allele_freq(np.arange(flag.sum())),
                           [np.nan if (r == NA).all() else r[r != NA].mean() * 0.5 for r in g[flag]], equal_nan=True)


def test_conv_bed_errors(hib, tmp_path):
    from hibag_amd import _lib
    g = np.ones((8, 8), np.int32)
    p = write_bed(str(tmp_path / "t.bed"), g, 1)
    L = _lib.lib()
    out = np.zeros((8, 8), np.int32)
    flag = np.ones(8, np.int32)
    def conv(fn, n_samp, n_snp, n_save):
        return L.hibag_hip_conv_bed(fn.encode(), n_samp, n_snp, n_save, flag.ctypes.data, out.ctypes.data)
    assert conv(p, 8, 8, 8) == 0 and (out == 1).all()
    assert conv(p, 8, 8, 7) == -1                                   # flag count != n_save_snp
    assert conv(p, 16, 8, 8) == -1 and b"fewer than" in L.hibag_hip_last_error()
    assert conv(str(tmp_path / "missing.bed"), 8, 8, 8) == -1
    assert L.hibag_hip_last_error().startswith(b"Fail to open the file")


@pytest.mark.parametrize("mode", [0, 1])
def test_predict_straight_from_bed_hapmap(hib, oracle, model_a, hapmap_geno, tmp_path, mode):
    """hlaPredict on the lazily opened BED file == hlaPredict on the decoded object
    == the oracle, for the bundled HLA-A model and all 90 HapMap samples; both BED
    storage modes (the shipped file is individual-major; SNP-major is re-written)."""
    full = hib.hlaBED2Geno(BED, FAM, BIM, import_chr="", assembly="hg19", verbose=False)
    bed_fn = BED
    if mode == 1:
        bed_fn = write_bed(str(tmp_path / "snpmajor.bed"), full.genotype, 1)
    lazy = hib.hlaBED2Geno(bed_fn, FAM, BIM, assembly="hg19", verbose=False, lazy=True)
    eager = hib.hlaBED2Geno(bed_fn, FAM, BIM, assembly="hg19", verbose=False)
    assert isinstance(lazy, hib.HlaBEDGeno) and lazy.mode == mode
    m = hib.hlaModelFromObj(model_a)
    for vote in ("prob", "majority"):
        a = hib.hlaPredict(m, lazy, type="response+prob", vote=vote, match_type="RefSNP", verbose=False)
        b = hib.hlaPredict(m, eager, type="response+prob", vote=vote, match_type="RefSNP", verbose=False)
        assert a.sample_id == b.sample_id and a.allele1 == b.allele1 and a.allele2 == b.allele2
        for k in ("prob", "matching", "dosage", "postprob"):
            assert np.array_equal(getattr(a, k), getattr(b, k), equal_nan=True), k
    # and against the oracle on the matrix the matching step builds
    col = [full.snp_id.index(s) for s in model_a.snp_id]
    G = np.ascontiguousarray(full.genotype[col].T)
    want = oracle.predict(oracle.flatten(model_a), G, vote_method=1)
    got = hib.hlaPredict(m, lazy, type="response+prob", match_type="RefSNP", allele_check=False, verbose=False)
    assert np.array_equal(got.h1, want["h1"]) and np.array_equal(got.h2, want["h2"])
    assert np.array_equal(got.postprob.T, want["postprob"], equal_nan=True)
    # a device list with a lazily opened BED file is refused, not silently ignored (the BED route decodes on one device)
    with pytest.raises(ValueError, match="device"):
        hib.hlaPredict(m, lazy, cl=[0], match_type="RefSNP", verbose=False)


@pytest.mark.parametrize("mode", [0, 1])
def test_predict_bed_missing_snps_flips_and_batches(hib, oracle, tmp_path, mode):
    """Synthetic HLA-A-sized model; cohort file lacks some model SNPs, has extra SNPs,
    a shuffled SNP order and reversed A/B alleles on a third of the SNPs."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-a-small", seed=11)
    G, _ = synth.make_samples(founders, af, 333, seed=12)            # [n_samp, S]
    S = model.n_snp
    rng = np.random.default_rng(13)
    keep = rng.random(S) < 0.9
    flip = rng.random(S) < 0.33
    extra = 17
    order = rng.permutation(int(keep.sum()) + extra)
    rows, ids, pos, alle = [], [], [], []
    for k in np.where(keep)[0]:
        g = G[:, k].copy()
        if flip[k]:
            g = np.where(g == NA, NA, 2 - g)
        rows.append(g); ids.append(model.snp_id[k]); pos.append(model.snp_position[k])
        alle.append("G/A" if flip[k] else "A/G")
    for e in range(extra):
        rows.append(rng.integers(0, 3, G.shape[0]).astype(np.int32)); ids.append(f"x{e}"); pos.append(1000 + e); alle.append("C/T")
    rows = [rows[i] for i in order]; ids = [ids[i] for i in order]; pos = [pos[i] for i in order]; alle = [alle[i] for i in order]
    pre = str(tmp_path / "cohort")
    write_fam_bim(pre, [f"s{i}" for i in range(G.shape[0])], ids, ["6"] * len(ids), pos, alle)
    write_bed(pre + ".bed", np.array(rows), mode)
    lazy = hib.hlaBED2Geno(pre + ".bed", pre + ".fam", pre + ".bim", import_chr="", assembly="hg19", verbose=False, lazy=True)
    m = hib.hlaModelFromObj(model)
    got = hib.hlaPredict(m, lazy, type="response+prob", verbose=False)
    Gm = G.copy()
    Gm[:, ~keep] = NA
    want = oracle.predict(oracle.flatten(model), Gm, vote_method=1)
    assert np.array_equal(got.h1, want["h1"]) and np.array_equal(got.h2, want["h2"])
    assert np.array_equal(got.prob, want["prob"], equal_nan=True)
    assert np.array_equal(got.matching, want["matching"], equal_nan=True)
    assert np.array_equal(got.postprob.T, want["postprob"], equal_nan=True)
    assert np.array_equal(got.dosage.T, want["dosage"], equal_nan=True)


def test_predict_bed_full_size_equals_matrix_path(hib, tmp_path):
    """BASELINE config 2 shape (HLA-B-like, 10k samples): the BED route and the int32
    route give the same bits."""
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b", seed=3)
    G, _ = synth.make_samples(founders, af, 10000, seed=4)
    p = write_bed(str(tmp_path / "big.bed"), G.T, 1)
    m = hib.hlaModelFromObj(model)
    a = m.predict_bed(p, 10000, model.n_snp, np.arange(model.n_snp), None, 1, want_dosage=True, want_prob=True)
    b = m.predict_raw(G, 1, want_dosage=True, want_prob=True)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize("n_samp", [333, 140_000])
def test_int32_route_selects_and_flips_on_the_device(hib, oracle, n_samp):
    """hlaPredict on an hlaSNPGenoClass whose SNP set differs from the model's (R/HIBAG.R:640-676): missing
    model SNPs, extra cohort SNPs, shuffled order, reversed A/B alleles on a third of the SNPs.  The library
    gets the cohort's own matrix plus the column map (hibag_hip_predict_mapped) and gathers / flips while
    packing; the oracle gets the matrix the reference's host code would have built.  140,000 samples cross
    the library's batch cut (the map must follow the slices)."""
    from hibag_amd import synth
    from hibag_amd.snpmatch import match_snps_for_predict
    model, founders, af = synth.make_model("hla-a-small", seed=11)
    G, _ = synth.make_samples(founders, af, n_samp, seed=12)          # [n_samp, S]
    S = model.n_snp
    rng = np.random.default_rng(13)
    keep = rng.random(S) < 0.9
    flip = rng.random(S) < 0.33
    extra = 17
    order = rng.permutation(int(keep.sum()) + extra)
    rows, ids, pos, alle = [], [], [], []
    for k in np.where(keep)[0]:
        g = G[:, k].copy()
        if flip[k]:
            g = np.where(g == NA, NA, 2 - g)
        rows.append(g); ids.append(model.snp_id[k]); pos.append(model.snp_position[k])
        alle.append("G/A" if flip[k] else "A/G")
    for e in range(extra):
        rows.append(rng.integers(0, 3, G.shape[0]).astype(np.int32)); ids.append(f"x{e}"); pos.append(1000 + e); alle.append("C/T")
    cohort = hib.HlaSNPGeno(genotype=np.array([rows[i] for i in order], np.int32), sample_id=[f"s{i}" for i in range(n_samp)],
                            snp_id=[ids[i] for i in order], snp_position=np.array([pos[i] for i in order], np.float64),
                            snp_allele=[alle[i] for i in order], assembly="hg19")
    m = hib.hlaModelFromObj(model)
    got = hib.hlaPredict(m, cohort, type="response+prob" if n_samp < 1000 else "response+dosage", verbose=False)
    # what the reference's host code builds, then the oracle (a subset at the large size)
    mat, _ = match_snps_for_predict(model, cohort, "Position", True, False, False, False)
    sub = np.arange(n_samp) if n_samp < 1000 else np.unique(np.concatenate([[0, n_samp - 1], rng.choice(n_samp, 300, replace=False)]))
    Gm = np.ascontiguousarray(mat.T[sub].astype(np.int32))
    Gexp = G[sub].copy(); Gexp[:, ~keep] = NA
    assert np.array_equal(Gm, Gexp)
    want = oracle.predict(oracle.flatten(model), Gm, vote_method=1)
    assert np.array_equal(got.h1[sub], want["h1"]) and np.array_equal(got.h2[sub], want["h2"])
    assert np.array_equal(got.prob[sub], want["prob"], equal_nan=True)
    assert np.array_equal(got.matching[sub], want["matching"], equal_nan=True)
    assert np.array_equal(got.dosage.T[sub], want["dosage"], equal_nan=True)
    if n_samp < 1000:
        assert np.array_equal(got.postprob.T, want["postprob"], equal_nan=True)
    # the mapped entry and the plain entry on the pre-built matrix agree bit for bit on every sample
    plain = m.predict_raw(np.ascontiguousarray(mat.T.astype(np.int32)), 1, want_dosage=True)
    assert np.array_equal(got.h1, plain["h1"]) and np.array_equal(got.prob, plain["prob"], equal_nan=True)
    assert np.array_equal(got.dosage.T, plain["dosage"], equal_nan=True)
    # argument errors
    from hibag_amd import _lib
    bad = np.full(model.n_snp, cohort.genotype.shape[0], np.int32)
    with pytest.raises(_lib.HibagHipError, match="outside the"):
        m.predict_mapped(np.zeros((2, cohort.genotype.shape[0]), np.int32), bad)
