"""The one numeric result the reference prints: the example of man/HIBAG-package.Rd:67-96,139-152

    set.seed(100); hlatab <- hlaSplitAllele(hla, train.prop=0.5)
    snpid <- hlaFlankingSNP(..., "A", 500*1000, assembly="hg19")          # 275 SNPs
    set.seed(100); model <- hlaAttrBagging(hlatab$training, train.geno, nclassifier=4)
    hapmap.ceu <- hlaBED2Geno(bed.fn, fam.fn, bim.fn, assembly="hg19")
    pred <- hlaPredict(model, hapmap.ceu, type="response"); head(pred$value)
    #   sample.id allele1 allele2      prob
    # 1   NA10859   01:01   03:01 0.9999992
    # 2   NA11882   01:01   29:02 1.0000000

replayed end to end.  It is the only value held by the reference that pins the ENSEMBLE probability
(missingness-weighted average over classifiers, src/LibHLA.cpp:2418-2480), the position-based SNP matching
of hlaPredict (R/HIBAG.R:550-686) and the BED import in one chain.  On the CPU with the oracle, and with
`-m gpu` through the HIP library (device-scored training, device BED decode, device prediction)."""

import math
import os

import numpy as np
import pytest

from conftest import REFDATA

BED = os.path.join(REFDATA, "HapMap_CEU.bed")
BIM = os.path.join(REFDATA, "HapMap_CEU.bim")
FAM = os.path.join(REFDATA, "HapMap_CEU.fam")
PRINTED = [("NA10859", "01:01", "03:01", 0.9999992), ("NA11882", "01:01", "29:02", 1.0000000)]


def _split_and_flank(hb, hapmap_geno, hla_type_table):
    ids, a1, a2 = list(hla_type_table["sample.id"]), list(hla_type_table["A.1"]), list(hla_type_table["A.2"])
    keep = [i for i in range(len(ids)) if a1[i] is not None and a2[i] is not None]
    hla = hb.hlaAllele([ids[i] for i in keep], [a1[i] for i in keep], [a2[i] for i in keep], locus="A", assembly="hg19")
    # set.seed(100); hlaSplitAllele(hla, 0.5) with the sample() of the R release the example was written under
    # (R < 3.6 "Rounding"; tests/test_run_tests_mirror.py pins this against OutOfBag.RData's training set)
    hlatab = hb.hlaSplitAllele(hla, train_prop=0.5, rng=hb.RRandom(100), sample_kind="Rounding")
    snpid = hb.hlaFlankingSNP(hapmap_geno.snp_id, hapmap_geno.snp_position, "A", 500 * 1000, assembly="hg19")
    assert len(snpid) == 275                                                   # man/HIBAG-package.Rd:81
    col = {s: i for i, s in enumerate(hapmap_geno.snp_id)}
    row = {s: i for i, s in enumerate(hapmap_geno.sample_id)}
    train_geno = hb.hlaGenoSubset(hapmap_geno, snp_sel=[col[s] for s in snpid],
                                  samp_sel=[row[s] for s in hlatab["training"].sample_id])
    return hlatab, train_geno


def _check(sample_id, allele1, allele2, prob):
    for i, (sid, x1, x2, p) in enumerate(PRINTED):
        assert sample_id[i] == sid and allele1[i] == x1 and allele2[i] == x2, (i, sample_id[i], allele1[i], allele2[i])
        assert round(float(prob[i]), 7) == p, (sid, prob[i])


def test_printed_example_with_the_oracle(oracle, hapmap_geno, hla_type_table):
    import hibag_amd as hb
    from hibag_amd import bed, snpmatch
    hlatab, train_geno = _split_and_flank(hb, hapmap_geno, hla_type_table)
    tr = hlatab["training"]
    # hlaAttrBagging's preparation (R/HIBAG.R:120-185): drop monomorphic SNPs, mtry = ceil(sqrt(n.snp))
    G = train_geno.genotype
    valid = (G >= 0) & (G <= 2)
    af = np.array([r[v].mean() * 0.5 if v.any() else np.nan for r, v in zip(G, valid)])
    use = np.where(np.isfinite(af) & (af > 0) & (af < 1))[0]
    assert len(use) == 266
    alleles = hb.hlaUniqueAllele(list(tr.allele1) + list(tr.allele2))
    lut = {a: i for i, a in enumerate(alleles)}
    h1 = np.array([lut[a] for a in tr.allele1], np.int32)
    h2 = np.array([lut[a] for a in tr.allele2], np.int32)
    out = oracle.train(np.ascontiguousarray(G[use].T.astype(np.int32)), h1, h2, len(alleles), nclassifier=4,
                       mtry=math.ceil(math.sqrt(len(use))), prune=True, seed=100)
    cls = [hb.Classifier(snpidx=o["snpidx"], freq=o["freq"], hla=o["hla"], haplo=o["haplo"], samp_num=o["samp_num"],
                         outofbag_acc=o["acc"]) for o in out]
    obj = hb.HlaAttrBagObj(n_samp=len(tr.sample_id), n_snp=len(use), hla_allele=alleles, classifiers=cls, hla_locus="A",
                           sample_id=list(tr.sample_id), snp_id=[train_geno.snp_id[i] for i in use],
                           snp_position=np.asarray(train_geno.snp_position)[use],
                           snp_allele=[train_geno.snp_allele[i] for i in use], snp_allele_freq=af[use], hla_freq=None,
                           assembly="hg19")
    # hlaBED2Geno(bed, fam, bim, assembly="hg19"): xMHC selection, decode (oracle's HIBAG_ConvBED)
    bim = bed._read_table(BIM, 6)
    fam = bed._read_table(FAM, 6)
    pos = np.array([float(r[3]) for r in bim])
    flag = bed._snp_selection("hg19", "xMHC", [r[0] for r in bim], pos, False)
    ix = np.where(flag)[0]
    g = oracle.conv_bed(open(BED, "rb").read(), len(fam), len(bim), flag.astype(np.int32))
    ceu = hb.HlaSNPGeno(genotype=np.ascontiguousarray(g.T), sample_id=[r[1] for r in fam], snp_id=[bim[i][1] for i in ix],
                        snp_position=pos[ix], snp_allele=[f"{bim[i][4]}/{bim[i][5]}" for i in ix], assembly="hg19")
    # hlaPredict(model, hapmap.ceu, type="response"): match.type="Position", allele check on
    mat, _ = snpmatch.match_snps_for_predict(obj, ceu, "Position", True, False, False, False)
    res = oracle.predict(oracle.flatten(obj), np.ascontiguousarray(mat.T.astype(np.int32)), vote_method=1)
    _check(ceu.sample_id, [alleles[i] for i in res["h1"][:2]], [alleles[i] for i in res["h2"][:2]], res["prob"])


@pytest.mark.gpu
@pytest.mark.parametrize("lazy", [False, True])
def test_printed_example_on_the_gpu(hapmap_geno, hla_type_table, lazy):
    import hibag_amd as hb
    hb.hlaSetKernelTarget("hip")
    hlatab, train_geno = _split_and_flank(hb, hapmap_geno, hla_type_table)
    hb.set_seed(100)
    model = hb.hlaAttrBagging(hlatab["training"], train_geno, nclassifier=4, verbose=False)
    assert len(model.obj.classifiers) == 4 and model.obj.n_snp == 266
    ceu = hb.hlaBED2Geno(BED, FAM, BIM, assembly="hg19", verbose=False, lazy=lazy)     # lazy: decoded by the kernels
    pred = hb.hlaPredict(model, ceu, type="response", verbose=False)
    _check(pred.sample_id, pred.allele1, pred.allele2, pred.prob)
    assert len(pred.sample_id) == 90
