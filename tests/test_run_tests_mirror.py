"""The reference's own test script, tests/runTests.R, line for line: for each of six HLA genes split the
typed HapMap CEU samples (set.seed(100); hlaSplitAllele), take the SNPs within 500 kb, train ten
classifiers (set.seed(100); hlaAttrBagging), predict the validation half and require the haplotype
accuracy to reach the script's lower bound.  Plus the known answer behind it: with the sampling of the
R release that wrote the fixture, the split reproduces the training set stored in OutOfBag.RData."""

import numpy as np
import pytest

HLA_LIST = ["A", "B", "C", "DQA1", "DQB1", "DRB1"]           # tests/runTests.R:14
HLA_ACC = [0.9, 0.8, 0.8, 0.8, 0.8, 0.7]                      # tests/runTests.R:17


def _hla(hb, table, gene):
    ids, a1, a2 = list(table["sample.id"]), list(table[f"{gene}.1"]), list(table[f"{gene}.2"])
    keep = [i for i in range(len(ids)) if a1[i] is not None and a2[i] is not None]      # hlaAllele(..., na.rm=TRUE)
    return hb.hlaAllele([ids[i] for i in keep], [a1[i] for i in keep], [a2[i] for i in keep], locus=gene, assembly="hg19")


def test_split_reproduces_the_stored_training_set(hla_type_table, model_oob):
    """vignettes/HIBAG.Rmd:196-198 with R 3.4's sample() ("Rounding"; the fixture was written by R 3.4.1)."""
    import hibag_amd as hb
    hla = _hla(hb, hla_type_table, "A")
    split = hb.hlaSplitAllele(hla, train_prop=0.5, rng=hb.RRandom(100), sample_kind="Rounding")
    assert split["training"].sample_id == list(model_oob.sample_id)
    assert len(split["validation"].sample_id) == len(hla.sample_id) - 34
    assert not set(split["training"].sample_id) & set(split["validation"].sample_id)


@pytest.mark.gpu
@pytest.mark.parametrize("gene,floor", list(zip(HLA_LIST, HLA_ACC)))
def test_run_tests_r(gene, floor, hapmap_geno, hla_type_table):
    import hibag_amd as hb
    hb.hlaSetKernelTarget("hip")
    hla = _hla(hb, hla_type_table, gene)
    hb.set_seed(100)
    hlatab = hb.hlaSplitAllele(hla, train_prop=0.5)
    snpid = hb.hlaFlankingSNP(hapmap_geno.snp_id, hapmap_geno.snp_position, gene, 500 * 1000, assembly="hg19")
    col = {s: i for i, s in enumerate(hapmap_geno.snp_id)}
    row = {s: i for i, s in enumerate(hapmap_geno.sample_id)}
    train_geno = hb.hlaGenoSubset(hapmap_geno, snp_sel=[col[s] for s in snpid],
                                  samp_sel=[row[s] for s in hlatab["training"].sample_id])
    test_geno = hb.hlaGenoSubset(hapmap_geno, samp_sel=[row[s] for s in hlatab["validation"].sample_id])
    hb.set_seed(100)
    model = hb.hlaAttrBagging(hlatab["training"], train_geno, nclassifier=10, verbose=False)
    pred = hb.hlaPredict(model, test_geno, type="response", verbose=False)
    comp = hb.hlaCompareAllele(hlatab["validation"], pred, allele_limit=model, call_threshold=0)
    assert comp["acc.haplo"] >= floor, (gene, comp)
    assert len(model.obj.classifiers) == 10 and np.all(np.isfinite(model.obj.matching))
