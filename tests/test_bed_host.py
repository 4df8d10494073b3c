"""CPU side of the PLINK BED input step: the oracle's
restatement of HIBAG_ConvBED pinned on the reference's own pair of fixtures, and
the host logic of hlaBED2Geno (annotation, SNP selection, prefix check)."""

import os

import numpy as np
import pytest

from conftest import REFDATA, write_bed

BED = os.path.join(REFDATA, "HapMap_CEU.bed")
BIM = os.path.join(REFDATA, "HapMap_CEU.bim")
FAM = os.path.join(REFDATA, "HapMap_CEU.fam")


def test_oracle_conv_bed_reproduces_the_reference_fixture(oracle, hapmap_geno):
    """inst/extdata/HapMap_CEU.bed (individual-major) decoded by the oracle equals
    data/HapMap_CEU_Geno.rdata -- genotypes, alleles and positions -- on the
    fixture's 1564 SNPs x 60 samples: pins code table, bit order and layout."""
    from hibag_amd import bed
    bim = bed._read_table(BIM, 6)
    fam = bed._read_table(FAM, 6)
    image = open(BED, "rb").read()
    assert image[2] == 0
    got = oracle.conv_bed(image, len(fam), len(bim), np.ones(len(bim), np.int32))      # [90, 5316]
    col = {r[1]: j for j, r in enumerate(bim)}
    row = {r[1]: i for i, r in enumerate(fam)}
    cj = [col[s] for s in hapmap_geno.snp_id]
    ri = [row[s] for s in hapmap_geno.sample_id]
    assert np.array_equal(got[np.ix_(ri, cj)].T, hapmap_geno.genotype)
    assert [f"{bim[j][4]}/{bim[j][5]}" for j in cj] == list(hapmap_geno.snp_allele)
    assert np.array_equal(np.array([float(bim[j][3]) for j in cj]), hapmap_geno.snp_position)


def test_oracle_conv_bed_modes_agree(oracle, tmp_path):
    rng = np.random.default_rng(5)
    g = rng.integers(0, 4, size=(77, 131)).astype(np.int32)
    g[g == 3] = -2147483648
    flag = rng.random(77) < 0.6
    out = []
    for mode in (0, 1):
        p = write_bed(str(tmp_path / f"m{mode}.bed"), g, mode)
        out.append(oracle.conv_bed(open(p, "rb").read(), 131, 77, flag))
        assert np.array_equal(out[-1].T, g[flag])
    assert np.array_equal(out[0], out[1])
    with pytest.raises(ValueError):
        oracle.conv_bed(b"\x6c\x1c\x01abcd", 4, 4, np.ones(4))
    with pytest.raises(ValueError):
        oracle.conv_bed(open(p, "rb").read()[:-1], 131, 77, flag)


def test_loci_info_and_snp_selection():
    from hibag_amd import bed
    info = bed.hlaLociInfo("hg19")
    assert info["A"] == (6, 29910247, 29913661) and info["DRB3"] == (6, None, None)
    assert bed.hlaLociInfo("unknown") is None
    with pytest.raises(ValueError):
        bed.hlaLociInfo("hg17")
    bim = bed._read_table(BIM, 6)
    chrom = [r[0] for r in bim]
    pos = np.array([float(r[3]) for r in bim])
    # xMHC = [min gene start - 1Mb, max gene end + 1Mb] over the chr-6 genes inside MHC +- 1Mb
    # (R/DataUtilities.R:650-670): hg19 -> DPB2/DPA3 end 33,099,120 ... HLA-F start 29,691,117
    f = bed._snp_selection("hg19", "xMHC", chrom, pos, False)
    inside = (pos >= 29691117 - 1000000) & (pos <= 33099120 + 1000000)
    assert np.array_equal(f, inside)
    assert bed._snp_selection("hg19", "", chrom, pos, False).all()
    assert bed._snp_selection("hg19", ["6"], chrom, pos, False).sum() == (pos > 0).sum()
    with pytest.raises(ValueError, match="no SNP imported"):
        bed._snp_selection("hg19", ["7"], chrom, pos, False)


def test_bed_flag_is_host_only_and_checks_the_prefix(tmp_path):
    """HIBAG_BEDFlag (src/HIBAG.cpp:1068-1081) needs no device."""
    from hibag_amd import _lib
    L = _lib.lib()
    assert L.hibag_hip_bed_flag(BED.encode()) == 0
    p = write_bed(str(tmp_path / "s.bed"), np.zeros((3, 5), np.int32), 1)
    assert L.hibag_hip_bed_flag(p.encode()) == 1
    bad = tmp_path / "bad.bed"
    bad.write_bytes(b"\x6c\x1a\x01\x00")
    assert L.hibag_hip_bed_flag(str(bad).encode()) == -1
    assert L.hibag_hip_last_error() == b"Invalid prefix in the PLINK BED file."
    assert L.hibag_hip_bed_flag(str(tmp_path / "none.bed").encode()) == -1
    assert L.hibag_hip_last_error().startswith(b"Cannot open the file")
