"""Seeded synthetic models and cohorts for the benchmark and the parity tests.

The reference publishes no HLA-B / HLA-DRB1 model with the package, so the
benchmark configurations of BASELINE.json are generated: an ``hlaAttrBagObj``
with the named shape (alleles x classifiers x haplotypes per classifier) whose
haplotypes come in families around one founder per allele, and samples DRAWN
FROM THE MODEL (two founders + genotyping error + missingness).  Uniformly
random genotypes would be useless: every haplotype pair is then dozens of
mismatches away and all posteriors underflow.
"""

from __future__ import annotations

from typing import Tuple

import numpy as np

from .model import NA_INTEGER, Classifier, HlaAttrBagObj, HlaSNPGeno

DEFAULT_SEED = 20260515

# named shapes of BASELINE.json's configs
SHAPES = {
    # cfg2/cfg3: "Pre-fit HLA-B (European, ~100 classifiers, ~150 SNPs)"
    "hla-b": dict(n_hla=50, n_classifier=100, n_snp=150, n_haplo=100, snps=(15, 30)),
    # cfg4: "HLA-DRB1 4-digit model (large allele set, ~500 haplotypes/classifier)"
    "hla-drb1": dict(n_hla=60, n_classifier=100, n_snp=200, n_haplo=500, snps=(18, 32)),
    # cfg1-like small shape for quick tests
    "hla-a-small": dict(n_hla=14, n_classifier=20, n_snp=80, n_haplo=40, snps=(10, 24)),
}


def _allele_counts(rng, n_hla: int, n_haplo: int) -> np.ndarray:
    """Zipf-like split of n_haplo haplotypes over alleles: a few alleles own many, some own none."""
    w = 1.0 / np.arange(1, n_hla + 1) ** 0.9
    w = rng.permutation(w)
    cnt = rng.multinomial(n_haplo, w / w.sum())
    return cnt.astype(np.int64)


def make_model(shape: str = "hla-b", seed: int = DEFAULT_SEED, wide_classifier: bool = True, **over
               ) -> Tuple[HlaAttrBagObj, np.ndarray, np.ndarray]:
    """Returns ``(model, founders[n_hla, n_snp] uint8, allele_freq[n_hla])``."""
    p = dict(SHAPES[shape])
    p.update(over)
    n_hla, C, S, H = p["n_hla"], p["n_classifier"], p["n_snp"], p["n_haplo"]
    lo, hi = p["snps"]
    rng = np.random.default_rng(seed)
    snp_af = rng.uniform(0.08, 0.92, S)
    founders = (rng.random((n_hla, S)) < snp_af).astype(np.uint8)
    allele_freq = rng.dirichlet(np.full(n_hla, 0.6))

    classifiers = []
    for c in range(C):
        k = int(rng.integers(lo, hi + 1))
        if wide_classifier and c == C // 2:
            k = min(100, S)                      # exercises the multi-word bit planes
        if p.get("snp_counts") is not None:      # explicit SNP count per classifier (tests)
            k = int(p["snp_counts"][c])
        snpidx = rng.choice(S, size=k, replace=False).astype(np.int32)
        cnt = _allele_counts(rng, n_hla, H)
        hla = np.repeat(np.arange(n_hla), cnt).astype(np.int32)
        rows = []
        for a in range(n_hla):
            base = founders[a, snpidx]
            for j in range(int(cnt[a])):
                h = base.copy()
                if j > 0:                        # siblings: 1-3 flipped sites
                    flips = rng.choice(k, size=min(k, int(rng.integers(1, 4))), replace=False)
                    h[flips] ^= 1
                rows.append(h)
        freq = rng.gamma(0.5, 1.0, len(rows)) + 1e-5
        freq = freq / freq.sum()
        haplo = ["".join("1" if b else "0" for b in r) for r in rows]
        classifiers.append(Classifier(snpidx=snpidx, freq=freq, hla=hla, haplo=haplo,
                                      samp_num=None, outofbag_acc=0.0))
    alleles = [f"{a // 4 + 1:02d}:{a % 4 + 1:02d}" for a in range(n_hla)]
    model = HlaAttrBagObj(
        n_samp=0, n_snp=S, hla_allele=alleles, classifiers=classifiers, hla_locus=shape.upper(),
        sample_id=[], snp_id=[f"rs{100000 + i}" for i in range(S)],
        snp_position=np.arange(S, dtype=np.float64) * 1000 + 30_000_000,
        snp_allele=["A/G"] * S, snp_allele_freq=snp_af, hla_freq=allele_freq, assembly="hg19")
    return model, founders, allele_freq


def make_samples(founders: np.ndarray, allele_freq: np.ndarray, n_samp: int, seed: int = DEFAULT_SEED + 1,
                 err: float = 0.002, miss: float = 0.01, heavy_missing_frac: float = 0.01
                 ) -> Tuple[np.ndarray, np.ndarray]:
    """int32 [n_samp, n_snp] genotypes (NA = INT_MIN) and the true allele pairs [n_samp, 2]."""
    rng = np.random.default_rng(seed)
    n_hla, S = founders.shape
    a = rng.choice(n_hla, size=(n_samp, 2), p=allele_freq)
    g = founders[a[:, 0]].astype(np.int32) + founders[a[:, 1]].astype(np.int32)
    e = rng.random((n_samp, S)) < err
    g = np.where(e, (g + rng.integers(1, 3, (n_samp, S))) % 3, g).astype(np.int32)
    m = rng.random((n_samp, S)) < miss
    heavy = rng.random(n_samp) < heavy_missing_frac
    m |= heavy[:, None] & (rng.random((n_samp, S)) < 0.30)
    g[m] = NA_INTEGER
    return np.ascontiguousarray(g), np.sort(a, axis=1).astype(np.int32)


def as_snp_geno(model: HlaAttrBagObj, genomat: np.ndarray, order: str = "C") -> HlaSNPGeno:
    """Wrap a sample-major matrix as an ``hlaSNPGenoClass`` over the model's SNPs: ``genotype`` [n_snp, n_samp] in
    numpy's row-major order (``"C"``, a transposed copy) or in R's column-major order (``"F"``: a view of ``genomat``)."""
    n = genomat.shape[0]
    return HlaSNPGeno(genotype=(np.ascontiguousarray(genomat.T) if order == "C" else np.ascontiguousarray(genomat).T), sample_id=[f"S{i + 1}" for i in range(n)],
                      snp_id=list(model.snp_id), snp_position=model.snp_position,
                      snp_allele=list(model.snp_allele), assembly=model.assembly)


def write_bed(path: str, genomat: np.ndarray) -> str:
    """The sample-major matrix ``genomat`` [n_samp, n_snp] as a SNP-major PLINK BED file (two bits per genotype: 2 -> 00,
    missing -> 01, 1 -> 10, 0 -> 11, four samples per byte from the low bits up: the format ``HIBAG_ConvBED`` reads,
    ``src/HIBAG.cpp:1094-1191``)."""
    g = np.asarray(genomat).T
    code = np.full(g.shape, 1, np.uint8)
    code[g == 2] = 0
    code[g == 1] = 2
    code[g == 0] = 3
    pad = np.zeros((g.shape[0], (g.shape[1] + 3) // 4 * 4), np.uint8)
    pad[:, :g.shape[1]] = code
    q = pad.reshape(g.shape[0], -1, 4)
    with open(path, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 1]))
        f.write((q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8).tobytes())
    return path


def as_bed_geno(model: HlaAttrBagObj, genomat: np.ndarray, path: str):
    """The same cohort as :func:`as_snp_geno`, kept in a PLINK BED file: what ``hlaBED2Geno(..., lazy=True)`` returns."""
    from .bed import HlaBEDGeno
    n, s = genomat.shape
    write_bed(path, genomat)
    return HlaBEDGeno(bed_fn=path, mode=1, n_bed_samp=n, n_bed_snp=s, bed_index=np.arange(s, dtype=np.int64),
                      sample_id=[f"S{i + 1}" for i in range(n)], snp_id=list(model.snp_id), snp_position=model.snp_position,
                      snp_allele=list(model.snp_allele), assembly=model.assembly)
