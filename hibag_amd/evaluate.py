"""The R-level helpers the reference's own test script is written with (``tests/runTests.R``):
``hlaSplitAllele`` (``R/DataUtilities.R:1688-1726``), ``hlaFlankingSNP`` (``:1732-1780``),
``hlaGenoSubset`` / ``hlaAlleleSubset`` (``:304-353``, ``:1249-1280``) and the ``overall`` block of
``hlaCompareAllele`` (``:1330-1560``).  Host-side bookkeeping around ``hlaAttrBagging`` and
``hlaPredict``; random draws follow R's ``sample()`` so that ``set_seed(100)`` gives R's split."""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

from .bed import hlaLociInfo
from .hibag import HlaAlleleClass
from .model import HlaSNPGeno
from .train import RRandom, _R, hlaUniqueAllele


def _unif_index(rng: RRandom, dn: int, sample_kind: str) -> int:
    """``R_unif_index`` (R ``src/main/RNG.c``): "Rounding" is R < 3.6.0's ``floor(dn * unif_rand())``,
    "Rejection" the current default (16-bit chunks, rejection above dn)."""
    if sample_kind == "Rounding":
        return int(math.floor(dn * rng.unif_rand()))
    if dn <= 0:
        return 0
    bits = int(math.ceil(math.log2(dn)))
    while True:
        v = 0
        for _ in range(0, bits + 1, 16):
            v = 65536 * v + int(math.floor(rng.unif_rand() * 65536))
        v &= (1 << bits) - 1
        if v < dn:
            return v


def r_sample(x: Sequence, size: int, rng: Optional[RRandom] = None, sample_kind: str = "Rejection") -> List:
    """``sample(x, size)`` without replacement (``do_sample``, R ``src/main/unique.c`` / ``random.c``)."""
    rng = _R if rng is None else rng
    n = len(x)
    idx = list(range(n))
    out = []
    for _ in range(size):
        j = _unif_index(rng, n, sample_kind)
        out.append(x[idx[j]])
        n -= 1
        idx[j] = idx[n]
    return out


def hlaAlleleSubset(hla: HlaAlleleClass, samp_sel: Sequence[int]) -> HlaAlleleClass:
    ix = list(samp_sel)
    pick = lambda v: None if v is None else np.asarray(v)[ix]      # noqa: E731
    return HlaAlleleClass(locus=hla.locus, sample_id=[hla.sample_id[i] for i in ix], allele1=[hla.allele1[i] for i in ix],
                          allele2=[hla.allele2[i] for i in ix], prob=pick(hla.prob), matching=pick(hla.matching),
                          assembly=hla.assembly)


def hlaSplitAllele(HLA: HlaAlleleClass, train_prop: float = 0.5, rng: Optional[RRandom] = None,
                   sample_kind: str = "Rejection") -> Dict[str, HlaAlleleClass]:
    """``hlaSplitAllele``: repeatedly take the rarest allele, send ``ceiling(n * train.prop)`` of its carriers
    to the training set, drop all its carriers, until no sample is left."""
    ids = [str(s) for s in HLA.sample_id]
    left = list(range(len(ids)))
    train: List[str] = []
    while left:
        a = [HLA.allele1[i] for i in left] + [HLA.allele2[i] for i in left]
        hua = hlaUniqueAllele(a)
        count = {h: 0 for h in hua}
        for v in a:
            if v is not None:
                count[v] += 1
        allele = min(hua, key=lambda h: count[h])          # order(count)[1]: stable, first minimum
        carriers = [i for i in left if HLA.allele1[i] == allele or HLA.allele2[i] == allele]
        n_train = int(math.ceil(len(carriers) * train_prop))
        train.extend(r_sample([ids[i] for i in carriers], n_train, rng, sample_kind))
        gone = set(carriers)
        left = [i for i in left if i not in gone]
    train.sort()                                             # train.set[order(train.set)] (C-locale order for these ids)
    pos = {s: i for i, s in enumerate(ids)}
    tset = set(train)
    return {"training": hlaAlleleSubset(HLA, [pos[s] for s in train]),
            "validation": hlaAlleleSubset(HLA, [i for i, s in enumerate(ids) if s not in tset])}


def hlaFlankingSNP(snp_id: Sequence, position: Sequence[float], locus: str, flank_bp: int = 500000,
                   assembly: str = "auto") -> List:
    info = hlaLociInfo(assembly)
    if info is None or locus not in info:
        raise ValueError("'locus' should be one of " + ", ".join(info or []))
    _, start, end = info[locus]
    if start is None or end is None:
        raise ValueError("The position information is not available!")
    pos = np.asarray(position, np.float64)
    keep = (start - flank_bp <= pos) & (pos <= end + flank_bp)
    return [s for s, k in zip(snp_id, keep) if k]


def hlaGenoSubset(geno: HlaSNPGeno, samp_sel: Optional[Sequence[int]] = None, snp_sel: Optional[Sequence[int]] = None) -> HlaSNPGeno:
    si = list(range(len(geno.sample_id))) if samp_sel is None else list(samp_sel)
    ki = list(range(len(geno.snp_id))) if snp_sel is None else list(snp_sel)
    return HlaSNPGeno(genotype=np.ascontiguousarray(np.asarray(geno.genotype)[np.ix_(ki, si)]),
                      sample_id=[geno.sample_id[i] for i in si], snp_id=[geno.snp_id[i] for i in ki],
                      snp_position=None if geno.snp_position is None else np.asarray(geno.snp_position)[ki],
                      snp_allele=[geno.snp_allele[i] for i in ki], assembly=geno.assembly)


def hlaCompareAllele(TrueHLA: HlaAlleleClass, PredHLA: HlaAlleleClass, allele_limit=None,
                     call_threshold: float = float("nan")) -> Dict[str, float]:
    """The ``overall`` row of ``hlaCompareAllele``: samples common to both objects whose true alleles are
    within ``allele_limit`` (a model or a list of alleles); a call counts if its probability reaches
    ``call_threshold``."""
    pred = {s: i for i, s in enumerate(PredHLA.sample_id)}
    rows = [(i, pred[s]) for i, s in enumerate(TrueHLA.sample_id) if s in pred]
    # R/DataUtilities.R:1376-1380: samples with an NA in the true OR the predicted pair are left out first
    rows = [(i, j) for i, j in rows if TrueHLA.allele1[i] is not None and TrueHLA.allele2[i] is not None
            and PredHLA.allele1[j] is not None and PredHLA.allele2[j] is not None]
    if allele_limit is not None:
        allowed = set(getattr(allele_limit, "hla_allele", allele_limit))
        rows = [(i, j) for i, j in rows if TrueHLA.allele1[i] in allowed and TrueHLA.allele2[i] in allowed]
    n = len(rows)
    cnt_ind = cnt_haplo = cnt_call = 0
    for i, j in rows:
        if math.isfinite(call_threshold) and PredHLA.prob is not None and not (PredHLA.prob[j] >= call_threshold):
            continue
        cnt_call += 1
        s = [TrueHLA.allele1[i], TrueHLA.allele2[i]]
        p = [PredHLA.allele1[j], PredHLA.allele2[j]]
        if (s[0] == p[0] and s[1] == p[1]) or (s[1] == p[0] and s[0] == p[1]):
            cnt_ind += 1
        if s[0] == p[0] or s[0] == p[1]:
            p[0 if s[0] == p[0] else 1] = ""
            cnt_haplo += 1
        if s[1] == p[0] or s[1] == p[1]:
            cnt_haplo += 1
    return {"total.num.ind": n, "crt.num.ind": cnt_ind, "crt.num.haplo": cnt_haplo,
            "acc.ind": cnt_ind / cnt_call if cnt_call else float("nan"),
            "acc.haplo": 0.5 * cnt_haplo / cnt_call if cnt_call else float("nan"),
            "call.threshold": 0.0 if not math.isfinite(call_threshold) else call_threshold,
            "n.call": cnt_call, "call.rate": cnt_call / n if n else float("nan")}
