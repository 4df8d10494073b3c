"""``hlaPredMerge`` (``R/HIBAG.R:825-1023``): combine the posterior matrices of several
``hlaPredict(..., type="response+prob")`` results (e.g. models trained for different SNP
arrays or ancestries) into one call per sample.  Host-side post-processing behind the hot
path: O(#pairs x #samples) adds, done with the reference's operation order
(``HIBAG_SumList`` / ``HIBAG_UpdateAddProbW`` / ``HIBAG_NormalizeProb``,
``src/HIBAG.cpp:1455-1547``)."""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from .hibag import HlaAlleleClass, _pair_names
from .train import hlaUniqueAllele

_RESOLUTION = {"2-digit": 1, "1-field": 1, "4-digit": 2, "2-field": 2, "6-digit": 3, "3-field": 3,
               "8-digit": 4, "4-field": 4, "allele": 1, "protein": 2}


def hlaAlleleDigit(alleles: Sequence[Optional[str]], max_resolution: str = "", rm_suffix: bool = False) -> List[Optional[str]]:
    """``hlaAlleleDigit`` for a character vector (``R/DataUtilities.R:1078-1116``)."""
    if max_resolution in ("full", "none", ""):
        return list(alleles)
    if max_resolution not in _RESOLUTION:
        raise ValueError("'max.resolution' should be one of " + ", ".join(f"'{k}'" for k in _RESOLUTION) + ", 'full', 'none', ''.")
    n = _RESOLUTION[max_resolution]
    out = []
    for a in alleles:
        if a is None:
            out.append(None)
            continue
        f = a.split(":")[:n]
        if rm_suffix:
            f[-1] = f[-1].rstrip("".join(c for c in set(f[-1]) if not c.isdigit()))
        out.append(":".join(f))
    return out


def hlaPredMerge(*pdlist: HlaAlleleClass, weight: Optional[Sequence[float]] = None,
                 equivalence: Optional[Dict[str, str]] = None, use_matching: bool = True, ret_dosage: bool = True,
                 ret_postprob: bool = False, max_resolution: str = "", rm_suffix: bool = False,
                 verbose: bool = True) -> HlaAlleleClass:
    """``equivalence`` maps an existing allele name to its replacement (the reference takes a
    two-column data frame: new name, old name)."""
    if not pdlist:
        raise ValueError("No hlaAlleleClass object passed to 'hlaPredMerge()'.")
    for pd in pdlist:
        if not isinstance(pd, HlaAlleleClass):
            raise TypeError("The object(s) passed to 'hlaPredMerge()' should be 'hlaAlleleClass'.")
        if pd.postprob is None:
            raise ValueError("The object(s) passed to 'hlaPredMerge()' should have a field of 'postprob' returned from "
                             "'hlaPredict(..., type=\"response+prob\")'.")
    samp_id, locus = list(pdlist[0].sample_id), pdlist[0].locus
    for pd in pdlist:
        if list(pd.sample_id) != samp_id:
            raise ValueError("The sample IDs should be the same.")
        if pd.locus != locus:
            raise ValueError("The locus should be the same.")
    k = len(pdlist)
    if weight is not None:
        w = np.asarray(weight, np.float64)
        if w.shape != (k,):
            raise ValueError("Invalid 'weight'.")
        if np.isnan(w).any():
            raise ValueError("'weight' should not have NA/NaN.")
        if (w < 0).any():
            raise ValueError("'weight' should not have a negative value.")
        w = w / w.sum()
    else:
        w = np.full(k, 1.0 / k)
    if use_matching and any(pd.matching is None for pd in pdlist):
        raise ValueError("The column 'matching' should be provided when use.matching=TRUE.")
    use_resolution = max_resolution != "" or rm_suffix

    def replace(alleles: List[str]) -> List[str]:
        if equivalence:
            alleles = [equivalence.get(a, a) for a in alleles]
        if use_resolution:
            alleles = hlaAlleleDigit(alleles, max_resolution, rm_suffix)
        return alleles

    if verbose:
        print(f"Aggregate {k} set{'s' if k > 1 else ''} of predictions:")
    pair_lists = []
    merged: List[str] = []
    for i, pd in enumerate(pdlist):
        pairs = [p.split("/") for p in pd.pair_names]
        pair_lists.append(pairs)
        h = list(dict.fromkeys(a for p in pairs for a in p))
        nh = replace(h)
        merged.extend(nh)
        if verbose:
            print(f"    {i + 1}. # of unique alleles: {len(h)}" + (f" ==>  {len(set(nh))}" if equivalence else ""))
    hla_allele = hlaUniqueAllele(merged)
    n_hla, n_samp = len(hla_allele), len(samp_id)
    if verbose:
        print(f"# of unique allele in the merged set = {n_hla}")
    names = _pair_names(hla_allele)
    row = {nm: j for j, nm in enumerate(names)}
    prob = np.zeros((len(names), n_samp), np.float64)

    matching = None
    if all(pd.matching is not None for pd in pdlist):
        matching = np.zeros(n_samp, np.float64)                        # HIBAG_SumList
        for wi, pd in zip(w, pdlist):
            matching += wi * np.asarray(pd.matching, np.float64)

    for wi, pd, pairs in zip(w, pdlist, pair_lists):
        flat = replace([a for p in pairs for a in p])
        idx = []
        for h1, h2 in zip(flat[0::2], flat[1::2]):
            j = row.get(f"{h1}/{h2}", row.get(f"{h2}/{h1}"))
            if j is None:
                raise AssertionError("allele pair missing from the merged set")
            idx.append(j)
        w2 = wi * np.asarray(pd.matching, np.float64) if use_matching else np.full(n_samp, wi)
        np.add.at(prob, np.asarray(idx), np.asarray(pd.postprob, np.float64) * w2[None, :])     # HIBAG_UpdateAddProbW

    total = np.zeros(n_samp, np.float64)                               # HIBAG_NormalizeProb: in-order column sums
    for j in range(prob.shape[0]):
        total += prob[j]
    with np.errstate(invalid="ignore", divide="ignore"):
        prob /= total[None, :]
    best = np.argmax(np.where(np.isnan(prob), -np.inf, prob), axis=0)  # which.max: first maximum
    pb = prob[best, np.arange(n_samp)]
    a2 = [names[j].split("/")[0] for j in best]                        # H2 = first name, H1 = second (R/HIBAG.R:997-998)
    a1 = [names[j].split("/")[1] for j in best]
    lut = {a: i for i, a in enumerate(hla_allele)}
    rv = HlaAlleleClass(locus=locus, sample_id=samp_id, allele1=a1, allele2=a2, prob=pb, matching=matching,
                        assembly=pdlist[0].assembly or "auto",
                        h1=np.array([lut[a] for a in a1], np.int32), h2=np.array([lut[a] for a in a2], np.int32))
    if ret_dosage:
        ds = np.zeros((n_hla, n_samp), np.float64)
        first = np.array([lut[nm.split("/")[0]] for nm in names])
        second = np.array([lut[nm.split("/")[1]] for nm in names])
        for i in range(n_hla):
            ds[i] = prob[first == i].sum(axis=0) + prob[second == i].sum(axis=0)
        rv.dosage = ds
    if ret_postprob:
        rv.postprob = prob
        rv.pair_names = names
    return rv
