"""SNP matching and strand / allele alignment in front of the kernel.

The step between a user's genotype object and the integer matrix the hot path
consumes (reference: ``hlaPredict`` ``R/HIBAG.R:533-686``, ``hlaSNPID`` and
``hlaGenoSwitchStrand`` ``R/DataUtilities.R:415-524``, ``HIBAG_AlleleStrand``
``src/HIBAG.cpp:221-342``).  Pure host logic, O(#SNPs); its product is the
row selection and the set of SNPs whose genotypes are flipped to ``2 - g``.
"""

from __future__ import annotations

import sys
import warnings
from dataclasses import dataclass
from functools import lru_cache
from itertools import repeat
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .model import NA_INTEGER, HlaAttrBagObj, HlaSNPGeno

MATCH_TYPES = ("Position", "Pos+Allele", "RefSNP+Position", "RefSNP")
_COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A"}


def _fmt_pos(p) -> str:
    p = float(p)
    return str(int(p)) if p.is_integer() else repr(p)


def hlaSNPID(obj, type: str = "Position") -> List:
    """``hlaSNPID`` (``R/DataUtilities.R:512-524``)."""
    if type not in MATCH_TYPES:
        raise ValueError("'arg' should be one of " + ", ".join(f'"{t}"' for t in MATCH_TYPES))
    pos = [] if obj.snp_position is None else [float(p) for p in obj.snp_position]
    if type == "Position":
        return pos
    if type == "Pos+Allele":
        return [f"{_fmt_pos(p)}-{a}" for p, a in zip(pos, obj.snp_allele)]
    if type == "RefSNP+Position":
        return [f"{i}-{_fmt_pos(p)}" for i, p in zip(obj.snp_id, pos)]
    return list(obj.snp_id)


def _ids(obj, type: str):
    """:func:`hlaSNPID` for comparisons: positions stay one float64 array (no per-SNP Python objects)."""
    if type == "Position" and obj.snp_position is not None:
        return np.asarray(obj.snp_position, np.float64)
    return hlaSNPID(obj, type)


def _split_allele(txt: Optional[str]) -> Tuple[str, str]:
    txt = "" if txt is None else txt
    a, sep, b = txt.partition("/")
    return a.upper(), (b.upper() if sep else "")


def _is_base(s: str) -> bool:
    return s in _COMPLEMENT


@lru_cache(maxsize=65536)
def _strand_rule(template: Optional[str], target: Optional[str], check_strand: bool) -> int:
    """The decision of ``HIBAG_AlleleStrand`` (``src/HIBAG.cpp:221-342``) for ONE SNP, which depends on the two allele strings
    only: bit 0 = swap A/B, bits 1-2 = 0 decided / 1 strand ambiguity / 2 allele mismatch (both: compare which allele is the
    minor one instead), bit 3 = counted as a swapped strand.  Cached: a cohort has a handful of distinct allele pairs."""
    s1, s2 = _split_allele(template)
    p1, p2 = _split_allele(target)
    by_freq = 0      # 1: strand ambiguity, 2: allele mismatch
    sw = swapped = False
    if all(_is_base(x) for x in (s1, s2, p1, p2)):
        if s1 == p1 and s2 == p2:
            if check_strand and s1 == _COMPLEMENT[p2]:
                by_freq = 1
        elif s1 == p2 and s2 == p1:
            if check_strand and s1 == _COMPLEMENT[p1]:
                by_freq = 1
            else:
                sw = True
        elif check_strand:
            if s1 == _COMPLEMENT[p1] and s2 == _COMPLEMENT[p2]:
                if s1 == p2:
                    by_freq = 1
                else:
                    swapped = True
            elif s1 == _COMPLEMENT[p2] and s2 == _COMPLEMENT[p1]:
                sw = True
                swapped = True
            else:
                by_freq = 2
        else:
            by_freq = 2
    else:
        if s1 == p1 and s2 == p2:
            if s1 == s2:
                by_freq = 1
        elif s1 == p2 and s2 == p1:
            if s1 == s2:
                by_freq = 1
            else:
                sw = True
        else:
            by_freq = 2
    return int(sw) | (by_freq << 1) | (int(swapped) << 3)


def allele_strand_flags(template_allele: Sequence[str], template_afreq, target_allele: Sequence[str], target_afreq,
                        same_strand: bool):
    """Decide per SNP whether the target's A/B alleles must be swapped to agree
    with the template (``HIBAG_AlleleStrand``, ``src/HIBAG.cpp:221-342``).

    Returns ``(flip[bool n], n_ambiguous, n_mismatch, n_swapped_strand)``.
    Ambiguous (e.g. C/G) and mismatching SNPs fall back to comparing which
    allele is the minor one.  ``template_afreq`` / ``target_afreq``: a sequence of A-allele frequencies, or a callable
    ``f(indices) -> frequencies`` that is asked for those SNPs only (the reference computes every row mean up front,
    ``R/DataUtilities.R:455-460``; the decision reads them for the undecided SNPs alone, so the result is the same)."""
    n = len(template_allele)
    check_strand = not same_strand
    code = np.fromiter(map(_strand_rule, template_allele, target_allele, repeat(check_strand)), np.uint8, n)
    flip = (code & 1).astype(bool)
    by_freq = (code >> 1) & 3
    und = np.flatnonzero(by_freq)
    if len(und):
        def freq_of(src):
            v = src(und) if callable(src) else np.asarray(src, np.float64)[und]
            return np.asarray(v, np.float64)
        # ALLELE_MINOR(f) = (f <= 0.5) ? 0 : 1 ; NaN compares false -> 1
        with np.errstate(invalid="ignore"):
            flip[und] = (freq_of(template_afreq) <= 0.5) != (freq_of(target_afreq) <= 0.5)
    return flip, int(np.count_nonzero(by_freq == 1)), int(np.count_nonzero(by_freq == 2)), int(np.count_nonzero(code & 8))


def _row_afreq(geno: np.ndarray) -> np.ndarray:
    """``rowMeans(genotype, na.rm=TRUE) * 0.5`` with NA = anything outside 0..2 stored as INT_MIN."""
    geno = np.asarray(geno)
    ok = geno != NA_INTEGER
    cnt = ok.sum(axis=1)
    tot = np.where(ok, geno, 0).sum(axis=1, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(cnt > 0, tot / cnt, np.nan) * 0.5


def _same_ids(a, b) -> bool:
    """``length(a) == length(b) && all(a == b)`` (``R/HIBAG.R:627-632``)."""
    if len(a) != len(b):
        return False
    if isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
        return bool(np.array_equal(a, b))
    return list(a) == list(b)


def hlaGenoSwitchStrand(target: HlaSNPGeno, template, match_type: str = "Position",
                        same_strand: bool = False, verbose: bool = True) -> HlaSNPGeno:
    """``hlaGenoSwitchStrand`` (``R/DataUtilities.R:415-505``): subset ``target`` to the
    SNPs it shares with ``template`` (a model or a genotype object) and flip
    genotypes to the template's allele orientation."""
    s1 = hlaSNPID(template, match_type)
    s2 = hlaSNPID(target, match_type)
    if _same_ids(s1, s2):
        I1 = I2 = list(range(len(s1)))
    else:
        first2 = {}
        for j, v in enumerate(s2):
            first2.setdefault(v, j)
        seen, I1, I2 = set(), [], []
        for i, v in enumerate(s1):
            if v in first2 and v not in seen:
                seen.add(v); I1.append(i); I2.append(first2[v])
        if not I1:
            raise ValueError("There is no common SNP.")
    out = sys.stdout
    if match_type != "Pos+Allele":
        a1, a2 = np.asarray(I1, np.int64), np.asarray(I2, np.int64)

        def m_af(idx):                     # (row means only of the SNPs the allele strings leave undecided)
            af = getattr(template, "snp_allele_freq", None)
            return _row_afreq(np.asarray(template.genotype)[a1[idx]]) if af is None else np.asarray(af, np.float64)[a1[idx]]
        flip, n_amb, n_mis, n_swap = allele_strand_flags(
            [template.snp_allele[i] for i in I1], m_af,
            [target.snp_allele[i] for i in I2], lambda idx: _row_afreq(np.asarray(target.genotype)[a2[idx]]), same_strand)
        if verbose:
            x = int(flip.sum())
            print(f"# of SNP loci with flipped alleles: {x}" if x > 0
                  else "No allelic strand or A/B allele is flipped.", file=out)
            if n_swap > 0:
                print(f"# of SNP loci with swapped strands: {n_swap}", file=out)
            if n_amb > 0:
                print(f"# of SNP loci with strand ambiguity (e.g., C/G): {n_amb} (comparing allele frequencies)", file=out)
            if n_mis > 0:
                print(f"# of SNP loci with mismatched alleles: {n_mis} (comparing allele frequencies)", file=out)
    else:
        if verbose:
            print("No allele is flipped since match.type='Pos+Allele'.", file=out)
        flip = np.zeros(len(I1), bool)
    geno = target.genotype[I2].copy()
    rows = np.where(flip)[0]
    if len(rows):
        sub = geno[rows]
        geno[rows] = np.where(sub == NA_INTEGER, NA_INTEGER, 2 - sub)
    return HlaSNPGeno(genotype=geno, sample_id=list(target.sample_id), snp_id=[target.snp_id[i] for i in I2],
                      snp_position=None if target.snp_position is None else np.asarray(target.snp_position)[I2],
                      snp_allele=[template.snp_allele[i] for i in I1],
                      assembly=getattr(template, "assembly", "unknown"))


@dataclass
class SNPPlan:
    """What the SNP-matching step of ``hlaPredict`` decides: for every model SNP
    the row of the user's data that supplies it (-1 = absent -> all missing) and
    whether its allele count is reversed (``g -> 2 - g``)."""
    sel: np.ndarray          # int64 [n.snp]
    flip: np.ndarray         # bool  [n.snp]
    assembly: str
    identity: bool = False   # the cohort's SNPs ARE the model's, in the model's order (sel = 0 .. n.snp - 1)


def plan_snps_for_predict(obj: HlaAttrBagObj, snp, afreq_of_rows: Callable[[np.ndarray], np.ndarray],
                          match_type: str, allele_check: bool, same_strand: bool, verbose: bool,
                          verbose_match: bool) -> SNPPlan:
    """The ``hlaSNPGenoClass`` branch of ``hlaPredict`` (``R/HIBAG.R:550-686``) up to, but
    not including, the construction of the genotype matrix.  ``snp`` only needs the
    annotation (``snp_id``, ``snp_position``, ``snp_allele``, ``assembly``);
    ``afreq_of_rows(rows)`` returns the A-allele frequency of the given data rows
    (``rowMeans(genotype, na.rm=TRUE) * 0.5``), which the strand check consults for
    ambiguous SNPs (``hlaGenoSwitchStrand``, ``R/DataUtilities.R:415-505``)."""
    if match_type not in MATCH_TYPES:
        raise ValueError("'arg' should be one of " + ", ".join(f'"{t}"' for t in MATCH_TYPES))
    out = sys.stdout
    model_asm = obj.assembly or "unknown"
    geno_asm = snp.assembly or "unknown"
    ref = f"Model assembly: {model_asm}, SNP assembly: {geno_asm}"
    if verbose:
        print(ref, file=out)
    if model_asm != geno_asm:
        if "unknown" in (model_asm, geno_asm):
            if verbose:
                print("The human genome references might not match!", file=sys.stderr)
            assembly = model_asm if geno_asm == "unknown" else geno_asm
        else:
            warnings.warn(f"The human genome references do not match! {ref}.")
            assembly = model_asm
    else:
        assembly = model_asm if model_asm != "unknown" else "auto"

    if verbose and verbose_match:
        print("Matching the SNPs between the model and the test data:", file=out)
        for tp in MATCH_TYPES:
            a, b = hlaSNPID(obj, tp), set(hlaSNPID(snp, tp))
            miss = len(a) - len(set(a) & b)
            print(f"   {tp:<16} missing SNPs # {miss} ({100.0 * miss / max(len(a), 1):.1f}%)"
                  f"{'  *being used' if tp == match_type else ''}", file=out)
    elif verbose:
        print(f"Using match.type='{match_type}' for SNP matching", file=out)

    obj_id = _ids(obj, match_type)
    geno_id = _ids(snp, match_type)
    identity = _same_ids(obj_id, geno_id)
    if identity:
        sel = np.arange(len(obj_id), dtype=np.int64)
        alleles = snp.snp_allele
    else:
        if isinstance(obj_id, np.ndarray):
            obj_id, geno_id = obj_id.tolist(), geno_id.tolist()
        first = {}
        for j, v in enumerate(geno_id):
            first.setdefault(v, j)
        picked, used = [], set()
        for v in obj_id:                       # match(); duplicated selections -> NA
            j = first.get(v)
            if j is not None and j in used:
                j = None
            if j is not None:
                used.add(j)
            picked.append(j)
        n_missing = sum(1 for j in picked if j is None)
        if n_missing == len(obj_id):
            raise ValueError("There is no overlapping of SNPs!")
        if n_missing > 0.5 * len(obj_id):
            warnings.warn("More than 50% of SNPs are missing!")
        sel = np.array([-1 if j is None else j for j in picked], np.int64)
        alleles = []
        for i, j in enumerate(picked):
            if j is None:                          # unmatched: takes the model's alleles (R/HIBAG.R:641-646)
                alleles.append(obj.snp_allele[i])
            else:                                  # matched with an NA allele string: "" -> "mismatched alleles" rule
                a = snp.snp_allele[j] if j < len(snp.snp_allele) else None
                alleles.append("" if a is None else a)

    flip = np.zeros(len(obj_id), bool)
    if allele_check:
        if match_type != "Pos+Allele":
            def cohort_afreq(idx):             # asked for the SNPs the allele strings leave undecided only
                af = np.full(len(idx), np.nan)
                have = sel[idx] >= 0
                if have.any():
                    af[have] = afreq_of_rows(sel[idx][have])
                return af
            flip, n_amb, n_mis, n_swap = allele_strand_flags(obj.snp_allele, lambda idx: np.asarray(_model_afreq(obj), np.float64)[idx],
                                                             alleles, cohort_afreq, same_strand)
            if verbose:
                x = int(flip.sum())
                print(f"# of SNP loci with flipped alleles: {x}" if x > 0
                      else "No allelic strand or A/B allele is flipped.", file=out)
                if n_swap > 0:
                    print(f"# of SNP loci with swapped strands: {n_swap}", file=out)
                if n_amb > 0:
                    print(f"# of SNP loci with strand ambiguity (e.g., C/G): {n_amb} (comparing allele frequencies)", file=out)
                if n_mis > 0:
                    print(f"# of SNP loci with mismatched alleles: {n_mis} (comparing allele frequencies)", file=out)
        elif verbose:
            print("No allele is flipped since match.type='Pos+Allele'.", file=out)
    return SNPPlan(sel=sel, flip=np.asarray(flip, bool), assembly=assembly, identity=identity)


def _model_afreq(obj):
    af = getattr(obj, "snp_allele_freq", None)
    return _row_afreq(obj.genotype) if af is None else af


def match_snps_for_predict(obj: HlaAttrBagObj, snp: HlaSNPGeno, match_type: str, allele_check: bool,
                           same_strand: bool, verbose: bool, verbose_match: bool):
    """:func:`plan_snps_for_predict` applied to an in-memory genotype object: returns
    ``(genotype matrix [n.snp, n.samp] in model SNP order, assembly)``; model SNPs
    absent from the data become all-missing rows (``R/HIBAG.R:640-660``)."""
    plan = plan_snps_for_predict(obj, snp, lambda rows: _row_afreq(snp.genotype[rows]), match_type,
                                 allele_check, same_strand, verbose, verbose_match)
    n_samp = snp.genotype.shape[1]
    g = np.full((len(plan.sel), n_samp), NA_INTEGER, np.int32)
    have = plan.sel >= 0
    g[have] = snp.genotype[plan.sel[have]]
    rows = np.where(plan.flip & have)[0]
    if len(rows):
        sub = g[rows]
        g[rows] = np.where(sub == NA_INTEGER, NA_INTEGER, 2 - sub)
    return g, plan.assembly
