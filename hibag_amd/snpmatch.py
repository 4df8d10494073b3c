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


def _split_allele(txt: Optional[str]) -> Tuple[str, str]:
    txt = "" if txt is None else txt
    a, sep, b = txt.partition("/")
    return a.upper(), (b.upper() if sep else "")


def _is_base(s: str) -> bool:
    return s in _COMPLEMENT


def allele_strand_flags(template_allele: Sequence[str], template_afreq: Sequence[float],
                        target_allele: Sequence[str], target_afreq: Sequence[float],
                        same_strand: bool):
    """Decide per SNP whether the target's A/B alleles must be swapped to agree
    with the template (``HIBAG_AlleleStrand``, ``src/HIBAG.cpp:221-342``).

    Returns ``(flip[bool n], n_ambiguous, n_mismatch, n_swapped_strand)``.
    Ambiguous (e.g. C/G) and mismatching SNPs fall back to comparing which
    allele is the minor one."""
    n = len(template_allele)
    flip = np.zeros(n, bool)
    n_amb = n_mis = n_swap = 0
    check_strand = not same_strand
    for i in range(n):
        s1, s2 = _split_allele(template_allele[i])
        p1, p2 = _split_allele(target_allele[i])
        by_freq = 0      # 1: strand ambiguity, 2: allele mismatch
        sw = False
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
                        n_swap += 1
                elif s1 == _COMPLEMENT[p2] and s2 == _COMPLEMENT[p1]:
                    sw = True
                    n_swap += 1
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
        if by_freq:
            f1, f2 = template_afreq[i], target_afreq[i]
            # ALLELE_MINOR(f) = (f <= 0.5) ? 0 : 1 ; NaN compares false -> 1
            sw = (0 if f1 <= 0.5 else 1) != (0 if f2 <= 0.5 else 1)
            if by_freq == 1:
                n_amb += 1
            else:
                n_mis += 1
        flip[i] = sw
    return flip, n_amb, n_mis, n_swap


def _row_afreq(geno: np.ndarray) -> np.ndarray:
    """``rowMeans(genotype, na.rm=TRUE) * 0.5`` with NA = anything outside 0..2 stored as INT_MIN."""
    g = geno.astype(np.float64)
    ok = geno != NA_INTEGER
    cnt = ok.sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(cnt > 0, (g * ok).sum(axis=1) / cnt, np.nan) * 0.5


def hlaGenoSwitchStrand(target: HlaSNPGeno, template, match_type: str = "Position",
                        same_strand: bool = False, verbose: bool = True) -> HlaSNPGeno:
    """``hlaGenoSwitchStrand`` (``R/DataUtilities.R:415-505``): subset ``target`` to the
    SNPs it shares with ``template`` (a model or a genotype object) and flip
    genotypes to the template's allele orientation."""
    s1 = hlaSNPID(template, match_type)
    s2 = hlaSNPID(target, match_type)
    if len(s1) == len(s2) and all(a == b for a, b in zip(s1, s2)):
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
        t_af = _row_afreq(target.genotype)
        m_af = getattr(template, "snp_allele_freq", None)
        if m_af is None:
            m_af = _row_afreq(template.genotype)
        flip, n_amb, n_mis, n_swap = allele_strand_flags(
            [template.snp_allele[i] for i in I1], [m_af[i] for i in I1],
            [target.snp_allele[i] for i in I2], [t_af[i] for i in I2], same_strand)
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

    obj_id = hlaSNPID(obj, match_type)
    geno_id = hlaSNPID(snp, match_type)
    if len(obj_id) == len(geno_id) and all(a == b for a, b in zip(obj_id, geno_id)):
        sel = np.arange(len(obj_id), dtype=np.int64)
        alleles = list(snp.snp_allele)
    else:
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
            af = np.full(len(obj_id), np.nan)
            have = sel >= 0
            if have.any():
                af[have] = afreq_of_rows(sel[have])
            flip, n_amb, n_mis, n_swap = allele_strand_flags(obj.snp_allele, _model_afreq(obj), alleles, af, same_strand)
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
    return SNPPlan(sel=sel, flip=np.asarray(flip, bool), assembly=assembly)


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
