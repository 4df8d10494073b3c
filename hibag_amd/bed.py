"""PLINK BED input: ``hlaBED2Geno`` and the data it needs.

Reference: ``hlaBED2Geno`` / ``.snp_selection`` / ``.clean_geno``
(``R/DataUtilities.R:610-780``), ``hlaLociInfo`` (``:1051-1070``), the C side
``HIBAG_BEDFlag`` / ``HIBAG_ConvBED`` (``src/HIBAG.cpp:1068-1191``).

Only the annotation (.fam / .bim text, SNP selection) is handled here; the
2-bit genotypes are decoded on the device by ``libhibag_hip.so``
(``hibag_hip_conv_bed``), or -- for ``hlaBED2Geno(..., lazy=True)`` -- not until
``hlaPredict`` hands the file to ``hibag_hip_predict_bed``, which unpacks it
straight into the kernels' packed form.
"""

from __future__ import annotations

import ctypes as C
import math
import os
import sys
import warnings
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .model import NA_INTEGER, HlaSNPGeno

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_ASSEMBLIES = ("auto", "auto-silent", "hg18", "hg19", "hg38", "unknown")


def _hla_assembly(assembly: str = "auto") -> str:
    """``.hla_assembly`` (``R/DataUtilities.R:71-82``)."""
    if assembly not in _ASSEMBLIES:
        raise ValueError("'arg' should be one of " + ", ".join(f'"{a}"' for a in _ASSEMBLIES))
    if assembly in ("auto", "auto-silent"):
        if assembly == "auto":
            print('using the default genome assembly (assembly="hg19")', file=sys.stderr)
        assembly = "hg19"
    return assembly


def hlaLociInfo(assembly: str = "auto") -> Optional[Dict[str, tuple]]:
    """``hlaLociInfo`` (``R/DataUtilities.R:1051-1070``): gene -> (chrom, start, end), ``None`` for NA
    coordinates, in table order (the first row is the MHC itself).  The coordinates are the
    reference's per-assembly gene tables (``inst/doc/GeneInfo_<assembly>.txt``), kept here as
    ``data/hla_loci.json``."""
    assembly = _hla_assembly(assembly)
    global _LOCI
    if _LOCI is None:
        import json
        with open(os.path.join(_DATA, "hla_loci.json")) as f:
            _LOCI = json.load(f)
    if assembly not in _LOCI:
        if assembly != "unknown":
            raise ValueError("Unknown human genome reference in 'assembly'!")
        return None
    return {name: tuple(v) for name, v in _LOCI[assembly].items()}


_LOCI = None


def _plural(n: int) -> str:
    return "s" if n > 1 else ""


def _snp_selection(assembly: str, import_chr: Sequence[str], chrom: Sequence[str], pos: np.ndarray,
                   verbose: bool) -> np.ndarray:
    """``.snp_selection`` (``R/DataUtilities.R:646-701``)."""
    chrom = np.asarray(chrom, dtype=object)
    import_chr = [import_chr] if isinstance(import_chr, str) else list(import_chr)
    rest: Optional[List[str]] = import_chr
    flag = None
    if len(import_chr) == 1:
        if import_chr[0] == "xMHC":
            info = hlaLociInfo(assembly)
            if info is None:
                raise ValueError("Unknown human genome reference in 'assembly'!")
            rows = [(s, e) for (c, s, e) in info.values() if c == 6]
            st, ed = rows[0][0] - 1000000, rows[0][1] + 1000000
            # NA coordinates compare as NA in R and are dropped by which()
            known = [(s, e) for (s, e) in rows if s is not None and e is not None]
            inmhc = [(s, e) for (s, e) in known if st <= s and e <= ed]
            outmhc = [(s, e) for (s, e) in known if not (st <= s and e <= ed)]
            st = min(s for s, _ in inmhc) - 1000000
            ed = max(e for _, e in inmhc) + 1000000
            on6 = chrom == "6"
            flag = on6 & (st <= pos) & (pos <= ed)
            for s, e in outmhc:
                flag |= on6 & (s - 1000000 <= pos) & (pos <= e + 1000000)
            if verbose:
                n = int(flag.sum())
                print(f"Import {n} SNP{_plural(n)} within the xMHC region on chromosome 6")
            rest = None
        elif import_chr[0] == "":
            flag = np.ones(len(pos), bool)
            if verbose:
                print(f"Import {len(pos)} SNP{_plural(len(pos))}")
            rest = None
    if rest is not None:
        want = {str(c) for c in rest}
        flag = np.array([c in want for c in chrom], bool) & (pos > 0)
        if verbose:
            n = int(flag.sum())
            print(f"Import {n} SNP{_plural(n)} from chromosome {','.join(str(c) for c in rest)}")
    if int(flag.sum()) <= 0:
        raise ValueError("There is no SNP imported.")
    return flag


def _read_table(fn: str, ncol: int) -> List[List[str]]:
    rows = []
    with open(fn) as f:
        for ln in f:
            p = ln.split()
            if not p:
                continue
            if len(p) != ncol:
                raise ValueError(f"line {len(rows) + 1} of {fn!r} did not have {ncol} elements")
            rows.append(p)
    return rows


def _lib_bed_flag(bed_fn: str) -> int:
    rc = _lib.lib().hibag_hip_bed_flag(os.fsencode(bed_fn))
    if rc < 0:
        _lib.check(rc)
    return rc


@dataclass
class HlaBEDGeno:
    """A ``hlaSNPGenoClass`` whose genotypes still live in the PLINK BED file
    (``hlaBED2Geno(..., lazy=True)``).  ``bed_index[j]`` is the 0-based position
    in the .bim file of the object's SNP j.  ``hlaPredict`` feeds it to
    ``hibag_hip_predict_bed``; ``load()`` materialises the ordinary object."""
    bed_fn: str
    mode: int
    n_bed_samp: int
    n_bed_snp: int
    bed_index: np.ndarray
    sample_id: List[str]
    snp_id: List[str]
    snp_position: Optional[np.ndarray] = None
    snp_allele: List[str] = field(default_factory=list)
    assembly: str = "unknown"

    def subset_snps(self, keep: np.ndarray) -> "HlaBEDGeno":
        keep = np.asarray(keep, bool)
        ix = np.where(keep)[0]
        return HlaBEDGeno(self.bed_fn, self.mode, self.n_bed_samp, self.n_bed_snp, self.bed_index[ix],
                          list(self.sample_id), [self.snp_id[i] for i in ix],
                          None if self.snp_position is None else np.asarray(self.snp_position)[ix],
                          [self.snp_allele[i] for i in ix], self.assembly)

    def load(self) -> HlaSNPGeno:
        """Decode on the device (``HIBAG_ConvBED``) -> ``hlaSNPGenoClass``."""
        flag = np.zeros(self.n_bed_snp, np.int32)
        flag[self.bed_index] = 1
        if int(flag.sum()) != len(self.bed_index):
            raise ValueError("duplicated SNP selection")
        geno = np.empty((self.n_bed_samp, len(self.bed_index)), np.int32)
        _lib.check(_lib.lib().hibag_hip_conv_bed(
            os.fsencode(self.bed_fn), self.n_bed_samp, self.n_bed_snp, len(self.bed_index),
            flag.ctypes.data_as(C.c_void_p), geno.ctypes.data_as(C.c_void_p)))
        # ConvBED emits the flagged SNPs in file order
        order = np.argsort(np.argsort(self.bed_index))
        g = np.ascontiguousarray(geno.T[order])
        return HlaSNPGeno(genotype=g, sample_id=list(self.sample_id), snp_id=list(self.snp_id),
                          snp_position=self.snp_position, snp_allele=list(self.snp_allele), assembly=self.assembly)

    def allele_freq(self, rows: np.ndarray) -> np.ndarray:
        """A-allele frequency of the object's SNPs ``rows`` -- ``rowMeans(genotype,
        na.rm=TRUE) * 0.5`` -- counted on the packed bytes (annotation-level host work
        the strand check of ``hlaGenoSwitchStrand`` needs, not part of the hot path)."""
        rows = np.asarray(rows, np.int64)
        mm = np.memmap(self.bed_fn, np.uint8, "r", offset=3)
        cols = self.bed_index[rows]
        if self.mode == 0:
            stride = (self.n_bed_snp + 3) // 4
            mat = mm[: stride * self.n_bed_samp].reshape(self.n_bed_samp, stride)
            two = (mat[:, cols >> 2] >> (2 * (cols & 3)).astype(np.uint8)) & 3          # [samp, rows]
            two = two.T
        else:
            stride = (self.n_bed_samp + 3) // 4
            mat = mm[: stride * self.n_bed_snp].reshape(self.n_bed_snp, stride)[cols]   # [rows, stride]
            two = np.stack([(mat >> s) & 3 for s in (0, 2, 4, 6)], axis=2).reshape(len(cols), -1)[:, : self.n_bed_samp]
        g = np.array([2, 0, 1, 0], np.int64)[two]
        ok = two != 1
        cnt = ok.sum(axis=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(cnt > 0, (g * ok).sum(axis=1) / cnt, np.nan) * 0.5


def _clean(v, verbose: bool):
    """``.clean_geno`` (``R/DataUtilities.R:610-644``) for either object kind."""
    def subset(obj, keep):
        if isinstance(obj, HlaBEDGeno):
            return obj.subset_snps(keep)
        ix = np.where(keep)[0]
        return HlaSNPGeno(genotype=obj.genotype[ix], sample_id=list(obj.sample_id),
                          snp_id=[obj.snp_id[i] for i in ix],
                          snp_position=None if obj.snp_position is None else np.asarray(obj.snp_position)[ix],
                          snp_allele=[obj.snp_allele[i] for i in ix], assembly=obj.assembly)
    seen, dup = set(), []
    for s in v.snp_id:
        dup.append(s in seen)
        seen.add(s)
    if any(dup):
        if verbose:
            print(f"{sum(dup)} SNP{_plural(sum(dup))} with duplicated ID have been removed.")
        v = subset(v, ~np.array(dup))
    ok = []
    for a in v.snp_allele:
        p = ("?/?" if a is None else a).split("/")
        ok.append(len(p) == 2 and all(x in ("A", "G", "C", "T") for x in p))
    ok = np.array(ok, bool)
    if (~ok).any():
        if verbose:
            n = int((~ok).sum())
            print(f"{n} SNP{_plural(n)} with invalid alleles have been removed.")
        v = subset(v, ok)
    return v


def hlaBED2Geno(bed_fn: str, fam_fn: str, bim_fn: str, rm_invalid_allele: bool = False,
                import_chr="xMHC", assembly: str = "auto", verbose: bool = True, lazy: bool = False):
    """``hlaBED2Geno`` (``R/DataUtilities.R:703-780``).

    Returns an :class:`HlaSNPGeno` whose matrix was decoded on the device; with
    ``lazy=True`` (extension) an :class:`HlaBEDGeno` that keeps the genotypes in
    the file for ``hlaPredict`` to decode directly into the kernels' layout."""
    for v, n in ((bed_fn, "bed.fn"), (fam_fn, "fam.fn"), (bim_fn, "bim.fn")):
        if not isinstance(v, (str, os.PathLike)):
            raise TypeError(f"is.character({n}) is not TRUE")
    assembly = _hla_assembly(assembly)
    mode = _lib_bed_flag(os.fspath(bed_fn))
    if verbose:
        print(f"Open '{bed_fn}' " + ("(the individual-major mode)" if mode == 0 else "(the SNP-major mode)"))

    fam = _read_table(os.fspath(fam_fn), 6)
    inv = [r[1] for r in fam]
    if len(set(inv)) == len(inv):
        sample_id = inv
    else:
        sample_id = [f"{r[0]}-{r[1]}" for r in fam]
        if len(set(sample_id)) != len(sample_id):
            raise ValueError("IDs in PLINK bed are not unique!")
    if verbose:
        print(f"Open '{fam_fn}'")

    bim = _read_table(os.fspath(bim_fn), 6)
    chrom = ["" if r[0] == "NA" else r[0] for r in bim]

    def to_pos(x: str) -> float:
        try:
            p = float(x)
        except ValueError:
            return 0.0
        return p if math.isfinite(p) else 0.0
    pos = np.array([to_pos(r[3]) for r in bim], np.float64)
    snp_id = [r[1] for r in bim]
    if len(set(snp_id)) != len(snp_id):
        raise ValueError("The SNP IDs in the PLINK binary file should be unique!")
    snp_allele = [f"{r[4]}/{r[5]}" for r in bim]
    if verbose:
        print(f"Open '{bim_fn}'")

    flag = _snp_selection(assembly, import_chr, chrom, pos, verbose)
    ix = np.where(flag)[0]
    geno = HlaBEDGeno(bed_fn=os.fspath(bed_fn), mode=mode, n_bed_samp=len(sample_id), n_bed_snp=len(snp_id),
                      bed_index=ix.astype(np.int64), sample_id=sample_id, snp_id=[snp_id[i] for i in ix],
                      snp_position=pos[ix], snp_allele=[snp_allele[i] for i in ix], assembly=assembly)
    if not lazy:
        geno = geno.load()
    if rm_invalid_allele:
        geno = _clean(geno, verbose)
    if len(set(geno.snp_id)) != len(geno.snp_id):
        warnings.warn("'snp.id' is not unique.")
    return geno
