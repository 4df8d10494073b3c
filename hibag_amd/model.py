"""In-memory form of HIBAG's model and genotype objects.

``HlaAttrBagObj`` mirrors the R list of class ``hlaAttrBagObj`` that
``hlaModelToObj()`` writes and ``hlaModelFromObj()`` reads (reference:
``R/HIBAG.R:1041-1062`` / ``:1135-1178``; field meaning in
``man/hlaAttrBagObj.Rd:9-38``), so that pre-fit public models load unchanged.
``HlaSNPGeno`` mirrors ``hlaSNPGenoClass`` (``R/DataUtilities.R:236-244``).

Index conventions are the C side's, not R's: ``Classifier.snpidx`` and
``Classifier.hla`` are 0-based here, exactly what ``hlaModelFromObj`` passes to
``HIBAG_NewClassifierHaplo`` (``tree$snpidx - 1L``, ``match(hla, allele) - 1L``,
``R/HIBAG.R:1147-1163``).
"""

from __future__ import annotations

import os

from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence

import numpy as np

from . import rdata

NA_INTEGER = rdata.NA_INTEGER
MAX_SNP_IN_CLASSIFIER = 128  # inst/include/LibHLA_ext.h:223


@dataclass
class Classifier:
    """One individual classifier of the ensemble (``man/hlaAttrBagObj.Rd:25-38``)."""
    snpidx: np.ndarray            # int32, 0-based indices into the model's SNP list
    freq: np.ndarray              # float64 haplotype frequencies
    hla: np.ndarray               # int32 allele index of each haplotype, ascending
    haplo: List[str]              # '0'/'1' strings, char s <-> SNP snpidx[s]
    samp_num: Optional[np.ndarray] = None   # bootstrap counts of the training samples
    outofbag_acc: float = 0.0

    def __post_init__(self):
        self.snpidx = np.ascontiguousarray(self.snpidx, np.int32)
        self.freq = np.ascontiguousarray(self.freq, np.float64)
        self.hla = np.ascontiguousarray(self.hla, np.int32)
        self.haplo = list(self.haplo)
        if not (len(self.freq) == len(self.hla) == len(self.haplo)):
            raise ValueError("haplotype columns 'freq', 'hla', 'haplo' differ in length")


@dataclass
class HlaAttrBagObj:
    n_samp: int
    n_snp: int
    hla_allele: List[str]
    classifiers: List[Classifier]
    hla_locus: str = ""
    sample_id: List[str] = field(default_factory=list)
    snp_id: List[str] = field(default_factory=list)
    snp_position: Optional[np.ndarray] = None
    snp_allele: List[str] = field(default_factory=list)
    snp_allele_freq: Optional[np.ndarray] = None
    hla_freq: Optional[np.ndarray] = None
    assembly: str = "unknown"
    matching: Optional[np.ndarray] = None
    appendix: Any = None

    @property
    def n_hla(self) -> int:
        return len(self.hla_allele)

    @property
    def n_cell(self) -> int:
        """Length of the allele-pair posterior vector, nHLA(nHLA+1)/2 (src/LibHLA.cpp:1481)."""
        return self.n_hla * (self.n_hla + 1) // 2

    def pair_evals_per_sample(self) -> int:
        """Sum over classifiers of H(H+1)/2 haplotype-pair evaluations."""
        return int(sum(len(c.freq) * (len(c.freq) + 1) // 2 for c in self.classifiers))


def engine_kind(n_snp_c: int) -> str:
    """The library's distance engine for a classifier with ``n_snp_c`` SNPs (``HIBAG_ENGINE_OF`` in
    csrc/hibag_device.h): ``"fp4"`` (up to 30 SNPs: one v_mfma_scale_f32_32x32x64_f8f6f4 per sample half and
    32-record block; 33..112 SNPs: one per 28 SNPs, chained through the accumulator -- ``engine_steps``), ``"i8"``
    (31..32 SNPs: two v_mfma_i32_32x32x32_i8), ``"valu"`` (more than 112 SNPs).  A description for documentation and
    tests; what a finalized model really uses is ``HlaAttrBagClass.engine(c)`` (``hibag_hip_model_engine``)."""
    k = int(n_snp_c)
    e = os.environ.get("HIBAG_ENGINE")
    if e == "valu":
        return "valu"
    if k <= 30 and e != "i8":
        return "fp4"
    if k <= 32:
        return "i8"
    if k <= 112 and e != "i8" and os.environ.get("HIBAG_PASS2") != "recompute":
        return "fp4"
    return "valu"


def engine_steps(n_snp_c: int) -> int:
    """K steps of the FP4 engine for a classifier with ``n_snp_c`` SNPs (1 for the other engines)."""
    k = int(n_snp_c)
    return (1 if k <= 30 else -(-k // 28)) if engine_kind(k) == "fp4" else 1


def engine_nkb(n_snp_c: int) -> int:
    """int8-equivalent K blocks (kept for callers of round 1's interface): 0 = VALU engine."""
    return {"valu": 0, "fp4": 1, "i8": 2}[engine_kind(n_snp_c)]


@dataclass
class HlaSNPGeno:
    """``hlaSNPGenoClass``: ``genotype`` is [n_snp, n_samp] like the R matrix
    (values 0/1/2 = number of A alleles, NA = INT_MIN).  Either memory order is first class: column-major is R's own
    (objects read from ``.RData`` keep it: no copy) and reaches the device through ``hibag_hip_predict[_mapped]``,
    row-major is numpy's default and goes through ``hibag_hip_predict_snp_major``; ``hlaPredict`` copies neither."""
    genotype: np.ndarray
    sample_id: List[str]
    snp_id: List[str]
    snp_position: Optional[np.ndarray] = None
    snp_allele: List[str] = field(default_factory=list)
    assembly: str = "unknown"

    def sample_major(self, snp_sel: Optional[Sequence[int]] = None) -> np.ndarray:
        """int32 [n_samp, n_snp] -- the memory order ``.Call(HIBAG_Predict_*, as.integer(snp), ...)``
        hands the C side (R matrices are column-major, ``R/HIBAG.R:715-725``)."""
        g = self.genotype if snp_sel is None else self.genotype[np.asarray(snp_sel)]
        return np.ascontiguousarray(g.T, np.int32)


def _strs(x) -> List[str]:
    if x is None:
        return []
    if isinstance(x, (list, tuple)):
        return [s for s in x]
    return rdata.factor_to_strings(x)


def model_from_robj(obj) -> HlaAttrBagObj:
    """Decoded R list of class ``hlaAttrBagObj`` -> :class:`HlaAttrBagObj`.

    Error behaviour follows ``hlaModelFromObj`` (``R/HIBAG.R:1135-1163``)."""
    cls_attr = obj.attrs.get("class") or []
    if "hlaAttrBagObj" not in cls_attr:
        raise TypeError("inherits(obj, \"hlaAttrBagObj\") is not TRUE")
    alleles = _strs(obj["hla.allele"])
    lut = {a: i for i, a in enumerate(alleles)}
    n_samp = int(np.asarray(obj["n.samp"])[0])
    out = []
    for tree in obj["classifiers"]:
        hp = tree["haplos"]
        if "haplo" not in hp:
            raise ValueError("No 'haplo' component. The model may be used by the higher version of HIBAG.")
        names = _strs(hp["hla"])
        if any(a not in lut for a in names):
            raise ValueError("Invalid HLA alleles in the individual classifier.")
        sn = tree.get("samp.num")
        out.append(Classifier(
            snpidx=np.asarray(tree["snpidx"], np.int64) - 1,
            freq=np.asarray(hp["freq"], np.float64),
            hla=np.array([lut[a] for a in names], np.int32),
            haplo=_strs(hp["haplo"]),
            samp_num=(np.ones(n_samp, np.int32) if sn is None else np.asarray(sn, np.int32)),
            outofbag_acc=float(np.asarray(tree["outofbag.acc"])[0]) if "outofbag.acc" in tree else 0.0))
    asm = _strs(obj.get("assembly"))
    m = obj.get("matching")
    return HlaAttrBagObj(
        n_samp=n_samp, n_snp=int(np.asarray(obj["n.snp"])[0]), hla_allele=alleles, classifiers=out,
        hla_locus=(_strs(obj.get("hla.locus")) or [""])[0],
        sample_id=_strs(obj.get("sample.id")), snp_id=_strs(obj.get("snp.id")),
        snp_position=None if obj.get("snp.position") is None else np.asarray(obj["snp.position"], np.float64),
        snp_allele=_strs(obj.get("snp.allele")),
        snp_allele_freq=None if obj.get("snp.allele.freq") is None else np.asarray(obj["snp.allele.freq"], np.float64),
        hla_freq=None if obj.get("hla.freq") is None else np.asarray(obj["hla.freq"], np.float64),
        assembly=(asm[0] if asm and asm[0] is not None else "unknown"),
        matching=None if m is None else np.asarray(m, np.float64),
        appendix=obj.get("appendix"))


def geno_from_robj(obj) -> HlaSNPGeno:
    if "hlaSNPGenoClass" not in (obj.attrs.get("class") or []):
        raise TypeError("inherits(obj, \"hlaSNPGenoClass\") is not TRUE")
    g = obj["genotype"]
    dim = [int(v) for v in np.asarray(g.attrs["dim"])]
    # R is column-major: [n_snp, n_samp] as a VIEW of R's own memory (Fortran order) -- what hlaPredict hands the C side as it is
    mat = np.asarray(g, np.int32).reshape(dim[1], dim[0]).T
    asm = _strs(obj.get("assembly"))
    return HlaSNPGeno(genotype=mat, sample_id=_strs(obj["sample.id"]),
                      snp_id=_strs(obj["snp.id"]),
                      snp_position=np.asarray(obj["snp.position"], np.float64),
                      snp_allele=_strs(obj["snp.allele"]),
                      assembly=(asm[0] if asm and asm[0] is not None else "unknown"))


def load_model(path: str, name: Optional[str] = None, gene: Optional[str] = None) -> HlaAttrBagObj:
    """Load an ``hlaAttrBagObj`` from an R workspace.  ``name`` picks the object,
    ``gene`` an element of a model list such as ``modellist$A``."""
    ws = rdata.load_rdata(path)
    obj = ws[name] if name else next(iter(ws.values()))
    if gene is not None:
        obj = obj[gene]
    return model_from_robj(obj)


def load_geno(path: str, name: Optional[str] = None) -> HlaSNPGeno:
    ws = rdata.load_rdata(path)
    obj = ws[name] if name else next(iter(ws.values()))
    return geno_from_robj(obj)


def model_to_robj(obj: HlaAttrBagObj) -> "rdata.RList":
    """:class:`HlaAttrBagObj` -> the R list ``hlaModelToObj()`` returns (``R/HIBAG.R:1040-1062``; classifier
    records as built by ``HIBAG_GetClassifierList``, ``src/HIBAG.cpp:871-953``), ready for
    :func:`hibag_amd.rdata.save_rdata` -- R's ``hlaModelFromObj(get(load(file)))`` reads it back."""
    R = rdata

    def strs(v):
        return None if v is None else R.RStrings(list(v))

    def nums(v, dt):
        return None if v is None else R.RArray(np.asarray(v, dt))

    cls = []
    for c in obj.classifiers:
        n = len(c.freq)
        haplos = R.RList([R.RArray(np.asarray(c.freq, np.float64)), R.RStrings([obj.hla_allele[int(a)] for a in c.hla]),
                          R.RStrings(list(c.haplo))],
                         {"names": ["freq", "hla", "haplo"],
                          "row.names": R.RArray(np.array([NA_INTEGER, -n], np.int32)),      # R's compact 1..n row names
                          "class": ["data.frame"]})
        samp = c.samp_num if c.samp_num is not None else np.ones(obj.n_samp, np.int32)
        cls.append(R.RList([R.RArray(np.asarray(samp, np.int32)), haplos, R.RArray(np.asarray(c.snpidx, np.int32) + 1),
                            R.RArray(np.array([c.outofbag_acc], np.float64))],
                           {"names": ["samp.num", "haplos", "snpidx", "outofbag.acc"]}))
    hla_freq = None
    if obj.hla_freq is not None:           # prop.table(table(H)): a named 1-d table
        hla_freq = R.RArray(np.asarray(obj.hla_freq, np.float64),
                            {"class": ["table"], "dim": R.RArray(np.array([obj.n_hla], np.int32)),
                             "dimnames": R.RList([R.RStrings(list(obj.hla_allele))], {"names": ["H"]})})
    fields = [
        ("n.samp", R.RArray(np.array([obj.n_samp], np.int32))), ("n.snp", R.RArray(np.array([obj.n_snp], np.int32))),
        ("sample.id", strs(obj.sample_id)), ("snp.id", strs(obj.snp_id)),
        ("snp.position", nums(obj.snp_position, np.float64)), ("snp.allele", strs(obj.snp_allele)),
        ("snp.allele.freq", nums(obj.snp_allele_freq, np.float64)), ("hla.locus", R.RStrings([obj.hla_locus])),
        ("hla.allele", strs(obj.hla_allele)), ("hla.freq", hla_freq), ("assembly", R.RStrings([obj.assembly])),
        ("classifiers", R.RList(cls)), ("matching", nums(obj.matching, np.float64)),
        ("appendix", obj.appendix if obj.appendix is not None else R.RList([]))]
    return R.RList([v for _, v in fields], {"names": [k for k, _ in fields], "class": ["hlaAttrBagObj"]})


def save_model(path: str, obj: HlaAttrBagObj, name: str = "mobj", compress=True, version: int = 2,
               altrep: bool = False) -> None:
    """``mobj <- hlaModelToObj(model); save(mobj, file=path, compress=, version=)``."""
    rdata.save_rdata(path, {name: model_to_robj(obj)}, compress=compress, version=version, altrep=altrep)
