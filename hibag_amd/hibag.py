"""Host-side mirror of the reference's R interface for the prediction path.

Same names, argument meaning and error behaviour as the reference's
``hlaSetKernelTarget`` (``R/HIBAG.R:1668-1674``), ``hlaModelFromObj`` /
``hlaModelToObj`` (``R/HIBAG.R:1135-1178`` / ``:1041-1062``) and ``hlaPredict``
(``R/HIBAG.R:481-818``); the compute goes through the C ABI of
``libhibag_hip.so`` (``include/hibag_hip.h``) and nowhere else.

R is not available on the GPU box, so the thin R layer is restated in Python
(the reference itself has no Python).  PyTorch appears only where a caller
hands over device tensors.
"""

from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib
from ._lib import HibagHipError
from .bed import HlaBEDGeno
from .model import (NA_INTEGER, Classifier, HlaAttrBagObj, HlaSNPGeno)

_TARGETS_CPU = ("max", "auto.avx2", "base", "sse2", "sse4", "avx", "avx2", "avx512f", "avx512bw",
                "avx512vpopcnt")   # src/LibHLA.cpp:1279-1456, man/hlaSetKernelTarget.Rd
_kernel_target: Optional[str] = None
_kernel_info: str = ""


def hlaSetKernelTarget(cpu: str = "hip") -> List[str]:
    """Select the kernel target.  The reference accepts CPU instruction sets
    (``src/LibHLA.cpp:1266-1475``); this build adds the value ``"hip"`` and
    implements only that: the CPU names raise, as the reference does for a
    target the build does not support (``Rf_error("Not support AVX2.")``)."""
    global _kernel_target, _kernel_info
    cpu = str(cpu)
    if cpu != "hip":
        if cpu in _TARGETS_CPU:
            raise HibagHipError(_lib.lib().hibag_hip_set_kernel_target(cpu.encode(), None, 0),
                                f"Not support {cpu.upper()}: hibag_amd implements the \"hip\" kernel target only.")
        raise ValueError(f"'arg' should be one of \"hip\", {', '.join(repr(t) for t in _TARGETS_CPU)}")
    buf = C.create_string_buffer(256)
    _lib.check(_lib.lib().hibag_hip_set_kernel_target(b"hip", buf, len(buf)))
    _kernel_target, _kernel_info = "hip", buf.value.decode()
    return [_kernel_info]


def _as_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HlaAttrBagClass:
    """``hlaAttrBagClass``: an ``hlaAttrBagObj`` plus the native model handle
    (the reference keeps an index into a handle table and frees it from a
    finalizer, ``src/HIBAG.cpp:409-475``)."""

    def __init__(self, obj: HlaAttrBagObj, device: Optional[int] = None,
                 snp_weight: Optional[np.ndarray] = None):
        L = _lib.lib()
        self.obj = obj
        self._h = None
        if device is not None:
            _lib.check(L.hibag_hip_set_device(int(device)))
        h = L.hibag_hip_model_new(int(obj.n_hla), int(obj.n_snp))
        if not h:
            raise HibagHipError(-1, L.hibag_hip_last_error().decode())
        self._h = C.c_void_p(h)
        try:
            for c in obj.classifiers:
                strs = (C.c_char_p * len(c.haplo))(*[s.encode() for s in c.haplo])
                _lib.check(L.hibag_hip_model_add_classifier(
                    self._h, len(c.snpidx), _as_ptr(c.snpidx), len(c.freq), _as_ptr(c.freq), _as_ptr(c.hla), strs))
            if snp_weight is not None:
                sw = np.ascontiguousarray(snp_weight, np.int32)
                if sw.shape != (obj.n_snp,):
                    raise ValueError("snp_weight must have one entry per model SNP")
                _lib.check(L.hibag_hip_model_set_snp_weights(self._h, _as_ptr(sw)))
            _lib.check(L.hibag_hip_model_finalize(self._h))
        except Exception:
            self.close()
            raise

    # attribute access like the R list: model$hla.allele -> model.hla_allele
    def __getattr__(self, name):
        if name in ("obj", "_h"):
            raise AttributeError(name)
        return getattr(self.obj, name)

    @property
    def handle(self) -> C.c_void_p:
        if self._h is None:
            raise HibagHipError(-4, "the model has been closed")
        return self._h

    def close(self):
        """``hlaClose`` (``R/HIBAG.R:1023-1035``)."""
        for r in self.__dict__.pop("_replicas", {}).values():      # replicas made for hlaPredict(cl=[devices])
            r.close()
        if getattr(self, "_h", None) is not None:
            _lib.lib().hibag_hip_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device(self) -> int:
        """The HIP device the model lives on."""
        return int(_lib.lib().hibag_hip_model_device(self.handle))

    def pair_evals(self) -> int:
        return int(_lib.lib().hibag_hip_model_pair_evals(self.handle))

    def stored_cells(self) -> int:
        """Cell sums per sample that pass 1 stores for pass 2 to read back."""
        return int(_lib.lib().hibag_hip_model_stored_cells(self.handle))

    def second_pass_pairs(self) -> int:
        """Haplotype pairs per sample that pass 2 evaluates again (those of the cells that are not stored)."""
        return int(_lib.lib().hibag_hip_model_second_pass_pairs(self.handle))

    # --- launch status (include/hibag_hip.h "launch status of the device-pointer entries") ---
    def status(self) -> int:
        """0, or the model's sticky fault code (waits for the model's outstanding launches first)."""
        return int(_lib.lib().hibag_hip_model_status(self.handle))

    def clear_status(self):
        _lib.check(_lib.lib().hibag_hip_model_clear_status(self.handle))

    def handover_faults(self) -> int:
        return int(_lib.lib().hibag_hip_model_handover_faults(self.handle))

    def inject_handover_fault(self, which_pass: int):
        """Tests only: the next batch drops the first hand-over of pass 1 or 2."""
        _lib.check(_lib.lib().hibag_hip_test_inject_handover_fault(self.handle, int(which_pass)))

    def engine(self, classifier: int):
        """(engine name, K steps) of a classifier as the library finalized it."""
        e, k = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().hibag_hip_model_engine(self.handle, int(classifier), C.byref(e), C.byref(k)))
        return {0: "valu", 1: "fp4", 2: "i8", 3: "i8"}[e.value], k.value

    def replicate(self, device: int) -> "HlaAttrBagClass":
        """A finalized copy of the model on another (or the same) device: ``hibag_hip_model_replicate``."""
        h = _lib.lib().hibag_hip_model_replicate(self.handle, int(device))
        if not h:
            raise HibagHipError(-2, _lib.lib().hibag_hip_last_error().decode())
        r = object.__new__(HlaAttrBagClass)
        r.obj = self.obj
        r._h = C.c_void_p(h)
        return r

    def shard(self, shard: int, n_shards: int, device: int) -> "HlaAttrBagClass":
        """Classifiers ``hibag_hip_shard_bounds(C, n_shards, shard)`` of the model as a model of their own on ``device``,
        with the full model's per-SNP classifier counts: ``hibag_hip_model_shard``."""
        h = _lib.lib().hibag_hip_model_shard(self.handle, int(shard), int(n_shards), int(device))
        if not h:
            raise HibagHipError(-2, _lib.lib().hibag_hip_last_error().decode())
        r = object.__new__(HlaAttrBagClass)
        r.obj = self.obj                      # (alleles, SNPs: the full model's; the shard's classifiers live in the library)
        r._h = C.c_void_p(h)
        return r

    def mutation_table(self) -> np.ndarray:
        t = np.empty(257, np.float64)
        _lib.check(_lib.lib().hibag_hip_model_mutation_table(self.handle, _as_ptr(t)))
        return t

    # --- timing of the kernels (HIP events on the launch stream) ---
    def set_timing(self, enabled=True):
        """True / False: HIP events around every kernel class or none; a sequence of kernel names (``"total"``, ``"accum"``,
        ``"pack"``, ``"finish"``): events around those only (``hibag_hip_set_timing``'s mask form)."""
        if isinstance(enabled, (list, tuple, set, frozenset)):
            ids = {v: k for k, v in _lib.KERNEL_NAMES.items()}
            code = 2 * sum(1 << ids[name] for name in set(enabled))
        else:
            code = int(bool(enabled))
        _lib.check(_lib.lib().hibag_hip_set_timing(self.handle, code))

    def reset_timing(self):
        _lib.check(_lib.lib().hibag_hip_reset_timing(self.handle))

    def get_timing(self) -> dict:
        out = {}
        for k, name in _lib.KERNEL_NAMES.items():
            ms, n = C.c_double(0), C.c_int64(0)
            _lib.check(_lib.lib().hibag_hip_get_timing(self.handle, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    # --- raw entry points -------------------------------------------------
    def predict_raw(self, genomat: np.ndarray, vote_method: int = 1, want_dosage: bool = True,
                    want_prob: bool = False) -> dict:
        """``CAttrBag_Model::PredictHLA`` on host arrays: ``genomat`` int32 [n_samp, n_snp]."""
        g = np.ascontiguousarray(genomat, np.int32)
        if g.ndim != 2 or g.shape[1] != self.obj.n_snp:
            raise ValueError("genomat must be [n_samp, n.snp] int32")
        n = g.shape[0]
        out = dict(h1=np.zeros(n, np.int32), h2=np.zeros(n, np.int32),
                   prob=np.zeros(n, np.float64), matching=np.zeros(n, np.float64))
        if want_dosage:
            out["dosage"] = np.zeros((n, self.obj.n_hla), np.float64)
        if want_prob:
            out["postprob"] = np.zeros((n, self.obj.n_cell), np.float64)
        _lib.check(_lib.lib().hibag_hip_predict(
            self.handle, _as_ptr(g), n, int(vote_method), _as_ptr(out["h1"]), _as_ptr(out["h2"]),
            _as_ptr(out["prob"]), _as_ptr(out["matching"]), _as_ptr(out.get("dosage")), _as_ptr(out.get("postprob"))))
        return out

    def predict_bed(self, bed_fn: str, n_samp: int, n_snp: int, snp_col: np.ndarray, flip: Optional[np.ndarray] = None,
                    vote_method: int = 1, want_dosage: bool = True, want_prob: bool = False) -> dict:
        """``PredictHLA`` on every sample of a PLINK BED file (``hibag_hip_predict_bed``):
        ``snp_col[k]`` = 0-based .bim index of model SNP k (-1 = absent), ``flip[k]`` =
        reverse the allele count of SNP k."""
        col = np.ascontiguousarray(snp_col, np.int32)
        if col.shape != (self.obj.n_snp,):
            raise ValueError("snp_col must have one entry per model SNP")
        fl = None if flip is None else np.ascontiguousarray(np.asarray(flip) != 0, np.int32)
        n = int(n_samp)
        out = dict(h1=np.zeros(n, np.int32), h2=np.zeros(n, np.int32),
                   prob=np.zeros(n, np.float64), matching=np.zeros(n, np.float64))
        if want_dosage:
            out["dosage"] = np.zeros((n, self.obj.n_hla), np.float64)
        if want_prob:
            out["postprob"] = np.zeros((n, self.obj.n_cell), np.float64)
        _lib.check(_lib.lib().hibag_hip_predict_bed(
            self.handle, os.fsencode(bed_fn), n, int(n_snp), _as_ptr(col), _as_ptr(fl), int(vote_method),
            _as_ptr(out["h1"]), _as_ptr(out["h2"]), _as_ptr(out["prob"]), _as_ptr(out["matching"]),
            _as_ptr(out.get("dosage")), _as_ptr(out.get("postprob"))))
        return out

    def predict_mapped(self, genomat: np.ndarray, snp_col: np.ndarray, flip: Optional[np.ndarray] = None,
                       vote_method: int = 1, want_dosage: bool = True, want_prob: bool = False) -> dict:
        """``PredictHLA`` on the COHORT's own matrix ``genomat`` [n_samp, n_geno_snp] (``hibag_hip_predict_mapped``):
        ``snp_col[k]`` = column of model SNP k (-1 = absent), ``flip[k]`` = reverse its allele count; the
        selection and the flip happen on the device while the genotypes are packed."""
        g = np.ascontiguousarray(genomat, np.int32)
        if g.ndim != 2:
            raise ValueError("genomat must be [n_samp, n_geno_snp]")
        col = np.ascontiguousarray(snp_col, np.int32)
        if col.shape != (self.obj.n_snp,):
            raise ValueError("snp_col must have one entry per model SNP")
        fl = None if flip is None else np.ascontiguousarray(np.asarray(flip) != 0, np.int32)
        n = g.shape[0]
        out = dict(h1=np.zeros(n, np.int32), h2=np.zeros(n, np.int32),
                   prob=np.zeros(n, np.float64), matching=np.zeros(n, np.float64))
        if want_dosage:
            out["dosage"] = np.zeros((n, self.obj.n_hla), np.float64)
        if want_prob:
            out["postprob"] = np.zeros((n, self.obj.n_cell), np.float64)
        _lib.check(_lib.lib().hibag_hip_predict_mapped(
            self.handle, _as_ptr(g), n, g.shape[1], _as_ptr(col), _as_ptr(fl), int(vote_method),
            _as_ptr(out["h1"]), _as_ptr(out["h2"]), _as_ptr(out["prob"]), _as_ptr(out["matching"]),
            _as_ptr(out.get("dosage")), _as_ptr(out.get("postprob"))))
        return out

    def predict_device(self, d_geno, n_samp: int, vote_method: int = 1, d_h1=None, d_h2=None, d_prob=None,
                       d_matching=None, d_dosage=None, d_postprob=None, stream=None):
        """Device-pointer form; arguments are ints (``tensor.data_ptr()``) or None."""
        def p(x):
            return None if x is None else C.c_void_p(int(x))
        _lib.check(_lib.lib().hibag_hip_predict_device(
            self.handle, p(d_geno), int(n_samp), int(vote_method), p(d_h1), p(d_h2), p(d_prob), p(d_matching),
            p(d_dosage), p(d_postprob), p(stream)))

    def batch_limit(self) -> int:
        """Samples one call of the partial entry takes (``hibag_hip_model_batch_limit``)."""
        return int(_lib.lib().hibag_hip_model_batch_limit(self.handle))

    def predict_partial_device(self, d_geno, n_samp: int, d_partial, stream=None):
        def p(x):
            return None if x is None else C.c_void_p(int(x))
        _lib.check(_lib.lib().hibag_hip_predict_partial_device(self.handle, p(d_geno), int(n_samp), p(d_partial), p(stream)))

    def finish_device(self, d_partial, n_samp: int, d_h1=None, d_h2=None, d_prob=None, d_matching=None,
                      d_dosage=None, d_postprob=None, stream=None):
        def p(x):
            return None if x is None else C.c_void_p(int(x))
        _lib.check(_lib.lib().hibag_hip_finish_device(
            self.handle, p(d_partial), int(n_samp), p(d_h1), p(d_h2), p(d_prob), p(d_matching), p(d_dosage),
            p(d_postprob), p(stream)))


def multi_slice(n_samp: int, n_models: int, i: int):
    """(first, count) of replica i's contiguous sample slice (``hibag_hip_multi_slice``; host arithmetic only)."""
    a, b = C.c_int(0), C.c_int(0)
    _lib.check(_lib.lib().hibag_hip_multi_slice(int(n_samp), int(n_models), int(i), C.byref(a), C.byref(b)))
    return a.value, b.value


def predict_multi(models: Sequence[HlaAttrBagClass], genomat: np.ndarray, vote_method: int = 1, want_dosage: bool = True,
                  want_prob: bool = False) -> dict:
    """``hibag_hip_predict_multi``: one cohort over several replicas of a model (one per device, one host thread each,
    contiguous sample slices, no collective) -- the counterpart of ``hlaPredict(cl=<cluster>)`` (``R/HIBAG.R:764-808``)."""
    if not models:
        raise ValueError("no models given")
    obj = models[0].obj
    g = np.ascontiguousarray(genomat, np.int32)
    if g.ndim != 2 or g.shape[1] != obj.n_snp:
        raise ValueError("genomat must be [n_samp, n.snp] int32")
    n = g.shape[0]
    out = dict(h1=np.zeros(n, np.int32), h2=np.zeros(n, np.int32),
               prob=np.zeros(n, np.float64), matching=np.zeros(n, np.float64))
    if want_dosage:
        out["dosage"] = np.zeros((n, obj.n_hla), np.float64)
    if want_prob:
        out["postprob"] = np.zeros((n, obj.n_cell), np.float64)
    hs = (C.c_void_p * len(models))(*[m.handle for m in models])
    _lib.check(_lib.lib().hibag_hip_predict_multi(
        hs, len(models), _as_ptr(g), n, int(vote_method), _as_ptr(out["h1"]), _as_ptr(out["h2"]),
        _as_ptr(out["prob"]), _as_ptr(out["matching"]), _as_ptr(out.get("dosage")), _as_ptr(out.get("postprob"))))
    return out


class ShardGroup:
    """``hibag_hip_shard_group``: the shards of one model, each on its device, merged per batch by ONE RCCL all-reduce
    issued by the library itself (``hibag_amd/csrc/hibag_shard.hip``).  ``devices``: one entry per shard (a device may
    repeat: its shards are added up on it before the all-reduce)."""

    def __init__(self, model: HlaAttrBagClass, devices: Sequence[int]):
        if not devices:
            raise ValueError("no devices given")
        self.obj = model.obj
        self.shards = [model.shard(i, len(devices), int(d)) for i, d in enumerate(devices)]
        hs = (C.c_void_p * len(self.shards))(*[m.handle for m in self.shards])
        h = _lib.lib().hibag_hip_shard_group_new(hs, len(self.shards))
        if not h:
            msg = _lib.lib().hibag_hip_last_error().decode()
            for m in self.shards:
                m.close()
            raise HibagHipError(-2, msg)
        self._h = C.c_void_p(h)

    @property
    def ranks(self) -> int:
        return int(_lib.lib().hibag_hip_shard_group_ranks(self._h))

    @property
    def allreduces(self) -> int:
        return int(_lib.lib().hibag_hip_shard_group_allreduces(self._h))

    def predict_raw(self, genomat: np.ndarray, want_dosage: bool = True, want_prob: bool = False) -> dict:
        g = np.ascontiguousarray(genomat, np.int32)
        if g.ndim != 2 or g.shape[1] != self.obj.n_snp:
            raise ValueError("genomat must be [n_samp, n.snp] int32")
        n = g.shape[0]
        out = dict(h1=np.zeros(n, np.int32), h2=np.zeros(n, np.int32), prob=np.zeros(n, np.float64), matching=np.zeros(n, np.float64))
        if want_dosage:
            out["dosage"] = np.zeros((n, self.obj.n_hla), np.float64)
        if want_prob:
            out["postprob"] = np.zeros((n, self.obj.n_cell), np.float64)
        _lib.check(_lib.lib().hibag_hip_shard_group_predict(
            self._h, _as_ptr(g), n, _as_ptr(out["h1"]), _as_ptr(out["h2"]), _as_ptr(out["prob"]), _as_ptr(out["matching"]),
            _as_ptr(out.get("dosage")), _as_ptr(out.get("postprob"))))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.lib().hibag_hip_shard_group_free(self._h)
            self._h = None
            for m in self.shards:
                m.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hlaModelFromObj(obj: HlaAttrBagObj, device: Optional[int] = None) -> HlaAttrBagClass:
    """``hlaModelFromObj`` (``R/HIBAG.R:1135-1178``)."""
    if not isinstance(obj, HlaAttrBagObj):
        raise TypeError("inherits(obj, \"hlaAttrBagObj\") is not TRUE")
    return HlaAttrBagClass(obj, device)


def hlaModelToObj(model: HlaAttrBagClass) -> HlaAttrBagObj:
    """``hlaModelToObj`` (``R/HIBAG.R:1041-1062``)."""
    if not isinstance(model, HlaAttrBagClass):
        raise TypeError("inherits(model, \"hlaAttrBagClass\") is not TRUE")
    return model.obj


def hlaClose(model: HlaAttrBagClass) -> None:
    model.close()


@dataclass
class HlaAlleleClass:
    """``hlaAlleleClass`` as returned by ``hlaPredict`` (``R/HIBAG.R:729-748``)."""
    locus: str
    sample_id: List
    allele1: List[Optional[str]]
    allele2: List[Optional[str]]
    prob: Optional[np.ndarray] = None
    matching: Optional[np.ndarray] = None
    assembly: str = "unknown"
    dosage: Optional[np.ndarray] = None        # [n_hla, n_samp], rows = hla.allele
    postprob: Optional[np.ndarray] = None      # [n_cell, n_samp], rows = pair_names
    pair_names: List[str] = field(default_factory=list)
    h1: Optional[np.ndarray] = None            # 0-based allele indices (NA = INT_MIN)
    h2: Optional[np.ndarray] = None


def _pair_names(alleles: Sequence[str]) -> List[str]:
    # outer(a, a, paste, sep="/")[lower.tri(, diag=TRUE)] (R/HIBAG.R:746-747): column-major
    # lower triangle = for h1, for h2 >= h1: a[h2]/a[h1] -- the posterior vector's order
    return [f"{alleles[j]}/{alleles[i]}" for i in range(len(alleles)) for j in range(i, len(alleles))]


def _snp_ids(obj, match_type: str) -> List:
    """``hlaSNPID`` (``R/DataUtilities.R:512-524``)."""
    pos = [None if p is None else (int(p) if float(p).is_integer() else float(p)) for p in (obj.snp_position if obj.snp_position is not None else [])]
    if match_type == "Position":
        return pos
    if match_type == "Pos+Allele":
        return [f"{p}-{a}" for p, a in zip(pos, obj.snp_allele)]
    if match_type == "RefSNP+Position":
        return [f"{i}-{p}" for i, p in zip(obj.snp_id, pos)]
    if match_type == "RefSNP":
        return list(obj.snp_id)
    raise ValueError("'arg' should be one of \"Position\", \"Pos+Allele\", \"RefSNP+Position\", \"RefSNP\"")


_TYPES = ("response+dosage", "response", "prob", "response+prob")
_VOTES = ("prob", "majority")


def hlaPredict(object: HlaAttrBagClass, snp: Union[HlaSNPGeno, HlaBEDGeno, np.ndarray], cl=False,
               type: str = "response+dosage", vote: str = "prob", allele_check: bool = True,
               match_type: str = "Position", same_strand: bool = False, verbose: bool = True,
               verbose_match: bool = True):
    """``hlaPredict`` (``R/HIBAG.R:481-818``).

    ``snp`` is an :class:`HlaSNPGeno` or a numeric matrix [n.snp, n.samp] (or a
    vector of length n.snp) laid out like the R argument.  ``cl``: the reference takes a
    ``parallel`` cluster and spreads contiguous sample slices over its workers
    (``R/HIBAG.R:764-808``); here a list of device indices does the same over the GPUs of
    the node (``hibag_hip_predict_multi``: one replica and one host thread per device, no
    collective, results identical to one device).  ``False`` / ``None`` / a thread count:
    the model's own device processes the whole cohort.
    Returns :class:`HlaAlleleClass`, or for ``type="prob"`` the posterior matrix
    [n_cell, n_samp] like the reference.
    """
    if not isinstance(object, HlaAttrBagClass):
        raise TypeError("inherits(object, \"hlaAttrBagClass\") is not TRUE")
    if type not in _TYPES:
        raise ValueError("'arg' should be one of " + ", ".join(f'"{t}"' for t in _TYPES))
    if vote not in _VOTES:
        raise ValueError("'arg' should be one of \"prob\", \"majority\"")
    vote_method = _VOTES.index(vote) + 1
    obj = object.obj
    out = sys.stdout

    if verbose:
        s = list(obj.hla_allele)
        if len(s) > 3:
            s = s[:3] + ["..."]
        n_c = len(obj.classifiers)
        print(f"HIBAG model for HLA-{obj.hla_locus}:\n    {n_c} individual classifier{'s' if n_c > 1 else ''}\n"
              f"    {len(obj.snp_id)} SNPs\n    {obj.n_hla} unique HLA alleles: {', '.join(s)}", file=out)
        print("Prediction:\n    " + ("based on the averaged posterior probabilities" if vote_method == 1
                                      else "by voting from all individual classifiers"), file=out)

    bed_plan = map_plan = None
    if isinstance(snp, HlaBEDGeno):
        # extension: the genotypes stay in the PLINK BED file; the SNP matching / strand check
        # (R/HIBAG.R:550-686) runs on the annotation and the device decodes the file directly
        from .snpmatch import plan_snps_for_predict
        bed_plan = plan_snps_for_predict(obj, snp, snp.allele_freq, match_type, allele_check, same_strand,
                                         verbose, verbose_match)
        assembly = bed_plan.assembly
        geno_sampid = list(snp.sample_id)
        mat = None
    elif not isinstance(snp, HlaSNPGeno):
        g = np.asarray(snp)
        if g.ndim == 1:
            if g.shape[0] != obj.n_snp:
                raise ValueError("length(snp) == object$n.snp is not TRUE")
            g = g.reshape(-1, 1)
        elif g.ndim != 2 or g.shape[0] != obj.n_snp:
            raise ValueError("nrow(snp) == object$n.snp is not TRUE")
        geno_sampid: List = list(range(1, g.shape[1] + 1))
        assembly = "auto-silent"
        mat = g
    else:
        # the SNP matching / strand check (R/HIBAG.R:550-686) decides on the annotation; the rows are
        # picked and flipped on the device while the genotypes are packed (hibag_hip_predict_mapped)
        from .snpmatch import _row_afreq, plan_snps_for_predict
        map_plan = plan_snps_for_predict(obj, snp, lambda rows: _row_afreq(snp.genotype[rows]), match_type,
                                         allele_check, same_strand, verbose, verbose_match)
        assembly = map_plan.assembly
        geno_sampid = list(snp.sample_id)
        mat = None

    if mat is not None and mat.shape[0] != obj.n_snp:
        raise ValueError("The number of SNPs is not valid, and it maybe due to duplicated 'snp.id' "
                         "or incorrect dimension of genotype matrix.")
    n_samp = len(geno_sampid) if mat is None else mat.shape[1]
    if verbose:
        print(f"# of samples: {n_samp}", file=out)
        print(f"Kernel target: {_kernel_info or 'hip'}", file=out)

    want_prob = type in ("prob", "response+prob")
    want_dosage = type != "response"
    devices = list(cl) if isinstance(cl, (list, tuple)) else None
    if devices is not None:
        # a device list: validated up front (an index out of range used to surface as ENODEV from deep inside replicate())
        n_dev = int(_lib.lib().hibag_hip_device_count())
        bad = [d for d in devices if not isinstance(d, (int, np.integer)) or isinstance(d, bool) or not (0 <= int(d) < n_dev)]
        if not devices or bad:
            raise ValueError(f"'cl' must be a non-empty list of HIP device indices below {n_dev}: {cl!r}")
        if bed_plan is not None:
            # the BED route decodes on ONE device (hibag_hip_predict_bed); silently ignoring the list would not be what
            # the caller asked for
            raise ValueError("hlaPredict(cl = [devices]) takes a genotype matrix or an hlaSNPGenoClass; for a lazily opened BED "
                             "file (hlaBED2Geno(lazy=True)) predict on one device, or load the genotypes first (hlaBED2Geno())")
    if devices is not None and bed_plan is None:
        # several devices: the model-order matrix is built on the host (the model's few hundred columns of the
        # cohort), then sliced over the replicas
        if map_plan is not None:
            g = np.asarray(snp.genotype)
            sel = np.asarray(map_plan.sel)
            rows = g[np.maximum(sel, 0)]
            if rows.dtype.kind == "f":
                rows = np.where(np.isfinite(rows), rows, NA_INTEGER)
            rows = rows.astype(np.int32)
            ok = (rows >= 0) & (rows <= 2)
            fl = np.asarray(map_plan.flip) != 0 if map_plan.flip is not None else np.zeros(len(sel), bool)
            rows = np.where(ok & fl[:, None], 2 - rows, rows)
            rows[sel < 0] = NA_INTEGER
            genomat = np.ascontiguousarray(rows.T, np.int32)
        else:
            gi = np.where(np.isfinite(mat), mat, NA_INTEGER) if mat.dtype.kind == "f" else mat
            genomat = np.ascontiguousarray(np.asarray(gi).T, np.int32)
        cache = object.__dict__.setdefault("_replicas", {})
        reps = []
        for d in devices:
            key = (int(d), len([r for r in reps if r[0] == int(d)]))      # (several replicas on one device are allowed)
            if key not in cache:
                cache[key] = object.replicate(int(d))
            reps.append((int(d), cache[key]))
        rv = predict_multi([r for _, r in reps], genomat, vote_method, want_dosage=want_dosage, want_prob=want_prob)
    elif bed_plan is not None:
        col = np.where(bed_plan.sel >= 0, snp.bed_index[np.maximum(bed_plan.sel, 0)], -1)
        rv = object.predict_bed(snp.bed_fn, snp.n_bed_samp, snp.n_bed_snp, col, bed_plan.flip, vote_method,
                                want_dosage=want_dosage, want_prob=want_prob)
    elif map_plan is not None:
        g = np.asarray(snp.genotype)
        if g.dtype.kind == "f":
            g = np.where(np.isfinite(g), g, NA_INTEGER)
        cohort = np.ascontiguousarray(g.T, np.int32)      # [n_samp, cohort SNPs]: R's memory order
        rv = object.predict_mapped(cohort, map_plan.sel, map_plan.flip, vote_method,
                                   want_dosage=want_dosage, want_prob=want_prob)
    else:
        # as.integer(snp): R's NA -> NA_integer_ ; the C side treats anything outside 0..2 as missing
        gi = np.where(np.isfinite(mat), mat, NA_INTEGER) if mat.dtype.kind == "f" else mat
        genomat = np.ascontiguousarray(np.asarray(gi).T, np.int32)     # [n_samp, n_snp]
        rv = object.predict_raw(genomat, vote_method, want_dosage=want_dosage, want_prob=want_prob)

    names = _pair_names(obj.hla_allele)
    if type == "prob":
        res = np.ascontiguousarray(rv["postprob"].T)
        na_cnt = int(np.nansum(res.sum(axis=0) <= 0))
    else:
        def nm(ix):
            return [None if int(k) == NA_INTEGER else obj.hla_allele[int(k)] for k in ix]
        res = HlaAlleleClass(locus=obj.hla_locus, sample_id=geno_sampid, allele1=nm(rv["h1"]), allele2=nm(rv["h2"]),
                             prob=rv["prob"], matching=rv["matching"], assembly=assembly,
                             dosage=(np.ascontiguousarray(rv["dosage"].T) if type != "response" else None),
                             postprob=(np.ascontiguousarray(rv["postprob"].T) if want_prob else None),
                             pair_names=names if want_prob else [], h1=rv["h1"], h2=rv["h2"])
        na_cnt = sum(1 for a, b in zip(res.allele1, res.allele2) if a is None or b is None)

    if na_cnt > 0:   # R/HIBAG.R:811-815
        import warnings
        warnings.warn(f"No prediction output{'s' if na_cnt > 1 else ''} for {na_cnt} individual"
                      f"{'s' if na_cnt > 1 else ''} (possibly due to missing SNPs).")
    return res
