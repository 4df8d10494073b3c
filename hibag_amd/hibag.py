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
        out = self._outputs(n, want_dosage, want_prob)
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
        out = self._outputs(n, want_dosage, want_prob)
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
        out = self._outputs(n, want_dosage, want_prob)
        _lib.check(_lib.lib().hibag_hip_predict_mapped(
            self.handle, _as_ptr(g), n, g.shape[1], _as_ptr(col), _as_ptr(fl), int(vote_method),
            _as_ptr(out["h1"]), _as_ptr(out["h2"]), _as_ptr(out["prob"]), _as_ptr(out["matching"]),
            _as_ptr(out.get("dosage")), _as_ptr(out.get("postprob"))))
        return out

    def predict_snp_major(self, genomat: np.ndarray, snp_col: Optional[np.ndarray] = None, flip: Optional[np.ndarray] = None,
                          vote_method: int = 1, want_dosage: bool = True, want_prob: bool = False) -> dict:
        """``PredictHLA`` on a SNP-MAJOR matrix ``genomat`` [n_geno_snp, n_samp] in C order -- numpy's own layout for the
        [SNP, sample] matrix of an ``hlaSNPGenoClass`` (``hibag_hip_predict_snp_major``): row ``snp_col[k]`` holds model SNP k
        (-1 = absent; ``None`` = row k), ``flip[k]`` reverses its allele count.  Nothing is transposed on the host."""
        g = np.asarray(genomat)
        if g.ndim != 2 or g.dtype != np.int32 or g.strides[1] != 4 or g.strides[0] % 4 or (g.shape[0] > 1 and g.strides[0] < 4 * g.shape[1]):
            g = np.ascontiguousarray(g, np.int32)
            if g.ndim != 2:
                raise ValueError("genomat must be [n_geno_snp, n_samp]")
        col = None
        if snp_col is not None:
            col = np.ascontiguousarray(snp_col, np.int32)
            if col.shape != (self.obj.n_snp,):
                raise ValueError("snp_col must have one entry per model SNP")
        elif g.shape[0] < self.obj.n_snp:
            raise ValueError("nrow(snp) == object$n.snp is not TRUE")
        fl = None if flip is None else np.ascontiguousarray(np.asarray(flip) != 0, np.int32)
        n = g.shape[1]
        out = self._outputs(n, want_dosage, want_prob)
        ld = g.strides[0] // 4 if g.shape[0] > 1 else max(n, 1)
        _lib.check(_lib.lib().hibag_hip_predict_snp_major(
            self.handle, _as_ptr(g), ld, n, max(g.shape[0], 1), _as_ptr(col), _as_ptr(fl), int(vote_method),
            _as_ptr(out["h1"]), _as_ptr(out["h2"]), _as_ptr(out["prob"]), _as_ptr(out["matching"]),
            _as_ptr(out.get("dosage")), _as_ptr(out.get("postprob"))))
        return out

    def _outputs(self, n: int, want_dosage: bool, want_prob: bool) -> dict:
        """The output arrays of ``PredictHLA`` for n samples (every element is written by the library)."""
        out = dict(h1=np.empty(n, np.int32), h2=np.empty(n, np.int32), prob=np.empty(n, np.float64), matching=np.empty(n, np.float64))
        if want_dosage:
            out["dosage"] = np.empty((n, self.obj.n_hla), np.float64)
        if want_prob:
            out["postprob"] = np.empty((n, self.obj.n_cell), np.float64)
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


class HlaAlleleClass:
    """``hlaAlleleClass`` as returned by ``hlaPredict`` (``R/HIBAG.R:729-748``): ``locus``, ``sample_id``, ``allele1`` /
    ``allele2`` (lists of allele names, ``None`` = NA), ``prob``, ``matching``, ``assembly``, ``dosage`` [n_hla, n_samp] (rows =
    ``hla.allele``), ``postprob`` [n_cell, n_samp] (rows = ``pair_names``), and the calls as 0-based allele indices ``h1`` /
    ``h2`` (NA = INT_MIN).

    R builds the two name columns with one vectorised gather, ``object$hla.allele[H1 + 1L]`` -- pointers into its string
    cache.  The counterpart here is a categorical: a result of ``hlaPredict`` keeps the indices and the model's allele
    names (``levels``) and makes the lists the first time ``allele1`` / ``allele2`` is read, so that a caller who wants the
    indices, the probabilities or the dosages does not pay 2 x n_samp Python objects per call."""

    def __init__(self, locus: str, sample_id: List, allele1: Optional[List[Optional[str]]] = None,
                 allele2: Optional[List[Optional[str]]] = None, prob: Optional[np.ndarray] = None,
                 matching: Optional[np.ndarray] = None, assembly: str = "unknown", dosage: Optional[np.ndarray] = None,
                 postprob: Optional[np.ndarray] = None, pair_names: Optional[List[str]] = None,
                 h1: Optional[np.ndarray] = None, h2: Optional[np.ndarray] = None, levels: Optional[Sequence[str]] = None):
        if (allele1 is None or allele2 is None) and (h1 is None or h2 is None or levels is None):
            raise TypeError("HlaAlleleClass needs allele1 and allele2, or h1, h2 and levels")
        self.locus, self.sample_id = locus, sample_id
        self._allele1, self._allele2, self._levels = allele1, allele2, levels
        self.prob, self.matching, self.assembly = prob, matching, assembly
        self.dosage, self.postprob = dosage, postprob
        self.pair_names = [] if pair_names is None else pair_names
        self.h1, self.h2 = h1, h2

    def _names_of(self, h: np.ndarray) -> List[Optional[str]]:
        n = len(self._levels)
        lv = np.empty(n + 1, dtype=np.object_)
        lv[:n] = list(self._levels)
        lv[n] = None
        return lv.take(np.where(np.asarray(h) == NA_INTEGER, n, h)).tolist()

    @property
    def allele1(self) -> List[Optional[str]]:
        if self._allele1 is None:
            self._allele1 = self._names_of(self.h1)
        return self._allele1

    @allele1.setter
    def allele1(self, v):
        self._allele1 = v

    @property
    def allele2(self) -> List[Optional[str]]:
        if self._allele2 is None:
            self._allele2 = self._names_of(self.h2)
        return self._allele2

    @allele2.setter
    def allele2(self, v):
        self._allele2 = v

    def __repr__(self):
        return f"HlaAlleleClass(locus={self.locus!r}, {len(self.sample_id)} samples, assembly={self.assembly!r})"


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


def _model_pair_names(model: "HlaAttrBagClass") -> List[str]:
    """:func:`_pair_names` of the model's alleles, made once per model (P = nHLA(nHLA+1)/2 strings)."""
    names = model.__dict__.get("_pair_names")
    if names is None:
        names = model.__dict__["_pair_names"] = _pair_names(model.obj.hla_allele)
    return list(names)


_TYPES = ("response+dosage", "response", "prob", "response+prob")
_VOTES = ("prob", "majority")


def _as_integer(g: np.ndarray) -> np.ndarray:
    """``as.integer(snp)`` for a numeric matrix (``R/HIBAG.R:715``): int32 in the array's OWN memory order, NA / NaN /
    anything an int cannot hold -> ``NA_integer_``.  An int32 array is returned as it is -- no copy."""
    g = np.asarray(g)
    if g.dtype == np.int32:
        return g
    if g.dtype.kind == "f":
        with np.errstate(invalid="ignore"):
            gi = g.astype(np.int32, order="K")
        bad = ~((g > -2147483648.0) & (g < 2147483648.0))         # NaN, +-inf, out of range: whatever the cast made of them
        if bad.any():
            gi[bad] = NA_INTEGER
        return gi
    if g.dtype.kind in "iub":
        if g.dtype.itemsize < 4 or g.dtype.kind == "b":
            return g.astype(np.int32, order="K")
        big = (g > 2147483647) | (g < -2147483647)
        gi = g.astype(np.int32, order="K")
        if big.any():
            gi[big] = NA_INTEGER
        return gi
    raise TypeError("is.numeric(snp) is not TRUE")


def _predict_matrix(model: "HlaAttrBagClass", g: np.ndarray, sel: Optional[np.ndarray], flip: Optional[np.ndarray],
                    vote_method: int, want_dosage: bool, want_prob: bool) -> dict:
    """``PredictHLA`` on the matrix ``g`` [SNP, sample] of an ``hlaSNPGenoClass`` (or the numeric matrix handed to
    ``hlaPredict``) WITHOUT building a second matrix on the host: row ``sel[k]`` holds model SNP k (-1 = absent, ``None`` =
    row k), ``flip[k]`` reverses its allele count -- both applied on the device while the genotypes are packed.  The entry
    follows the array's memory: column-major (R's own order: the transpose view is the C side's sample-major matrix) ->
    ``hibag_hip_predict`` / ``_mapped``; row-major (numpy's default) -> ``hibag_hip_predict_snp_major``.  Bit-identical."""
    g = _as_integer(g)
    if flip is not None and not np.any(flip):
        flip = None
    if g.flags.f_contiguous:
        cohort = g.T                          # a view: [n_samp, cohort SNPs], C-contiguous
        if sel is None and flip is None:
            return model.predict_raw(cohort, vote_method, want_dosage=want_dosage, want_prob=want_prob)
        if sel is None:
            sel = np.arange(model.obj.n_snp, dtype=np.int32)
        return model.predict_mapped(cohort, sel, flip, vote_method, want_dosage=want_dosage, want_prob=want_prob)
    if not g.flags.c_contiguous:
        g = np.ascontiguousarray(g)
    return model.predict_snp_major(g, sel, flip, vote_method, want_dosage=want_dosage, want_prob=want_prob)


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

    Cost: like the reference (``R/HIBAG.R:715-748``: one ``.Call`` and O(1) R-level work per cohort besides the result's
    data frame) the host side does nothing per sample in the interpreter and copies no matrix: the genotypes go to the
    device from the caller's own memory in either memory order, SNP selection and allele flips happen on the device, and
    ``dosage`` / ``postprob`` are returned as [row, sample] VIEWS of the C side's sample-major output -- R's own memory
    order for those matrices.  What `bench.py` reports as ``api_inclusive``.
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
        if g.dtype.kind not in "iufb":
            raise TypeError("is.numeric(snp) is not TRUE")
        if g.ndim == 1:
            if g.shape[0] != obj.n_snp:
                raise ValueError("length(snp) == object$n.snp is not TRUE")
            g = g.reshape(-1, 1)
        elif g.ndim != 2 or g.shape[0] != obj.n_snp:
            raise ValueError("nrow(snp) == object$n.snp is not TRUE")
        geno_sampid = range(1, g.shape[1] + 1)
        assembly = "auto-silent"
        mat = g
    else:
        # the SNP matching / strand check (R/HIBAG.R:550-686) decides on the annotation; the rows are
        # picked and flipped on the device while the genotypes are packed (hibag_hip_predict_mapped / _snp_major)
        from .snpmatch import _row_afreq, plan_snps_for_predict
        mat = np.asarray(snp.genotype)
        if mat.ndim != 2:
            raise ValueError("'snp$genotype' must be a matrix [n.snp, n.samp]")
        map_plan = plan_snps_for_predict(obj, snp, lambda rows: _row_afreq(_as_integer(mat[rows])), match_type,
                                         allele_check, same_strand, verbose, verbose_match)
        assembly = map_plan.assembly
        geno_sampid = snp.sample_id
        if len(geno_sampid) != mat.shape[1]:
            raise ValueError("length(snp$sample.id) == ncol(snp$genotype) is not TRUE")

    n_samp = len(geno_sampid) if mat is None else mat.shape[1]
    if verbose:
        print(f"# of samples: {n_samp}", file=out)
        print(f"Kernel target: {_kernel_info or 'hip'}", file=out)

    want_prob = type in ("prob", "response+prob")
    want_dosage = type != "response"
    sel = flip = None
    if map_plan is not None:
        sel = None if map_plan.identity else map_plan.sel
        flip = map_plan.flip if (map_plan.flip is not None and np.any(map_plan.flip)) else None
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
    if devices is not None:
        # several devices: hibag_hip_predict_multi slices ONE sample-major matrix in model order over the replicas.  R's memory
        # order with the model's own SNPs is that matrix already; anything else is put in that form on the host (the model's
        # few hundred rows of the cohort)
        g = _as_integer(mat)
        if sel is None and flip is None and g.flags.f_contiguous:
            genomat = g.T
        else:
            idx = np.arange(obj.n_snp) if sel is None else np.asarray(sel)
            rows = g[np.maximum(idx, 0)]
            if flip is not None:
                fl = np.asarray(flip, bool)[:, None]
                rows = np.where(fl & (rows >= 0) & (rows <= 2), 2 - rows, rows)
            rows[idx < 0] = NA_INTEGER
            genomat = np.ascontiguousarray(rows.T, np.int32)
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
    else:
        rv = _predict_matrix(object, mat, sel, flip, vote_method, want_dosage, want_prob)

    if type == "prob":
        res = rv["postprob"].T                # [n_cell, n_samp]: a view of the sample-major output = R's memory order
        with np.errstate(invalid="ignore"):
            na_cnt = int(np.count_nonzero(rv["postprob"].sum(axis=1) <= 0))
    else:
        h1, h2 = rv["h1"], rv["h2"]
        na_cnt = int(np.count_nonzero((h1 == NA_INTEGER) | (h2 == NA_INTEGER)))
        # (allele1 / allele2: object$hla.allele[H1 + 1L] (R/HIBAG.R:729-736), made from h1 / h2 and the levels when first read)
        res = HlaAlleleClass(locus=obj.hla_locus, sample_id=list(geno_sampid), h1=h1, h2=h2, levels=obj.hla_allele,
                             prob=rv["prob"], matching=rv["matching"], assembly=assembly,
                             dosage=(rv["dosage"].T if type != "response" else None),
                             postprob=(rv["postprob"].T if want_prob else None),
                             pair_names=_model_pair_names(object) if want_prob else [])

    if na_cnt > 0:   # R/HIBAG.R:811-815
        import warnings
        warnings.warn(f"No prediction output{'s' if na_cnt > 1 else ''} for {na_cnt} individual"
                      f"{'s' if na_cnt > 1 else ''} (possibly due to missing SNPs).")
    return res
