"""The caller's side of HIBAG's GPU plugin table, for benchmarks and examples.

An unmodified HIBAG drives a plugin through ``TypeGPUExtProc`` (``inst/include/LibHLA_ext.h:358-388``):
``predict_init`` once per ``PredictHLA`` with the host's haplotype lists (``src/LibHLA.cpp:2498-2523``), then
``predict_avg_prob`` ONCE PER SAMPLE with the sample's genotype packed per classifier (``TGenotype::IntToSNP``,
``src/LibHLA.cpp:662-706``) and the classifier weights (``:2418-2431``), then ``predict_done``.  :class:`PluginHost` does
what that host code does -- nothing else -- so that the table returned by ``hibag_hip_gpu_ext_proc()`` can be driven and
timed without R.  The product's own throughput path is the batched entry (``hibag_hip_predict``); this is the drop-in."""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .model import HlaAttrBagObj


class THaplotype(C.Structure):          # inst/include/LibHLA_ext.h:261-299
    _fields_ = [("packed", C.c_uint64 * 2), ("freq", C.c_double), ("freq_f32", C.c_float), ("hla", C.c_int)]


class TGenotype(C.Structure):           # inst/include/LibHLA_ext.h:311-352
    _fields_ = [("s1", C.c_uint64 * 2), ("s2", C.c_uint64 * 2), ("boot", C.c_int), ("a1", C.c_int),
                ("a2", C.c_int), ("pad", C.c_int)]


class GPUExtProc(C.Structure):          # inst/include/LibHLA_ext.h:358-388
    _fields_ = [(n, C.c_void_p) for n in ("build_init", "build_done", "build_set_bootstrap", "build_haplomatch",
                                           "build_set_haplo_geno", "build_acc_oob", "build_acc_ib")] + [
        ("predict_init", C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(C.POINTER(THaplotype)),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int))),
        ("predict_done", C.CFUNCTYPE(None)),
        ("predict_avg_prob", C.CFUNCTYPE(None, C.POINTER(TGenotype), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)))]


assert C.sizeof(THaplotype) == 32 and C.sizeof(TGenotype) == 48


def _pack_bits(flags: np.ndarray) -> np.ndarray:
    """bool [n, 128] -> uint64 [n, 2], bit s of word s // 64 = flags[:, s]."""
    b = np.packbits(flags.astype(np.uint8), axis=1, bitorder="little")          # [n, 16] bytes
    return b.view("<u8").reshape(len(flags), 2)


class PluginHost:
    """``CAttrBag_Model::_Init_GPU_PredHLA`` / ``_PredictHLA``'s GPU branch / ``_Done_GPU_PredHLA`` for one model."""

    def __init__(self, obj: HlaAttrBagObj):
        self.obj = obj
        self.table = GPUExtProc.from_address(_lib.lib().hibag_hip_gpu_ext_proc())
        nC = len(obj.classifiers)
        self._lists = []
        for c in obj.classifiers:
            arr = (THaplotype * len(c.freq))()
            bits = np.zeros((len(c.freq), 128), bool)
            for i, h in enumerate(c.haplo):
                bits[i, :len(h)] = np.frombuffer(h.encode(), np.uint8) == ord("1")
            packed = _pack_bits(bits)
            for i in range(len(c.freq)):
                arr[i].packed[0], arr[i].packed[1] = int(packed[i, 0]), int(packed[i, 1])
                arr[i].freq = float(c.freq[i]); arr[i].freq_f32 = float(c.freq[i]); arr[i].hla = int(c.hla[i])
            self._lists.append(arr)
        ptrs = (C.POINTER(THaplotype) * nC)(*[C.cast(a, C.POINTER(THaplotype)) for a in self._lists])
        n_hap = (C.c_int * nC)(*[len(a) for a in self._lists])
        n_snp = (C.c_int * nC)(*[len(c.snpidx) for c in obj.classifiers])
        self.table.predict_init(obj.n_hla, nC, ptrs, n_hap, n_snp)
        self._open = True
        sw = np.zeros(max(obj.n_snp, 1), np.int64)               # _GetSNPWeights, src/LibHLA.cpp:2484-2496
        for c in obj.classifiers:
            sw[c.snpidx] += 1
        self._snp_weight = sw

    def pack(self, genomat: np.ndarray):
        """Host work of ``_PredictHLA`` for every sample at once: (TGenotype [n_samp][n_classifier], weights
        float64 [n_samp][n_classifier])."""
        G = np.asarray(genomat)
        n, nC = len(G), len(self.obj.classifiers)
        geno = np.zeros((n, nC, 6), np.uint64)                    # s1[2], s2[2], 16 bytes of book-keeping
        wt = np.zeros((n, nC), np.float64)
        for ci, c in enumerate(self.obj.classifiers):
            g = G[:, c.snpidx]
            ok = (g >= 0) & (g <= 2)
            s1 = np.zeros((n, 128), bool); s2 = np.ones((n, 128), bool)      # beyond the classifier's SNPs: missing (0, 1)
            k = len(c.snpidx)
            s1[:, :k] = ok & (g >= 1)                              # 0 -> (0,0), 1 -> (1,0), 2 -> (1,1), missing -> (0,1)
            s2[:, :k] = ~ok | (g == 2)
            geno[:, ci, 0:2] = _pack_bits(s1)
            geno[:, ci, 2:4] = _pack_bits(s2)
            sw = self._snp_weight[c.snpidx]
            tot = int(sw.sum())
            wt[:, ci] = (ok * sw).sum(axis=1) / tot if tot > 0 else 0.0
        return geno, wt

    def avg_prob(self, geno_row: np.ndarray, wt_row: np.ndarray, out_prob: np.ndarray, out_match: np.ndarray) -> None:
        """One ``predict_avg_prob`` call (``src/LibHLA.cpp:2433-2441``)."""
        self.table.predict_avg_prob(geno_row.ctypes.data_as(C.POINTER(TGenotype)), wt_row.ctypes.data_as(C.POINTER(C.c_double)),
                                    out_prob.ctypes.data_as(C.POINTER(C.c_double)), out_match.ctypes.data_as(C.POINTER(C.c_double)))

    def avg_prob_loop(self, geno: np.ndarray, wt: np.ndarray):
        """The host's per-sample loop in C (``hibag_hip_test_time_avg_prob``): ``predict_avg_prob`` for every row of
        ``geno`` / ``wt`` followed by the arg-max scan -- no interpreter between the calls, which is how a compiled host
        sees the route.  Returns (best posterior cell per sample, matching per sample, seconds)."""
        g = np.ascontiguousarray(geno, np.uint64)
        w = np.ascontiguousarray(wt, np.float64)
        n, nC = w.shape
        best = np.zeros(n, np.int32); match = np.zeros(n, np.float64)
        sec = C.c_double(0)
        _lib.check(_lib.lib().hibag_hip_test_time_avg_prob(g.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), n, nC,
                                                           self.obj.n_cell, best.ctypes.data_as(C.c_void_p),
                                                           match.ctypes.data_as(C.c_void_p), C.byref(sec)))
        return best, match, sec.value

    def close(self):
        if getattr(self, "_open", False):
            self.table.predict_done()
            self._open = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
