"""ctypes front-end of the CPU oracle (``oracle/libhibag_oracle.so``).

TEST INFRASTRUCTURE ONLY.  Imported by ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` -- never by ``hibag_amd``.  See the
header of ``hibag_oracle.c`` for what it restates and how it is pinned.

A model is passed as any object with ``n_hla``, ``n_snp`` and ``classifiers``
(each with ``snpidx`` 0-based, ``freq``, ``hla`` allele indices in ascending
order, ``haplo`` '0'/'1' strings) -- the in-memory form of an ``hlaAttrBagObj``
(reference: ``man/hlaAttrBagObj.Rd:9-38``).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIBAG_ORACLE_LIBRARY selects another build of the oracle (the sanitizer builds of oracle/Makefile, tools/run_sanitizers.sh)
_LIB_PATH = os.environ.get("HIBAG_ORACLE_LIBRARY") or os.path.join(_HERE, "libhibag_oracle.so")
NA_INTEGER = -2147483648

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``)."""
    srcs = [os.path.join(_HERE, f) for f in ("hibag_oracle.c", "hibag_oracle_avx2.c", "hibag_oracle_train.c", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if (force or stale) and not os.environ.get("HIBAG_ORACLE_LIBRARY"):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_mutation_table.argtypes = [_f64p]
        L.oracle_haplo_from_string.argtypes = [C.c_char_p, _u64p]
        L.oracle_int_to_snp.argtypes = [C.c_int, _i32p, _i32p, _u64p, _u64p]
        L.oracle_hamm_d.argtypes = [C.c_int, _u64p, _u64p, _u64p, _u64p]
        cls = [C.c_int, C.c_int, _i32p, _u64p, _f64p, _u64p, _u64p]
        L.oracle_post_prob2.argtypes = cls + [_f64p]
        L.oracle_post_prob2.restype = C.c_double
        L.oracle_best_guess.argtypes = cls + [_i32p]
        L.oracle_post_prob.argtypes = cls + [C.c_int, C.c_int]
        L.oracle_post_prob.restype = C.c_double
        L.oracle_prep_haplo_match.argtypes = [C.c_int, _u64p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              _u64p, _u64p, C.c_void_p]
        L.oracle_compare_hla.argtypes = [C.c_int] * 4
        L.oracle_snp_weights.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p]
        pred = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, _i32p, _u64p, _f64p,
                _i32p, C.c_int, C.c_int]
        outs = [C.c_void_p] * 6
        L.oracle_predict.argtypes = pred + outs
        L.oracle_predict_avx2_mt.argtypes = pred + [C.c_int] + outs
        _lib = L
    return _lib


def mutation_table() -> np.ndarray:
    t = np.empty(257, np.float64)
    lib().oracle_mutation_table(t)
    return t


def haplo_bits(s: str) -> np.ndarray:
    b = np.zeros(2, np.uint64)
    if lib().oracle_haplo_from_string(s.encode(), b) != 0:
        raise ValueError(f"bad haplotype string {s!r}")
    return b


def int_to_snp(geno_row: np.ndarray, index: np.ndarray):
    g = np.ascontiguousarray(geno_row, np.int32)
    ix = np.ascontiguousarray(index, np.int32)
    s1 = np.zeros(2, np.uint64)
    s2 = np.zeros(2, np.uint64)
    lib().oracle_int_to_snp(len(ix), g, ix, s1, s2)
    return s1, s2


def hamm_d(n_snp, s1, s2, h1, h2) -> int:
    a = [np.ascontiguousarray(x, np.uint64) for x in (s1, s2, h1, h2)]
    return lib().oracle_hamm_d(int(n_snp), *a)


@dataclass
class FlatModel:
    """Flat arrays the C entry points take (layout documented in hibag_oracle.c)."""
    n_hla: int
    n_classifier: int
    n_snp_total: int
    n_snp_c: np.ndarray
    snp_off: np.ndarray
    snp_index: np.ndarray
    hap_off: np.ndarray
    len_per_hla: np.ndarray
    bits: np.ndarray
    freq: np.ndarray

    def classifier(self, c: int):
        lo = int(self.hap_off[c])
        n = int(self.len_per_hla[c].sum())
        return (int(self.n_snp_c[c]), np.ascontiguousarray(self.len_per_hla[c]),
                np.ascontiguousarray(self.bits[lo:lo + n]), np.ascontiguousarray(self.freq[lo:lo + n]),
                np.ascontiguousarray(self.snp_index[self.snp_off[c]:self.snp_off[c] + self.n_snp_c[c]]))


def flatten(model) -> FlatModel:
    cls = model.classifiers
    n_hla = int(model.n_hla)
    n_snp_c = np.array([len(c.snpidx) for c in cls], np.int32)
    n_hap = np.array([len(c.freq) for c in cls], np.int32)
    snp_off = np.concatenate([[0], np.cumsum(n_snp_c)]).astype(np.int32)
    hap_off = np.concatenate([[0], np.cumsum(n_hap)]).astype(np.int32)
    snp_index = (np.concatenate([np.asarray(c.snpidx, np.int32) for c in cls])
                 if cls else np.zeros(0, np.int32)).astype(np.int32)
    freq = (np.concatenate([np.asarray(c.freq, np.float64) for c in cls])
            if cls else np.zeros(0, np.float64)).astype(np.float64)
    bits = np.zeros((int(hap_off[-1]), 2), np.uint64)
    len_per_hla = np.zeros((len(cls), n_hla), np.int32)
    for ci, c in enumerate(cls):
        hla = np.asarray(c.hla, np.int64)
        if len(hla) and (np.any(np.diff(hla) < 0) or hla.min() < 0 or hla.max() >= n_hla):
            raise ValueError("haplotypes must be grouped by ascending allele index")
        len_per_hla[ci] = np.bincount(hla, minlength=n_hla)
        for k, s in enumerate(c.haplo):
            bits[hap_off[ci] + k] = haplo_bits(s)
    return FlatModel(n_hla, len(cls), int(model.n_snp), n_snp_c, snp_off, np.ascontiguousarray(snp_index),
                     hap_off, len_per_hla, bits, np.ascontiguousarray(freq))


def post_prob2(fm: FlatModel, c: int, s1, s2):
    n_snp, lens, bits, freq, _ = fm.classifier(c)
    P = fm.n_hla * (fm.n_hla + 1) // 2
    prob = np.zeros(P, np.float64)
    total = lib().oracle_post_prob2(fm.n_hla, n_snp, lens, bits.reshape(-1), freq,
                                    np.ascontiguousarray(s1, np.uint64), np.ascontiguousarray(s2, np.uint64), prob)
    return prob, total


def best_guess(fm: FlatModel, c: int, s1, s2):
    n_snp, lens, bits, freq, _ = fm.classifier(c)
    out = np.zeros(2, np.int32)
    lib().oracle_best_guess(fm.n_hla, n_snp, lens, bits.reshape(-1), freq,
                            np.ascontiguousarray(s1, np.uint64), np.ascontiguousarray(s2, np.uint64), out)
    return int(out[0]), int(out[1])


def post_prob(fm: FlatModel, c: int, s1, s2, a1: int, a2: int) -> float:
    n_snp, lens, bits, freq, _ = fm.classifier(c)
    return lib().oracle_post_prob(fm.n_hla, n_snp, lens, bits.reshape(-1), freq,
                                  np.ascontiguousarray(s1, np.uint64), np.ascontiguousarray(s2, np.uint64),
                                  int(a1), int(a2))


def prep_haplo_match(fm: FlatModel, c: int, s1, s2, a1: int, a2: int) -> np.ndarray:
    n_snp, lens, bits, _, _ = fm.classifier(c)
    if a1 > a2:
        a1, a2 = a2, a1
    st = np.concatenate([[0], np.cumsum(lens)])
    args = (n_snp, bits.reshape(-1), int(st[a1]), int(lens[a1]), int(st[a2]), int(lens[a2]),
            np.ascontiguousarray(s1, np.uint64), np.ascontiguousarray(s2, np.uint64))
    k = lib().oracle_prep_haplo_match(*args, None)
    out = np.zeros((k, 2), np.int32)
    if k:
        lib().oracle_prep_haplo_match(*args, out.ctypes.data_as(C.c_void_p))
    return out


def compare_hla(p1, p2, t1, t2) -> int:
    return lib().oracle_compare_hla(int(p1), int(p2), int(t1), int(t2))


def predict(fm: FlatModel, genomat: np.ndarray, vote_method: int = 1, want_dosage=True, want_prob=True,
            avx2: bool = False, n_threads: int = 1):
    """CAttrBag_Model::PredictHLA restated.  ``genomat`` is [n_samp, n_snp] int32."""
    g = np.ascontiguousarray(genomat, np.int32)
    if g.ndim != 2 or g.shape[1] != fm.n_snp_total:
        raise ValueError("genomat must be [n_samp, n_snp]")
    n = g.shape[0]
    P = fm.n_hla * (fm.n_hla + 1) // 2
    out = dict(h1=np.zeros(n, np.int32), h2=np.zeros(n, np.int32), prob=np.zeros(n, np.float64),
               matching=np.zeros(n, np.float64))
    if want_dosage:
        out["dosage"] = np.zeros((n, fm.n_hla), np.float64)
    if want_prob:
        out["postprob"] = np.zeros((n, P), np.float64)

    def ptr(k):
        return out[k].ctypes.data_as(C.c_void_p) if k in out else None

    head = (fm.n_hla, fm.n_classifier, fm.n_snp_total, fm.n_snp_c, fm.snp_off[:-1].copy(), fm.snp_index,
            fm.hap_off[:-1].copy(), fm.len_per_hla.reshape(-1), fm.bits.reshape(-1), fm.freq,
            g.reshape(-1), n, int(vote_method))
    tail = (ptr("h1"), ptr("h2"), ptr("prob"), ptr("matching"), ptr("dosage"), ptr("postprob"))
    if avx2:
        rc = lib().oracle_predict_avx2_mt(*head, int(n_threads), *tail)
    else:
        rc = lib().oracle_predict(*head, *tail)
    if rc == -1:
        raise ValueError("Invalid 'vote_method'.")
    if rc != 0:
        raise RuntimeError(f"oracle_predict failed ({rc})")
    return out


def conv_bed(image: bytes, n_samp: int, n_snp: int, snp_flag) -> np.ndarray:
    """``HIBAG_ConvBED`` on the bytes of a BED file -> int32 [n_samp, n_save]."""
    L = lib()
    L.oracle_conv_bed.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, _i32p, _i32p]
    flag = np.ascontiguousarray(np.asarray(snp_flag) != 0, np.int32)
    n_save = int(flag.sum())
    out = np.zeros((n_samp, max(n_save, 1)), np.int32)
    rc = L.oracle_conv_bed(image, len(image), n_samp, n_snp, n_save, flag, out)
    if rc != 0:
        raise ValueError({-1: "Invalid prefix in the PLINK BED file.", -2: "BED image too short"}[rc])
    return out[:, :n_save]


def train(genomat: np.ndarray, h1, h2, n_hla: int, nclassifier: int, mtry: int, prune: bool = True, seed: int = 100):
    """``set.seed(seed)`` + ``CAttrBag_Model::BuildClassifiers`` (hibag_oracle_train.c).
    ``genomat`` int32 [n_samp, n_snp]; ``h1``/``h2`` 0-based allele indices.  Returns one dict per
    classifier: snpidx (0-based), samp_num, freq, hla, bits [n_haplo, 2] uint64, haplo strings, acc."""
    L = lib()
    L.oracle_train_new.restype = C.c_void_p
    L.oracle_train_new.argtypes = [C.c_int, C.c_int, _i32p, C.c_int, _i32p, _i32p, C.c_uint]
    L.oracle_train_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.oracle_train_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.oracle_train_get.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _f64p, _i32p, _u64p]
    L.oracle_train_free.argtypes = [C.c_void_p]
    g = np.ascontiguousarray(genomat, np.int32)
    a1 = np.ascontiguousarray(h1, np.int32)
    a2 = np.ascontiguousarray(h2, np.int32)
    n_samp, n_snp = g.shape
    h = C.c_void_p(L.oracle_train_new(n_snp, n_samp, g, n_hla, a1, a2, seed & 0xFFFFFFFF))
    try:
        n = L.oracle_train_run(h, nclassifier, mtry, int(bool(prune)))
        out = []
        for i in range(n):
            ns, nh, acc = C.c_int(0), C.c_int(0), C.c_double(0)
            L.oracle_train_info(h, i, C.byref(ns), C.byref(nh), C.byref(acc))
            snpidx = np.zeros(max(ns.value, 1), np.int32)
            samp = np.zeros(n_samp, np.int32)
            freq = np.zeros(max(nh.value, 1), np.float64)
            hla = np.zeros(max(nh.value, 1), np.int32)
            bits = np.zeros((max(nh.value, 1), 2), np.uint64)
            L.oracle_train_get(h, i, snpidx, samp, freq, hla, bits)
            k = ns.value
            bits = bits[:nh.value]
            haplo = ["".join("1" if (int(b[j >> 6]) >> (j & 63)) & 1 else "0" for j in range(k)) for b in bits]
            out.append(dict(snpidx=snpidx[:k], samp_num=samp, freq=freq[:nh.value], hla=hla[:nh.value], bits=bits,
                            haplo=haplo, acc=acc.value))
        return out
    finally:
        L.oracle_train_free(h)

