"""Host-side mirror of the reference's training entry ``hlaAttrBagging``
(``R/HIBAG.R:48-275``): sample / SNP preparation in front of the native driver
(``hibag_hip_trainer_*`` in ``include/hibag_hip.h`` -- bootstrap, greedy SNP selection
and EM on the host in C++, haplotype-pair scoring on the device), and the assembly of the
resulting ``hlaAttrBagObj``.

The reference draws from R's global random stream; :func:`set_seed` /
:class:`RRandom` reproduce ``set.seed()`` and R's default Mersenne-Twister, so
``set_seed(100); hlaAttrBagging(...)`` gives the model R gives (the reference's
``inst/extdata/OutOfBag.RData`` is reproduced bit for bit in the tests).
"""

from __future__ import annotations

import ctypes as C
import math
import os
import sys
import warnings
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib
from ._lib import HibagHipError
from .hibag import HlaAlleleClass, HlaAttrBagClass, hlaPredict
from .model import NA_INTEGER, Classifier, HlaAttrBagObj, HlaSNPGeno


class RRandom:
    """R's default uniform generator: Mersenne-Twister MT19937 seeded like ``set.seed()``
    (R ``src/main/RNG.c``: ``Randomize`` scrambling, ``MT_genrand``, ``fixup``)."""

    def __init__(self, seed: Optional[int] = None):
        self.mt = [0] * 624
        self.mti = 625
        if seed is not None:
            self.set_seed(seed)

    def set_seed(self, seed: int) -> None:
        seed &= 0xFFFFFFFF
        for _ in range(50):
            seed = (69069 * seed + 1) & 0xFFFFFFFF
        for j in range(625):
            seed = (69069 * seed + 1) & 0xFFFFFFFF
            if j > 0:
                self.mt[j - 1] = seed
        self.mti = 624

    def unif_rand(self) -> float:
        mt = self.mt
        if self.mti >= 624:
            if self.mti == 625:
                self.set_seed(4357)
            for kk in range(624):
                y = (mt[kk] & 0x80000000) | (mt[(kk + 1) % 624] & 0x7FFFFFFF)
                mt[kk] = mt[(kk + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.mti = 0
        y = mt[self.mti]
        self.mti += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        v = y * 2.3283064365386963e-10
        eps = 2.328306437080797e-10
        if v <= 0.0:
            return 0.5 * eps
        if 1.0 - v <= 0.0:
            return 1.0 - 0.5 * eps
        return v


_R = RRandom()


def set_seed(seed: int) -> None:
    """``set.seed(seed)`` for the stream :func:`hlaAttrBagging` draws from."""
    _R.set_seed(int(seed))


def _allele_key(s: str):
    """Sort key of ``HIBAG_SortAlleleStr`` (``src/HIBAG.cpp:79-148``): fields split at ':',
    each compared as (leading integer, suffix); a missing number sorts last."""
    key = []
    for f in s.split(":"):
        i = 0
        while i < len(f) and f[i].isdigit():
            i += 1
        key.append((int(f[:i]) if i else 2 ** 31 - 1, f[i:]))
    return key


def hlaUniqueAllele(hla: Sequence[Optional[str]]) -> List[str]:
    """``hlaUniqueAllele`` for a character vector (``R/DataUtilities.R:1139-1150``)."""
    seen, out = set(), []
    for a in hla:
        if a is not None and a not in seen:
            seen.add(a)
            out.append(a)
    return sorted(out, key=_allele_key)


def hlaAllele(sample_id: Sequence, H1: Sequence[Optional[str]], H2: Sequence[Optional[str]], locus: str = "any",
              assembly: str = "auto-silent") -> HlaAlleleClass:
    """``hlaAllele`` (``R/DataUtilities.R:1176-1240``), the part training needs."""
    if not (len(sample_id) == len(H1) == len(H2)):
        raise ValueError("length(sample.id) == length(H1) is not TRUE")
    from .bed import _hla_assembly
    return HlaAlleleClass(locus=locus, sample_id=list(sample_id), allele1=list(H1), allele2=list(H2),
                          assembly=_hla_assembly(assembly))


def _mtry(mtry, n_snp: int) -> int:
    """``R/HIBAG.R:180-208``."""
    if isinstance(mtry, str):
        if mtry == "sqrt":
            m = math.ceil(math.sqrt(n_snp))
        elif mtry == "all":
            m = n_snp
        elif mtry == "one":
            m = 1
        else:
            raise ValueError("Invalid mtry!")
    else:
        v = float(mtry)
        if math.isfinite(v):
            if 0 < v < 1:
                v = n_snp * v
            m = min(math.ceil(v), n_snp)
        else:
            m = math.ceil(math.sqrt(n_snp))
    return max(int(m), 1)


_UNIF = C.CFUNCTYPE(C.c_double, C.c_void_p)


class _Trainer:
    """``hibag_hip_trainer`` handle (``HIBAG_Training`` ... ``HIBAG_Close``)."""

    def __init__(self, genomat: np.ndarray, h1: np.ndarray, h2: np.ndarray, n_hla: int):
        L = _lib.lib()
        g = np.ascontiguousarray(genomat, np.int32)
        self.n_samp, self.n_snp = g.shape
        a1 = np.ascontiguousarray(h1, np.int32)
        a2 = np.ascontiguousarray(h2, np.int32)
        h = L.hibag_hip_trainer_new(self.n_snp, self.n_samp, g.ctypes.data_as(C.c_void_p), int(n_hla),
                                    a1.ctypes.data_as(C.c_void_p), a2.ctypes.data_as(C.c_void_p))
        if not h:
            raise HibagHipError(-1, L.hibag_hip_last_error().decode())
        self._h = C.c_void_p(h)
        self._cb = None

    def close(self):
        if self._h is not None:
            _lib.lib().hibag_hip_trainer_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_threads(self, n_threads: int):
        """Host threads for the EM fits (``nthread`` of ``hlaAttrBagging``); <= 0 restores the default."""
        _lib.check(_lib.lib().hibag_hip_trainer_set_threads(self._h, int(n_threads)))

    def set_em_mode(self, mode: str):
        """Where the EM fits run: "auto" (the device where the trainer has two host threads or fewer), "host", "device"."""
        _lib.check(_lib.lib().hibag_hip_trainer_set_em_mode(self._h, {"auto": 0, "host": 1, "device": 2}[mode]))

    def set_shared(self, shared: bool = True):
        """The trainer runs beside others of this process: its device work goes through the device's combiners -- one fused
        launch per kind of operation for all of them (``csrc/hibag_combine.h``).  Same classifiers, bit for bit."""
        _lib.check(_lib.lib().hibag_hip_trainer_set_shared(self._h, int(bool(shared))))

    @property
    def threads(self) -> int:
        return int(_lib.lib().hibag_hip_trainer_threads(self._h))

    def set_seed(self, seed: int):
        _lib.check(_lib.lib().hibag_hip_trainer_set_seed(self._h, C.c_uint32(int(seed) & 0xFFFFFFFF)))

    def set_rng(self, rng: RRandom):
        self._cb = _UNIF(lambda _ctx: rng.unif_rand())
        _lib.check(_lib.lib().hibag_hip_trainer_set_rng(self._h, self._cb, None))

    def new_classifiers(self, nclassifier: int, mtry: int, prune: bool, verbose: bool, verbose_detail: bool):
        sys.stdout.flush()
        _lib.check(_lib.lib().hibag_hip_trainer_new_classifiers(self._h, int(nclassifier), int(mtry), int(bool(prune)),
                                                                int(bool(verbose)), int(bool(verbose_detail))))

    def classifiers(self) -> List[Classifier]:
        L = _lib.lib()
        out = []
        for i in range(L.hibag_hip_trainer_n_classifier(self._h)):
            k, nh = C.c_int(0), C.c_int(0)
            _lib.check(L.hibag_hip_trainer_classifier_dims(self._h, i, C.byref(k), C.byref(nh)))
            snpidx = np.zeros(max(k.value, 1), np.int32)
            samp = np.zeros(self.n_samp, np.int32)
            freq = np.zeros(max(nh.value, 1), np.float64)
            hla = np.zeros(max(nh.value, 1), np.int32)
            bits = np.zeros((max(nh.value, 1), 2), np.uint64)
            acc = C.c_double(0)
            _lib.check(L.hibag_hip_trainer_classifier_get(
                self._h, i, snpidx.ctypes.data_as(C.c_void_p), samp.ctypes.data_as(C.c_void_p),
                freq.ctypes.data_as(C.c_void_p), hla.ctypes.data_as(C.c_void_p), bits.ctypes.data_as(C.c_void_p),
                C.byref(acc)))
            kk = k.value
            haplo = ["".join("1" if (int(b[j >> 6]) >> (j & 63)) & 1 else "0" for j in range(kk)) for b in bits[:nh.value]]
            out.append(Classifier(snpidx=snpidx[:kk], freq=freq[:nh.value], hla=hla[:nh.value], haplo=haplo,
                                  samp_num=samp, outofbag_acc=acc.value))
        return out


def _row_mean_half(g: np.ndarray) -> np.ndarray:
    ok = (g >= 0) & (g <= 2) if g.dtype.kind != "f" else np.isfinite(g)
    cnt = ok.sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(cnt > 0, np.where(ok, g, 0).sum(axis=1) / cnt, np.nan) * 0.5


def hlaAttrBagging(hla: HlaAlleleClass, snp: HlaSNPGeno, nclassifier: int = 100,
                   mtry: Union[str, float, int] = "sqrt", prune: bool = True, na_rm: bool = True,
                   mono_rm: bool = True, maf: float = float("nan"), nthread: Optional[int] = None, verbose: bool = True,
                   verbose_detail: bool = False, rng: Optional[RRandom] = None, grow=None,
                   device: Optional[int] = None) -> HlaAttrBagClass:
    """``hlaAttrBagging`` (``R/HIBAG.R:48-275``).  ``nthread``: host threads for the EM fits of a growth step
    (the scoring runs on the device); ``None`` = the library's default, the usable CPUs divided by the ranks
    sharing the host (``hibag_hip_trainer_set_threads``).  ``rng`` defaults to the module's
    R-compatible stream (see :func:`set_seed`).  ``grow`` (internal) replaces the single-device
    call of the native driver, see :func:`hlaParallelAttrBagging`.  ``device`` (extension): the HIP
    device that trains and holds the returned model (default: the thread's current selection)."""
    if device is not None:
        _lib.check(_lib.lib().hibag_hip_set_device(int(device)))
    if not isinstance(hla, HlaAlleleClass):
        raise TypeError("inherits(hla, \"hlaAlleleClass\") is not TRUE")
    if not isinstance(snp, HlaSNPGeno):
        raise TypeError("inherits(snp, \"hlaSNPGenoClass\") is not TRUE")
    if verbose_detail:
        verbose = True
    nclassifier = 0 if nclassifier is None else int(nclassifier)
    with_matching = nclassifier > 0
    if not with_matching:
        nclassifier = -nclassifier or 1

    in_snp = {s: i for i, s in enumerate(snp.sample_id)}
    samp_id = [s for s in dict.fromkeys(hla.sample_id) if s in in_snp]          # intersect()
    pos = {}
    for i, s in enumerate(hla.sample_id):
        pos.setdefault(s, i)
    a1 = [hla.allele1[pos[s]] for s in samp_id]
    a2 = [hla.allele2[pos[s]] for s in samp_id]
    if any(x is None or y is None for x, y in zip(a1, a2)):
        if not na_rm:
            raise ValueError("There are missing HLA alleles!")
        warnings.warn("There are missing HLA alleles, and the corresponding samples have been removed.")
        keep = [x is not None and y is not None for x, y in zip(a1, a2)]
        samp_id = [s for s, k in zip(samp_id, keep) if k]
        a1 = [x for x, k in zip(a1, keep) if k]
        a2 = [x for x, k in zip(a2, keep) if k]
    if not samp_id:
        raise ValueError("There is no common sample between 'hla' and 'snp'.")

    geno = np.asarray(snp.genotype)[:, [in_snp[s] for s in samp_id]]
    if geno.dtype.kind == "f":
        geno = np.where(np.isfinite(geno), geno, NA_INTEGER)
    geno = geno.astype(np.int32)
    snp_id, snp_pos, snp_allele = list(snp.snp_id), snp.snp_position, list(snp.snp_allele)

    msg = ""
    if mono_rm or math.isfinite(maf):
        msg = f"    MAF threshold: {'NaN' if math.isnan(maf) else maf}\n"
        mf = _row_mean_half(geno)
        mf = np.minimum(mf, 1 - mf)
        mf[~np.isfinite(mf)] = 0
        sel = np.ones(len(mf), bool)
        if mono_rm:
            n0 = int(sel.sum())
            sel &= mf > 0
            a = n0 - int(sel.sum())
            if a > 0:
                msg += f"    excluding {a} monomorphic SNP{'s' if a > 1 else ''}\n"
        if math.isfinite(maf):
            n0 = int(sel.sum())
            sel &= mf >= maf
            a = n0 - int(sel.sum())
            if a > 0:
                msg += f"    excluding {a} SNP{'s' if a > 1 else ''} for MAF threshold\n"
        if not sel.all():
            ix = np.where(sel)[0]
            snp_id = [snp_id[i] for i in ix]
            snp_pos = None if snp_pos is None else np.asarray(snp_pos)[ix]
            snp_allele = [snp_allele[i] for i in ix]
            geno = geno[ix]

    n_snp, n_samp = geno.shape
    if n_snp <= 0:
        raise ValueError("There is no valid SNP markers.")
    HUA = hlaUniqueAllele(a1 + a2)
    lut = {a: i for i, a in enumerate(HUA)}
    H1 = np.array([lut[a] for a in a1], np.int32)
    H2 = np.array([lut[a] for a in a2], np.int32)
    m = _mtry(mtry, n_snp)

    if verbose:
        print(f"Build a HIBAG model with {nclassifier} individual classifier{'s' if nclassifier > 1 else ''}:")
        print(msg, end="")
        print(f"    # of SNPs randomly sampled as candidates for each selection: {m}")
        print(f"    # of SNPs: {n_snp}\n    # of samples: {n_samp}")
        print(f"    # of unique {'KIR' if hla.locus.startswith('KIR') else 'HLA'} alleles: {len(HUA)}")

    if grow is None:
        tr = _Trainer(np.ascontiguousarray(geno.T), H1, H2, len(HUA))
        try:
            tr.set_rng(_R if rng is None else rng)
            if nthread is not None:
                tr.set_threads(int(nthread))
            tr.new_classifiers(nclassifier, m, prune, verbose, verbose_detail)
            classifiers = tr.classifiers()
        finally:
            tr.close()
    else:
        classifiers = grow(np.ascontiguousarray(geno.T), H1, H2, len(HUA), nclassifier, m, prune)

    counts = np.bincount(np.concatenate([H1, H2]), minlength=len(HUA)).astype(np.float64)
    obj = HlaAttrBagObj(
        n_samp=n_samp, n_snp=n_snp, hla_allele=HUA, classifiers=classifiers, hla_locus=hla.locus,
        sample_id=samp_id, snp_id=snp_id, snp_position=snp_pos, snp_allele=snp_allele,
        snp_allele_freq=_row_mean_half(geno), hla_freq=counts / counts.sum(),
        assembly=snp.assembly or "unknown", matching=None, appendix=None)
    mod = HlaAttrBagClass(obj, device=device)
    if with_matching:
        if verbose:
            print("Calculating matching proportion:")
        pd = hlaPredict(mod, snp, match_type="Pos+Allele", verbose=False)
        obj.matching = np.asarray(pd.matching)
        if verbose:
            acc = np.mean([c.outofbag_acc for c in classifiers]) * 100
            print(f"Out-of-bag accuracy: {acc:.2f}%")
    return mod


def grow_classifier_sharded(grow_fn, nclassifier: int, group=None) -> List[Classifier]:
    """Classifiers are independent given their random draws (``src/LibHLA.cpp:2274-2304``):
    every rank grows its share with ``grow_fn(count, rank)`` and the shares are concatenated
    in rank order on every rank (one ``all_gather_object`` of the small classifier records --
    the counterpart of ``hlaCombineModelObj``, ``R/HIBAG.R:1069-1114``).  No collective on the
    compute path."""
    import torch.distributed as dist
    from .dist import shard_bounds

    live = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if live else 1
    rank = dist.get_rank(group) if live else 0
    lo, hi = shard_bounds(int(nclassifier), world, rank)
    mine = grow_fn(hi - lo, rank)
    if world == 1:
        return list(mine)
    parts: List = [None] * world
    dist.all_gather_object(parts, list(mine), group=group)
    return [c for part in parts for c in part]


def _usable_cpus() -> int:
    """CPUs the process may use: the affinity mask capped by the cgroup CPU quota (what the native trainer's default is made of)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:                                                                        # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def grow_concurrently(genomat, h1, h2, n_hla: int, nclassifier: int, mtry: int, prune: bool, n_trainers: int,
                      threads_per_trainer: int, seed: int, device: Optional[int] = None, em: str = "auto",
                      combine: bool = True, thread_budget: int = 0) -> List[Classifier]:
    """``n_trainers`` independent trainers of ONE process side by side on one device, each driven by its own host thread
    (the native calls release the interpreter lock; the library keeps its training state and its default stream per host
    thread) with ``threads_per_trainer`` host threads of its own for the EM fits: while one trainer's candidates are
    fitted on the host, another's are scored on the device.  Trainer r grows ``shard_bounds(nclassifier, n_trainers, r)``
    classifiers from R's Mersenne-Twister seeded with ``seed + r`` -- one stream per worker like ``hlaParallelAttrBagging``'s
    (``R/HIBAG.R:329-390``), though not R's own cluster streams (L'Ecuyer-CMRG via ``clusterSetRNGStream``) -- and the shares are concatenated in
    trainer order.  Every classifier equals what a serial trainer grows from the same stream.

    ``combine`` (default): the trainers share the device through its combiners -- one fused launch per kind of operation
    (pair lists, EM fits, scoring) for all of them instead of a stream each (``csrc/hibag_combine.h``).  ``thread_budget`` > 0:
    at most that many of the trainers' host threads are runnable at a time -- a trainer waiting for the device gives its
    slot up -- so ``n_trainers`` may be far larger than the host threads the process is allowed (sixteen trainers on the two
    threads a rank of an eight-GPU node gets)."""
    import threading
    from .dist import shard_bounds
    k = max(1, min(int(n_trainers), max(int(nclassifier), 1)))
    if device is None:
        device = int(_lib.lib().hibag_hip_get_device())      # the caller's selection: the workers' threads start from the default
    # a list of devices: trainer r on devices[r % len] -- the GPUs of a node from ONE process (each device has combiners of its own)
    devices = [int(d) for d in device] if isinstance(device, (list, tuple)) else [int(device)]
    if not devices:
        raise ValueError("no device given")
    parts: List = [None] * k
    errs: List = [None] * k

    def work(r: int) -> None:
        try:
            _lib.check(_lib.lib().hibag_hip_set_device(devices[r % len(devices)]))     # (the selection is per host thread)
            lo, hi = shard_bounds(int(nclassifier), k, r)
            if hi <= lo:
                parts[r] = []
                return
            tr = _Trainer(genomat, h1, h2, n_hla)
            try:
                tr.set_em_mode(em)
                tr.set_threads(int(threads_per_trainer))
                tr.set_shared(bool(combine) and k > 1)
                tr.set_seed(int(seed) + r)
                tr.new_classifiers(hi - lo, mtry, prune, False, False)
                parts[r] = tr.classifiers()
            finally:
                tr.close()
        except BaseException as e:                                             # noqa: BLE001 -- re-raised by the caller's thread
            errs[r] = e

    _lib.lib().hibag_hip_train_set_thread_budget(int(thread_budget) if combine and k > 1 else 0)
    ths = [threading.Thread(target=work, args=(r,), name=f"hibag-trainer-{r}") for r in range(k)]
    try:
        for t in ths:
            t.start()
    finally:
        for t in ths:
            if t.ident is not None:
                t.join()
        _lib.lib().hibag_hip_train_set_thread_budget(0)
    for e in errs:
        if e is not None:
            raise e
    return [c for part in parts for c in part]


def hlaConcurrentAttrBagging(hla: HlaAlleleClass, snp: HlaSNPGeno, nclassifier: int = 100,
                             mtry: Union[str, float, int] = "sqrt", prune: bool = True, na_rm: bool = True,
                             mono_rm: bool = True, maf: float = float("nan"), n_trainers: Optional[int] = None,
                             nthread: Optional[int] = None, seed: Optional[int] = None, verbose: bool = True,
                             device: Optional[int] = None) -> HlaAttrBagClass:
    """``hlaParallelAttrBagging``'s decomposition (``R/HIBAG.R:293-440``: independent workers, one random stream each, the
    classifiers combined in worker order) inside ONE process on ONE device: ``n_trainers`` trainers side by side
    (:func:`grow_concurrently`), their device work fused into one launch per kind of operation, their EM fits on the device,
    and at most ``nthread`` of their host threads runnable at a time (default: the usable CPUs, at most four -- more buys
    nothing: a trainer's thread sleeps while the device works).  A single trainer leaves the device idle most of a growth
    step; sixteen keep it busy on the host threads of four.  ``n_trainers=None``: per device one trainer for every six
    classifiers, at most 32 -- sixteen for a model of 100 (with fewer classifiers each, the trainers that finish first leave the
    device to the last ones: 16 trainers 178-199 classifiers/s, 32 trainers 155), 32 for jobs of 192 classifiers or more (250
    against 216; ``profiles/r06_notes.txt`` item 4k).  The model depends on the number of trainers (stream r grows share r).  ``device``: a device index, or a LIST of them -- trainer r then runs
    on ``device[r % len(device)]``: the GPUs of a node from one process (BASELINE config 5's "in parallel across 8 GPUs" without a
    process per GPU; give ``n_trainers`` a multiple of the list's length, sixteen per device, and ``nthread`` accordingly); the
    model is kept on the first.

    Random streams: trainer r draws from R's Mersenne-Twister seeded with ``seed + r``.  These are NOT the streams of R's
    cluster workers -- ``hlaParallelAttrBagging`` sets those with ``parallel::clusterSetRNGStream`` (L'Ecuyer-CMRG,
    ``R/HIBAG.R:338``) -- so the model differs from what R's workers would grow, as two runs of the reference with
    different cluster sizes differ from each other.  ``seed=None`` (default): one integer drawn from the module's stream
    (:func:`set_seed`), so that repeated calls give different models and ``set_seed(s)`` before the call makes it
    reproducible -- like ``hlaAttrBagging``."""
    total = int(nthread) if nthread else min(4, _usable_cpus())
    if seed is None:
        seed = int(_R.unif_rand() * 2147483647.0)
    if n_trainers is None:
        n_dev = len(device) if isinstance(device, (list, tuple)) and device else 1
        n_trainers = n_dev * min(32, max(1, int(nclassifier) // (6 * n_dev)))

    def grow(genomat, h1, h2, n_hla, n, m, pr):
        return grow_concurrently(genomat, h1, h2, n_hla, n, m, pr, n_trainers, 1, seed, device, em="device",
                                 combine=True, thread_budget=total)

    first = device[0] if isinstance(device, (list, tuple)) and device else device
    return hlaAttrBagging(hla, snp, nclassifier=nclassifier, mtry=mtry, prune=prune, na_rm=na_rm, mono_rm=mono_rm,
                          maf=maf, verbose=verbose, grow=grow, device=first)


def hlaParallelAttrBagging(cl, hla: HlaAlleleClass, snp: HlaSNPGeno, nclassifier: int = 100,
                           mtry: Union[str, float, int] = "sqrt", prune: bool = True, na_rm: bool = True,
                           mono_rm: bool = True, maf: float = float("nan"), verbose: bool = True,
                           verbose_detail: bool = False, seed: Optional[int] = None, group=None,
                           device: Optional[int] = None) -> HlaAttrBagClass:
    """``hlaParallelAttrBagging`` (``R/HIBAG.R:293-440``) for one process per GPU: ``cl`` is
    accepted for signature compatibility; the "cluster" is the initialised
    ``torch.distributed`` process group (RCCL on a GPU box).  Like the reference's workers
    (``clusterSetRNGStream``) every rank draws from its own stream: R's Mersenne-Twister seeded
    with ``seed + rank`` -- so the model differs from a serial run with the same seed, as it
    does in the reference.  Every rank returns the complete model.  Each rank trains and keeps its model
    on its own GPU: ``device`` defaults to ``LOCAL_RANK`` when a process group is live."""
    import os
    import torch.distributed as dist
    live = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if live else 0
    base = 0 if seed is None else int(seed)
    if device is None and live and _lib.lib().hibag_hip_device_count() > 1:
        device = int(os.environ.get("LOCAL_RANK", "0"))

    def grow(genomat, h1, h2, n_hla, n, m, pr):
        def grow_fn(count, r):
            if count <= 0:
                return []
            tr = _Trainer(genomat, h1, h2, n_hla)
            try:
                tr.set_seed(base + r)
                tr.new_classifiers(count, m, pr, verbose and rank == 0, verbose_detail and rank == 0)
                return tr.classifiers()
            finally:
                tr.close()
        return grow_classifier_sharded(grow_fn, n, group)

    return hlaAttrBagging(hla, snp, nclassifier=nclassifier, mtry=mtry, prune=prune, na_rm=na_rm, mono_rm=mono_rm,
                          maf=maf, verbose=verbose and rank == 0, verbose_detail=verbose_detail, grow=grow, device=device)

