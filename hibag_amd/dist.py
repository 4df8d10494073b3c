"""Multi-GPU host logic: one process per GPU (`torch.distributed`, backend
"nccl" = RCCL on ROCm; "gloo" in the CPU tests).

Two ways to spread `hlaPredict()` over ranks:

* **sample sharding** (default): samples are independent
  (``src/LibHLA.cpp:2362-2411``), so every rank predicts a contiguous slice with
  the full model.  No collective on the data path; the only communication is
  the optional gather of the results.  Bit-identical to a single GPU.
* **classifier sharding**: every rank holds a subset of the classifiers, writes
  the un-normalised partial ensemble sums, ONE sum all-reduce merges them, then
  every rank finishes (arg-max, dosage, ...).  Changes the order in which
  classifier contributions are added, so it is held to 1e-10 relative with
  identical calls (wherever the two best cells of a sample do not tie to within
  rounding: a call is the first strict maximum), not to bit equality.

The compute is injected as callables so that the orchestration can be tested
with gloo on CPU; on a GPU box the callables are the HIP entry points of
:class:`hibag_amd.hibag.HlaAttrBagClass`.
"""

from __future__ import annotations

import copy
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from .model import HlaAttrBagObj


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `n` items owned by `rank` (sizes differ by at most 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def snp_weights(obj: HlaAttrBagObj) -> np.ndarray:
    """Number of classifiers using each SNP (``_GetSNPWeights``, ``src/LibHLA.cpp:2484-2496``)."""
    w = np.zeros(obj.n_snp, np.int32)
    for c in obj.classifiers:
        np.add.at(w, c.snpidx, 1)
    return w


def classifier_shard(obj: HlaAttrBagObj, world: int, rank: int) -> Tuple[HlaAttrBagObj, np.ndarray]:
    """The rank's sub-model (a contiguous run of classifiers, order kept) and the
    FULL model's SNP weights, which the classifier weights of
    ``src/LibHLA.cpp:2418-2431`` depend on."""
    lo, hi = shard_bounds(len(obj.classifiers), world, rank)
    sub = copy.copy(obj)
    sub.classifiers = obj.classifiers[lo:hi]
    return sub, snp_weights(obj)


def predict_sample_sharded(predict_fn: Callable[[np.ndarray], Dict[str, np.ndarray]], genomat: np.ndarray,
                           gather: bool = True, group=None) -> Optional[Dict[str, np.ndarray]]:
    """Every rank runs ``predict_fn`` on its slice of ``genomat`` ([n_samp, n_snp]);
    with ``gather`` the slices are concatenated on every rank (all_gather of
    padded blocks).  Works with an uninitialised process group (world = 1)."""
    import torch
    import torch.distributed as dist

    live = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if live else 1
    rank = dist.get_rank(group) if live else 0
    n = genomat.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    mine = predict_fn(genomat[lo:hi])
    if not gather or world == 1:
        return mine
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    cap = max(shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world))
    out = {}
    for key, arr in mine.items():
        pad = np.zeros((cap,) + arr.shape[1:], arr.dtype)
        pad[:hi - lo] = arr
        t = torch.from_numpy(pad).to(dev)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        chunks = []
        for r, p in enumerate(parts):
            a, b = shard_bounds(n, world, r)
            chunks.append(p[:b - a].cpu().numpy())
        out[key] = np.concatenate(chunks, axis=0)
    return out


def predict_classifier_sharded(partial_fn: Callable[[np.ndarray], "object"],
                               finish_fn: Callable[["object"], Dict[str, np.ndarray]],
                               genomat: np.ndarray, group=None) -> Dict[str, np.ndarray]:
    """``partial_fn(genomat)`` returns this rank's partial sums as a torch tensor
    [P+3, n_pad] (device memory on a GPU box); ONE all-reduce(SUM) merges the
    ranks; ``finish_fn(merged)`` turns them into the PredictHLA outputs."""
    import torch.distributed as dist

    part = partial_fn(genomat)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return finish_fn(part)


def hip_classifier_sharded_fns(obj: HlaAttrBagObj, device: int, world: int, rank: int):
    """(partial_fn, finish_fn) backed by the HIP library for this rank's classifier shard."""
    import torch
    from .hibag import HlaAttrBagClass

    sub, sw = classifier_shard(obj, world, rank)
    model = HlaAttrBagClass(sub, device=device, snp_weight=sw)
    dev = torch.device("cuda", device)
    P, n_hla = obj.n_cell, obj.n_hla
    state = {}

    def partial_fn(genomat: np.ndarray):
        n = genomat.shape[0]
        n_pad = (max(n, 1) + 63) // 64 * 64
        g = torch.from_numpy(np.ascontiguousarray(genomat, np.int32)).to(dev)
        part = torch.zeros((P + 3, n_pad), dtype=torch.float64, device=dev)
        model.predict_partial_device(g.data_ptr(), n, part.data_ptr(),
                                     stream=torch.cuda.current_stream(dev).cuda_stream)
        state["n"] = n
        state["g"] = g          # keep alive until the stream has consumed it
        return part

    def finish_fn(part) -> Dict[str, np.ndarray]:
        n = state["n"]
        h1 = torch.empty(n, dtype=torch.int32, device=dev)
        h2 = torch.empty(n, dtype=torch.int32, device=dev)
        prob = torch.empty(n, dtype=torch.float64, device=dev)
        match = torch.empty(n, dtype=torch.float64, device=dev)
        dos = torch.empty((n, n_hla), dtype=torch.float64, device=dev)
        pp = torch.empty((n, P), dtype=torch.float64, device=dev)
        model.finish_device(part.data_ptr(), n, h1.data_ptr(), h2.data_ptr(), prob.data_ptr(), match.data_ptr(),
                            dos.data_ptr(), pp.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        return dict(h1=h1.cpu().numpy(), h2=h2.cpu().numpy(), prob=prob.cpu().numpy(), matching=match.cpu().numpy(),
                    dosage=dos.cpu().numpy(), postprob=pp.cpu().numpy())

    return partial_fn, finish_fn, model
