"""The N > 1 paths.  CPU: two gloo processes run the orchestration of
hibag_amd/dist.py with the oracle standing in for the per-rank compute.
GPU: the classifier-sharded entry points of the HIP library, two shards
emulated on one device, against the unsharded result."""

import os
import socket

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _result(q, procs, timeout=180):
    """First item a worker puts on `q`; fails (instead of hanging) when a worker dies or nothing arrives."""
    import queue
    import time
    end = time.time() + timeout
    while time.time() < end:
        try:
            return q.get(timeout=1.0)
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead:
                raise AssertionError(f"a worker exited with code {dead[0]} before producing a result")
    for p in procs:
        p.terminate()
    raise AssertionError("no result from the workers within the time limit")


def _oracle_partial(O, fm_full, model, lo, hi, G):
    """[P+3, n] partial sums of classifiers lo..hi-1, as the HIP partial entry defines them."""
    P = fm_full.n_hla * (fm_full.n_hla + 1) // 2
    n = G.shape[0]
    sw = np.zeros(model.n_snp, np.int64)
    for c in model.classifiers:
        np.add.at(sw, c.snpidx, 1)
    part = np.zeros((P + 3, n))
    for i in range(n):
        for c in range(lo, hi):
            cl = model.classifiers[c]
            g = G[i][cl.snpidx]
            ok = (g >= 0) & (g <= 2)
            tot = int(sw[cl.snpidx].sum())
            w = float(int(sw[cl.snpidx][ok].sum())) / tot if tot > 0 else 0.0
            if w <= 0:
                continue
            s1, s2 = O.int_to_snp(G[i], cl.snpidx)
            prob, total = O.post_prob2(fm_full, c, s1, s2)
            part[:P, i] += prob * w
            part[P, i] += w
            part[P + 1, i] += total * w
            part[P + 2, i] += w
    return part


def _finish(part, n_hla):
    P = n_hla * (n_hla + 1) // 2
    S = part[:P].copy()
    sw = part[P]
    S[:, sw > 0] /= sw[sw > 0]
    h1 = np.repeat(np.arange(n_hla), np.arange(n_hla, 0, -1))
    h2 = np.concatenate([np.arange(i, n_hla) for i in range(n_hla)])
    am = S.argmax(axis=0)
    return dict(h1=h1[am], h2=h2[am], postprob=S.T, matching=part[P + 1] / part[P + 2])


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from hibag_amd import dist as hd, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, founders, af = synth.make_model("hla-a-small", n_classifier=7)
    G, _ = synth.make_samples(founders, af, 37)
    fm = O.flatten(model)
    # sample sharding: slices predicted independently, gathered on every rank
    got = hd.predict_sample_sharded(lambda g: O.predict(fm, g), G)
    # classifier sharding: one all-reduce of the partial sums
    lo, hi = hd.shard_bounds(len(model.classifiers), world, rank)
    sub, sw = hd.classifier_shard(model, world, rank)
    assert len(sub.classifiers) == hi - lo and sw.sum() == sum(len(c.snpidx) for c in model.classifiers)
    got2 = hd.predict_classifier_sharded(
        lambda g: torch.from_numpy(_oracle_partial(O, fm, model, lo, hi, g)),
        lambda p: _finish(p.numpy(), model.n_hla), G)
    if rank == 0:
        q.put((got, got2))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_orchestration(oracle):
    import torch.multiprocessing as mp
    from hibag_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, got2 = _result(q, procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    model, founders, af = synth.make_model("hla-a-small", n_classifier=7)
    G, _ = synth.make_samples(founders, af, 37)
    want = oracle.predict(oracle.flatten(model), G)
    for k in want:                       # sample sharding: bit-identical
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    # classifier sharding: same calls, posteriors within 1e-10 relative
    assert np.array_equal(got2["h1"], want["h1"]) and np.array_equal(got2["h2"], want["h2"])
    np.testing.assert_allclose(got2["postprob"], want["postprob"], rtol=1e-10, atol=0)
    np.testing.assert_allclose(got2["matching"], want["matching"], rtol=1e-10, atol=0)


def test_shard_bounds_cover_everything():
    from hibag_amd.dist import shard_bounds
    for n in (0, 1, 7, 64, 10_000):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.gpu
def test_classifier_sharded_entry_points_on_one_gpu(oracle):
    """Two classifier shards on one device: partial sums added like the all-reduce would,
    then finished.  Calls identical, posteriors within 1e-10 relative of the unsharded run."""
    import torch
    import hibag_amd
    from hibag_amd import dist as hd, synth
    hibag_amd.hlaSetKernelTarget("hip")
    model, founders, af = synth.make_model("hla-b", n_classifier=24)
    G, _ = synth.make_samples(founders, af, 300)
    G[4, :] = hibag_amd.NA_INTEGER
    full = hibag_amd.hlaModelFromObj(model).predict_raw(G, 1, want_dosage=True, want_prob=True)
    parts, fins, keep = [], [], []
    for r in range(2):
        pf, ff, m = hd.hip_classifier_sharded_fns(model, 0, 2, r)
        parts.append(pf(G)); fins.append(ff); keep.append(m)
    torch.cuda.synchronize()
    merged = parts[0] + parts[1]
    got = fins[0](merged)
    assert np.array_equal(got["h1"], full["h1"]) and np.array_equal(got["h2"], full["h2"])
    ok = np.isfinite(full["postprob"]).all(axis=1)
    for k in ("prob", "matching", "dosage", "postprob"):
        np.testing.assert_allclose(got[k][ok], full[k][ok], rtol=1e-10, atol=1e-300, err_msg=k)
    # one shard holding every classifier is the unsharded computation: bit-identical
    pf, ff, m = hd.hip_classifier_sharded_fns(model, 0, 1, 0)
    same = ff(pf(G))
    for k in same:
        assert np.array_equal(same[k], full[k], equal_nan=True), k


def _train_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hibag_amd import synth, train
    from hibag_amd.model import Classifier
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, founders, af = synth.make_model("hla-a-small", seed=5, n_snp=40)
    G, truth = synth.make_samples(founders, af, 80, seed=6)

    def grow_fn(count, r):          # the oracle's CPU trainer stands in for the device-scored driver
        out = O.train(G, truth[:, 0], truth[:, 1], model.n_hla, count, 6, True, 300 + r)
        return [Classifier(snpidx=o["snpidx"], freq=o["freq"], hla=o["hla"], haplo=o["haplo"], samp_num=o["samp_num"],
                           outofbag_acc=o["acc"]) for o in out]

    got = train.grow_classifier_sharded(grow_fn, 5)
    if rank == 0:
        q.put([(c.snpidx.tolist(), c.freq.tolist(), c.haplo, c.samp_num.tolist()) for c in got])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_training_shards(oracle):
    """hlaParallelAttrBagging's orchestration: ranks grow 3 + 2 classifiers from their own streams,
    every rank ends up with all 5 in rank order."""
    import torch.multiprocessing as mp
    from hibag_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = _result(q, procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    model, founders, af = synth.make_model("hla-a-small", seed=5, n_snp=40)
    G, truth = synth.make_samples(founders, af, 80, seed=6)
    want = oracle.train(G, truth[:, 0], truth[:, 1], model.n_hla, 3, 6, True, 300) + \
        oracle.train(G, truth[:, 0], truth[:, 1], model.n_hla, 2, 6, True, 301)
    assert len(got) == 5
    for (snpidx, freq, haplo, samp), w in zip(got, want):
        assert snpidx == w["snpidx"].tolist() and freq == w["freq"].tolist() and haplo == w["haplo"]
        assert samp == w["samp_num"].tolist()
