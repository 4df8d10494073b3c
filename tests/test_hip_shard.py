"""Classifier-sharded prediction driven by the C library itself: hibag_hip_model_shard + hibag_hip_shard_group_* with the
posterior merge as ONE ncclAllReduce issued by libhibag_hip.so (hibag_amd/csrc/hibag_shard.hip).  The sum being split is
CAttrBag_Model::_PredictHLA's ensemble sum (src/LibHLA.cpp:2448-2476, :1497-1518); the reference's own multi-worker
branch is hlaPredict(cl=) (R/HIBAG.R:764-808).  Splitting the classifiers changes the order of their additions, so
the bar is identical calls and 1e-10 relative on the posteriors against the unsharded run -- and bit equality where the
split is trivial (one shard)."""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-10


@pytest.fixture(scope="module")
def hib():
    import hibag_amd
    hibag_amd.hlaSetKernelTarget("hip")
    return hibag_amd


@pytest.fixture(scope="module")
def case(hib):
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b")
    G, _ = synth.make_samples(founders, af, 3000)
    G[11, :] = hib.NA_INTEGER
    m = hib.hlaModelFromObj(model)
    full = m.predict_raw(G, 1, want_dosage=True, want_prob=True)
    return model, G, m, full


def close_to_unsharded(got, full):
    assert np.array_equal(got["h1"], full["h1"]) and np.array_equal(got["h2"], full["h2"])
    fin = np.isfinite(full["postprob"]).all(axis=1)
    assert np.array_equal(np.isnan(got["postprob"]), np.isnan(full["postprob"]))
    for k in ("prob", "matching", "dosage", "postprob"):
        np.testing.assert_allclose(got[k][fin], full[k][fin], rtol=TOL, atol=1e-300, err_msg=k)


def test_one_shard_one_rccl_rank_equals_the_unsharded_run_bit_for_bit(hib, case):
    """A 1-device communicator: the all-reduce is the identity, the partial + finish route must reproduce the plain entry."""
    from hibag_amd.hibag import ShardGroup
    model, G, m, full = case
    assert hib._lib.lib().hibag_hip_rccl_version() >= 20000
    grp = ShardGroup(m, [0])
    assert grp.ranks == 1
    got = grp.predict_raw(G, want_dosage=True, want_prob=True)
    assert grp.allreduces >= 1
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        assert np.array_equal(got[k], full[k], equal_nan=True), k
    for s in grp.shards:
        assert s.handover_faults() == 0
    grp.close()


@pytest.mark.parametrize("n_shards", [2, 8])
def test_shards_sharing_the_device_merge_like_ranks_would(hib, case, n_shards):
    """Several shards on the one device of the box: added up on the device in shard order, then the (1-rank) all-reduce."""
    from hibag_amd.hibag import ShardGroup
    model, G, m, full = case
    grp = ShardGroup(m, [0] * n_shards)
    assert grp.ranks == 1 and len(grp.shards) == n_shards
    counts = [hib._lib.lib().hibag_hip_model_n_classifier(s.handle) for s in grp.shards]
    assert sum(counts) == len(model.classifiers) and max(counts) - min(counts) <= 1
    got = grp.predict_raw(G, want_dosage=True, want_prob=True)
    close_to_unsharded(got, full)
    # only the calls
    only = grp.predict_raw(G[:100], want_dosage=False, want_prob=False)
    assert np.array_equal(only["h1"], full["h1"][:100]) and np.array_equal(only["h2"], full["h2"][:100])
    grp.close()


def test_one_shot_entry_and_errors(hib, case):
    model, G, m, full = case
    L = hib._lib.lib()
    s0, s1 = m.shard(0, 2, 0), m.shard(1, 2, 0)
    n = 500
    g = np.ascontiguousarray(G[:n], np.int32)
    h1 = np.zeros(n, np.int32); h2 = np.zeros(n, np.int32); pr = np.zeros(n); mt = np.zeros(n)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    hs = (C.c_void_p * 2)(s0.handle, s1.handle)
    hib._lib.check(L.hibag_hip_predict_multi_sharded(hs, 2, p(g), n, p(h1), p(h2), p(pr), p(mt), None, None))
    assert np.array_equal(h1, full["h1"][:n]) and np.array_equal(h2, full["h2"][:n])
    np.testing.assert_allclose(pr, full["prob"][:n], rtol=TOL)
    np.testing.assert_allclose(mt, full["matching"][:n], rtol=TOL)
    # H1 without H2, no shards
    assert L.hibag_hip_predict_multi_sharded(hs, 2, p(g), n, p(h1), None, None, None, None, None) != 0
    assert L.hibag_hip_predict_multi_sharded(hs, 0, p(g), n, p(h1), p(h2), None, None, None, None) != 0
    with pytest.raises(hib.HibagHipError):
        m.shard(0, 2, 99)
    with pytest.raises(hib.HibagHipError):
        m.shard(2, 2, 0)
    s0.close(); s1.close()


def test_a_failed_handover_on_one_shard_is_repaired_not_returned(hib):
    """The poisoned scalar rows of one shard reach the merged sums as NaN; the group entry must notice (sticky status of
    the shard), run the batch again without hand-overs, and return the repaired numbers."""
    from hibag_amd import synth
    from hibag_amd.hibag import ShardGroup
    model, founders, af = synth.make_model("hla-b")
    G, _ = synth.make_samples(founders, af, 10_000)
    m = hib.hlaModelFromObj(model)
    full = m.predict_raw(G, 1, want_dosage=True, want_prob=False)
    grp = ShardGroup(m, [0])                      # one shard = the whole model: its items are chunked at 10,000 samples
    grp.shards[0].inject_handover_fault(2)
    got = grp.predict_raw(G, want_dosage=True, want_prob=False)
    assert grp.shards[0].handover_faults() == 1 and grp.shards[0].status() == 0
    for k in ("h1", "h2", "prob", "matching", "dosage"):
        assert np.array_equal(got[k], full[k], equal_nan=True), k
    grp.close(); m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--no-cpu-baseline"], ["--shard", "classifiers"]])
def test_bench_two_rank_rehearsal_on_one_device(extra):
    """`python bench.py --gpus 2` end to end on a one-GPU box: bench.py starts its own two ranks (torch.distributed.run), both
    ranks share device 0 and the collectives go over gloo (HIBAG_BENCH_DRY_RANKS=1 -- RCCL refuses two ranks on one device).
    What is checked is the orchestration the driver's 8-GPU run goes through: slice bounds, the classifier-sharded merge
    against the unsharded posterior, the one JSON line."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIBAG_BENCH_DRY_RANKS="1")
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, stdin=subprocess.DEVNULL)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["samples_per_step"] == 100_000
    assert "dry_run" in d and d["rccl_ranks"] is None
    assert d["call_accuracy_vs_truth"] > 0.99
    chk = d["classifier_shard_check"] if "--shard" in extra else d["classifier_sharded"]["check"]
    assert chk["calls_identical_to_unsharded"] and chk["nan_pattern_identical"]
    assert chk["max_rel_dev_posterior_vs_unsharded"] < chk["tolerance"]
    assert all(v == 0 for v in d["handover_faults"].values())
