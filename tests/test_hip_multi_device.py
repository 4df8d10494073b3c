"""Everything that needs TWO OR MORE devices, so that the first execution on a multi-GPU node is a test and not the driver's
scaling run (BASELINE config 3; the reference's own multi-worker branch is hlaPredict(cl = ), R/HIBAG.R:764-808).  Skipped on
a one-GPU box -- where tests/test_hip_shard.py and tests/test_hip_configs.py rehearse the same code with several replicas /
shards / ranks on the one device.

  * hibag_hip_predict_multi over replicas on DISTINCT devices: samples are independent (src/LibHLA.cpp:2362-2411), so every
    output must equal the one-device run bit for bit;
  * hibag_hip_shard_group_predict with one RCCL rank per device: the classifiers' terms are added in another order, so
    identical calls and 1e-10 relative on the posteriors (north_star's tolerance);
  * `python bench.py --gpus 2` over real RCCL: the line's rccl_ranks must say 2;
  * concurrent trainers of ONE process spread over the devices (train.grow_concurrently(device=[0, 1, ...]): a combiner per
    device): every classifier equal to the oracle's serial run from its stream."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _n_devices():
    try:
        from hibag_amd import _lib
        return int(_lib.lib().hibag_hip_device_count())
    except Exception:
        return 0


def needs_two(fn):
    """Skip at RUN time, not at collection: asking the library for the device count initialises HIP, which must not happen
    before torch has initialised its own context (tests/conftest.py, pytest_collection_finish)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        if _n_devices() < 2:
            pytest.skip("needs two or more devices (hibag_hip_device_count() < 2)")
        return fn(*a, **k)
    return wrapper


@pytest.fixture(scope="module")
def hib():
    import hibag_amd
    hibag_amd.hlaSetKernelTarget("hip")
    return hibag_amd


@pytest.fixture(scope="module")
def case(hib):
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b")
    G, _ = synth.make_samples(founders, af, 6000)
    G[17, :] = hib.NA_INTEGER
    m = hib.hlaModelFromObj(model, device=0)
    full = m.predict_raw(G, 1, want_dosage=True, want_prob=True)
    return model, G, m, full


def test_device_count_is_reported(hib):
    """(runs everywhere: says in the log how many devices the box has, i.e. whether the tests below ran)"""
    n = _n_devices()
    assert n >= 1
    print(f"hibag_hip_device_count() = {n}")


@needs_two
def test_replicas_on_distinct_devices_equal_one_device_bit_for_bit(hib, case):
    model, G, m, full = case
    devs = list(range(min(_n_devices(), 8)))
    reps = [m] + [m.replicate(d) for d in devs[1:]]
    assert [r.device() for r in reps] == devs
    from hibag_amd.hibag import predict_multi
    got = predict_multi(reps, G, 1, want_dosage=True, want_prob=True)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        assert np.array_equal(got[k], full[k], equal_nan=True), k
    for r in reps:
        assert r.handover_faults() == 0
    for r in reps[1:]:
        r.close()
    # the Python mirror of hlaPredict(cl = <cluster>)
    from hibag_amd import synth
    res = hib.hlaPredict(m, synth.as_snp_geno(model, G), cl=devs[:2], type="response", verbose=False)
    assert np.array_equal(res.h1, full["h1"]) and np.array_equal(res.h2, full["h2"])


@needs_two
@pytest.mark.parametrize("ranks", [2, 0])
def test_one_rccl_rank_per_device_merges_the_classifier_shards(hib, case, ranks):
    """ranks = 0: every device of the box."""
    from hibag_amd.hibag import ShardGroup
    model, G, m, full = case
    devs = list(range(min(_n_devices(), 8) if ranks == 0 else ranks))
    grp = ShardGroup(m, devs)
    assert grp.ranks == len(devs) and len(grp.shards) == len(devs)
    got = grp.predict_raw(G, want_dosage=True, want_prob=True)
    assert grp.allreduces >= 1
    assert np.array_equal(got["h1"], full["h1"]) and np.array_equal(got["h2"], full["h2"])
    fin = np.isfinite(full["postprob"]).all(axis=1)
    assert np.array_equal(np.isnan(got["postprob"]), np.isnan(full["postprob"]))
    for k in ("prob", "matching", "dosage", "postprob"):
        np.testing.assert_allclose(got[k][fin], full[k][fin], rtol=TOL, atol=1e-300, err_msg=k)
    for s in grp.shards:
        assert s.handover_faults() == 0
    grp.close()


@needs_two
@pytest.mark.parametrize("extra", [[], ["--shard", "classifiers"], ["--launcher", "threads"]])
def test_bench_over_two_real_ranks(extra):
    """`python bench.py --gpus 2` as the driver runs it for SCALE_rNN.json, but short: one process (or thread) per GPU, the
    collectives over RCCL -- `rccl_ranks` counts them with a real all-reduce."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "HIBAG_BENCH_DRY_RANKS"}
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, stdin=subprocess.DEVNULL)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "dry_run" not in d
    assert d["value"] > 0 and d["call_accuracy_vs_truth"] > 0.99
    if "--launcher" not in extra:
        assert d["rccl_ranks"] == 2
    assert "error" not in d
    if "--shard" in extra:
        chk = d["classifier_shard_check"]
        assert chk["calls_identical_to_unsharded"] and chk["max_rel_dev_posterior_vs_unsharded"] < chk["tolerance"]
    elif "--launcher" not in extra:
        assert "error" not in d.get("classifier_sharded", {}), d["classifier_sharded"]


@needs_two
def test_concurrent_trainers_over_the_devices_of_the_node(hib, oracle):
    """hlaConcurrentAttrBagging's decomposition over a LIST of devices from one process: trainer r on device r % n, each device
    with combiners of its own (csrc/hibag_combine.h); classifiers are independent given their streams, so every one must equal
    the oracle's serial run from seed + r."""
    from hibag_amd import synth, train
    from hibag_amd.dist import shard_bounds
    from test_hip_train_driver import _as_dict
    from test_oracle_train import assert_same_classifier
    model, founders, af = synth.make_model("hla-a-small", seed=35, n_snp=60)
    G, truth = synth.make_samples(founders, af, 160, seed=36, miss=0.02)
    nd = _n_devices()
    k, ncl, mtry = 4 * nd, 8 * nd, 8
    got = train.grow_concurrently(G, truth[:, 0], truth[:, 1], model.n_hla, ncl, mtry, True, n_trainers=k, threads_per_trainer=1,
                                  seed=900, device=list(range(nd)), em="device", combine=True, thread_budget=4)
    assert len(got) == ncl
    at = 0
    for r in range(k):
        lo, hi = shard_bounds(ncl, k, r)
        for w in oracle.train(G, truth[:, 0], truth[:, 1], model.n_hla, nclassifier=hi - lo, mtry=mtry, prune=True, seed=900 + r):
            c = hib.Classifier(snpidx=w["snpidx"], freq=w["freq"], hla=w["hla"], haplo=w["haplo"], samp_num=w["samp_num"], outofbag_acc=w["acc"])
            assert_same_classifier(_as_dict(got[at]), c, at)
            at += 1
