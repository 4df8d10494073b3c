"""Launch status of the device-pointer entries (include/hibag_hip.h, "launch status"): a hand-over between the
chunks of a work item that never arrives must reach the caller -- poisoned outputs, a sticky model status -- and the
second attempt, in the same process and without hand-overs, must equal the oracle.  The reference's convention is that an
entry never returns garbage (CORE_TRY / CORE_CATCH, src/HIBAG.cpp:41-60; try_final_* guards, src/LibHLA.cpp:2307-2315).
Also: one workspace per model, calls on two streams are chained on the device."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E_HANDOVER = -5
N = 10_000          # the benchmark batch: both passes have more work items than resident workgroups, so their tails are cut


@pytest.fixture(scope="module")
def hib():
    import hibag_amd
    hibag_amd.hlaSetKernelTarget("hip")
    return hibag_amd


@pytest.fixture(scope="module")
def bench_case(hib, oracle):
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b")
    G, _ = synth.make_samples(founders, af, N)
    sub = np.arange(0, N, 417)[:24]
    want = oracle.predict(oracle.flatten(model), G[sub], vote_method=1, want_prob=False, avx2=True, n_threads=8)
    return model, G, sub, want


def device_call(m, torch, dev, G, stream=None):
    n = len(G)
    dg = torch.from_numpy(G).to(dev)
    # (torch.empty: no fill kernel on torch's own stream that could race with the library's kernels on `stream`)
    out = dict(h1=torch.empty(n, dtype=torch.int32, device=dev), h2=torch.empty(n, dtype=torch.int32, device=dev),
               prob=torch.empty(n, dtype=torch.float64, device=dev), matching=torch.empty(n, dtype=torch.float64, device=dev),
               dosage=torch.empty((n, m.obj.n_hla), dtype=torch.float64, device=dev))
    st = torch.cuda.current_stream(dev) if stream is None else stream
    m.predict_device(dg.data_ptr(), n, 1, out["h1"].data_ptr(), out["h2"].data_ptr(), out["prob"].data_ptr(),
                     out["matching"].data_ptr(), out["dosage"].data_ptr(), None, stream=st.cuda_stream)
    out["_geno"] = dg
    return out


def same_as_oracle(out, sub, want):
    for k in ("h1", "h2", "prob", "matching", "dosage"):
        got = out[k].cpu().numpy()[sub]
        assert np.array_equal(got, want[k], equal_nan=True), k


@pytest.mark.parametrize("which_pass", [1, 2])
def test_failed_handover_reaches_the_caller_of_the_device_entry(hib, bench_case, which_pass):
    import torch
    model, G, sub, want = bench_case
    dev = torch.device("cuda", 0)
    m = hib.hlaModelFromObj(model)
    out = device_call(m, torch, dev, G)
    torch.cuda.synchronize(dev)
    assert m.status() == 0 and m.handover_faults() == 0
    same_as_oracle(out, sub, want)

    m.inject_handover_fault(which_pass)
    out = device_call(m, torch, dev, G)            # returns 0: enqueued
    torch.cuda.synchronize(dev)
    # the outputs are poisoned on the device: nothing that looks like a result
    assert np.all(out["h1"].cpu().numpy() == hib.NA_INTEGER) and np.all(out["h2"].cpu().numpy() == hib.NA_INTEGER)
    for k in ("prob", "matching", "dosage"):
        assert np.all(np.isnan(out[k].cpu().numpy())), k
    # ... and the model says so, stickily, through every way in
    assert m.status() == E_HANDOVER
    assert m.status() == E_HANDOVER
    assert m.handover_faults() == 1
    with pytest.raises(hib.HibagHipError) as e:
        device_call(m, torch, dev, G)
    assert e.value.code == E_HANDOVER and "hand-over" in str(e.value)
    with pytest.raises(hib.HibagHipError):
        m.get_timing()
    with pytest.raises(hib.HibagHipError):
        m.predict_raw(G[:64])
    # the caller's second attempt: same process, same model, now without hand-overs
    m.clear_status()
    assert m.status() == 0
    out = device_call(m, torch, dev, G)
    torch.cuda.synchronize(dev)
    assert m.status() == 0 and m.handover_faults() == 1
    same_as_oracle(out, sub, want)
    m.close()


@pytest.mark.parametrize("which_pass", [1, 2])
def test_host_entry_repairs_a_failed_handover_itself(hib, bench_case, which_pass):
    model, G, sub, want = bench_case
    m = hib.hlaModelFromObj(model)
    m.inject_handover_fault(which_pass)
    got = m.predict_raw(G, 1, want_dosage=True)                  # rc 0: the library ran the call again without hand-overs
    assert m.handover_faults() == 1 and m.status() == 0
    for k in ("h1", "h2", "prob", "matching", "dosage"):
        assert np.array_equal(got[k][sub], want[k], equal_nan=True), k
    assert not np.isnan(got["prob"]).any()
    m.close()


def test_partial_sums_of_a_failed_launch_are_poisoned_through_the_merge(hib, bench_case):
    """Classifier-sharded route: the three scalar rows of the partial sums are NaN, which survives the all-reduce (a sum)
    and makes hibag_hip_finish_device write NA / NaN."""
    import torch
    model, G, sub, want = bench_case
    dev = torch.device("cuda", 0)
    m = hib.hlaModelFromObj(model)
    n, P = len(G), model.n_cell
    n_pad = (n + 63) // 64 * 64
    dg = torch.from_numpy(G).to(dev)
    part = torch.zeros((P + 3, n_pad), dtype=torch.float64, device=dev)
    m.inject_handover_fault(2)
    m.predict_partial_device(dg.data_ptr(), n, part.data_ptr())
    torch.cuda.synchronize(dev)
    assert m.status() == E_HANDOVER
    assert torch.isnan(part[P:, :n]).all()
    m.clear_status()
    merged = part + torch.zeros_like(part)                       # what a sum with another rank's partials leaves of it
    h1 = torch.zeros(n, dtype=torch.int32, device=dev); h2 = torch.zeros_like(h1)
    pr = torch.zeros(n, dtype=torch.float64, device=dev)
    m.finish_device(merged.data_ptr(), n, h1.data_ptr(), h2.data_ptr(), pr.data_ptr(), None, None, None)
    torch.cuda.synchronize(dev)
    assert (h1.cpu().numpy() == hib.NA_INTEGER).all() and np.isnan(pr.cpu().numpy()).all()
    m.close()


def test_calls_on_two_streams_share_the_workspace_safely(hib, oracle):
    """Two device-entry calls on two different streams, enqueued back to back with different inputs: the library chains
    them on the device (one workspace per model), both equal the oracle."""
    import torch
    from hibag_amd import synth
    model, founders, af = synth.make_model("hla-b")
    dev = torch.device("cuda", 0)
    m = hib.hlaModelFromObj(model)
    flat = oracle.flatten(model)
    n = 6000
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    Ga, _ = synth.make_samples(founders, af, n, seed=501)
    Gb, _ = synth.make_samples(founders, af, n, seed=502)
    torch.cuda.synchronize(dev)
    outs = []
    for rep in range(3):
        outs.append((Ga, device_call(m, torch, dev, Ga, s1)))
        outs.append((Gb, device_call(m, torch, dev, Gb, s2)))
    s1.synchronize(); s2.synchronize()
    assert m.status() == 0 and m.handover_faults() == 0
    sub = np.arange(0, n, 250)
    for G, out in outs:
        want = oracle.predict(flat, G[sub], vote_method=1, want_prob=False, avx2=True, n_threads=8)
        same_as_oracle(out, sub, want)
    m.close()


def test_two_models_run_concurrently_on_two_streams_without_a_fault(hib, oracle):
    """Two DIFFERENT models (their own workspaces, flags and epochs) launched back to back on two streams, so that their
    chunked work items share the device: the hand-over scheme rests on observed dispatch order (chunks of an item on one
    XCD, first chunks dispatched before second chunks) -- with two grids in flight it must still hold: zero faults, and
    both models' outputs bit-equal to the oracle.  (src/HIBAG.cpp:41-60: an entry never returns numbers it cannot vouch for.)"""
    import torch
    from hibag_amd import synth
    dev = torch.device("cuda", 0)
    ma_obj, fa, afa = synth.make_model("hla-b")
    mb_obj, fb, afb = synth.make_model("hla-b", seed=synth.DEFAULT_SEED + 5, n_classifier=60, wide_classifier=False)
    ma, mb = hib.hlaModelFromObj(ma_obj), hib.hlaModelFromObj(mb_obj)
    n = N
    Ga, _ = synth.make_samples(fa, afa, n, seed=601)
    Gb, _ = synth.make_samples(fb, afb, n, seed=602)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    outs = []
    for rep in range(4):
        outs.append((ma_obj, Ga, device_call(ma, torch, dev, Ga, s1)))
        outs.append((mb_obj, Gb, device_call(mb, torch, dev, Gb, s2)))
    s1.synchronize(); s2.synchronize()
    assert ma.status() == 0 and mb.status() == 0
    assert ma.handover_faults() == 0 and mb.handover_faults() == 0
    sub = np.arange(0, n, 417)[:24]
    want = {}
    for obj, G, out in outs:
        if id(obj) not in want:
            want[id(obj)] = oracle.predict(oracle.flatten(obj), G[sub], vote_method=1, want_prob=False, avx2=True, n_threads=8)
        same_as_oracle(out, sub, want[id(obj)])
    ma.close(); mb.close()
