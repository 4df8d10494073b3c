#!/usr/bin/env python3
"""Benchmark of the hlaPredict() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): hlaPredict() samples/sec (+ achieved HBM GB/s) on the
10k-sample x 100-classifier HLA-B configuration.  A *step* is one pass of the
hot path (CAttrBag_Model::PredictHLA, src/LibHLA.cpp:2317-2412, with the
default outputs of type="response+dosage": calls, probability, matching,
dosage) over one batch of synthetic samples that already sit in HBM; `value`
is device-resident (the PCIe-inclusive rate of the host-pointer entry is
`host_inclusive`, SURVEY.md 8(d)'s protocol number).
  N = 1   BASELINE config 2: 10,000 samples.
  N > 1   BASELINE config 3: 100,000 samples split N ways (contiguous slices, samples are independent: no data-path
          collective; strong scaling) -- one process per GPU, launched by torch.distributed.run, rendezvous over RCCL --
          and, in the same line under "classifier_sharded", config 3's other reading: every rank a slice of the
          CLASSIFIERS on all the samples, merged by one RCCL all-reduce of the partial posterior sums per 25,000 samples.
          `--samples n` instead gives every rank its own n samples (weak scaling).

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  roofline     : the dominant kernel against the HBM roof the metric names,
                 from HIP events recorded on the launch stream inside the timed
                 region (and, under "issue", the SIMD time per wavefront-pair
                 against the FP64 + matrix-core issue floor measured in this process, the ceiling
                 that actually binds this path -- DESIGN.md section 5);
  cpu_baseline : the AVX2 + threads CPU port of the reference's kernel
                 (oracle/, kind "port") timed on this box's host cores on a
                 bounded sample of the same workload (rank 0, N = 1 only).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLES_PER_GPU = 10_000       # BASELINE config 2
CFG3_SAMPLES = 100_000         # BASELINE config 3: the cohort N GPUs share
CFG3_SLICE = 25_000            # ... and what one all-reduce of the classifier-sharded route merges
SHAPE = "hla-b"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
# Issue floor used for the "issue" block (DESIGN.md "Rooflines"), measured on MI355X
# with tools/ubench_mfma.hip: a wave64 FP64 mul or add holds its SIMD for 2.29 ns, a
# v_mfma_i32_32x32x32_i8 for 16.0 ns, and the two do not overlap across wavefronts of
# one SIMD (half/half mix runs at 86 % of the serial sum).
N_SIMD = 256 * 4
FP64_OP_NS = 2.29
MFMA_NS = 16.0
UBENCH_FILE = os.path.join(ROOT, "profiles", "ubench_constants.json")   # parsed from the raw tools/ubench_* logs next to it


def issue_constants():
    """FP64 / MFMA issue costs (ns per wave-instruction per SIMD) measured NOW on this process's device
    (hibag_hip_measure_issue_costs, ~50 ms: boards differ by up to 10 % under an FP64 / matrix load); if the
    measurement cannot run, the committed micro-benchmark logs of an earlier round (profiles/ubench_constants.json)."""
    import ctypes as C
    c = {"fp64_op_ns": FP64_OP_NS, "mfma_i8_32x32x32_ns": MFMA_NS, "mfma_fp4_32x32x64_ns": 18.3, "mfma_fp64_overlap": None,
         "source": "built-in (round 1)"}
    try:
        from hibag_amd import _lib
        v = [C.c_double(0) for _ in range(4)]
        if _lib.lib().hibag_hip_measure_issue_costs(*[C.byref(x) for x in v]) == 0 and v[0].value > 0:
            c.update({"fp64_op_ns": round(v[0].value, 3), "mfma_i8_32x32x32_ns": round(v[1].value, 2),
                      "mfma_fp4_32x32x64_ns": round(v[2].value, 2),
                      "mfma_fp64_overlap": {"frac_of_serial_sum": round(v[3].value, 3)},
                      "source": "measured in this process (hibag_hip_measure_issue_costs: 8 wavefronts on every SIMD)"})
            return c
    except Exception:
        pass
    try:
        d = json.load(open(UBENCH_FILE))
        c.update({k: d[k] for k in ("fp64_op_ns", "mfma_i8_32x32x32_ns", "mfma_fp4_32x32x64_ns", "mfma_fp64_overlap") if k in d})
        c["source"] = "profiles/ubench_constants.json (" + d.get("tag", "?") + ")"
    except (OSError, ValueError, KeyError):
        pass
    return c


def issue_constants_file():
    """The constants earlier rounds priced the floor with (profiles/ubench_constants.json, measured in round 2 on another
    board): kept so that the fraction can be followed across rounds next to the one from this process's own measurement."""
    c = {"fp64_op_ns": 2.2, "mfma_i8_32x32x32_ns": 16.1, "mfma_fp4_32x32x64_ns": 18.3}
    try:
        d = json.load(open(UBENCH_FILE))
        c.update({k: d[k] for k in c if k in d})
    except (OSError, ValueError):
        pass
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--samples", type=int, default=None,
                    help="samples per GPU per step (weak scaling).  Default: 10,000 at one GPU (BASELINE config 2); with "
                         "--gpus N > 1 the 100,000 samples of config 3 split N ways (strong scaling)")
    ap.add_argument("--shape", default=SHAPE)
    ap.add_argument("--prob", action="store_true", help="also return the full posterior matrix (type='response+prob')")
    ap.add_argument("--vote", choices=("prob", "majority"), default="prob", help="hlaPredict(vote=): averaged posteriors (default) or majority vote")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-wide", action="store_true", help="diagnostic: drop the one 100-SNP classifier of the synthetic model")
    ap.add_argument("--shard", choices=("samples", "classifiers"), default="samples",
                    help="multi-GPU decomposition: every rank its own samples (default, no collective), or every rank a "
                         "slice of the classifiers on the SAME samples, merged by one RCCL all-reduce of the partial "
                         "posterior sums per step (BASELINE config 3's 'RCCL posterior merge'; strong scaling)")
    ap.add_argument("--no-extras", action="store_true", help="skip host_inclusive / other_configs (profiling runs)")
    ap.add_argument("--launcher", choices=("ranks", "threads"), default="ranks",
                    help="--gpus N > 1: one process per GPU under torch.distributed (default; RCCL rendezvous), or ONE process "
                         "with one host thread per GPU through the C ABI (hibag_hip_model_replicate + the device entry), "
                         "the way an R / C++ host would drive a node (INTEGRATION.md section B)")
    args = ap.parse_args()

    if args.launcher == "threads" and "WORLD_SIZE" not in os.environ:
        return main_threads(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` starts its own ranks: one child process per GPU through
        # torch.distributed.run (a CHILD, never an exec: nothing here has touched the GPU yet, but the
        # rule of the pool is to spawn).  Rank 0's JSON line and the exit code are relayed.
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: hibag_amd has no CPU fallback")
    # HIBAG_BENCH_DRY_RANKS=1: rehearsal of the N-rank orchestration on a box with fewer GPUs than ranks -- the ranks share
    # the devices there are and the collectives go over gloo (RCCL refuses two ranks on one device).  The line says so
    # ("dry_run"); its numbers are not a measurement of anything but the logic.
    dry = os.environ.get("HIBAG_BENCH_DRY_RANKS") == "1" and world > torch.cuda.device_count()
    if dry:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = None
    if dry:
        _all_reduce = dist.all_reduce

        def staged_all_reduce(t, op=dist.ReduceOp.SUM):
            h = t.cpu()
            _all_reduce(h, op=op)
            t.copy_(h)
        dist.all_reduce = staged_all_reduce
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm
        probe = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(probe, op=dist.ReduceOp.SUM)     # a real collective over RCCL: every rank contributes 1
        rccl_ranks = int(round(float(probe.item())))
        assert rccl_ranks == dist.get_world_size() == world

    import hibag_amd
    from hibag_amd import synth

    hibag_amd._lib.check(hibag_amd._lib.lib().hibag_hip_set_device(local_rank))
    target = hibag_amd.hlaSetKernelTarget("hip")[0]
    model_obj, founders, afreq = synth.make_model(args.shape, wide_classifier=not args.no_wide)
    by_classifier = args.shard == "classifiers"
    vote_method = 2 if args.vote == "majority" else 1
    from hibag_amd import dist as hdist
    # What the ranks share.  Default: one GPU -> config 2's 10,000 samples; N GPUs -> config 3's 100,000 samples, a
    # contiguous slice per rank (strong scaling).  --samples n: every rank its own n samples (weak scaling).
    strong = args.samples is None and world > 1 and args.shard != "classifiers"
    if by_classifier:
        # every rank sees ALL the samples and holds a slice of the classifiers
        n = args.samples if args.samples is not None else (CFG3_SAMPLES if world > 1 else SAMPLES_PER_GPU)
        n_total = n
        geno, truth = synth.make_samples(founders, afreq, n, seed=synth.DEFAULT_SEED + 1)
    elif strong:
        n_total = CFG3_SAMPLES
        lo, hi = hdist.shard_bounds(n_total, world, rank)
        n = hi - lo
        geno_all, truth_all = synth.make_samples(founders, afreq, n_total, seed=synth.DEFAULT_SEED + 1)
        geno, truth = np.ascontiguousarray(geno_all[lo:hi]), truth_all[lo:hi]
    else:
        n = args.samples if args.samples is not None else SAMPLES_PER_GPU
        n_total = n * world
        geno, truth = synth.make_samples(founders, afreq, n, seed=synth.DEFAULT_SEED + 1 + rank)
    scaling = "strong" if (by_classifier or strong) else "weak"
    if by_classifier:
        sub, sw = hdist.classifier_shard(model_obj, world, rank)
        model = hibag_amd.HlaAttrBagClass(sub, device=local_rank, snp_weight=sw)
    else:
        model = hibag_amd.hlaModelFromObj(model_obj, device=local_rank)

    n_hla, P, S = model_obj.n_hla, model_obj.n_cell, model_obj.n_snp
    d_geno = torch.from_numpy(geno).to(dev)
    d_h1 = torch.empty(n, dtype=torch.int32, device=dev)
    d_h2 = torch.empty(n, dtype=torch.int32, device=dev)
    d_prob = torch.empty(n, dtype=torch.float64, device=dev)
    d_match = torch.empty(n, dtype=torch.float64, device=dev)
    d_dos = torch.empty((n, n_hla), dtype=torch.float64, device=dev)
    d_pp = torch.empty((n, P), dtype=torch.float64, device=dev) if args.prob else None
    stream = torch.cuda.current_stream(dev)

    def sharded_step_fn(mdl, n_s, g, h1, h2, pr, mt, ds, pp):
        """One step of the classifier-sharded route on n_s samples: per slice of <= CFG3_SLICE samples the partial ensemble
        sums of this rank's classifiers -> ONE all-reduce -> arg-max / dosage (on every rank)."""
        sl = min(CFG3_SLICE, mdl.batch_limit(), max(n_s, 1))
        part = torch.zeros((P + 3, (sl + 63) // 64 * 64), dtype=torch.float64, device=dev)

        def run():
            for s0 in range(0, n_s, sl):
                k = min(sl, n_s - s0)
                kp = (k + 63) // 64 * 64
                view = part if kp == part.shape[1] else part[:, :kp].contiguous()     # (the entry wants rows of n_pad doubles)
                mdl.predict_partial_device(g.data_ptr() + 4 * s0 * S, k, view.data_ptr(), stream=stream.cuda_stream)
                if world > 1:
                    dist.all_reduce(view, op=dist.ReduceOp.SUM)
                mdl.finish_device(view.data_ptr(), k, h1.data_ptr() + 4 * s0, h2.data_ptr() + 4 * s0, pr.data_ptr() + 8 * s0,
                                  mt.data_ptr() + 8 * s0, ds.data_ptr() + 8 * s0 * n_hla,
                                  None if pp is None else pp.data_ptr() + 8 * s0 * P, stream=stream.cuda_stream)
        return run, int((P + 3) * ((sl + 63) // 64 * 64) * 8)

    allreduce_bytes = None
    if by_classifier:
        step, allreduce_bytes = sharded_step_fn(model, n, d_geno, d_h1, d_h2, d_prob, d_match, d_dos, d_pp)
    else:
        def step():
            model.predict_device(d_geno.data_ptr(), n, vote_method, d_h1.data_ptr(), d_h2.data_ptr(), d_prob.data_ptr(),
                                 d_match.data_ptr(), d_dos.data_ptr(), None if d_pp is None else d_pp.data_ptr(),
                                 stream=stream.cuda_stream)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(fn, warmup, steps, mdl):
        """W untimed warm-up steps, then exactly K timed steps between fences.  Inside the timed region HIP events bracket
        the DOMINANT kernel only (the roofline's `avg_launch_ms` is measured live there, as the contract asks): an event
        record is a packet of its own on the queue, and five of them per step cost 2 % of it.  Which kernel dominates is
        found in two untimed steps before the warm-up; the other kernels' times come from up to ten more steps with every
        kernel class bracketed, AFTER the timed region."""
        mdl.set_timing(True); mdl.reset_timing()
        for _ in range(2):
            fn()
        fence()
        pre = mdl.get_timing()
        dom_k = max(("total", "accum"), key=lambda k: pre[k][0])
        mdl.set_timing(False)
        for _ in range(warmup):
            fn()
        fence()
        mdl.set_timing([dom_k])
        mdl.reset_timing()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        fence()
        dt_ = time.perf_counter() - t0
        timed.local_dt = dt_                   # (this rank's own clock; the line's figure is the max over ranks)
        tm_dom = mdl.get_timing()
        extra = max(1, min(steps, 10))
        mdl.set_timing(True); mdl.reset_timing()
        for _ in range(extra):
            fn()
        fence()
        tm_all = mdl.get_timing()
        mdl.set_timing(False)
        # per kernel (summed ms over `steps` launches, launches): the dominant one as measured in the timed region, the others
        # scaled from the steps behind it
        tm = {k: (v[0] / max(v[1], 1) * steps, steps) for k, v in tm_all.items()}
        tm[dom_k] = tm_dom[dom_k]
        timed.events = f"timed region: HIP events around k_{dom_k} only; other kernels from {extra} further steps"
        if world > 1:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, tm

    dt, timing = timed(step, args.warmup, args.steps, model)
    dt_local = timed.local_dt
    faults = {"timed_model": int(model.handover_faults())}

    # sanity on the timed outputs: calls of samples drawn from the model
    h1 = d_h1.cpu().numpy(); h2 = d_h2.cpu().numpy()
    call_acc = float(np.mean((h1 == truth[:, 0]) & (h2 == truth[:, 1])))

    value = n_total * args.steps / dt
    pair_evals = model_obj.pair_evals_per_sample()
    pe_rank = sub.pair_evals_per_sample() if by_classifier else pair_evals      # what one rank's kernels evaluate

    # ---- roofline of the dominant kernel (HIP events on the launch stream) ----
    dom = max(("total", "accum"), key=lambda k: timing[k][0])
    dom_ms, dom_launches = timing[dom]
    avg_ms = dom_ms / max(dom_launches, 1)
    bytes_per_sample = 4 * S + 24 + 8 * n_hla + (8 * P if args.prob else 0)
    alg_bytes = bytes_per_sample * n                                           # one launch covers the batch
    achieved_gbs = alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf) and n == SAMPLES_PER_GPU and args.shape == SHAPE and not by_classifier:
        try:
            tj = json.load(open(tf))
            traffic = tj.get(f"k_{dom}", {}).get("hbm_bytes_per_launch")
            traffic_source = ("not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, "
                              "summary committed as profiles/pmc_traffic.json (" + str(tj.get("_source", {}).get("tag")) + ")")
        except Exception:
            traffic = None
    K = issue_constants()
    n_pad = (n + 63) // 64 * 64
    # SIMD issue: both passes together against the pairs they actually evaluate (pass 1 all of them; pass 2 those of
    # the cells whose sums pass 1 did not store), and each pass by itself
    ms1 = timing["total"][0] / max(timing["total"][1], 1)
    ms2 = timing["accum"][0] / max(timing["accum"][1], 1)
    pairs2 = model.second_pass_pairs() if vote_method == 1 else pe_rank        # (majority vote: k_vote_best walks every pair)
    stored = model.stored_cells()
    obj_rank = sub if by_classifier else model_obj
    floor_ns, ns1 = issue_floor(obj_rank, ms1, n, K, model)
    ns2 = N_SIMD * ms2 * 1e6 / (max(pairs2, 1) * n / 64.0)
    ns_both = N_SIMD * (ms1 + ms2) * 1e6 / ((pe_rank + pairs2) * n / 64.0)
    # The dominant kernel against the ceiling that BINDS it -- SIMD issue: per evaluated haplotype pair and wavefront one FP64
    # multiply + one FP64 add in the reference's order plus the pair's share of the matrix instructions, priced with the costs
    # measured in this process -- in pair evaluations per second; the HBM figure the metric names stays beside it (hbm_*).
    dom_ns, dom_pairs = (ns1, pe_rank) if dom == "total" else (ns2, pairs2)
    dom_rate = dom_pairs * n / (avg_ms * 1e-3) / 1e9                      # G pair evaluations / s of the dominant kernel
    dom_peak = dom_rate * dom_ns / floor_ns                               # ... at the issue floor
    roofline = {
        "kernel": f"k_{dom}", "bound": "simd-issue", "achieved": round(dom_rate, 2), "peak": round(dom_peak, 2),
        "unit": "G pair-evals/s", "frac": round(floor_ns / dom_ns, 4),
        "issue_frac_both_passes": round(floor_ns / ns_both, 4),
        "k_total_issue_frac": round(floor_ns / ns1, 4), "k_accum_issue_frac": round(floor_ns / ns2, 4),
        "k_total_ms": round(ms1, 4), "k_accum_ms": round(ms2, 4),
        "floor_ns_per_wave_pair": round(floor_ns, 2), "fp64_op_ns": K["fp64_op_ns"], "mfma_fp4_ns": K["mfma_fp4_32x32x64_ns"],
        "hbm_achieved_gbs": round(achieved_gbs, 3), "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac": achieved_gbs / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": int(alg_bytes),
        "traffic": traffic, "traffic_source": traffic_source,
        "avg_launch_ms": round(avg_ms, 4), "launches": int(dom_launches), "events": timed.events,
        "bound_note": "not HBM: ~1e3 pair evaluations per algorithmic byte (SURVEY 8d 'Which roofline binds'); `frac` = issue floor / "
                      "measured SIMD time per wavefront-pair of the dominant kernel, hbm_frac = algorithmic bytes / launch time / 8 TB/s",
        "issue": {"pair_evals_per_s": (pe_rank + pairs2) * n / ((ms1 + ms2) * 1e-3),
                  "pairs_evaluated_per_sample": {"pass1": pe_rank, "pass2": pairs2},
                  "cell_sums_stored_per_sample": stored,
                  "simd_ns_per_wave_pair": round(ns_both, 2),
                  "floor_ns_per_wave_pair": round(floor_ns, 2),
                  "frac": round(floor_ns / ns_both, 4),
                  "k_total": {"ms": round(ms1, 4), "simd_ns_per_wave_pair": round(ns1, 2), "frac": round(floor_ns / ns1, 4)},
                  "k_accum": {"ms": round(ms2, 4), "simd_ns_per_wave_pair": round(ns2, 2), "frac": round(floor_ns / ns2, 4),
                              "stored_cells_read_gb": round(stored * 8.0 * n_pad / 1e9, 3)},
                  "constants": K,
                  "k_total_frac_with_round2_constants": round(issue_floor(obj_rank, ms1, n, issue_constants_file(), model)[0] /
                                                              max(issue_floor(obj_rank, ms1, n, issue_constants_file(), model)[1], 1e-9), 4),
                  "note": "the binding ceiling is SIMD issue, not HBM: per evaluated pair one FP64 mul "
                          "+ one FP64 add in the reference's order plus its share of the MFMAs, which serialise with FP64 on a SIMD "
                          "(constants: measured in this process, see `constants.source`; DESIGN.md section 5).  Pass 2 evaluates only the pairs "
                          "of the cells with few pairs (the others' sums come from pass 1 through HBM), so its time is mostly per-cell "
                          "and per-(classifier, tile) work, not pair evaluation"},
        "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in timing.items()},
    }

    def shard_check_of(mdl, n_s, g, G_host, calls):
        """How far the classifier-sharded route's merged posterior is from the unsharded run (on at most 10,000 samples)."""
        k = min(n_s, 10_000)
        full = hibag_amd.hlaModelFromObj(model_obj, device=local_rank)
        ref = full.predict_raw(G_host[:k], 1, want_dosage=False, want_prob=True)
        full.close()
        o = [torch.empty(k, dtype=torch.int32, device=dev), torch.empty(k, dtype=torch.int32, device=dev),
             torch.empty(k, dtype=torch.float64, device=dev), torch.empty(k, dtype=torch.float64, device=dev),
             torch.empty((k, n_hla), dtype=torch.float64, device=dev), torch.empty((k, P), dtype=torch.float64, device=dev)]
        run, _ = sharded_step_fn(mdl, k, g, *o)
        run()
        torch.cuda.synchronize(dev)
        got_pp = o[5].cpu().numpy()
        denom = np.maximum(np.abs(ref["postprob"]), 1e-300)
        with np.errstate(invalid="ignore"):
            # below 1e-200 both sides are sums of denormals; NaN / inf entries: samples no classifier can call, whose ensemble
            # sum is zero or so small that its reciprocal overflows (the reference's own behaviour) -- those must agree as a pattern
            live = np.isfinite(ref["postprob"]) & (ref["postprob"] > 1e-200)
            rel = float(np.max(np.abs(got_pp - ref["postprob"])[live] / denom[live])) if live.any() else 0.0
        nan_same = bool(np.array_equal(np.isnan(got_pp), np.isnan(ref["postprob"])) and
                        np.array_equal(np.isinf(got_pp), np.isinf(ref["postprob"])))
        return {"samples_checked": k, "nan_pattern_identical": nan_same, "max_rel_dev_posterior_vs_unsharded": rel,
                "calls_identical_to_unsharded": bool(np.array_equal(ref["h1"], calls[0][:k]) and np.array_equal(ref["h2"], calls[1][:k])),
                "tolerance": 1e-10}

    # classifier-sharded mode: how far the merged posterior is from the unsharded (= sample-sharded) result
    shard_check = shard_check_of(model, n, d_geno, geno, (h1, h2)) if by_classifier else None

    # N > 1, default workload: config 3's other reading in the same line -- every rank a slice of the classifiers on ALL
    # 100,000 samples, the partial posterior sums merged by one RCCL all-reduce per 25,000 samples
    sharded_line = None
    if strong and not args.no_extras and vote_method == 1:
        # Fail-soft as far as SET-UP goes: everything that can fail without a collective (the shard's model, its buffers) is
        # done first and the ranks agree -- one all-reduce of a flag -- on whether to run the leg at all.  A failure INSIDE the
        # timed section (it contains the all-reduce) cannot be made soft: the peers are inside the collective, so the failing
        # rank exits non-zero and the launcher ends the group.  The main line is therefore written to stderr first: a run that
        # dies in this leg still leaves its measurement in the log.
        if rank == 0:
            print("[bench] main line before the classifier-sharded leg: " + json.dumps(
                {"value": value, "ms_per_step": dt / args.steps * 1e3, "n_gpus": world, "samples": n_total}), file=sys.stderr, flush=True)
        err, m2 = None, None
        try:
            sub2, sw2 = hdist.classifier_shard(model_obj, world, rank)
            m2 = hibag_amd.HlaAttrBagClass(sub2, device=local_rank, snp_weight=sw2)
            g2 = torch.from_numpy(geno_all).to(dev)
            o2 = [torch.empty(n_total, dtype=torch.int32, device=dev), torch.empty(n_total, dtype=torch.int32, device=dev),
                  torch.empty(n_total, dtype=torch.float64, device=dev), torch.empty(n_total, dtype=torch.float64, device=dev),
                  torch.empty((n_total, n_hla), dtype=torch.float64, device=dev)]
            run2, ar_bytes = sharded_step_fn(m2, n_total, g2, *o2, None)
        except Exception as e:                          # noqa: BLE001 -- reported in the line
            err = repr(e)
        ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            sharded_line = {"error": err or "another rank could not set this leg up"}
        else:
            try:
                k2 = max(1, min(args.steps, 5))
                dt2, tm2 = timed(run2, 1, k2, m2)
                c1, c2 = o2[0].cpu().numpy(), o2[1].cpu().numpy()
                sharded_line = {"value": n_total * k2 / dt2, "unit": "samples/s", "ms_per_step": dt2 / k2 * 1e3, "steps": k2,
                                "samples": n_total, "scaling": "strong",
                                "parallelism": f"classifier-sharded x{world}: {len(sub2.classifiers)} of {len(model_obj.classifiers)} classifiers on this rank, "
                                               f"one RCCL all-reduce of {ar_bytes} bytes per {CFG3_SLICE} samples",
                                "allreduce_bytes_per_step": ar_bytes * ((n_total + CFG3_SLICE - 1) // CFG3_SLICE),
                                "kernels_ms_per_step": {k: round(v[0] / k2, 4) for k, v in tm2.items()},
                                "call_accuracy_vs_truth": float(np.mean((c1 == truth_all[:, 0]) & (c2 == truth_all[:, 1])))}
                try:
                    sharded_line["check"] = shard_check_of(m2, n_total, g2, geno_all, (c1, c2))
                except Exception as e:                  # noqa: BLE001
                    sharded_line["check"] = {"error": repr(e)}
                faults["classifier_sharded_model"] = int(m2.handover_faults())
            except Exception as e:                      # noqa: BLE001 -- see above: not soft
                print(f"[bench] rank {rank}: the classifier-sharded leg failed inside its timed section: {e!r}", file=sys.stderr, flush=True)
                os._exit(3)
        try:
            if m2 is not None:
                m2.close()
        except Exception:                               # noqa: BLE001
            pass
        g2 = o2 = None

    # what every rank saw: the first real multi-GPU run has to diagnose itself from its one line
    mine = {"rank": rank, "local_rank": local_rank, "device": int(model.device()), "devices_visible": int(hibag_amd._lib.lib().hibag_hip_device_count()),
            "samples": int(n), "ms_per_step": round(dt_local / args.steps * 1e3, 4),
            "k_total_ms": round(ms1, 4), "k_accum_ms": round(ms2, 4), "handover_faults": int(model.handover_faults())}
    ranks_seen = [mine]
    if world > 1:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, mine)

    out = {
        "metric": "hlaPredict() samples/sec, 10k samples x 100-classifier HLA-B",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload_text(args, model_obj, n, n_total, world, strong or by_classifier),
                   "samples_per_gpu": n, "samples_per_step": n_total,
                   "parallelism": (f"classifier-sharded x{world} ({len(sub.classifiers)} of {len(model_obj.classifiers)} classifiers on this rank; "
                                   f"one RCCL all-reduce of {allreduce_bytes} bytes per {CFG3_SLICE} samples)"
                                   if by_classifier else f"sample-sharded x{world} (contiguous slices, no collective)"),
                   "kernel_target": target,
                   "engine": os.environ.get("HIBAG_ENGINE", "mfma") + " (FP4 / int8 MFMA distances + FP64 VALU accumulation in reference order)"},
        "pair_evals_per_s": value * pair_evals,
        "call_accuracy_vs_truth": call_acc,
        "rccl_ranks": rccl_ranks,
        "ranks": ranks_seen,
        "roofline": roofline,
    }
    if dry:
        out["dry_run"] = (f"{world} ranks sharing {torch.cuda.device_count()} device(s), collectives over gloo through host "
                          "staging: a rehearsal of the orchestration, not a measurement")
        out["rccl_ranks"] = None
    if shard_check is not None:
        out["classifier_shard_check"] = shard_check
    if sharded_line is not None:
        out["classifier_sharded"] = sharded_line

    if rank == 0 and world == 1 and not args.no_extras and not by_classifier:
        # SURVEY.md 8(d) protocol: median of >= 10 individually timed repetitions; the device-resident step (what `value`
        # is made of) next to the host-pointer entry, which includes H2D of the genotypes and D2H of the requested outputs
        def one():
            step(); torch.cuda.synchronize(dev)
        out["protocol"] = {"repetitions": 12, "device_resident_median_ms": median_ms(one, 12)}
        out["protocol"]["device_resident_median_samples_per_s"] = n / out["protocol"]["device_resident_median_ms"] * 1e3
        out["host_inclusive"] = host_inclusive(model, geno, n)
        out["host_inclusive"]["frac_of_device_resident"] = out["host_inclusive"]["value"] / out["protocol"]["device_resident_median_samples_per_s"]
        # ... and the API the metric is named after: hibag_amd.hlaPredict() itself (the mirror of R/HIBAG.R:481-818) on the same cohort
        out["api_inclusive"] = api_inclusive(model, model_obj, geno, n, (h1, h2))
        out["api_inclusive"]["frac_of_host_inclusive"] = out["api_inclusive"]["value"] / out["host_inclusive"]["value"]
        # (scalars the driver's parsed record keeps: SURVEY 8(d)'s protocol number and the API's, beside the device-resident `value`)
        out["config"].update({
            "host_inclusive_samples_per_s": round(out["host_inclusive"]["value"], 1), "host_inclusive_ms_per_step": round(out["host_inclusive"]["ms_per_step"], 4),
            "api_inclusive_samples_per_s": round(out["api_inclusive"]["value"], 1), "api_inclusive_ms_per_step": round(out["api_inclusive"]["ms_per_step"], 4),
            "api_inclusive_frac_of_host_inclusive": round(out["api_inclusive"]["frac_of_host_inclusive"], 4),
            "api_inclusive_numpy_order_samples_per_s": round(out["api_inclusive"]["numpy_order"]["value"], 1),
            "api_inclusive_plink_bed_samples_per_s": round(out["api_inclusive"]["plink_bed"]["value"], 1)})
        faults["host_inclusive"] = int(model.handover_faults()) - faults["timed_model"]
        if args.shape == SHAPE and n == SAMPLES_PER_GPU:
            out["host_inclusive_100k"] = host_inclusive_cohort(model, model_obj, founders, afreq, dev, 100_000)
            faults["host_inclusive_100k"] = int(model.handover_faults()) - faults["timed_model"] - faults["host_inclusive"]
            model.close()
            out["other_configs"] = other_configs(K, faults)

    if rank == 0 and world == 1 and not args.no_cpu_baseline and vote_method == 1:
        out["cpu_baseline"] = cpu_baseline(model_obj, geno, h1, h2)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        out["speedup_vs_one_cpu_thread"] = value / out["cpu_baseline"]["one_thread"]["value"]

    # hand-overs that failed (and were repaired or reported) in any model this run timed: a fault costs a repeated batch,
    # so a non-zero count explains a slow line -- it must be 0
    out["handover_faults"] = faults

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def workload_text(args, model_obj, n, n_total, world, shared):
    what = (f"BASELINE config 3: {n_total} samples shared by {world} GPUs" if (shared and world > 1)
            else "BASELINE config 2" if (n == SAMPLES_PER_GPU and world == 1) else f"{n} samples per GPU")
    return (f"{what}: synthetic {args.shape} model ({model_obj.n_hla} alleles, {len(model_obj.classifiers)} classifiers, "
            f"{model_obj.n_snp} SNPs, {model_obj.pair_evals_per_sample()} haplotype-pair evaluations/sample), {n} samples per GPU per step, "
            f"type={'response+prob' if args.prob else 'response+dosage'}, vote={args.vote}; genotypes and outputs resident in HBM "
            f"(transfers excluded: `host_inclusive` is the PCIe-inclusive rate of the host-pointer entry)")


def issue_floor(obj, avg_ms, n, K, dev_model):
    """(floor, achieved) SIMD time in ns per wavefront-pair (64 samples x one haplotype pair) for a kernel that
    took avg_ms over n samples: two FP64 ops per pair plus the pair's share of the MFMAs of the distance dot product --
    per 32-record block two FP4 instructions (K = 64) per K step, or four int8 ones (K = 32 each).  Which engine a
    classifier runs on is asked of the finalized model (hibag_hip_model_engine), not re-derived here."""
    per_block = {"valu": 0.0, "fp4": 2 * K["mfma_fp4_32x32x64_ns"], "i8": 4 * K["mfma_i8_32x32x32_ns"]}
    mfma_ns, w = 0.0, 0
    for ci, c in enumerate(obj.classifiers):
        h = len(c.freq)
        kind, steps = dev_model.engine(ci)
        mfma_ns += h * (h + 1) // 2 * per_block[kind] * steps / 32.0
        w += h * (h + 1) // 2
    floor = 2 * K["fp64_op_ns"] + mfma_ns / max(w, 1)
    achieved = N_SIMD * avg_ms * 1e6 / (w * n / 64.0)
    return floor, achieved


def median_ms(fn, reps=12, warm=2):
    import numpy as np
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3


def host_inclusive(model, geno, n, reps=12):
    """hibag_hip_predict with host pointers: pageable int32 genotypes up, calls / prob / matching / dosage down; one
    call = one 10k-sample batch, so nothing of it can overlap (the upload precedes the first kernel, the download
    follows the last).  Median of `reps` calls."""
    import ctypes as C
    import numpy as np
    from hibag_amd import _lib
    L = _lib.lib()
    nh = model.obj.n_hla
    h1 = np.zeros(n, np.int32); h2 = np.zeros(n, np.int32); pr = np.zeros(n); mt = np.zeros(n); ds = np.zeros((n, nh))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    g = np.ascontiguousarray(geno, np.int32)

    def call():
        _lib.check(L.hibag_hip_predict(model.handle, p(g), n, 1, p(h1), p(h2), p(pr), p(mt), p(ds), None))
    ms = median_ms(call, reps)
    return {"value": n / ms * 1e3, "unit": "samples/s", "ms_per_step": ms, "repetitions": reps,
            "what": "hibag_hip_predict: H2D of the int32 genotype matrix + kernels + D2H of H1, H2, prob, matching, dosage "
                    "(pageable host memory), median; never `value`"}


def api_inclusive(model, model_obj, geno, n, calls, reps=12):
    """hibag_amd.hlaPredict(model, hlaSNPGenoClass, type="response+dosage", verbose=FALSE) -- the function the metric is
    named after -- on the timed cohort: SNP matching and strand check on the annotation, the int32 genotypes from the
    caller's memory to the device (no host copy), kernels, calls / prob / matching / dosage back, the result object.
    Median of `reps` calls for the genotype matrix in R's memory order (column-major [SNP, sample]: what an R host holds,
    hibag_hip_predict) and in numpy's (row-major: hibag_hip_predict_snp_major)."""
    import numpy as np
    import hibag_amd
    from hibag_amd import synth
    res = {}
    for key, order in (("r_order", "F"), ("numpy_order", "C")):
        snp = synth.as_snp_geno(model_obj, geno, order=order)
        last = [None]

        def call():
            last[0] = hibag_amd.hlaPredict(model, snp, type="response+dosage", verbose=False)
        ms = median_ms(call, reps)
        r = last[0]
        res[key] = {"value": n / ms * 1e3, "ms_per_step": ms,
                    "calls_identical_to_timed_step": bool(np.array_equal(r.h1, calls[0]) and np.array_equal(r.h2, calls[1])),
                    "result": f"hlaAlleleClass: {len(r.sample_id)} samples, dosage {tuple(r.dosage.shape)}"}
    # the cohort in a PLINK BED file (two bits per genotype: a sixteenth of the int32 upload), decoded on the device
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        bed = synth.as_bed_geno(model_obj, geno, os.path.join(tmp, "cohort.bed"))
        last = [None]

        def call_bed():
            last[0] = hibag_amd.hlaPredict(model, bed, type="response+dosage", verbose=False)
        ms = median_ms(call_bed, reps)
        r = last[0]
        res["plink_bed"] = {"value": n / ms * 1e3, "ms_per_step": ms,
                            "calls_identical_to_timed_step": bool(np.array_equal(r.h1, calls[0]) and np.array_equal(r.h2, calls[1])),
                            "what": "hlaPredict(model, hlaBED2Geno(..., lazy=TRUE)): the file read (page cache), H2D of the packed bytes, "
                                    "decode + kernels (hibag_hip_predict_bed), D2H, the result object"}
    out = dict(res["r_order"])
    out.update({"unit": "samples/s", "repetitions": reps, "numpy_order": res["numpy_order"], "plink_bed": res["plink_bed"],
                "what": "hibag_amd.hlaPredict(model, hlaSNPGenoClass, type='response+dosage', verbose=False): annotation matching + "
                        "H2D of the int32 genotypes (pageable, the caller's own array) + kernels + D2H + the result object, median; "
                        "`value`: genotype matrix in R's memory order, `numpy_order`: row-major [SNP, sample] through "
                        "hibag_hip_predict_snp_major; never the line's `value`"})
    return out


def host_inclusive_cohort(model, model_obj, founders, afreq, dev, n_big):
    """The same entry on a cohort of several slices (BASELINE config 3's size on one GPU): upload of slice i + 1 and
    download of slice i - 1 run beside the kernels of slice i (pinned staging, three streams), against the same
    cohort resident in HBM."""
    import ctypes as C
    import numpy as np
    import torch
    from hibag_amd import _lib, synth
    L = _lib.lib()
    G, _ = synth.make_samples(founders, afreq, n_big, seed=synth.DEFAULT_SEED + 77)
    nh = model_obj.n_hla
    h1 = np.zeros(n_big, np.int32); h2 = np.zeros(n_big, np.int32); pr = np.zeros(n_big); mt = np.zeros(n_big); ds = np.zeros((n_big, nh))
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def call():
        _lib.check(L.hibag_hip_predict(model.handle, p(G), n_big, 1, p(h1), p(h2), p(pr), p(mt), p(ds), None))
    ms_host = median_ms(call, 5, 1)
    dg = torch.from_numpy(G).to(dev)
    d = [torch.empty(n_big, dtype=torch.int32, device=dev), torch.empty(n_big, dtype=torch.int32, device=dev),
         torch.empty(n_big, dtype=torch.float64, device=dev), torch.empty(n_big, dtype=torch.float64, device=dev),
         torch.empty((n_big, nh), dtype=torch.float64, device=dev)]
    st = torch.cuda.current_stream(dev).cuda_stream

    def resident():
        model.predict_device(dg.data_ptr(), n_big, 1, *[x.data_ptr() for x in d], None, stream=st)
        torch.cuda.synchronize(dev)
    ms_dev = median_ms(resident, 5, 1)
    return {"samples": n_big, "host_pointers_samples_per_s": n_big / ms_host * 1e3, "device_resident_samples_per_s": n_big / ms_dev * 1e3,
            "frac_of_device_resident": ms_dev / ms_host,
            "what": "hibag_hip_predict on 100,000 samples in host memory (slices pipelined: H2D of slice i+1 and D2H of slice i-1 "
                    "beside the kernels of slice i) against hibag_hip_predict_device on the same cohort in HBM; medians of 5"}


def m_pad(n):
    return (n + 63) // 64 * 64


def other_configs(K, faults=None):
    """BASELINE configs 4 and 5 at reduced repetition (the metric's config is the main line): cfg4 = the
    HLA-DRB1 shape (500 haplotypes per classifier), cfg5 = hlaAttrBagging() at 1,000 samples x 300 SNPs."""
    import numpy as np
    import torch
    import hibag_amd
    from hibag_amd import synth, train
    res = {}
    dev = torch.device("cuda", torch.cuda.current_device())
    # the metric's configuration with the other output sets of hlaPredict(): type = "response+prob" (the whole posterior
    # matrix, 10.2 KB per sample) and vote = "majority"; and through the reference's actual GPU hook, one sample per call
    try:
        obj, founders, af = synth.make_model(SHAPE)
        n2 = SAMPLES_PER_GPU
        G, truth = synth.make_samples(founders, af, n2, seed=synth.DEFAULT_SEED + 1)
        m = hibag_amd.hlaModelFromObj(obj)
        dg = torch.from_numpy(G).to(dev)
        h1 = torch.empty(n2, dtype=torch.int32, device=dev); h2 = torch.empty_like(h1)
        pr = torch.empty(n2, dtype=torch.float64, device=dev); mt = torch.empty_like(pr)
        ds = torch.empty((n2, obj.n_hla), dtype=torch.float64, device=dev)
        pp = torch.empty((n2, obj.n_cell), dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        for name, vote, post in (("cfg2_prob", 1, pp), ("cfg2_vote2", 2, None)):
            def one():
                m.predict_device(dg.data_ptr(), n2, vote, h1.data_ptr(), h2.data_ptr(), pr.data_ptr(), mt.data_ptr(), ds.data_ptr(),
                                 None if post is None else post.data_ptr(), stream=st)
                torch.cuda.synchronize(dev)
            ms = median_ms(one, 10)
            acc = float(np.mean((h1.cpu().numpy() == truth[:, 0]) & (h2.cpu().numpy() == truth[:, 1])))
            res[name] = {"samples_per_s": n2 / ms * 1e3, "ms_per_step": ms, "repetitions": 10, "call_accuracy_vs_truth": acc,
                         "handover_faults": int(m.handover_faults()),
                         "what": ("type='response+prob': also the posterior matrix, %d B per sample" % (8 * obj.n_cell)) if post is not None
                                 else "vote='majority' (k_vote_best + k_vote_tally instead of pass 2)"}
        m.close()
    except Exception as e:
        res["cfg2_prob"] = {"error": repr(e)}
    try:
        from hibag_amd.plugin import PluginHost
        obj, founders, af = synth.make_model(SHAPE)
        ns = 400
        G, truth = synth.make_samples(founders, af, ns, seed=synth.DEFAULT_SEED + 1)
        host = PluginHost(obj)                       # predict_init: the host's haplotype lists
        geno, wt = host.pack(G)                      # the host's own work per sample (IntToSNP, weights), done up front
        prob = np.zeros(obj.n_cell); match = np.zeros(1)
        for i in range(20):
            host.avg_prob(geno[i], wt[i], prob, match)
        nh = obj.n_hla
        cell = {(a, b): b + a * (2 * nh - a - 1) // 2 for a in range(nh) for b in range(a, nh)}
        want = np.array([cell[(min(a, b), max(a, b))] for a, b in truth])
        # the host's loop as a compiled host runs it (src/LibHLA.cpp:2362-2411: predict_avg_prob, then the arg-max scan, per sample)
        best, _, sec = host.avg_prob_loop(geno, wt)
        best, _, sec = host.avg_prob_loop(geno, wt)
        # ... and driven from Python, one ctypes call per sample (what earlier rounds reported)
        t = time.perf_counter()
        for i in range(ns):
            host.avg_prob(geno[i], wt[i], prob, match)
        dt = time.perf_counter() - t
        host.close()
        res["plugin_per_sample"] = {"samples_per_s": ns / sec, "us_per_call": sec / ns * 1e6, "samples": ns,
                                    "call_accuracy_vs_truth": float(np.mean(best == want)),
                                    "driven_from_python": {"samples_per_s": ns / dt, "us_per_call": dt / ns * 1e6},
                                    "what": "the reference's own GPU hook, TypeGPUExtProc.predict_avg_prob: ONE sample per call "
                                            "(src/LibHLA.cpp:2433-2441), genotypes packed by the host beforehand; the host's loop in C "
                                            "(hibag_hip_test_time_avg_prob: the call, then the arg-max scan of the posterior).  A call is one kernel "
                                            "(workgroup = classifier), genotypes / weights / posterior in host-mapped memory, its end polled -- "
                                            "latency-bound by construction, the batched entry is the product"}
    except Exception as e:
        res["plugin_per_sample"] = {"error": repr(e)}
    try:
        obj, founders, af = synth.make_model("hla-drb1")
        n4 = 4096
        G, truth = synth.make_samples(founders, af, n4)
        t = time.perf_counter()
        m = hibag_amd.hlaModelFromObj(obj)
        t_fin = time.perf_counter() - t
        dg = torch.from_numpy(G).to(dev)
        h1 = torch.empty(n4, dtype=torch.int32, device=dev); h2 = torch.empty_like(h1)
        pr = torch.empty(n4, dtype=torch.float64, device=dev); mt = torch.empty_like(pr)
        ds = torch.empty((n4, obj.n_hla), dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        run = lambda: m.predict_device(dg.data_ptr(), n4, 1, h1.data_ptr(), h2.data_ptr(), pr.data_ptr(), mt.data_ptr(),
                                       ds.data_ptr(), None, stream=st)
        run(); torch.cuda.synchronize(dev)
        m.set_timing(True); m.reset_timing()
        steps = 3
        t = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t) / steps
        tm = m.get_timing(); m.set_timing(False)
        acc_ms = tm["accum"][0] / max(tm["accum"][1], 1)
        tot_ms = tm["total"][0] / max(tm["total"][1], 1)
        floor, ach_tot = issue_floor(obj, tot_ms, n4, K, m)
        stored = m.stored_cells()
        acc = float(np.mean((h1.cpu().numpy() == truth[:, 0]) & (h2.cpu().numpy() == truth[:, 1])))
        res["cfg4_hla_drb1"] = {"samples_per_s": n4 / dt, "samples": n4, "ms_per_step": dt * 1e3,
                                "pair_evals_per_sample": obj.pair_evals_per_sample(),
                                "pair_evals_per_s": n4 / dt * obj.pair_evals_per_sample(),
                                "kernels_ms_per_step": {k: round(v[0] / steps, 3) for k, v in tm.items()},
                                "k_total_issue_frac": round(floor / ach_tot, 4),
                                "model_finalize_s": round(t_fin, 2), "call_accuracy_vs_truth": acc}
        if stored:          # pass 1 stores the cell sums, pass 2 (k_accum_cells) reads them back: HBM-bound
            gb = stored * 8.0 * m_pad(n4) / 1e9
            res["cfg4_hla_drb1"].update({"pass2": "reads back the cell sums pass 1 stored", "stored_cells_per_sample": stored,
                                         "stored_gb_per_step": round(gb, 3),
                                         "pass2_hbm_read_gb_per_s": round(gb / (acc_ms * 1e-3), 1),
                                         "pass2_hbm_frac_of_8tb_s": round(gb / (acc_ms * 1e-3) / 8000.0, 3)})
        else:
            floor2, ach2 = issue_floor(obj, acc_ms, n4, K, m)
            res["cfg4_hla_drb1"].update({"pass2": "evaluates every haplotype pair again", "k_accum_issue_frac": round(floor2 / ach2, 4)})
        # the same model on twice the batch: at 4,096 samples pass 1 is 1,600 work items on ~1,300 resident workgroups, i.e.
        # as long as its longest item; from two rounds on the denser build of pass 1 takes over
        n8 = 2 * n4
        G8, _ = synth.make_samples(founders, af, n8)
        dg8 = torch.from_numpy(G8).to(dev)
        o8 = [torch.empty(n8, dtype=torch.int32, device=dev), torch.empty(n8, dtype=torch.int32, device=dev),
              torch.empty(n8, dtype=torch.float64, device=dev), torch.empty(n8, dtype=torch.float64, device=dev),
              torch.empty((n8, obj.n_hla), dtype=torch.float64, device=dev)]
        run8 = lambda: m.predict_device(dg8.data_ptr(), n8, 1, *[x.data_ptr() for x in o8], None, stream=st)
        run8(); torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(2):
            run8()
        torch.cuda.synchronize(dev)
        dt8 = (time.perf_counter() - t) / 2
        res["cfg4_hla_drb1"]["at_twice_the_batch"] = {"samples": n8, "samples_per_s": n8 / dt8, "ms_per_step": dt8 * 1e3}
        res["cfg4_hla_drb1"]["handover_faults"] = int(m.handover_faults())
        m.close()
        # the same model on the host cores: the AVX2 + threads port of the reference's kernel on a bounded sample
        from oracle import oracle as O
        O.build()
        cores, cores_note = usable_cores()
        nc = 768
        t = time.perf_counter()
        ref = O.predict(O.flatten(obj), G[:nc], avx2=True, n_threads=cores, want_dosage=True, want_prob=False)
        dtc = time.perf_counter() - t
        res["cfg4_hla_drb1"]["cpu_baseline"] = {
            "value": nc / dtc, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"the first {nc} samples of the timed batch ({dtc:.1f} s), same model, same outputs; AVX2 4-wide inner loop + "
                      f"{cores} threads over samples ({cores_note})",
            "calls_identical_to_gpu": bool(np.array_equal(ref["h1"], h1.cpu().numpy()[:nc]) and np.array_equal(ref["h2"], h2.cpu().numpy()[:nc]))}
        res["cfg4_hla_drb1"]["speedup_vs_cpu_baseline"] = res["cfg4_hla_drb1"]["samples_per_s"] / res["cfg4_hla_drb1"]["cpu_baseline"]["value"]
    except Exception as e:                       # an extra must not take the metric line down
        res["cfg4_hla_drb1"] = {"error": repr(e)}
    try:
        mdl, founders, af = synth.make_model("hla-b", seed=9, n_snp=300, n_classifier=1, wide_classifier=False)
        G, truth = synth.make_samples(founders, af, 1000, seed=10)
        mtry = int(np.ceil(np.sqrt(300)))

        def train_rate(threads, ncl, em="auto"):
            tr = train._Trainer(G, truth[:, 0], truth[:, 1], mdl.n_hla)
            tr.set_em_mode(em)
            if threads:
                tr.set_threads(threads)
            used = tr.threads
            tr.set_seed(100)
            tr.new_classifiers(1, mtry, True, False, False)           # warm-up (allocations, first launches)
            t = time.perf_counter()
            tr.new_classifiers(ncl, mtry, True, False, False)
            dt = (time.perf_counter() - t) / ncl
            everything = tr.classifiers()
            tr.close()
            return dt, used, everything
        ncl = 100                                                     # BASELINE config 5: 100 individual classifiers
        dt, used, grown = train_rate(0, ncl)
        cls = grown[1:]
        res["cfg5_training"] = {"s_per_classifier": dt, "classifiers_per_s": 1.0 / dt, "model_of_100_classifiers_s": dt * ncl,
                                "threads": used, "n_samp": 1000, "n_snp": 300,
                                "n_hla": mdl.n_hla, "mtry": mtry, "classifiers_timed": ncl,
                                "mean_snps": float(np.mean([len(c.snpidx) for c in cls])),
                                "mean_haplo": float(np.mean([len(c.freq) for c in cls]))}
        # What a rank gets of the host when eight of them share it (usable CPUs / 8, the default under LOCAL_WORLD_SIZE=8),
        # and the 8-rank rate that projects to: classifiers are independent, every rank grows its share (no collective on
        # the compute path), so the model of 100 takes ceil(100 / 8) classifiers' time at that thread count.
        cores, _ = usable_cores()
        per_rank = max(1, cores // 8)
        dt8, used8, _ = train_rate(per_rank, 24)
        dt8h, _, _ = train_rate(per_rank, 24, "host")
        res["cfg5_training"]["em_fits"] = ("host threads (more than two of them: the faster place)" if used > 2 else "device (hibag_em.hip)")
        res["cfg5_training"]["at_one_eighth_of_the_host"] = {
            "threads": used8, "s_per_classifier": dt8, "slowdown_vs_all_cores": dt8 / dt, "classifiers_timed": 24,
            "em_fits": "device (two host threads or fewer: hibag_em.hip, workgroup = candidate)" if used8 <= 2 else "host threads",
            "s_per_classifier_with_the_fits_on_the_host_threads": dt8h,
            "projected_8_ranks": {"classifiers_per_s": 8.0 / dt8, "model_of_100_classifiers_s": -(-ncl // 8) * dt8,
                                  "speedup_vs_one_rank_with_all_cores": (dt * ncl) / (-(-ncl // 8) * dt8),
                                  "what": "eight ranks of one node, each a trainer on its own GPU with 1/8 of the host's usable CPUs "
                                          "(hibag_hip_trainer_set_threads; the default under LOCAL_WORLD_SIZE=8); projected from the "
                                          "one-GPU measurement at that thread count, no 8-GPU node was available to the build"}}
        from oracle import oracle as O
        O.build()
        # Several trainers side by side on the one device (train.grow_concurrently), their device work fused into one launch
        # per kind of operation (csrc/hibag_combine.h) and their host threads under a BUDGET: sixteen trainers (one thread each,
        # EM fits on the device, R's stream seeded with seed + r like hlaParallelAttrBagging's workers), of which at most
        # `budget` are runnable at a time -- a trainer that waits for the device sleeps.  budget 4 = the headline; budget =
        # cores / 8 = what a rank of an eight-GPU node gets; and for reference round 5's arrangement (4 trainers x 4 threads,
        # EM fits on the host threads, a stream per trainer).
        try:
            import resource
            import threading
            from hibag_amd.dist import shard_bounds
            conc = {}
            best_rate, best_cls, best_k, best_key = 0.0, None, 1, None

            def concurrent(key, k, per, em, budget, combine=True, n=None):
                nonlocal best_rate, best_cls, best_k, best_key
                n = n or ncl
                kw = dict(em=em, combine=combine, thread_budget=budget)
                train.grow_concurrently(G, truth[:, 0], truth[:, 1], mdl.n_hla, k, mtry, True, k, per, 100, **kw)      # warm-up
                r0, t = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
                got = train.grow_concurrently(G, truth[:, 0], truth[:, 1], mdl.n_hla, n, mtry, True, k, per, 100, **kw)
                dtk = time.perf_counter() - t
                r1 = resource.getrusage(resource.RUSAGE_SELF)
                conc[key] = {"trainers": k, "threads_per_trainer": per, "em_fits": em, "host_thread_budget": budget or None,
                             "launches": "fused across trainers" if combine else "a stream per trainer",
                             "classifiers": n, "seconds": dtk, "classifiers_per_s": n / dtk,
                             "host_cores_busy": round((r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime) / dtk, 2)}
                return got
            cls4 = concurrent("16_trainers_budget_4", 16, 1, "device", min(4, cores))
            concurrent("16_trainers_budget_cores_over_8", 16, 1, "device", per_rank)
            concurrent("4x4_host_em_own_streams", 4, max(1, cores // 4), "host", 0, combine=False)
            # the same two budgets on a job several models long (40 classifiers per trainer): with the configuration's 100
            # classifiers a trainer grows six or seven, and the trainers that finish first leave the device to the last ones
            concurrent("steady_16_trainers_budget_4", 16, 1, "device", min(4, cores), n=640)
            concurrent("steady_16_trainers_budget_cores_over_8", 16, 1, "device", per_rank, n=640)
            concurrent("steady_32_trainers_budget_4", 32, 1, "device", min(4, cores), n=960)
            concurrent("steady_32_trainers_no_budget", 32, 1, "device", 0, n=960)
            best_k, best_cls, best_key = 16, cls4, "16_trainers_budget_4"
            best_rate = conc[best_key]["classifiers_per_s"]
            res["cfg5_training"]["concurrent_trainers"] = conc
            res["cfg5_training"]["single_trainer_classifiers_per_s"] = 1.0 / dt
            # checked against the oracle: the first two classifiers of EVERY trainer's stream (one oracle run per stream,
            # the streams side by side on the host's threads)
            chk = [None] * best_k

            def check(r):
                lo, hi = shard_bounds(ncl, best_k, r)
                want = O.train(G, truth[:, 0], truth[:, 1], mdl.n_hla, min(2, hi - lo), mtry, True, 100 + r)
                chk[r] = all(np.array_equal(a.snpidx, b["snpidx"]) and np.array_equal(a.freq, b["freq"]) and a.haplo == b["haplo"]
                             and np.array_equal(a.samp_num, b["samp_num"]) for a, b in zip(best_cls[lo:lo + 2], want))
            th = [threading.Thread(target=check, args=(r,)) for r in range(best_k)]
            [x.start() for x in th]; [x.join() for x in th]
            one8 = conc["16_trainers_budget_cores_over_8"]
            res["cfg5_training"].update({
                "classifiers_per_s": best_rate, "s_per_classifier": 1.0 / best_rate, "model_of_100_classifiers_s": ncl / best_rate,
                "trainers": best_k, "threads": min(4, cores), "host_cores_busy": conc[best_key]["host_cores_busy"],
                "what": f"{best_k} trainers side by side on one MI355X, one fused launch per kind of device work for all of them, EM fits on "
                        f"the device, at most {min(4, cores)} of their host threads runnable at a time; stream r seeded with 100 + r; a single "
                        f"trainer with all {cores} threads: {1.0 / dt:.1f} classifiers/s",
                "oracle_check": {"classifiers_compared": 2 * best_k, "what": "the first two classifiers of every trainer's stream",
                                 "identical": bool(all(chk))}})
            res["cfg5_training"]["at_one_eighth_of_the_host"].update({
                "concurrent": one8, "classifiers_per_s": one8["classifiers_per_s"],
                "slowdown_vs_budget_4": best_rate / one8["classifiers_per_s"],
                "steady_state_slowdown_vs_budget_4": conc["steady_16_trainers_budget_4"]["classifiers_per_s"] / conc["steady_16_trainers_budget_cores_over_8"]["classifiers_per_s"],
                "projected_8_ranks_concurrent": {"classifiers_per_s": 8 * one8["classifiers_per_s"],
                                                 "what": f"eight ranks, each sixteen trainers on its own GPU under a budget of {per_rank} host threads "
                                                         "(measured on one GPU at that budget; no 8-GPU node was available to the build)"}})
        except Exception as e:
            res["cfg5_training"]["concurrent_trainers"] = {"error": repr(e)}
        # the oracle's one-core restatement of the reference's training driver on the same data, same random stream
        t = time.perf_counter()
        oc = O.train(G, truth[:, 0], truth[:, 1], mdl.n_hla, 2, mtry, True, 100)
        dto = (time.perf_counter() - t) / 2
        same = all(np.array_equal(a.snpidx, b["snpidx"]) and np.array_equal(a.freq, b["freq"]) and a.haplo == b["haplo"]
                   for a, b in zip(grown[:2], oc))
        res["cfg5_training"]["cpu_baseline"] = {
            "value": 1.0 / dto, "unit": "classifiers/s", "s_per_classifier": dto, "cores": 1, "kind": "port",
            "sample": f"the first 2 classifiers of the same training run (seed 100, {dto * 2:.1f} s): oracle/hibag_oracle_train.c, the "
                      "reference's driver restated for ONE core",
            "classifiers_identical_to_gpu": bool(same)}
        # The reference threads training too (HIBAG_NewClassifiers runs BuildClassifiers inside tbb::task_arena(nthread),
        # src/HIBAG.cpp:599-634; PARALLEL_FOR in the EM, src/LibHLA.cpp:1104/:1159/:1204, and in _OutOfBagAccuracy / _InBagLogLik,
        # :1944/:1966), so the GPU line's host-thread budget is matched on the CPU side: `cores` independent oracle streams side
        # by side, one classifier each (classifiers are independent: the best the host can do with those cores).
        import threading
        tcpu = time.perf_counter()
        th = [threading.Thread(target=lambda r=r: O.train(G, truth[:, 0], truth[:, 1], mdl.n_hla, 1, mtry, True, 200 + r)) for r in range(cores)]
        [x.start() for x in th]; [x.join() for x in th]
        dtc = time.perf_counter() - tcpu
        res["cfg5_training"]["cpu_baseline_all_cores"] = {
            "value": cores / dtc, "unit": "classifiers/s", "cores": cores, "kind": "port",
            "sample": f"{cores} oracle trainers side by side, one classifier each, seeds 200.. ({dtc:.1f} s)"}
        res["cfg5_training"]["speedup_vs_one_cpu_thread"] = dto * res["cfg5_training"]["classifiers_per_s"]
        res["cfg5_training"]["speedup_vs_cpu_baseline"] = res["cfg5_training"]["classifiers_per_s"] / (cores / dtc)
        res["cfg5_training"]["speedup_note"] = ("speedup_vs_cpu_baseline: the GPU line (device + a budget of `threads` host threads) against ALL "
                                                f"{cores} usable cores running the CPU port; speedup_vs_one_cpu_thread: against one core")
    except Exception as e:
        res["cfg5_training"] = dict(res.get("cfg5_training", {}), error=repr(e))
    # A REAL model: the reference's bundled HLA-A model (inst/extdata/ModelList.RData: 100 classifiers, 14 alleles, 18-87
    # haplotypes and 10-24 SNPs per classifier, 91,645 haplotype pairs per sample) on 10,000 samples resampled from its 60
    # HapMap genotypes (data/HapMap_CEU_Geno.rdata) -- every tuning decision of the two passes was made on the synthetic
    # HLA-B shape, whose haplotype counts are permuted per classifier; a real model's common alleles own the haplotypes in
    # every classifier.  (Both files are the repository's committed fixtures, byte-identical to the reference's.)
    try:
        res["cfg1_real_model"] = real_model_config(K, dev)
    except Exception as e:
        res["cfg1_real_model"] = {"error": repr(e)}
    # classifiers of 33 .. 112 SNPs only (FP4 in two to four K steps: k_total_wide + k_total_scan, every cell stored): the
    # shape of tests/test_hip_parity.py::test_only_wide_classifiers at the benchmark's batch size
    try:
        ks = [33, 40, 56, 57, 84, 85, 100, 112] * 4
        obj, founders, af = synth.make_model("hla-b", seed=31, n_classifier=len(ks), n_snp=150, snp_counts=ks, wide_classifier=False)
        nw = SAMPLES_PER_GPU
        G, truth = synth.make_samples(founders, af, nw, seed=32)
        m = hibag_amd.hlaModelFromObj(obj)
        dg = torch.from_numpy(G).to(dev)
        o = [torch.empty(nw, dtype=torch.int32, device=dev), torch.empty(nw, dtype=torch.int32, device=dev),
             torch.empty(nw, dtype=torch.float64, device=dev), torch.empty(nw, dtype=torch.float64, device=dev),
             torch.empty((nw, obj.n_hla), dtype=torch.float64, device=dev)]
        st = torch.cuda.current_stream(dev).cuda_stream
        run = lambda: m.predict_device(dg.data_ptr(), nw, 1, *[x.data_ptr() for x in o], None, stream=st)
        run(); torch.cuda.synchronize(dev)
        m.set_timing(True); m.reset_timing()
        steps = 5
        t = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t) / steps
        tm = m.get_timing(); m.set_timing(False)
        tot_ms = tm["total"][0] / max(tm["total"][1], 1)
        floor, ach = issue_floor(obj, tot_ms, nw, K, m)
        res["wide_classifiers"] = {"samples_per_s": nw / dt, "samples": nw, "ms_per_step": dt * 1e3,
                                   "snps_per_classifier": ks[:8], "classifiers": len(ks),
                                   "pair_evals_per_sample": obj.pair_evals_per_sample(),
                                   "kernels_ms_per_step": {k: round(v[0] / steps, 3) for k, v in tm.items()},
                                   "pass1_frac_of_multi_step_fp4_floor": round(floor / ach, 4),
                                   "handover_faults": int(m.handover_faults()),
                                   "call_accuracy_vs_truth": float(np.mean((o[0].cpu().numpy() == truth[:, 0]) & (o[1].cpu().numpy() == truth[:, 1]))),
                                   "what": "every classifier 33..112 SNPs: FP4 distances in two to four K steps chained through the accumulator; "
                                           "pass 1 = k_total_wide + k_total_scan (the 'total' timer), every cell sum stored, pass 2 reads them back"}
        m.close()
    except Exception as e:
        res["wide_classifiers"] = {"error": repr(e)}
    if faults is not None:
        for k, v in res.items():
            if isinstance(v, dict) and "handover_faults" in v:
                faults[k] = v["handover_faults"]
    return res


def real_model_config(K, dev):
    """other_configs.cfg1_real_model: the bundled HLA-A model on 10,000 resampled HapMap samples (see other_configs)."""
    import numpy as np
    import torch
    import hibag_amd
    from hibag_amd import model as Mdl
    ref = os.path.join(ROOT, "tests", "golden", "reference_data")
    obj = Mdl.load_model(os.path.join(ref, "ModelList.RData"), "modellist", "A")
    geno = Mdl.load_geno(os.path.join(ref, "HapMap_CEU_Geno.rdata"))
    gi = {sid: i for i, sid in enumerate(geno.snp_id)}
    G60 = geno.sample_major([gi[sid] for sid in obj.snp_id])              # [60][266], the model's SNPs by rs id (same alleles)
    n = SAMPLES_PER_GPU
    rng = np.random.default_rng(20260515)
    G = np.ascontiguousarray(G60[rng.integers(0, len(G60), n)])
    miss = rng.random(G.shape) < 0.01                                      # a cohort's missing calls: 1 % of the genotypes
    G[miss] = hibag_amd.NA_INTEGER
    m = hibag_amd.hlaModelFromObj(obj)
    dg = torch.from_numpy(G).to(dev)
    o = [torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
         torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
         torch.empty((n, obj.n_hla), dtype=torch.float64, device=dev)]
    st = torch.cuda.current_stream(dev).cuda_stream
    run = lambda: m.predict_device(dg.data_ptr(), n, 1, *[x.data_ptr() for x in o], None, stream=st)
    for _ in range(3):
        run()
    torch.cuda.synchronize(dev)
    m.set_timing(True); m.reset_timing()
    steps = 20
    t = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t) / steps
    tm = m.get_timing(); m.set_timing(False)
    ms1 = tm["total"][0] / max(tm["total"][1], 1)
    ms2 = tm["accum"][0] / max(tm["accum"][1], 1)
    floor, ach1 = issue_floor(obj, ms1, n, K, m)
    pairs = obj.pair_evals_per_sample()
    pairs2, stored = m.second_pass_pairs(), m.stored_cells()
    cells = sum(int(np.sum(np.bincount(c.hla, minlength=obj.n_hla) > 0)) for c in obj.classifiers)   # (alleles present per classifier)
    out = {"samples_per_s": n / dt, "samples": n, "ms_per_step": dt * 1e3, "steps": steps,
           "model": "inst/extdata/ModelList.RData $A (HLA-A, 100 classifiers, 14 alleles)", "pair_evals_per_sample": pairs,
           "pair_evals_per_s": n / dt * pairs,
           "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in tm.items()},
           "pass2": {"pairs_evaluated_again_per_sample": pairs2, "cell_sums_stored_per_sample": stored,
                     "mode": "every cell stored (k_accum_cells)" if pairs2 == 0 and stored else "hybrid (k_accum)"},
           "alleles_present_per_classifier_mean": cells / max(len(obj.classifiers), 1),
           "issue": {"floor_ns_per_wave_pair": round(floor, 2), "k_total_ns_per_wave_pair": round(ach1, 2),
                     "k_total_frac": round(floor / ach1, 4),
                     "both_passes_frac": round(floor / (N_SIMD * (ms1 + ms2) * 1e6 / ((pairs + pairs2) * n / 64.0)), 4)},
           "handover_faults": int(m.handover_faults())}
    # against the oracle: 200 of the timed samples, every output
    from oracle import oracle as O
    O.build()
    sub = np.arange(0, n, n // 200)[:200]
    got = m.predict_raw(G[sub], 1, want_dosage=True, want_prob=True)
    want = O.predict(O.flatten(obj), G[sub], vote_method=1, avx2=True, n_threads=usable_cores()[0], want_dosage=True, want_prob=True)
    out["oracle_check"] = {"samples": int(len(sub)),
                           "bit_identical": bool(all(np.array_equal(got[k], want[k], equal_nan=True)
                                                     for k in ("h1", "h2", "prob", "matching", "dosage", "postprob")))}
    h1 = o[0].cpu().numpy(); h2 = o[1].cpu().numpy()
    m.close()
    cores, cores_note = usable_cores()
    fm = O.flatten(obj)
    t = time.perf_counter()
    r1 = O.predict(fm, G, avx2=True, n_threads=cores, want_dosage=True, want_prob=False)
    rate = n / max(time.perf_counter() - t, 1e-6)
    reps = int(min(32, max(1, round(rate * 8.0 / n))))
    big = np.ascontiguousarray(np.tile(G, (reps, 1)))
    t = time.perf_counter()
    O.predict(fm, big, avx2=True, n_threads=cores, want_dosage=True, want_prob=False)
    dtc = time.perf_counter() - t
    out["cpu_baseline"] = {"value": len(big) / dtc, "unit": "samples/s", "cores": cores, "kind": "port",
                           "sample": f"the {n} timed samples repeated {reps}x ({dtc:.1f} s), same model, same outputs; AVX2 4-wide inner loop + "
                                     f"{cores} threads over samples ({cores_note})",
                           "calls_identical_to_gpu": bool(np.array_equal(r1["h1"], h1) and np.array_equal(r1["h2"], h2))}
    out["speedup_vs_cpu_baseline"] = out["samples_per_s"] / out["cpu_baseline"]["value"]
    return out


def main_threads(args):
    """`--launcher threads`: ONE process, no torch.distributed -- what an R / C++ host does with a node's GPUs through the
    C ABI (INTEGRATION.md sections B and C).
      --shard samples (default)   one host thread and one replica of the model per GPU (hibag_hip_model_replicate), each on
                                  its slice of config 3's 100,000 samples (or its own `--samples` n) through the device
                                  entry; a barrier over the threads and a device synchronisation on both sides of exactly K
                                  steps, the slowest thread's time
      --shard classifiers         hibag_hip_model_shard + hibag_hip_shard_group_predict: every device a slice of the
                                  classifiers, the partial posterior sums merged by the ONE ncclAllReduce per batch that
                                  libhibag_hip.so issues itself; host pointers in and out, so transfers are inside the time."""
    import threading
    import numpy as np
    import torch
    import hibag_amd
    from hibag_amd import synth
    from hibag_amd import dist as hdist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: hibag_amd has no CPU fallback")
    n_dev = torch.cuda.device_count()
    N = args.gpus
    devices = [i % n_dev for i in range(N)]          # (more replicas than devices: several on one, for a dry run on a 1-GPU box)
    target = hibag_amd.hlaSetKernelTarget("hip")[0]
    model_obj, founders, afreq = synth.make_model(args.shape, wide_classifier=not args.no_wide)
    vote_method = 2 if args.vote == "majority" else 1
    strong = args.samples is None and N > 1
    n_total = CFG3_SAMPLES if strong else (args.samples if args.samples is not None else SAMPLES_PER_GPU) * (1 if args.shard == "classifiers" else N)
    pair_evals = model_obj.pair_evals_per_sample()
    first = hibag_amd.hlaModelFromObj(model_obj, device=devices[0])

    if args.shard == "classifiers":
        from hibag_amd.hibag import ShardGroup
        n = n_total
        geno, truth = synth.make_samples(founders, afreq, n, seed=synth.DEFAULT_SEED + 1)
        grp = ShardGroup(first, devices)
        for _ in range(max(1, min(args.warmup, 2))):
            got = grp.predict_raw(geno, want_dosage=True)
        steps = max(1, min(args.steps, 10))
        t0 = time.perf_counter()
        for _ in range(steps):
            got = grp.predict_raw(geno, want_dosage=True)
        dt = time.perf_counter() - t0
        ref = first.predict_raw(geno[:10_000], 1, want_dosage=False, want_prob=False)
        out = {
            "metric": "hlaPredict() samples/sec, 10k samples x 100-classifier HLA-B",
            "value": n * steps / dt, "unit": "samples/s", "n_gpus": N, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload_text(args, model_obj, n, n, N, True).replace(
                           "genotypes and outputs resident in HBM (transfers excluded: `host_inclusive` is the PCIe-inclusive rate of the host-pointer entry)",
                           "HOST pointers in and out (upload of the genotypes to every rank and download of the outputs are inside the time)"),
                       "samples_per_step": n,
                       "parallelism": f"classifier-sharded x{N} inside libhibag_hip.so: {len(grp.shards)} shards on devices {devices}, "
                                      f"{grp.ranks} RCCL rank(s), one ncclAllReduce per batch of the library's choosing",
                       "kernel_target": target,
                       "launcher": "threads: one process, the calling thread drives every device (hibag_hip_shard_group_predict)"},
            "pair_evals_per_s": n * steps / dt * pair_evals,
            "call_accuracy_vs_truth": float(np.mean((got["h1"] == truth[:, 0]) & (got["h2"] == truth[:, 1]))),
            "rccl_ranks": grp.ranks, "rccl_allreduces": grp.allreduces,
            "rccl_version_code": int(hibag_amd._lib.lib().hibag_hip_rccl_version()),
            "classifier_shard_check": {"calls_identical_to_unsharded": bool(np.array_equal(ref["h1"], got["h1"][:10_000]) and
                                                                            np.array_equal(ref["h2"], got["h2"][:10_000])),
                                       "max_rel_dev_prob_vs_unsharded": float(np.nanmax(np.abs(got["prob"][:10_000] - ref["prob"]) /
                                                                                        np.maximum(np.abs(ref["prob"]), 1e-300))),
                                       "tolerance": 1e-10},
            "handover_faults": {"shards": int(sum(m.handover_faults() for m in grp.shards))},
        }
        print(json.dumps(out), flush=True)
        grp.close(); first.close()
        return

    models = [first] + [first.replicate(d) for d in devices[1:]]
    geno_all = truth_all = None
    if strong:
        geno_all, truth_all = synth.make_samples(founders, afreq, n_total, seed=synth.DEFAULT_SEED + 1)
    bar = threading.Barrier(N)
    times, accs, errs, sizes = [0.0] * N, [0.0] * N, [], [0] * N

    def worker(r):
        try:
            dev = torch.device("cuda", devices[r])
            torch.cuda.set_device(dev)
            if strong:
                lo, hi = hdist.shard_bounds(n_total, N, r)
                geno, truth = np.ascontiguousarray(geno_all[lo:hi]), truth_all[lo:hi]
            else:
                geno, truth = synth.make_samples(founders, afreq, n_total // N, seed=synth.DEFAULT_SEED + 1 + r)
            n = len(geno)
            sizes[r] = n
            m = models[r]
            dg = torch.from_numpy(geno).to(dev)
            h1 = torch.empty(n, dtype=torch.int32, device=dev); h2 = torch.empty_like(h1)
            pr = torch.empty(n, dtype=torch.float64, device=dev); mt = torch.empty_like(pr)
            ds = torch.empty((n, model_obj.n_hla), dtype=torch.float64, device=dev)
            st = torch.cuda.Stream(dev)               # (a stream per thread: replicas on one device run side by side)

            def step():
                m.predict_device(dg.data_ptr(), n, vote_method, h1.data_ptr(), h2.data_ptr(), pr.data_ptr(), mt.data_ptr(),
                                 ds.data_ptr(), None, stream=st.cuda_stream)
            for _ in range(args.warmup):
                step()
            st.synchronize(); bar.wait(); st.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            st.synchronize(); bar.wait(); st.synchronize()
            times[r] = time.perf_counter() - t0
            if m.status() != 0:
                raise RuntimeError("a launch reported a failed hand-over")
            accs[r] = float(np.mean((h1.cpu().numpy() == truth[:, 0]) & (h2.cpu().numpy() == truth[:, 1])))
        except Exception as e:                        # noqa: BLE001
            errs.append(repr(e))
            bar.abort()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(N)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        sys.exit("bench.py --launcher threads: " + "; ".join(errs))
    dt = max(times)
    total = sum(sizes)
    out = {
        "metric": "hlaPredict() samples/sec, 10k samples x 100-classifier HLA-B",
        "value": total * args.steps / dt, "unit": "samples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload_text(args, model_obj, sizes[0], total, N, strong),
                   "samples_per_gpu": sizes[0], "samples_per_step": total,
                   "parallelism": f"sample-sharded x{N} (contiguous slices, no collective)", "kernel_target": target,
                   "launcher": f"threads: one process, one host thread and one model replica per GPU through the C ABI "
                               f"(devices {devices})"},
        "pair_evals_per_s": total * args.steps / dt * pair_evals,
        "call_accuracy_vs_truth": float(np.mean(accs)), "rccl_ranks": None,
        "handover_faults": {"replicas": int(sum(m.handover_faults() for m in models))},
    }
    print(json.dumps(out), flush=True)
    for m in models[1:]:
        m.close()
    first.close()


def spawn_ranks(n_gpus):
    """Start `n_gpus` ranks of this script with torch.distributed.run as a child process and relay rank 0's line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    return p.returncode if (p.returncode != 0 or line is not None) else 1


def usable_cores():
    """Threads worth starting: the CPU affinity mask capped by the cgroup CPU quota (the GPU box
    exposes all hardware threads but limits the container's CPU time)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    note = f"{n} hardware threads visible"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period) + 0.5))
            if q < n:
                note += f", cgroup CPU quota {q}"
                n = q
    except (OSError, ValueError):
        pass
    return n, note


def cpu_baseline(model_obj, geno, gpu_h1, gpu_h2):
    """AVX2 + threads port of the reference's CPU kernel (oracle/hibag_oracle_avx2.c) on all host
    cores.  The timed sample is the benchmark batch repeated until it holds about 12 s of CPU work
    (so that every thread has enough samples to amortise its start-up); the first pass over the
    batch also cross-checks the GPU's calls."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    fm = O.flatten(model_obj)
    cores, cores_note = usable_cores()
    n = len(geno)
    t = time.perf_counter()
    ref = O.predict(fm, geno, avx2=True, n_threads=cores, want_dosage=True, want_prob=False)
    rate = n / max(time.perf_counter() - t, 1e-6)
    same = bool(np.array_equal(ref["h1"], gpu_h1) and np.array_equal(ref["h2"], gpu_h2))
    reps = int(min(64, max(1, round(rate * 12.0 / n))))
    big = np.ascontiguousarray(np.tile(geno, (reps, 1)))
    t = time.perf_counter()
    O.predict(fm, big, avx2=True, n_threads=cores, want_dosage=True, want_prob=False)
    dt = time.perf_counter() - t
    # BASELINE.md section 3's other column: ONE thread of the same port, on ~5 s of the same batch (the AVX2 kernel of the
    # reference on one core; rate from the threaded run / cores sizes the sample)
    n1 = int(min(n, max(64, round(len(big) / dt / max(cores, 1) * 5.0))))
    t = time.perf_counter()
    one = O.predict(fm, geno[:n1], avx2=True, n_threads=1, want_dosage=True, want_prob=False)
    dt1 = time.perf_counter() - t
    same1 = bool(np.array_equal(one["h1"], gpu_h1[:n1]) and np.array_equal(one["h2"], gpu_h2[:n1]))
    return {"value": len(big) / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"the {n} samples of the timed batch repeated {reps}x ({len(big)} samples, {dt:.1f} s), same model, "
                      f"same outputs; AVX2 4-wide inner loop + {cores} threads over samples ({cores_note})",
            "calls_identical_to_gpu": same,
            "one_thread": {"value": n1 / dt1, "unit": "samples/s", "cores": 1, "kind": "port",
                           "sample": f"the first {n1} samples of the timed batch ({dt1:.1f} s), same model, same outputs; the AVX2 "
                                     "4-wide inner loop on one thread", "calls_identical_to_gpu": same1}}


if __name__ == "__main__":
    main()
