"""Training at BASELINE config 5's shape with K trainers side by side on one device (train.grow_concurrently), T host threads
each, EM fits on the host threads or on the device: classifiers per second for a list of (K, T, em) settings.

    python tools/train_concurrency.py "1x2:device 8x2:device 16x1:device:2 4x4:host 8x1:device:0:nc" [classifiers per setting]

A setting is trainers x threads-per-trainer : em [: host-thread budget (0 = none) [: nc = every trainer on its own stream,
no combined launches]].  Prints classifiers/s and the process's CPU seconds per wall second (cores kept busy).
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import hibag_amd
from hibag_amd import synth, train
hibag_amd.hlaSetKernelTarget("hip")
mdl, founders, af = synth.make_model("hla-b", seed=9, n_snp=300, n_classifier=1, wide_classifier=False)
G, truth = synth.make_samples(founders, af, 1000, seed=10)
mtry = int(np.ceil(np.sqrt(300)))
settings = (sys.argv[1] if len(sys.argv) > 1 else "1x2:device 4x4:host 8x2:device").split()
ncl = int(sys.argv[2]) if len(sys.argv) > 2 else 64
import ctypes as C
import resource
from hibag_amd import _lib
for s in settings:
    f = s.split(":")
    kt, em = f[0], f[1]
    budget = int(f[2]) if len(f) > 2 else 0
    combine = not (len(f) > 3 and f[3] == "nc")
    k, t = (int(v) for v in kt.split("x"))
    kw = dict(em=em, combine=combine, thread_budget=budget)
    train.grow_concurrently(G, truth[:, 0], truth[:, 1], mdl.n_hla, k, mtry, True, k, t, 100, **kw)      # warm-up
    _lib.lib().hibag_hip_train_combine_stats(None, None, 1)
    _lib.lib().hibag_hip_train_combine_times(None, 1)
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    got = train.grow_concurrently(G, truth[:, 0], truth[:, 1], mdl.n_hla, max(ncl, k), mtry, True, k, t, 100, **kw)
    dt = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime) / dt
    sys_share = (r1.ru_stime - r0.ru_stime) / max(r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime, 1e-9)
    nl, no = (C.c_longlong * 8)(), (C.c_longlong * 8)()
    _lib.lib().hibag_hip_train_combine_stats(nl, no, 0)
    kinds = ((4, "pairs"), (2, "score"), (3, "em"))
    fused = " ".join(f"{n}:{(no[i] / nl[i] if nl[i] else 0):.1f}" for i, n in kinds)
    tm = (C.c_double * 12)()
    _lib.lib().hibag_hip_train_combine_times(tm, 0)
    if combine and k > 1:
        lat = " ".join(f"{n} {1e3 * tm[i] / max(no[i], 1):.2f}" for i, n in kinds)
        print(f"      latency per operation (ms): {lat}; batches: short {int(tm[10])} x {1e3 * tm[8] / max(tm[10], 1):.3f} ms, EM {int(tm[11])} x {1e3 * tm[9] / max(tm[11], 1):.3f} ms", flush=True)
    print(f"{k:3d} trainers x {t:2d} threads, EM {em:6s}, budget {budget}, {'combined' if combine and k > 1 else 'own streams'}: "
          f"{len(got) / dt:7.1f} classifiers/s  ({dt:.2f} s for {len(got)}; {cpu:.1f} cores busy, {100 * sys_share:.0f} % of it in the kernel; ops per fused launch {fused})", flush=True)
