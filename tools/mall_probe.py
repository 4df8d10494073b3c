"""Could pass 2 gain from finding pass 1's stored cell sums in a cache?  (DESIGN.md section 8, round 5's "next" item 2.)

The sums of n samples take 115.7 KB each (14,460 sums x 8 B): 10,000 samples = 1.16 GB, far beyond the 256 MB Infinity
Cache; 2,048 samples = 237 MB, which fits.  Each batch size is run as it is and with HIBAG_DEBUG_THRASH_MB (1 GB of scratch
overwritten between the passes: nothing of pass 1's output survives in L2 or the Infinity Cache), one child process per
setting (the variable is read once per process); pass 2's time comes from HIP events around it.

    python tools/mall_probe.py            # parent: runs the children, prints the table
"""
import json, os, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)


def child(n, reps):
    import numpy as np
    import torch
    import hibag_amd
    from hibag_amd import synth
    hibag_amd.hlaSetKernelTarget("hip")
    obj, founders, af = synth.make_model("hla-b")
    G, _ = synth.make_samples(founders, af, n, seed=synth.DEFAULT_SEED + 1)
    m = hibag_amd.hlaModelFromObj(obj)
    dev = torch.device("cuda", 0)
    dg = torch.from_numpy(G).to(dev)
    o = [torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
         torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
         torch.empty((n, obj.n_hla), dtype=torch.float64, device=dev)]
    st = torch.cuda.current_stream(dev).cuda_stream
    run = lambda: m.predict_device(dg.data_ptr(), n, 1, *[x.data_ptr() for x in o], None, stream=st)
    for _ in range(5):
        run()
    torch.cuda.synchronize(dev)
    m.set_timing(True); m.reset_timing()
    for _ in range(reps):
        run()
    torch.cuda.synchronize(dev)
    tm = m.get_timing()
    print(json.dumps({"n": n, "thrash_mb": int(os.environ.get("HIBAG_DEBUG_THRASH_MB", "0")),
                      "k_total_ms": tm["total"][0] / tm["total"][1], "k_accum_ms": tm["accum"][0] / tm["accum"][1],
                      "stored_sums_mb": m.stored_cells() * 8 * ((n + 63) // 64 * 64) / 1e6}))


if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(int(sys.argv[2]), int(sys.argv[3]))
else:
    for n in (1024, 2048, 4096, 10000):
        for rep in range(2):
            for mb in (0, 1024):
                env = dict(os.environ, HIBAG_DEBUG_THRASH_MB=str(mb))
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n), "30"], env=env, capture_output=True, text=True)
                line = [l for l in out.stdout.splitlines() if l.startswith("{")]
                print(line[-1] if line else out.stderr[-300:], flush=True)
