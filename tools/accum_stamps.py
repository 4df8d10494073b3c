#!/usr/bin/env python3
"""Where a wavefront of pass 2 (k_accum) spends its time, block phase by block phase.

Needs a library built with -DHIBAG_ACCUM_STAMPS (tools/build_variant.sh stamps -DHIBAG_ACCUM_STAMPS, then
HIBAG_HIP_LIBRARY=$PWD/gpurun_var_stamps.so python tools/accum_stamps.py): the kernel reads the clock at the phase
boundaries of every block and sums the differences (hibag_hip_test_read_diag).  The stamps drain lgkmcnt, so the step is
slower than the shipped one; the split says where the time is."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import hibag_amd
from hibag_amd import synth, _lib

PHASES = ["top of the block (this block's loads waited for, the next headers requested)", "stored sums added",
          "matrix instructions issued", "next block requested", "lane swaps (matrix results waited for)", "pairs added up", "-", "-"]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
hibag_amd.hlaSetKernelTarget("hip")
model_obj, founders, afreq = synth.make_model("hla-b")
geno, _ = synth.make_samples(founders, afreq, n, seed=synth.DEFAULT_SEED + 1)
dev = torch.device("cuda", 0)
model = hibag_amd.hlaModelFromObj(model_obj, device=0)
d_geno = torch.from_numpy(geno).to(dev)
d_h1 = torch.empty(n, dtype=torch.int32, device=dev); d_h2 = torch.empty(n, dtype=torch.int32, device=dev)
d_prob = torch.empty(n, dtype=torch.float64, device=dev); d_match = torch.empty(n, dtype=torch.float64, device=dev)
d_dos = torch.empty((n, model_obj.n_hla), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev)


def step():
    model.predict_device(d_geno.data_ptr(), n, 1, d_h1.data_ptr(), d_h2.data_ptr(), d_prob.data_ptr(), d_match.data_ptr(),
                         d_dos.data_ptr(), None, stream=st.cuda_stream)


for _ in range(3):
    step()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
_lib.check(_lib.lib().hibag_hip_test_read_diag(model.handle, buf, 16))      # zero the sums
steps = 10
model.set_timing(True); model.reset_timing()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tm = model.get_timing()
_lib.check(_lib.lib().hibag_hip_test_read_diag(model.handle, buf, 16))
v = [int(x) for x in buf]
blocks = v[8]
tot = sum(v[:8])
print(f"n={n}  step {dt / steps * 1e3:.3f} ms   kernels ms/step: " + ", ".join(f"{k} {x[0] / steps:.3f}" for k, x in tm.items()))
if blocks == 0:
    print("no stamps: the library was not built with -DHIBAG_ACCUM_STAMPS")
    sys.exit(0)
print(f"wavefront-blocks {blocks / steps:.0f} per step; clock ticks per wavefront-block {tot / blocks:.1f} (s_memtime ticks)")
for i in range(6):
    print(f"  phase {i}  {v[i] / blocks:8.1f} ticks/block  {100.0 * v[i] / max(tot, 1):5.1f} %   {PHASES[i]}")
