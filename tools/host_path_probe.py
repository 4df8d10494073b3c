#!/usr/bin/env python3
"""Where the host-pointer entry's time goes on the benchmark batch: hibag_hip_predict with fresh / reused output arrays
against the device-resident entry.  python tools/host_path_probe.py [n_samples]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hibag_amd as hb
from hibag_amd import synth, _lib
import torch
hb.hlaSetKernelTarget("hip")
obj, founders, af = synth.make_model("hla-b")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
G, _ = synth.make_samples(founders, af, n)
m = hb.hlaModelFromObj(obj)
L = _lib.lib()
def med(fn, reps=15):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3
h1 = np.zeros(n, np.int32); h2 = np.zeros(n, np.int32); pr = np.zeros(n); mt = np.zeros(n); ds = np.zeros((n, obj.n_hla))
p = lambda a: a.ctypes.data_as(C.c_void_p)
def reused():
    _lib.check(L.hibag_hip_predict(m.handle, p(G), n, 1, p(h1), p(h2), p(pr), p(mt), p(ds), None))
dev = torch.device("cuda", 0)
dg = torch.from_numpy(G).to(dev)
d = [torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
     torch.empty(n, dtype=torch.float64, device=dev), torch.empty((n, obj.n_hla), dtype=torch.float64, device=dev)]
st = torch.cuda.current_stream(dev).cuda_stream
def resident():
    m.predict_device(dg.data_ptr(), n, 1, *[x.data_ptr() for x in d], None, stream=st); torch.cuda.synchronize(dev)
sub = int(os.environ.get("PROBE_SUB", "0"))
if sub:
    def resident_sub():
        for a in range(0, n, sub):
            k = min(sub, n - a)
            m.predict_device(dg[a:].data_ptr(), k, 1, d[0][a:].data_ptr(), d[1][a:].data_ptr(), d[2][a:].data_ptr(), d[3][a:].data_ptr(),
                             d[4][a:].data_ptr(), None, stream=st)
        torch.cuda.synchronize(dev)
    print(f"device-resident in sub-batches of {sub}: {med(resident_sub):.3f} ms")
print(f"n={n}  fresh outputs {med(lambda: m.predict_raw(G, 1, want_dosage=True)):.3f} ms   reused outputs {med(reused):.3f} ms   device-resident {med(resident):.3f} ms"
      f"   (HIBAG_STAGED_NULL={os.environ.get('HIBAG_STAGED_NULL')})")
