#!/usr/bin/env python3
"""One 100,000-sample hibag_hip_predict call (host pointers, sliced pipeline) for a rocprofv3 timeline."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hibag_amd as hb
from hibag_amd import synth, _lib
hb.hlaSetKernelTarget("hip")
obj, founders, af = synth.make_model("hla-b")
n = 100000
G, _ = synth.make_samples(founders, af, n)
m = hb.hlaModelFromObj(obj)
L = _lib.lib()
h1 = np.zeros(n, np.int32); h2 = np.zeros(n, np.int32); pr = np.zeros(n); mt = np.zeros(n); ds = np.zeros((n, obj.n_hla))
p = lambda a: a.ctypes.data_as(C.c_void_p)
for rep in range(3):
    t = time.perf_counter()
    _lib.check(L.hibag_hip_predict(m.handle, p(G), n, 1, p(h1), p(h2), p(pr), p(mt), p(ds), None))
    print("call", rep, (time.perf_counter() - t) * 1e3, "ms", flush=True)
