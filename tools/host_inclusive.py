#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer entries on the benchmark workload
(DESIGN.md section 5): hibag_hip_predict (int32 matrix in host memory -> results in
host memory) and hibag_hip_predict_bed (PLINK BED file, page cache -> results in host
memory).  Never the benchmark's `value`, which starts with inputs in HBM."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import hibag_amd as hb                      # noqa: E402
from hibag_amd import synth                  # noqa: E402
from conftest import write_bed               # noqa: E402

hb.hlaSetKernelTarget("hip")
obj, founders, af = synth.make_model("hla-b")
n = 10000
G, _ = synth.make_samples(founders, af, n)
m = hb.hlaModelFromObj(obj)
out = {}
with tempfile.TemporaryDirectory() as d:
    bed = write_bed(os.path.join(d, "c.bed"), G.T, 1)
    col = np.arange(obj.n_snp)
    for name, fn in (("int32_matrix", lambda: m.predict_raw(G, 1, want_dosage=True)),
                     ("bed_file", lambda: m.predict_bed(bed, n, obj.n_snp, col, None, 1, want_dosage=True))):
        for _ in range(3):
            fn()
        t = time.perf_counter()
        reps = 20
        for _ in range(reps):
            fn()
        out[name + "_samples_per_s"] = n * reps / (time.perf_counter() - t)
print(json.dumps(out))
