#!/usr/bin/env python3
"""hlaAttrBagging() training throughput (BASELINE.json config 5 shape: 1,000 samples x 300
SNPs): the library's device-scored driver next to the oracle's CPU restatement (one core),
on the same synthetic cohort and random stream; also checks that both grow the same
classifiers.  Usage: python tools/train_bench.py [n_classifier] [n_samp] [n_snp]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import hibag_amd as hb                      # noqa: E402
from hibag_amd import synth, train           # noqa: E402
from oracle import oracle                    # noqa: E402

n_cls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_samp = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_snp = int(sys.argv[3]) if len(sys.argv) > 3 else 300
hb.hlaSetKernelTarget("hip")
model, founders, af = synth.make_model("hla-b", seed=9, n_snp=n_snp, n_classifier=1, wide_classifier=False)
G, truth = synth.make_samples(founders, af, n_samp, seed=10)
mtry = int(np.ceil(np.sqrt(n_snp)))

tr = train._Trainer(G, truth[:, 0], truth[:, 1], model.n_hla)
tr.set_seed(100)
t = time.perf_counter()
tr.new_classifiers(n_cls, mtry, True, False, False)
t_gpu = time.perf_counter() - t
got = tr.classifiers()
tr.close()

t = time.perf_counter()
want = oracle.train(G, truth[:, 0], truth[:, 1], model.n_hla, n_cls, mtry, True, 100)
t_cpu = time.perf_counter() - t
same = all(np.array_equal(a.snpidx, b["snpidx"]) and np.array_equal(a.freq, b["freq"]) and a.haplo == b["haplo"]
           for a, b in zip(got, want))
print(json.dumps({"n_classifier": n_cls, "n_samp": n_samp, "n_snp": n_snp, "n_hla": model.n_hla, "mtry": mtry,
                  "gpu_driver_s_per_classifier": t_gpu / n_cls, "cpu_oracle_1core_s_per_classifier": t_cpu / n_cls,
                  "identical_classifiers": bool(same),
                  "mean_snps": float(np.mean([len(c.snpidx) for c in got])),
                  "mean_haplo": float(np.mean([len(c.freq) for c in got]))}))
