import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import hibag_amd
from hibag_amd import synth, train
hibag_amd.hlaSetKernelTarget("hip")
mdl, founders, af = synth.make_model("hla-b", seed=9, n_snp=300, n_classifier=1, wide_classifier=False)
G, truth = synth.make_samples(founders, af, 1000, seed=10)
mtry = int(np.ceil(np.sqrt(300)))
def run(threads, ncl):
    tr = train._Trainer(G, truth[:, 0], truth[:, 1], mdl.n_hla)
    if threads: tr.set_threads(threads)
    tr.set_seed(100)
    tr.new_classifiers(1, mtry, True, False, False)
    t = time.perf_counter()
    tr.new_classifiers(ncl, mtry, True, False, False)
    dt = (time.perf_counter() - t) / ncl
    cls = tr.classifiers()
    tr.close()
    return dt, cls
for thr in (0, 2):
    dt, cls = run(thr, 24)
    print("threads", thr or "default", "s/classifier", round(dt, 4), flush=True)
