"""Re-run the classifier-shard leg of tests/test_hip_fuzz.py::test_entry_points_campaign for one seed and say what differs.

    python tools/fuzz_repro_shards.py SEED
"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import hibag_amd as hib
from test_hip_fuzz import _campaign_case
hib.hlaSetKernelTarget("hip")
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
model, G = _campaign_case(hib, rng, big=False)
n, S = G.shape
vote = int(rng.integers(1, 3))
m = hib.hlaModelFromObj(model)
ref = m.predict_raw(G, vote, want_dosage=True, want_prob=True)
# (consume the generator as the campaign does up to the shard leg)
n_extra = int(rng.integers(0, 12)); rng.random(S); rng.random(S); rng.permutation(S + n_extra); rng.integers(0, 3, size=(n, S + n_extra)); rng.integers(0, 9)
if seed % 3 == 0: rng.integers(0, 2)
if seed % 4 == 1: rng.integers(1, 4)
want1 = ref if vote == 1 else m.predict_raw(G, 1, want_dosage=True, want_prob=True)
k = int(rng.integers(2, min(len(model.classifiers), 6) + 1))
print(f"seed {seed}: {n} samples, {S} SNPs, {len(model.classifiers)} classifiers, {model.n_hla} alleles, vote {vote}, {k} shards")
for shards in sorted({k, 2, min(len(model.classifiers), 6)}):
    grp = hib.hibag.ShardGroup(m, [0] * shards)
    got = grp.predict_raw(G, want_dosage=True, want_prob=True)
    grp.close()
    calls = np.where((got["h1"] != want1["h1"]) | (got["h2"] != want1["h2"]))[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        fin = np.isfinite(want1["postprob"]) & (want1["postprob"] > 1e-200)
        rel = np.abs(got["postprob"] - want1["postprob"]) / np.where(fin, want1["postprob"], 1.0)
        rel = np.where(fin, rel, 0.0)
    nanp = np.where(np.isnan(got["postprob"]) != np.isnan(want1["postprob"]))
    print(f"  {shards} shards: calls differ at samples {calls.tolist()[:10]}; max rel {rel.max():.3e} at {np.unravel_index(rel.argmax(), rel.shape)}; NaN pattern differs at {len(nanp[0])} cells")
    for s in calls[:3]:
        c1 = np.argsort(want1["postprob"][s])[-3:][::-1]
        print(f"    sample {s}: one model calls ({want1['h1'][s]}, {want1['h2'][s]}) prob {want1['prob'][s]!r}; shards ({got['h1'][s]}, {got['h2'][s]}) prob {got['prob'][s]!r}; "
              f"top cells one model {[(int(c), float(want1['postprob'][s][c])) for c in c1]}; shards {[(int(c), float(got['postprob'][s][c])) for c in c1]}")
    if rel.max() >= 1e-10:
        s, c = np.unravel_index(rel.argmax(), rel.shape)
        print(f"    worst cell: sample {s} cell {c}: one model {want1['postprob'][s][c]!r}, shards {got['postprob'][s][c]!r}; sums of the rows: {np.nansum(want1['postprob'][s])!r} {np.nansum(got['postprob'][s])!r}")
