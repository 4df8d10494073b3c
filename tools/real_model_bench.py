#!/usr/bin/env python3
"""other_configs.cfg1_real_model of bench.py by itself: the reference's bundled HLA-A model on 10,000 samples resampled from its 60
HapMap genotypes (HIBAG_TILE_CAP=<cells per tile> to see what the tile size does to pass 2 of a model of few alleles)."""
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import hibag_amd

hibag_amd.hlaSetKernelTarget("hip")
dev = torch.device("cuda", 0)
K = bench.issue_constants()
r = bench.real_model_config(K, dev)
print(json.dumps({k: r[k] for k in ("samples_per_s", "ms_per_step", "kernels_ms_per_step", "pass2", "issue", "oracle_check")}))
