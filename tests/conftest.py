import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFDATA = os.path.join(GOLDEN, "reference_data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_finish(session):
    """GPU runs: let torch initialise its HIP context before anything else does.  With the test files in another order
    than the alphabetical one (tests/test_hip_parity.py before tests/test_hip_configs.py) torch's lazy initialisation
    came after the library's and the oracle's worker threads and failed with "No HIP GPUs are available"."""
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hapmap_geno():
    from hibag_amd import model as M
    return M.load_geno(os.path.join(REFDATA, "HapMap_CEU_Geno.rdata"))


@pytest.fixture(scope="session")
def hla_type_table():
    from hibag_amd import rdata
    return rdata.load_rdata(os.path.join(REFDATA, "HLA_Type_Table.rdata"))["HLA_Type_Table"]


@pytest.fixture(scope="session")
def model_oob():
    """inst/extdata/OutOfBag.RData: HLA-A, 34 training samples, 100 classifiers, has `matching`."""
    from hibag_amd import model as M
    return M.load_model(os.path.join(REFDATA, "OutOfBag.RData"), "mobj")


@pytest.fixture(scope="session")
def model_a():
    """inst/extdata/ModelList.RData: modellist$A, 60 training samples, 100 classifiers."""
    from hibag_amd import model as M
    return M.load_model(os.path.join(REFDATA, "ModelList.RData"), "modellist", "A")


def align_geno(model, geno, samples=None):
    """int32 [n_samp, model.n_snp] genotypes for `samples` (default: the model's
    training samples), SNPs matched by rs id -- identity strand (the fixtures'
    allele strings agree)."""
    import numpy as np
    gi = {s: i for i, s in enumerate(geno.snp_id)}
    sel = [gi[s] for s in model.snp_id]
    assert [geno.snp_allele[i] for i in sel] == list(model.snp_allele)
    si = {s: i for i, s in enumerate(geno.sample_id)}
    ids = model.sample_id if samples is None else samples
    return np.ascontiguousarray(geno.sample_major(sel)[[si[s] for s in ids]])


def write_bed(path, geno, mode):
    """PLINK BED file from genotypes [n_snp, n_samp] (0/1/2, anything else = missing):
    mode 0 = individual-major, 1 = SNP-major.  Codes: 2 -> 0, missing -> 1, 1 -> 2, 0 -> 3."""
    import numpy as np
    g = np.asarray(geno)
    code = np.full(g.shape, 1, np.uint8)
    code[g == 2] = 0
    code[g == 1] = 2
    code[g == 0] = 3
    rows = code if mode != 0 else code.T                  # [row, col]
    n_col = rows.shape[1]
    pad = np.zeros((rows.shape[0], (n_col + 3) // 4 * 4), np.uint8)
    pad[:, :n_col] = rows
    q = pad.reshape(rows.shape[0], -1, 4)
    packed = (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 1 if mode != 0 else 0]))
        f.write(packed.tobytes())
    return path


def write_fam_bim(prefix, sample_id, snp_id, chrom, pos, alleles):
    with open(prefix + ".fam", "w") as f:
        for s in sample_id:
            f.write(f"F{s} {s} 0 0 1 -9\n")
    with open(prefix + ".bim", "w") as f:
        for i, c, p, a in zip(snp_id, chrom, pos, alleles):
            a1, a2 = a.split("/")
            f.write(f"{c}\t{i}\t0\t{int(p)}\t{a1}\t{a2}\n")
