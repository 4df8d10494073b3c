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
