"""The C ABI from C: include/hibag_hip.h must be valid C99 (CPU check: gcc -pedantic compiles a client
against it), and a compiled client must get the same bits as the Python binding and the oracle (GPU)."""

import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c")
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "hibag_amd", "csrc")


def test_header_is_plain_c99(tmp_path):
    obj = str(tmp_path / "abi_smoke.o")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", INC, "-c", SRC, "-o", obj])


@pytest.mark.gpu
def test_c_client_matches_python_binding_and_oracle(tmp_path, oracle):
    import hibag_amd as hb
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I", INC, SRC, "-o", exe, "-L", LIBDIR, "-lhibag_hip",
                           f"-Wl,-rpath,{LIBDIR}"])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert out[0] == "n_hla 3 n_snp 6 n_classifier 2 pair_evals 16"
    assert out[-1] == "error: Invalid 'vote_method'."
    model = hb.HlaAttrBagObj(n_samp=0, n_snp=6, hla_allele=["a", "b", "c"], classifiers=[
        hb.Classifier(snpidx=[0, 2, 5], freq=[0.4, 0.1, 0.3, 0.2], hla=[0, 0, 1, 2], haplo=["010", "110", "001", "111"]),
        hb.Classifier(snpidx=[1, 2, 3, 4], freq=[0.5, 0.25, 0.25], hla=[0, 1, 2], haplo=["1010", "0101", "1111"])])
    G = np.array([[0, 1, 2, 1, 0, 1], [2] * 6, [1, hb.NA_INTEGER, 1, 0, 1, 0], [0] * 6], np.int32)
    want = oracle.predict(oracle.flatten(model), G, vote_method=1)
    for i, line in enumerate(out[1:5]):
        f = line.split()
        assert int(f[0]) == want["h1"][i] and int(f[1]) == want["h2"][i]
        vals = np.array([float.fromhex(x) for x in f[2:]])
        ref = np.concatenate([[want["prob"][i], want["matching"][i]], want["dosage"][i], want["postprob"][i]])
        assert np.array_equal(vals, ref, equal_nan=True), (i, vals, ref)
