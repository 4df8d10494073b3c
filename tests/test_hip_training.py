"""GPU parity of the training-side plugin entries (build_* of TypeGPUExtProc,
inst/include/LibHLA_ext.h:358-388), driven the way the reference's greedy SNP
search drives a plugin (src/LibHLA.cpp:1913-1979, :1002-1073):

* build_acc_oob against the `outofbag.acc` values the REFERENCE stored in its two
  bundled models (2 x 100 classifiers) -- known answers, not oracle outputs;
* build_acc_ib and build_haplomatch against the oracle's restatements of
  _PostProb / _PrepHaploMatch, bit for bit.
"""

import ctypes as C
import math

import numpy as np
import pytest

from conftest import align_geno
from test_hip_parity import _TGenotype, _THaplotype

pytestmark = pytest.mark.gpu


class _FullTable(C.Structure):           # inst/include/LibHLA_ext.h:358-388
    _fields_ = [
        ("build_init", C.CFUNCTYPE(None, C.c_int, C.c_int)),
        ("build_done", C.CFUNCTYPE(None)),
        ("build_set_bootstrap", C.CFUNCTYPE(None, C.POINTER(C.c_int))),
        ("build_haplomatch", C.CFUNCTYPE(C.POINTER(C.c_uint32), C.POINTER(_THaplotype), C.POINTER(C.c_size_t), C.c_int,
                                         C.POINTER(_TGenotype), C.POINTER(C.c_size_t))),
        ("build_set_haplo_geno", C.CFUNCTYPE(None, C.POINTER(_THaplotype), C.c_int, C.POINTER(_TGenotype), C.c_int)),
        ("build_acc_oob", C.CFUNCTYPE(C.c_int)),
        ("build_acc_ib", C.CFUNCTYPE(C.c_double)),
        ("predict_init", C.c_void_p), ("predict_done", C.c_void_p), ("predict_avg_prob", C.c_void_p)]


def _i64(v):
    v = int(v)
    return v - (1 << 64) if v >> 63 else v


def _truth(model, table):
    ti = {s: i for i, s in enumerate(table["sample.id"])}
    lut = {a: i for i, a in enumerate(model.hla_allele)}
    return ([lut[table["A.1"][ti[s]]] for s in model.sample_id], [lut[table["A.2"][ti[s]]] for s in model.sample_id])


@pytest.mark.parametrize("which", ["oob", "a"])
def test_training_entries(which, oracle, hapmap_geno, hla_type_table, model_oob, model_a):
    import hibag_amd
    from hibag_amd import _lib
    hibag_amd.hlaSetKernelTarget("hip")
    model = model_oob if which == "oob" else model_a
    tab = _FullTable.from_address(_lib.lib().hibag_hip_gpu_ext_proc())
    fm = oracle.flatten(model)
    G = align_geno(model, hapmap_geno)
    a1, a2 = _truth(model, hla_type_table)
    n, nh = model.n_samp, model.n_hla
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]

    tab.build_init(nh, n)
    for c, cl in enumerate(model.classifiers):
        k, lens, bits, freq, _ = fm.classifier(c)
        H = len(freq)
        hap = (_THaplotype * H)()
        hla = np.repeat(np.arange(nh), lens)
        for i in range(H):
            hap[i].packed[0] = _i64(bits[i, 0]); hap[i].packed[1] = -1      # garbage above n_snp like the reference
            hap[i].freq = freq[i]; hap[i].freq_f32 = freq[i]; hap[i].hla = int(hla[i])
        geno = (_TGenotype * n)()
        planes = []
        for s in range(n):
            s1, s2 = oracle.int_to_snp(G[s], cl.snpidx)
            planes.append((s1, s2))
            for w in range(2):
                geno[s].s1[w] = _i64(s1[w]); geno[s].s2[w] = _i64(s2[w])
            geno[s].boot = int(cl.samp_num[s])
            geno[s].a1, geno[s].a2 = (a2[s], a1[s]) if s % 2 else (a1[s], a2[s])   # either order must work
        boot = (C.c_int * n)(*[int(v) for v in cl.samp_num])
        tab.build_set_bootstrap(boot)
        tab.build_set_haplo_geno(hap, H, geno, k)

        oob = np.where(cl.samp_num == 0)[0]
        inbag = np.where(cl.samp_num > 0)[0]
        # known answer stored by the reference: 0.5 * correct / nOOB (src/LibHLA.cpp:2121)
        assert tab.build_acc_oob() == round(cl.outofbag_acc * 2 * len(oob)), f"classifier {c}"

        if c % 10 == 0:                    # the oracle loops below are Python-slow; every 10th classifier
            loglik = 0.0
            for s in inbag:
                loglik += int(cl.samp_num[s]) * math.log(oracle.post_prob(fm, c, planes[s][0], planes[s][1], a1[s], a2[s]))
            assert tab.build_acc_ib() == -2 * loglik, f"classifier {c}"

            n_per = (C.c_size_t * nh)(*[int(v) for v in lens])
            out_n = C.c_size_t(0)
            buf = tab.build_haplomatch(hap, n_per, k, geno, C.byref(out_n))
            cnt = buf[0] // 2
            assert out_n.value == 1 + 2 * cnt
            got = {}
            for q in range(cnt):
                kk, v = buf[1 + 2 * q], buf[2 + 2 * q]
                got.setdefault(kk, []).append((v & 0xFFFF, v >> 16))
            libc.free(buf)
            st = np.concatenate([[0], np.cumsum(lens)])
            for kk, s in enumerate(inbag):
                lo, hi = sorted((a1[s], a2[s]))
                want = oracle.prep_haplo_match(fm, c, planes[s][0], planes[s][1], lo, hi)
                want = [(int(i) - int(st[lo]), int(j) - int(st[hi])) for i, j in want]
                assert got.get(kk, []) == want, f"classifier {c}, in-bag sample {kk}"
                assert len(want) >= 1            # the host insists on a non-empty list (src/LibHLA.cpp:1066-1072)
    tab.build_done()
