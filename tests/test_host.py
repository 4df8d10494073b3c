"""CPU-only checks of the host logic: the C-ABI library loads and exports what
include/hibag_hip.h declares, model-building errors follow the reference, the R
workspace reader and the SNP matching / strand logic behave like the R code."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REFDATA, ROOT


def test_library_exports_every_declared_symbol():
    from hibag_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "hibag_hip.h")).read()
    declared = set(re.findall(r"\b(hibag_hip_[a-z_0-9]+)\s*\(", hdr))
    declared.discard("hibag_hip_model")      # the opaque struct tag
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.hibag_hip_abi_version() == 7


def test_no_cpu_fallback_in_product():
    """The product must not reach into oracle/ (checker only)."""
    for d, _, files in os.walk(os.path.join(ROOT, "hibag_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                src = open(os.path.join(d, f), errors="replace").read()
                assert "oracle" not in src.replace("the CPU oracle", ""), os.path.join(d, f)


def test_model_building_errors_follow_the_reference():
    from hibag_amd import _lib
    L = _lib.lib()
    err = lambda: L.hibag_hip_last_error().decode()
    assert not L.hibag_hip_model_new(0, 10)
    m = C.c_void_p(L.hibag_hip_model_new(3, 10))
    i32 = lambda v: np.ascontiguousarray(v, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    f = np.array([0.5, 0.5])
    strs = lambda *s: (C.c_char_p * len(s))(*[x.encode() for x in s])
    # characters other than 0/1 (src/LibHLA.cpp:333-334)
    assert L.hibag_hip_model_add_classifier(m, 2, p(i32([0, 1])), 2, p(f), p(i32([0, 1])), strs("01", "0x")) == -1
    assert "should be '0' or '1'" in err()
    # more than 128 SNPs (src/LibHLA.cpp:328-329, inst/include/LibHLA_ext.h:223)
    assert L.hibag_hip_model_add_classifier(m, 129, p(i32(np.zeros(129))), 0, None, None, None) == -1
    # alleles out of range / not grouped ascending (src/HIBAG.cpp:915-924 emits them grouped)
    assert L.hibag_hip_model_add_classifier(m, 2, p(i32([0, 1])), 2, p(f), p(i32([0, 3])), strs("01", "00")) == -1
    assert L.hibag_hip_model_add_classifier(m, 2, p(i32([0, 1])), 2, p(f), p(i32([1, 0])), strs("01", "00")) == -1
    assert L.hibag_hip_model_add_classifier(m, 2, p(i32([0, 10])), 2, p(f), p(i32([0, 1])), strs("01", "00")) == -1
    assert L.hibag_hip_model_add_classifier(m, 2, p(i32([0, 1])), 2, p(f), p(i32([0, 1])), strs("01", "00")) == 0
    assert L.hibag_hip_model_n_classifier(m) == 1 and L.hibag_hip_model_pair_evals(m) == 3
    # predicting before finalize is a state error, a bad vote_method carries the reference's text
    g = i32(np.zeros((1, 10)))
    assert L.hibag_hip_predict(m, p(g), 1, 1, None, None, None, None, None, None) == -4
    tab = np.zeros(257)
    assert L.hibag_hip_model_mutation_table(m, p(tab)) == 0 and tab[0] == 1 and tab[65] == 0
    L.hibag_hip_model_free(m)
    buf = C.create_string_buffer(64)
    assert L.hibag_hip_set_kernel_target(b"avx2", buf, 64) == -1


def test_mutation_table_matches_oracle(oracle):
    from hibag_amd import _lib
    L = _lib.lib()
    m = C.c_void_p(L.hibag_hip_model_new(2, 1))
    tab = np.zeros(257)
    L.hibag_hip_model_mutation_table(m, tab.ctypes.data_as(C.c_void_p))
    L.hibag_hip_model_free(m)
    assert np.array_equal(tab, oracle.mutation_table())


def test_rdata_reader_on_reference_fixtures(hapmap_geno, hla_type_table, model_a, model_oob):
    assert hapmap_geno.genotype.shape == (1564, 60) and hapmap_geno.assembly == "hg19"
    assert set(np.unique(hapmap_geno.genotype)) == {-2147483648, 0, 1, 2}
    assert len(hla_type_table["sample.id"]) == 60 and hla_type_table["B.2"][2] is None      # NA_character_
    for m, n in ((model_a, 60), (model_oob, 34)):
        assert (m.n_samp, m.n_snp, m.n_hla, len(m.classifiers)) == (n, 266, 14, 100)
        for c in m.classifiers:
            assert len(c.haplo[0]) == len(c.snpidx) and c.snpidx.min() >= 0 and c.snpidx.max() < 266
            assert np.all(np.diff(c.hla) >= 0) and abs(c.freq.sum() - 1) < 1e-9
    assert model_a.pair_evals_per_sample() == 91645
    assert model_oob.matching.shape == (34,) and model_a.matching is None


def test_strand_logic():
    from hibag_amd.snpmatch import allele_strand_flags
    tpl = ["A/G", "A/G", "A/G", "A/G", "C/G", "C/G", "A/G", "A/T", "I/D", "I/D"]
    tgt = ["A/G", "G/A", "T/C", "C/T", "C/G", "G/C", "A/C", "T/A", "I/D", "D/I"]
    f1 = [0.2] * 10
    f2 = [0.2, 0.8, 0.2, 0.8, 0.8, 0.2, 0.7, 0.3, 0.2, 0.8]
    flip, amb, mis, swap = allele_strand_flags(tpl, f1, tgt, f2, same_strand=False)
    #        same  swap  strand strand+swap  C/G amb(by freq)  mismatch(by freq)  A/T amb  indel same / swapped
    assert flip.tolist() == [False, True, False, True, True, False, True, False, False, True]
    assert (amb, mis, swap) == (3, 1, 2)
    flip2, amb2, mis2, swap2 = allele_strand_flags(tpl[:4], f1[:4], tgt[:4], f2[:4], same_strand=True)
    assert flip2.tolist() == [False, True, False, True] and (amb2, mis2, swap2) == (0, 2, 0)


def test_snp_matching_subset_and_flip(model_a, hapmap_geno):
    from hibag_amd.snpmatch import match_snps_for_predict
    from hibag_amd.model import HlaSNPGeno
    mat, asm = match_snps_for_predict(model_a, hapmap_geno, "Position", True, False, False, False)
    assert asm == "hg19" and mat.shape == (266, 60)
    gi = {s: i for i, s in enumerate(hapmap_geno.snp_id)}
    assert np.array_equal(mat, hapmap_geno.genotype[[gi[s] for s in model_a.snp_id]])
    # drop 100 model SNPs from the data and swap the alleles of 20 others
    keep = np.ones(1564, bool)
    keep[[gi[s] for s in model_a.snp_id[:100]]] = False
    sub = HlaSNPGeno(hapmap_geno.genotype[keep].copy(), hapmap_geno.sample_id,
                     [s for s, k in zip(hapmap_geno.snp_id, keep) if k], hapmap_geno.snp_position[keep],
                     [a for a, k in zip(hapmap_geno.snp_allele, keep) if k], "hg19")
    pos = {s: i for i, s in enumerate(sub.snp_id)}
    for s in model_a.snp_id[100:120]:
        j = pos[s]
        a, b = sub.snp_allele[j].split("/")
        sub.snp_allele[j] = f"{b}/{a}"
        row = sub.genotype[j]
        sub.genotype[j] = np.where(row == -2147483648, row, 2 - row)
    mat2, _ = match_snps_for_predict(model_a, sub, "RefSNP", True, False, False, False)
    assert np.all(mat2[:100] == -2147483648)
    unamb = [i for i in range(100, 266) if set(model_a.snp_allele[i].split("/")) not in ({"A", "T"}, {"C", "G"})]
    assert np.array_equal(mat2[unamb], mat[unamb])
    with pytest.raises(ValueError, match="no overlapping"):
        none = HlaSNPGeno(sub.genotype[:3], sub.sample_id, ["x1", "x2", "x3"], np.array([1., 2., 3.]), ["A/G"] * 3, "hg19")
        match_snps_for_predict(model_a, none, "RefSNP", True, False, False, False)


def test_no_gpu_means_loud_failure_not_fallback():
    """On a machine without an MI355X the product refuses to compute (there is no CPU path)."""
    import hibag_amd
    from hibag_amd import _lib, synth
    if _lib.lib().hibag_hip_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(hibag_amd.HibagHipError) as e:
        hibag_amd.hlaSetKernelTarget("hip")
    assert e.value.code == -2                                   # HIBAG_HIP_ENODEV
    model, _, _ = synth.make_model("hla-a-small", n_classifier=2)
    with pytest.raises(hibag_amd.HibagHipError):                # finalize needs the device
        hibag_amd.hlaModelFromObj(model)


def test_pred_merge_host_logic():
    """hlaPredMerge (R/HIBAG.R:825-1023): weighted, matching-scaled sum of posterior matrices over
    the union of alleles, renormalised per sample."""
    import numpy as np
    import hibag_amd as hb
    from hibag_amd.hibag import HlaAlleleClass, _pair_names
    rng = np.random.default_rng(1)

    def fake(alleles, n=5):
        names = _pair_names(alleles)
        p = rng.random((len(names), n))
        p /= p.sum(axis=0)
        return HlaAlleleClass(locus="A", sample_id=list(range(n)), allele1=[None] * n, allele2=[None] * n,
                              prob=p.max(axis=0), matching=rng.random(n), assembly="hg19", postprob=p, pair_names=names)

    a = fake(["01:01", "02:01", "03:01"])
    b = fake(["02:01", "03:01", "24:02"])
    same = hb.hlaPredMerge(a, verbose=False, ret_postprob=True)
    assert same.pair_names == a.pair_names and np.allclose(same.postprob, a.postprob, rtol=1e-15)
    assert [f"{y}/{x}" for x, y in zip(same.allele1, same.allele2)] == [a.pair_names[j] for j in a.postprob.argmax(axis=0)]
    m = hb.hlaPredMerge(a, b, weight=[3, 1], verbose=False, ret_postprob=True)
    assert m.pair_names == _pair_names(["01:01", "02:01", "03:01", "24:02"])
    assert np.allclose(m.postprob.sum(axis=0), 1) and np.allclose(m.dosage.sum(axis=0), 2)
    j = m.pair_names.index("03:01/02:01")
    want = 0.75 * a.matching * a.postprob[a.pair_names.index("03:01/02:01")] + 0.25 * b.matching * b.postprob[b.pair_names.index("03:01/02:01")]
    tot = 0.75 * a.matching * 1 + 0.25 * b.matching * 1
    assert np.allclose(m.postprob[j], want / tot) and np.allclose(m.matching, 0.75 * a.matching + 0.25 * b.matching)
    low = hb.hlaPredMerge(a, b, max_resolution="2-digit", verbose=False, ret_postprob=True)
    assert low.pair_names == _pair_names(["01", "02", "03", "24"]) and np.allclose(low.postprob.sum(axis=0), 1)
    assert hb.hlaAlleleDigit(["01:01:01G", "02:01N", None], "4-digit", rm_suffix=True) == ["01:01", "02:01", None]
    with pytest.raises(ValueError, match="sample IDs"):
        hb.hlaPredMerge(a, fake(["01:01"], n=4), verbose=False)


def test_pred_merge_known_answer():
    """hlaPredMerge on two tiny posteriors, expected values worked out by hand from the reference's source
    (R/HIBAG.R:825-1023: weights normalised to 3/4 and 1/4 (:897); merged alleles sorted (:946), pair rows
    outer(x, x, paste)[lower.tri] = "x_i/x_j", i >= j, column by column (:953-954); HIBAG_SumList, HIBAG_UpdateAddProbW
    (w2 = weight * matching[sample], out[row] += p * w2) and HIBAG_NormalizeProb (column sum in row order, then a
    division) in src/HIBAG.cpp:1463-1547; H1 = second name of the winning row, H2 = the first (:997-998); dosage =
    column sums over the rows naming the allele first plus those naming it second (:1012-1016)).
    Every input is a dyadic fraction, so each product and sum below is exact and each quotient is ONE rounded division:
    the comparison is `==`, not a tolerance.

        set 1 (weight 3): alleles 01:01, 02:01       rows 01:01/01:01, 02:01/01:01, 02:01/02:01
            sample 1   0.5    0.25   0.25     matching 0.5
            sample 2   0.125  0.75   0.125    matching 1.0
        set 2 (weight 1): alleles 02:01, 03:01       rows 02:01/02:01, 03:01/02:01, 03:01/03:01
            sample 1   0.25   0.5    0.25     matching 1.0
            sample 2   0.0    0.5    0.5      matching 0.25
        w2: set 1 (0.375, 0.75), set 2 (0.25, 0.0625)
        merged rows A/A, B/A, C/A, B/B, C/B, C/C  (A = 01:01, B = 02:01, C = 03:01), before the normalisation:
            sample 1   0.1875   0.09375  0  0.09375 + 0.0625 = 0.15625   0.125    0.0625     sum 0.625
            sample 2   0.09375  0.5625   0  0.09375 + 0      = 0.09375   0.03125  0.03125    sum 0.8125
    """
    import numpy as np
    import hibag_amd as hb
    from hibag_amd.hibag import HlaAlleleClass, _pair_names

    def pred(alleles, post, matching):
        p = np.array(post, np.float64).T                             # [rows][samples]
        return HlaAlleleClass(locus="A", sample_id=["s1", "s2"], allele1=[None, None], allele2=[None, None],
                              prob=p.max(axis=0), matching=np.array(matching), assembly="hg19", postprob=p,
                              pair_names=_pair_names(alleles))

    one = pred(["01:01", "02:01"], [[0.5, 0.25, 0.25], [0.125, 0.75, 0.125]], [0.5, 1.0])
    two = pred(["02:01", "03:01"], [[0.25, 0.5, 0.25], [0.0, 0.5, 0.5]], [1.0, 0.25])
    assert one.pair_names == ["01:01/01:01", "02:01/01:01", "02:01/02:01"]
    m = hb.hlaPredMerge(one, two, weight=[3, 1], verbose=False, ret_postprob=True)
    assert m.pair_names == ["01:01/01:01", "02:01/01:01", "03:01/01:01", "02:01/02:01", "03:01/02:01", "03:01/03:01"]
    s1 = [0.1875 / 0.625, 0.09375 / 0.625, 0.0, 0.15625 / 0.625, 0.125 / 0.625, 0.0625 / 0.625]
    s2 = [0.09375 / 0.8125, 0.5625 / 0.8125, 0.0, 0.09375 / 0.8125, 0.03125 / 0.8125, 0.03125 / 0.8125]
    assert s1 == [0.3, 0.15, 0.0, 0.25, 0.2, 0.1]                     # (correctly rounded quotients of exact operands)
    assert m.postprob[:, 0].tolist() == s1 and m.postprob[:, 1].tolist() == s2
    assert m.matching.tolist() == [0.625, 0.8125]                     # 0.75 * 0.5 + 0.25 * 1.0,  0.75 * 1.0 + 0.25 * 0.25
    assert m.prob.tolist() == [0.3, 0.5625 / 0.8125]
    assert (m.allele1, m.allele2) == (["01:01", "01:01"], ["01:01", "02:01"])      # rows "01:01/01:01" and "02:01/01:01"
    # dosage: rows naming the allele first + rows naming it second (the homozygous row counts twice)
    want = np.array([[s1[0] + (s1[0] + s1[1] + s1[2]), s2[0] + (s2[0] + s2[1] + s2[2])],
                     [(s1[1] + s1[3]) + (s1[3] + s1[4]), (s2[1] + s2[3]) + (s2[3] + s2[4])],
                     [(s1[2] + s1[4] + s1[5]) + s1[5], (s2[2] + s2[4] + s2[5]) + s2[5]]])
    assert np.allclose(m.dosage, want, rtol=4e-16, atol=0) and np.allclose(m.dosage.sum(axis=0), 2.0, rtol=1e-15)
    assert np.allclose(m.dosage[:, 0], [0.75, 0.85, 0.4], rtol=1e-15)
    # without the matching scale the weights alone count: sample 1's unnormalised sums are 0.375 0.1875 0 0.25 0.125 0.0625 = 1
    u = hb.hlaPredMerge(one, two, weight=[3, 1], use_matching=False, verbose=False, ret_postprob=True)
    assert u.postprob[:, 0].tolist() == [0.375, 0.1875, 0.0, 0.25, 0.125, 0.0625]
    assert u.postprob[:, 1].tolist() == [0.09375, 0.5625, 0.0, 0.09375, 0.125, 0.125]


def test_rdata_writer_round_trip(tmp_path, model_oob):
    """save_model writes the workspace hlaModelToObj() + save() would; the reader gets the same model back,
    and re-serialising the reference's own fixture reproduces its decoded structure."""
    import numpy as np
    from hibag_amd import model as M, rdata
    p = str(tmp_path / "m.RData")
    M.save_model(p, model_oob, "mobj")
    raw = __import__("gzip").open(p, "rb").read()
    assert raw[:7] == b"RDX2\nX\n" and raw[7:19] == bytes([0, 0, 0, 2, 0, 3, 5, 0, 0, 2, 3, 0])
    back = M.load_model(p, "mobj")
    assert back.hla_allele == model_oob.hla_allele and back.snp_id == model_oob.snp_id and back.sample_id == model_oob.sample_id
    assert np.array_equal(back.snp_position, model_oob.snp_position) and back.assembly == model_oob.assembly
    assert np.array_equal(back.matching, model_oob.matching) and np.array_equal(back.hla_freq, model_oob.hla_freq)
    for a, b in zip(back.classifiers, model_oob.classifiers):
        assert np.array_equal(a.snpidx, b.snpidx) and np.array_equal(a.freq, b.freq) and a.haplo == b.haplo
        assert np.array_equal(a.hla, b.hla) and np.array_equal(a.samp_num, b.samp_num) and a.outofbag_acc == b.outofbag_acc
    # generic objects: what the reader decoded from the reference's file survives a write + read
    import os
    from conftest import REFDATA
    ws = rdata.load_rdata(os.path.join(REFDATA, "HLA_Type_Table.rdata"))
    q = str(tmp_path / "t.rda")
    rdata.save_rdata(q, ws)
    ws2 = rdata.load_rdata(q)
    t1, t2 = ws["HLA_Type_Table"], ws2["HLA_Type_Table"]
    assert t1.names == t2.names and t1.attrs["class"] == t2.attrs["class"]
    def plain(v):
        return list(rdata.factor_to_strings(v)) if "levels" in getattr(v, "attrs", {}) else list(v)
    for k in t1.names:
        assert plain(t1[k]) == plain(t2[k])
    rdata.save_rds(str(tmp_path / "x.rds"), rdata.RArray(np.array([1.5, np.nan]), {"names": ["a", "b"]}))
    x = rdata.load_rds(str(tmp_path / "x.rds"))
    assert x[0] == 1.5 and np.isnan(x[1]) and x.attrs["names"] == ["a", "b"]


def test_rdata_writer_reproduces_r_bytes(tmp_path, model_oob):
    """Known answer for the writer: re-serialising what the reader decoded from the reference's own
    files gives back R's bytes exactly (all four fixtures, apart from the 4-byte "written by R x.y.z"
    stamp), and the exported form of a model equals R's hlaModelToObj() + save() output."""
    import os
    from conftest import REFDATA
    from hibag_amd import model as M, rdata

    def body(path):
        b = bytearray(rdata._decompress(open(path, "rb").read()))
        b[11:15] = b"\0\0\0\0"
        return bytes(b)

    for fn in ("OutOfBag.RData", "ModelList.RData", "HLA_Type_Table.rdata", "HapMap_CEU_Geno.rdata"):
        src = os.path.join(REFDATA, fn)
        out = str(tmp_path / fn)
        rdata.save_rdata(out, rdata.load_rdata(src), compress=False)
        assert body(out) == body(src), fn
    # the file was written by a release that left the last element (appendix) unnamed
    out = str(tmp_path / "export.RData")
    M.save_model(out, model_oob, "mobj")
    want = body(os.path.join(REFDATA, "OutOfBag.RData"))
    got = body(out).replace(b"\x00\x04\x00\x09\x00\x00\x00\x08appendix", b"\x00\x04\x00\x09\x00\x00\x00\x00")
    assert got == want



def test_multi_device_slices_cover_the_cohort_contiguously():
    """hibag_hip_multi_slice: the contiguous sample slices hibag_hip_predict_multi hands its replicas (the reference cuts the
    cohort the same way for its cluster workers, R/HIBAG.R:767-781) -- disjoint, in order, complete, boundaries on
    multiples of 64 samples except the cohort's end.  Pure host arithmetic: runs without a GPU."""
    from hibag_amd.hibag import multi_slice
    import hibag_amd
    for n in (0, 1, 63, 64, 65, 1000, 10_000, 100_000, 123_457):
        for k in (1, 2, 3, 4, 8):
            at = 0
            for i in range(k):
                first, count = multi_slice(n, k, i)
                assert first == at and count >= 0
                assert first % 64 == 0
                at += count
            assert at == n
            sizes = [multi_slice(n, k, i)[1] for i in range(k)]
            assert max(sizes) - min(sizes) <= 64 or n < 64 * k
    with pytest.raises(hibag_amd.HibagHipError):
        multi_slice(10, 0, 0)
    with pytest.raises(hibag_amd.HibagHipError):
        multi_slice(10, 2, 2)


def test_status_entries_reject_bad_arguments_without_a_gpu():
    from hibag_amd import _lib
    L = _lib.lib()
    assert L.hibag_hip_model_status(None) == -1
    assert L.hibag_hip_model_clear_status(None) == -1
    assert L.hibag_hip_model_handover_faults(None) == 0
    assert L.hibag_hip_test_inject_handover_fault(None, 1) == -1
    assert L.hibag_hip_predict_multi(None, 0, None, 0, 1, None, None, None, None, None, None) == -1
    m = L.hibag_hip_model_new(4, 10)
    assert m
    m = C.c_void_p(m)
    assert L.hibag_hip_test_inject_handover_fault(m, 3) == -1
    e, k = C.c_int(0), C.c_int(0)
    assert L.hibag_hip_model_engine(m, 0, C.byref(e), C.byref(k)) == -4          # not finalized
    assert L.hibag_hip_model_status(m) in (0, -2)                                # no launches; -2 where there is no device
    L.hibag_hip_model_free(m)


def test_classifier_shard_bounds_cover_the_model_in_order():
    """hibag_hip_shard_bounds (host logic of the RCCL-merged route): contiguous, ordered, sizes within one of each other --
    the same split hibag_amd.dist.shard_bounds makes for the one-process-per-GPU route."""
    import ctypes as C
    from hibag_amd import _lib
    from hibag_amd.dist import shard_bounds
    L = _lib.lib()
    for n, w in ((100, 8), (7, 3), (3, 8), (0, 2), (1, 1)):
        at = 0
        for r in range(w):
            f, c = C.c_int(-1), C.c_int(-1)
            assert L.hibag_hip_shard_bounds(n, w, r, C.byref(f), C.byref(c)) == 0
            assert (f.value, f.value + c.value) == shard_bounds(n, w, r)
            assert f.value == at
            at += c.value
        assert at == n
    assert L.hibag_hip_shard_bounds(10, 0, 0, None, None) != 0 and L.hibag_hip_shard_bounds(10, 2, 2, None, None) != 0
    assert L.hibag_hip_shard_group_new(None, 0) is None          # (no GPU needed to be refused)


def test_hlaPredict_host_side_routes_by_memory_order_and_copies_nothing():
    """The host side of hlaPredict (R/HIBAG.R:481-818) without a device: a stub model records which entry of the C ABI would
    be called and with what.  int32 genotypes reach it as VIEWS of the caller's array in either memory order (R's column-major:
    the sample-major entry; numpy's row-major: hibag_hip_predict_snp_major), doubles are converted once (NaN -> NA), the SNP
    selection / flips are handed on instead of being applied on the host, dosage / postprob come back as [row, sample] views,
    allele names are made when first read, NA calls are counted and warned about."""
    import warnings
    import hibag_amd
    from hibag_amd import hibag, synth
    model, founders, af = synth.make_model("hla-a-small", seed=5)
    G, _ = synth.make_samples(founders, af, 37, seed=6)
    calls = []

    class Stub(hibag.HlaAttrBagClass):
        def __init__(self, obj):
            self.obj, self._h = obj, None

        def _out(self, n, want_dosage, want_prob):
            o = self._outputs(n, want_dosage, want_prob)
            o["h1"][:] = 1; o["h2"][:] = 2; o["h1"][0] = o["h2"][0] = hibag.NA_INTEGER
            o["prob"][:] = 0.5; o["matching"][:] = 0.25
            for k in ("dosage", "postprob"):
                if k in o:
                    o[k][:] = np.arange(o[k].size, dtype=np.float64).reshape(o[k].shape)
            return o

        def predict_raw(self, g, vote_method=1, want_dosage=True, want_prob=False):
            calls.append(("raw", g, None, None)); return self._out(g.shape[0], want_dosage, want_prob)

        def predict_mapped(self, g, col, flip, vote_method=1, want_dosage=True, want_prob=False):
            calls.append(("mapped", g, col, flip)); return self._out(g.shape[0], want_dosage, want_prob)

        def predict_snp_major(self, g, col=None, flip=None, vote_method=1, want_dosage=True, want_prob=False):
            calls.append(("snp_major", g, col, flip)); return self._out(g.shape[1], want_dosage, want_prob)

    m = Stub(model)
    for order, entry in (("F", "raw"), ("C", "snp_major")):
        snp = synth.as_snp_geno(model, G, order=order)
        calls.clear()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            res = hibag_amd.hlaPredict(m, snp, type="response+prob", verbose=False)
        assert [c[0] for c in calls] == [entry]
        assert np.shares_memory(calls[0][1], snp.genotype)                  # the caller's own array went down, not a copy
        assert any("No prediction output for 1 individual" in str(x.message) for x in w)
        assert res.dosage.shape == (model.n_hla, 37) and res.postprob.shape == (model.n_cell, 37)
        assert res.dosage.base is not None and res.dosage.flags.f_contiguous          # views of the sample-major outputs
        assert res._allele1 is None                                         # not made yet ...
        assert res.allele1[:3] == [None, model.hla_allele[1], model.hla_allele[1]] and res.allele2[1] == model.hla_allele[2]
        assert res.pair_names[1] == f"{model.hla_allele[1]}/{model.hla_allele[0]}" and len(res.pair_names) == model.n_cell
    # doubles with NaN: one conversion, NaN -> NA_integer_, layout kept
    D = np.ascontiguousarray(np.where(G.T == hibag.NA_INTEGER, np.nan, G.T.astype(np.float64)))      # row-major [SNP, sample]
    calls.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hibag_amd.hlaPredict(m, D, type="response", verbose=False)
    assert calls[0][0] == "snp_major" and calls[0][1].dtype == np.int32
    assert np.array_equal(calls[0][1], np.where(np.isnan(D), hibag.NA_INTEGER, D).astype(np.int32))
    # a cohort whose SNPs differ from the model's: the selection and the flips are passed on, not applied on the host
    keep = np.ones(model.n_snp, bool); keep[[3, 11]] = False
    alle = ["G/A" if k % 5 == 0 else "A/G" for k in range(model.n_snp)]
    sub = hibag_amd.HlaSNPGeno(genotype=np.ascontiguousarray(G.T[keep]), sample_id=[f"s{i}" for i in range(37)],
                               snp_id=[i for i, k in zip(model.snp_id, keep) if k], snp_position=np.asarray(model.snp_position)[keep],
                               snp_allele=[a for a, k in zip(alle, keep) if k], assembly="hg19")
    calls.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hibag_amd.hlaPredict(m, sub, type="response", verbose=False)
    kind, g, col, flip = calls[0]
    assert kind == "snp_major" and np.shares_memory(g, sub.genotype)
    assert list(np.where(np.asarray(col) < 0)[0]) == [3, 11] and np.asarray(col)[4] == 3
    assert [bool(f) for f in flip] == [k % 5 == 0 and k not in (3, 11) for k in range(model.n_snp)]
    with pytest.raises(TypeError):
        hibag_amd.hlaPredict(m, np.array([["a"] * 3] * model.n_snp), verbose=False)


def test_synth_bed_writer_matches_the_test_writer(tmp_path):
    """``synth.write_bed`` (bench.py's PLINK leg) writes the same bytes as the tests' own writer, which the reference's
    BED <-> .rdata fixture pair pins (tests/test_hip_bed.py)."""
    from conftest import write_bed
    from hibag_amd import synth
    rng = np.random.default_rng(5)
    for n, s in ((1, 1), (7, 3), (64, 10), (101, 5)):
        g = rng.integers(-1, 4, (n, s)).astype(np.int32)
        a = synth.write_bed(str(tmp_path / "a.bed"), g)
        b = write_bed(str(tmp_path / "b.bed"), g.T, 1)
        assert open(a, "rb").read() == open(b, "rb").read()
