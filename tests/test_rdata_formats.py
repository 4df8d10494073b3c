""""pre-fit public models load unchanged" beyond the reference's own RDX2 fixtures --
serialisation version 3 (RDX3, R >= 3.5), ALTREP items (compact integer sequences, wrapped vectors,
deferred strings), bzip2 / xz / uncompressed containers, models anonymised by hlaPublish()
(R/DataUtilities.R:2010-2017: no sample.id, no samp.num -> hlaModelFromObj substitutes 1s,
R/HIBAG.R:1151-1154) and models with a non-empty `appendix` (R/HIBAG.R:1050-1060).

There is no R here and the reference ships version-2 files only, so the version-3 inputs are (a) byte
strings assembled in this file straight from the layout of R's serialize.c / altclasses.c and (b) the
reference's fixtures re-written by hibag_amd.rdata's version-3 writer."""

import os
import struct

import numpy as np
import pytest

from conftest import REFDATA


def _i(*v):
    return b"".join(struct.pack(">i", x) for x in v)


def _chr(s):
    b = s.encode()
    return _i(9 | (64 << 12), len(b)) + b


def _sym(s):
    return _i(1) + _chr(s)


def _altrep_info(cls_item, pkg_item, rtype):
    # ALTREP_SERIALIZED_CLASS: pairlist (class symbol, package symbol, type)
    return _i(2) + cls_item + _i(2) + pkg_item + _i(2) + _i(13, 1, rtype) + _i(254)


def _rds3(body):
    enc = b"UTF-8"
    return b"X\n" + _i(3, 0x00040201, 0x00030500, len(enc)) + enc + body


def test_handmade_version3_stream(tmp_path):
    """list(a = 1:5, b = <wrap_real of c(1.5, 2.5) with names>, c = as.character(c(7L, 9L, 8L)), d = 10:8) laid out
    by hand: ALTREP = flags 238, info pairlist, state, attributes (serialize.c, WriteItem)."""
    from hibag_amd import rdata
    real3 = lambda a, b, c: _i(14, 3) + struct.pack(">3d", a, b, c)
    a = _i(238) + _altrep_info(_sym("compact_intseq"), _sym("base"), 13) + real3(5, 1, 1) + _i(254)
    # symbols seen so far: 1 compact_intseq, 2 base  -> REFSXP = (index << 8) | 255
    names_attr = _i(2 | 0x400) + _sym("names") + _i(16, 2) + _chr("x") + _chr("y") + _i(254)      # symbol 4 = names
    b = _i(238) + _altrep_info(_sym("wrap_real"), _i((2 << 8) | 255), 14) + \
        _i(2) + _i(14, 2) + struct.pack(">2d", 1.5, 2.5) + _i(13, 2, 1, 1) + names_attr           # CONS(x, meta): dotted pair
    c = _i(238) + _altrep_info(_sym("deferred_string"), _i((2 << 8) | 255), 16) + \
        _i(2) + _i(13, 3, 7, 9, 8) + _i(13, 1, 0) + _i(254)
    d = _i(238) + _altrep_info(_i((1 << 8) | 255), _i((2 << 8) | 255), 13) + real3(3, 10, -1) + _i(254)
    top_names = _i(2 | 0x400) + _i((4 << 8) | 255) + _i(16, 4) + _chr("a") + _chr("b") + _chr("c") + _chr("d") + _i(254)
    body = _i(19 | 0x200, 4) + a + b + c + d + top_names
    p = tmp_path / "x.rds"
    p.write_bytes(_rds3(body))
    x = rdata.load_rds(str(p))
    assert x.names == ["a", "b", "c", "d"]
    assert np.array_equal(x["a"], [1, 2, 3, 4, 5]) and x["a"].dtype == np.int32
    assert np.array_equal(x["b"], [1.5, 2.5]) and x["b"].attrs["names"] == ["x", "y"]
    assert list(x["c"]) == ["7", "9", "8"]
    assert np.array_equal(x["d"], [10, 9, 8])
    # the same object through the version-3 writer gives the same bytes
    obj = rdata.RList([rdata.RArray(np.arange(1, 6, dtype=np.int32)),
                       rdata.Wrapped(rdata.RArray(np.array([1.5, 2.5]), {"names": ["x", "y"]})),
                       rdata.DeferredString(np.array([7, 9, 8], np.int32)),
                       rdata.RArray(np.array([10, 9, 8], np.int32))], {"names": ["a", "b", "c", "d"]})
    q = tmp_path / "y.rds"
    rdata.save_rds(str(q), obj, compress=False, version=3, altrep=True)
    assert q.read_bytes() == p.read_bytes()


@pytest.mark.parametrize("compress,magic", [("bzip2", b"BZh"), ("xz", b"\xfd7zXZ\x00"), ("gzip", b"\x1f\x8b"), (False, b"RDX3\n")])
def test_reference_models_through_version3(tmp_path, model_a, model_oob, compress, magic):
    """The reference's two fixtures re-written as RDX3 (integer sequences as ALTREP, every container) load
    back to the same model; classifier SNP indices such as 1:k and sorted positions take the ALTREP route."""
    from hibag_amd import model as M, rdata
    for name, src in (("A", model_a), ("oob", model_oob)):
        robj = M.model_to_robj(src)
        names = robj.names
        robj[names.index("snp.position")] = rdata.Wrapped(robj["snp.position"])          # sort()ed positions
        robj[names.index("sample.id")] = rdata.Wrapped(robj["sample.id"], is_sorted=0, no_na=1)
        robj[names.index("appendix")] = rdata.RList([rdata.RArray(np.arange(1, src.n_snp + 1, dtype=np.int32))],
                                                    {"names": ["snp.index"]})               # list(snp.index = 1:n.snp)
        p = str(tmp_path / f"{name}.RData")
        rdata.save_rdata(p, {"mobj": robj}, compress=compress, version=3, altrep=True)
        blob = open(p, "rb").read()
        assert blob.startswith(magic)
        got = M.load_model(p, "mobj")
        assert got.snp_id == src.snp_id and got.sample_id == src.sample_id and got.hla_allele == src.hla_allele
        assert np.array_equal(got.snp_position, src.snp_position)
        assert len(got.classifiers) == len(src.classifiers)
        for a, b in zip(got.classifiers, src.classifiers):
            assert np.array_equal(a.snpidx, b.snpidx) and np.array_equal(a.freq, b.freq) and a.haplo == b.haplo
            assert np.array_equal(a.hla, b.hla) and np.array_equal(a.samp_num, b.samp_num) and a.outofbag_acc == b.outofbag_acc
        if src.matching is not None:
            assert np.array_equal(got.matching, src.matching)
        assert np.array_equal(got.appendix["snp.index"], np.arange(1, src.n_snp + 1))
    # the ALTREP route was really taken: the stream holds the class symbols
    raw = rdata._decompress(open(p, "rb").read())
    assert b"compact_intseq" in raw and b"wrap_real" in raw and b"wrap_string" in raw


def _anonymised(robj, rdata):
    """hlaPublish(..., anonymize=TRUE): mobj$sample.id <- NULL; classifiers[[i]]$samp.num <- NULL."""
    keep = [i for i, n in enumerate(robj.names) if n != "sample.id"]
    out = rdata.RList([robj[i] for i in keep], dict(robj.attrs, names=[robj.names[i] for i in keep]))
    cls = []
    for tree in out["classifiers"]:
        k = [i for i, n in enumerate(tree.names) if n != "samp.num"]
        cls.append(rdata.RList([tree[i] for i in k], dict(tree.attrs, names=[tree.names[i] for i in k])))
    out[out.names.index("classifiers")] = rdata.RList(cls)
    return out


def test_anonymised_model_and_appendix(tmp_path, model_a):
    from hibag_amd import model as M, rdata
    robj = _anonymised(M.model_to_robj(model_a), rdata)
    appendix = rdata.RList([rdata.RStrings(["European"]), rdata.RStrings(["Illumina 1M Duo"]), rdata.RStrings(["demo, 2026"])],
                           {"names": ["ancestry", "platform", "information"]})
    robj[robj.names.index("appendix")] = appendix
    p = str(tmp_path / "pub.RData")
    rdata.save_rdata(p, {"mobj": robj}, compress="xz", version=3, altrep=True)
    got = M.load_model(p, "mobj")
    assert got.sample_id == [] and got.n_samp == model_a.n_samp
    for c in got.classifiers:
        assert np.array_equal(c.samp_num, np.ones(model_a.n_samp, np.int32))         # R/HIBAG.R:1151-1152
    assert list(got.appendix["platform"]) == ["Illumina 1M Duo"] and got.appendix.names == ["ancestry", "platform", "information"]
    # and it survives our own writer again (appendix kept)
    q = str(tmp_path / "again.RData")
    M.save_model(q, got, compress="bzip2", version=3)
    again = M.load_model(q)
    assert list(again.appendix["ancestry"]) == ["European"]


def test_writer_rejects_altrep_in_version2(tmp_path):
    from hibag_amd import rdata
    with pytest.raises(ValueError):
        rdata.save_rds(str(tmp_path / "z.rds"), rdata.Wrapped(rdata.RArray(np.array([1.0]))), version=2)
    with pytest.raises(ValueError):
        rdata.save_rdata(str(tmp_path / "z.RData"), {"x": 1}, compress="lz4")


@pytest.mark.gpu
def test_version3_anonymised_model_predicts_like_the_original(tmp_path, hapmap_geno, model_a, oracle):
    """A published-style file (RDX3, xz, ALTREP, anonymised, with appendix) loaded and run on the GPU gives the
    outputs of the original model bit for bit (samp.num does not enter prediction)."""
    import hibag_amd as hb
    from hibag_amd import model as M, rdata
    from conftest import align_geno
    hb.hlaSetKernelTarget("hip")
    robj = _anonymised(M.model_to_robj(model_a), rdata)
    p = str(tmp_path / "pub.RData")
    rdata.save_rdata(p, {"mobj": robj}, compress="xz", version=3, altrep=True)
    pub = M.load_model(p, "mobj")
    G = align_geno(model_a, hapmap_geno)
    got = hb.hlaModelFromObj(pub).predict_raw(G, 1, want_dosage=True, want_prob=True)
    want = oracle.predict(oracle.flatten(model_a), G, vote_method=1)
    for k in ("h1", "h2", "prob", "matching", "dosage", "postprob"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k
