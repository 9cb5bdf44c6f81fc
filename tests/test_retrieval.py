"""Feature retrieval (reference src/vc_infer_pipeline.py:409-431, index load :497-512) on the device with faiss'
search semantics: IndexIVFFlat (coarse quantizer -> nprobe lists -> direct-L2 scan -> k = 8 heap) and IndexFlatL2 (exhaustive),
checked against oracle/faiss_ivf.py (float64 restatement of faiss 1.7.x), the inverse-square blend, the `.index` reader against
(a) a writer kept in THIS file -- circular, says so -- and (b) faiss itself when it can be imported (skipped here: not installable)."""
import os
import struct

import numpy as np
import pytest
import torch

from aicovergen_amd import retrieval
from oracle import faiss_ivf


def _clustered(rng, n, dim, nlist, small_list=None):
    """Training-feature-like vectors around `nlist` centres, each assigned to its NEAREST centroid (what IndexIVFFlat.add does),
    stored list by list.  `small_list`: that list keeps only 3 vectors (fewer than k)."""
    cent = rng.standard_normal((nlist, dim)).astype(np.float32) * 2.0
    x = (cent[rng.integers(0, nlist, n)] + rng.standard_normal((n, dim)).astype(np.float32) * 0.9).astype(np.float32)
    assign = ((x[:, None, :].astype(np.float64) - cent[None].astype(np.float64)) ** 2).sum(-1).argmin(1)
    if small_list is not None:
        keep = np.ones(n, bool)
        keep[np.nonzero(assign == small_list)[0][3:]] = False
        x, assign = x[keep], assign[keep]
    ids_by_list = [np.nonzero(assign == l)[0].astype(np.int64) for l in range(nlist)]
    sizes = np.array([len(i) for i in ids_by_list], np.int64)
    ids = np.concatenate(ids_by_list)
    return cent, sizes, ids, x[ids], x      # x[ids]: storage order; x: label order


@pytest.mark.parametrize("nprobe", [1, 3])
def test_ivf_search_follows_faiss(dev, nprobe):
    rng = np.random.default_rng(11 + nprobe)
    dim, nlist = 64, 12
    cent, sizes, ids, stored, by_label = _clustered(rng, 1500, dim, nlist, small_list=5)
    t = 97
    feats = (by_label[rng.integers(0, len(by_label), t)] + rng.standard_normal((t, dim)).astype(np.float32) * 0.7).astype(np.float32)
    feats[0] = cent[5] + 0.01                                   # lands in the 3-vector list: 5 missing neighbours at nprobe 1
    idx = retrieval.FeatureIndex(stored, dev.device, lists=(cent, sizes, ids, nprobe), exact=False)
    assert idx.ivf and idx.nprobe == nprobe
    d, lab = idx.search(dev.t(torch.from_numpy(feats)))
    d, lab = d.cpu().numpy().astype(np.float64), lab.cpu().numpy()
    D, I = faiss_ivf.ivf_search(feats, cent, sizes, stored, ids, nprobe)
    if nprobe == 1:
        assert (I[0, 3:] == -1).all() and (lab[0, 3:] == -1).all() and np.isinf(d[0, 3:]).all()
    found = I >= 0
    assert np.array_equal(found, lab >= 0)
    assert np.allclose(d[found], D[found], rtol=2e-6, atol=1e-6)           # direct fp32 sum of squares vs float64
    # labels agree wherever the next-best candidate is not within fp32 noise (ranks inside the 8 AND the 8th / 9th boundary)
    firm = found.copy()
    with np.errstate(invalid="ignore"):
        firm[:, :-1] &= ~found[:, 1:] | (np.diff(D, axis=1) > 1e-4 * D[:, 1:].clip(1e-30))
    assert (lab[firm] == I[firm]).mean() > 0.999 and (lab[found] == I[found]).mean() > 0.99
    for r in range(t):
        assert set(lab[r][lab[r] >= 0]) == set(I[r][I[r] >= 0]) or np.min(np.abs(np.diff(np.sort(D[r][np.isfinite(D[r])])))) < 1e-3
    # IVF is NOT the exhaustive search: with one probe some true neighbours sit in other lists
    De, Ie = faiss_ivf.flat_search(feats, by_label)
    if nprobe == 1:
        assert any(set(I[r]) != set(Ie[r]) for r in range(t))
    # blend (reference :417-431)
    f = dev.t(torch.from_numpy(feats.copy()))
    mixed = idx.mix_(f, 0.6).cpu().numpy()
    want = faiss_ivf.mix(feats, by_label, D, I, 0.6)
    assert np.abs(mixed - want).max() < 2e-5 * np.abs(want).max()


def test_exhaustive_search_and_mix(dev):
    """IndexFlatL2 semantics (a flat index file, a bare total_fea.npy, or AICG_KNN=exact): all vectors, chunked distance GEMM, ties
    by the lower label, weights from directly recomputed distances."""
    rng = np.random.default_rng(3)
    n, dim, t = 3000, 64, 37
    big = rng.standard_normal((n, dim)).astype(np.float32)
    big[100] = big[7]                                           # an exact duplicate: tie broken by label
    feats = rng.standard_normal((t, dim)).astype(np.float32)
    feats[3] = big[7] + 1e-3
    old = retrieval.CHUNK
    retrieval.CHUNK = 1024                                      # three column chunks, the last one ragged
    try:
        idx = retrieval.FeatureIndex(big, dev.device)
        assert not idx.ivf
        d, i = idx.search(dev.t(torch.from_numpy(feats)))
        D, I = faiss_ivf.flat_search(feats, big)
        got_i = i.cpu().numpy()
        d64 = ((feats[:, None, :].astype(np.float64) - big[None].astype(np.float64)) ** 2).sum(-1)
        nxt = np.take_along_axis(d64, np.lexsort((np.arange(n)[None].repeat(t, 0), d64), axis=1)[:, 1:9], 1)
        firm = nxt - D > 1e-3                                   # the GEMM expansion's fp32 noise at |x|^2 ~ 64
        assert (got_i[firm] == I[firm]).all()
        assert np.allclose(d.cpu().numpy(), D, rtol=1e-4, atol=1e-3)
        assert got_i[3, 0] == 7 and got_i[3, 1] == 100
        f = dev.t(torch.from_numpy(feats.copy()))
        mixed = idx.mix_(f, 0.6).cpu().numpy()
        Dg = np.take_along_axis(d64, got_i, 1)                  # the blend uses DIRECT distances of the neighbours found
        want = faiss_ivf.mix(feats, big, Dg, got_i, 0.6)
        assert np.abs(mixed - want).max() < 2e-5 * np.abs(want).max()
    finally:
        retrieval.CHUNK = old


def test_duplicates_and_tiny_indexes_stay_finite(dev):
    """A query that EQUALS a stored vector (distance exactly 0: the reference's weights become inf / inf) and an index with fewer
    than 8 vectors: finite output, all weight on the zero-distance rows / on the rows that exist."""
    rng = np.random.default_rng(5)
    big = rng.standard_normal((40, 32)).astype(np.float32)
    big[9] = big[2]
    feats = rng.standard_normal((6, 32)).astype(np.float32)
    feats[1] = big[2]
    for kw in (dict(), dict(lists=(big[:1] * 0, np.array([40]), np.arange(40), 1), exact=False)):
        idx = retrieval.FeatureIndex(big, dev.device, **kw)
        out = idx.mix_(dev.t(torch.from_numpy(feats.copy())), 1.0).cpu().numpy()
        assert np.isfinite(out).all()
        assert np.allclose(out[1], big[2], atol=1e-6)           # the two exact copies share the weight
    small = retrieval.FeatureIndex(big[:5], dev.device)
    d, i = small.search(dev.t(torch.from_numpy(feats)))
    i = i.cpu().numpy()
    assert (np.sort(i[:, :5], 1) == np.arange(5)).all() and (i[:, 5:] == -1).all()
    out = small.mix_(dev.t(torch.from_numpy(feats.copy())), 0.5).cpu().numpy()
    assert np.isfinite(out).all()
    with pytest.raises(NotImplementedError):
        retrieval.FeatureIndex(big, dev.device, lists=(np.zeros((20, 32), np.float32), np.full(20, 2), np.arange(40), 9), exact=False)


def test_pipeline_features_with_an_ivf_index_file(tmp_path, dev):
    """VC.pipeline's retrieval step (reference :409-431) end to end: `file_index` -> load_index -> per-chunk search + blend of the
    HuBERT features, against the float64 faiss restatement applied to the same un-mixed features."""
    import test_pipeline as tp
    from synthetic import weights
    from synthetic.inputs import vocal_like
    nets = weights.small_model_set(1234)
    vc, hub, net_g, tgt_sr = tp.build(dev, nets)
    audio = torch.from_numpy(vocal_like(1.5, 16000, 3))
    plain, _ = vc._vc_features(hub, audio, None, None, 0.0, "v2", False)
    dim = plain.shape[-1]
    rng = np.random.default_rng(8)
    train = (plain[0].cpu().numpy()[rng.integers(0, plain.shape[1], 400)] + 0.2 * rng.standard_normal((400, dim))).astype(np.float32)
    cent = train[:7].copy()
    assign = ((train[:, None].astype(np.float64) - cent[None]) ** 2).sum(-1).argmin(1)
    ids = np.concatenate([np.nonzero(assign == l)[0] for l in range(7)]).astype(np.int64)
    sizes = np.array([(assign == l).sum() for l in range(7)], np.int64)
    p = str(tmp_path / "added_IVF7_Flat_nprobe_1_x_v2.index")
    _write_ivf_flat(p, cent, sizes, ids, train[ids])
    index = retrieval.load_index(p, dev.device)
    mixed, _ = vc._vc_features(hub, audio, index, None, 0.75, "v2", False)
    D, I = faiss_ivf.ivf_search(plain[0].cpu().numpy(), cent, sizes, train[ids], ids, 1)
    want = faiss_ivf.mix(plain[0].cpu().numpy(), train, D, I, 0.75)
    assert np.abs(mixed[0].cpu().numpy() - want).max() < 1e-4 * np.abs(want).max()
    # and through pipeline(): the file is read, the output changes, nothing else breaks
    a = vocal_like(1.2, 16000, 4)
    base = vc.pipeline(hub, net_g, 0, a, "x.wav", [0, 0, 0], 0, "rmvpe", "", 0.75, 1, 3, tgt_sr, 0, 0.25, "v2", 0.33, 128, noise_seed=1)
    with_ix = vc.pipeline(hub, net_g, 0, a, "x.wav", [0, 0, 0], 0, "rmvpe", p, 0.75, 1, 3, tgt_sr, 0, 0.25, "v2", 0.33, 128, noise_seed=1)
    assert base.shape == with_ix.shape and not np.array_equal(base, with_ix)


# ---- the file format ---------------------------------------------------------------------------------------------------------
def _hdr(d, ntotal, metric=1):
    return struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, 1, metric)


def _write_ivf_flat(path, cent, sizes, ids, stored, nprobe=1, sparse=False, direct_map=0):
    """An IndexIVFFlat file laid out as aicovergen_amd/retrieval.py's docstring restates faiss' index_write.cpp.  CIRCULAR: this
    writer and the reader share one reading of the format; only the faiss round trip below pins it."""
    n, d = stored.shape
    nlist = len(sizes)
    out = bytearray(b"IwFl" + _hdr(d, n) + struct.pack("<QQ", nlist, nprobe))
    out += b"IxF2" + _hdr(d, nlist) + struct.pack("<Q", cent.size) + cent.astype(np.float32).tobytes()
    if direct_map == 0:
        out += struct.pack("<BQ", 0, 0)
    else:
        out += struct.pack("<BQ", 1, n) + np.arange(n, dtype=np.int64).tobytes()
    out += b"ilar" + struct.pack("<QQ", nlist, 4 * d)
    if sparse:
        nz = [(l, s) for l, s in enumerate(sizes) if s]
        out += b"sprs" + struct.pack("<Q", 2 * len(nz)) + np.array(nz, dtype=np.uint64).tobytes()
    else:
        out += b"full" + struct.pack("<Q", nlist) + np.asarray(sizes, np.uint64).tobytes()
    at = 0
    for s in sizes:
        out += stored[at:at + s].astype(np.float32).tobytes() + np.asarray(ids[at:at + s], np.int64).tobytes()
        at += s
    open(path, "wb").write(bytes(out))


@pytest.mark.parametrize("sparse,direct_map", [(False, 0), (True, 1)])
def test_index_file_reader_and_loader(tmp_path, dev, sparse, direct_map):
    rng = np.random.default_rng(4)
    cent, sizes, ids, stored, by_label = _clustered(rng, 300, 16, 6)
    if sparse:                                                  # faiss writes the sparse form when most lists are empty
        nl = 20
        cent = np.concatenate([cent, rng.standard_normal((nl - 6, 16)).astype(np.float32) * 50.0 + 500.0])
        sizes = np.concatenate([sizes, np.zeros(nl - 6, np.int64)])
    p = str(tmp_path / "added_IVF6_Flat_nprobe_1_test_v2.index")
    _write_ivf_flat(p, cent, sizes, ids, stored, nprobe=1, sparse=sparse, direct_map=direct_map)
    ix = retrieval.read_faiss_index(p)
    assert ix["kind"] == "ivf" and ix["nprobe"] == 1 and ix["nlist"] == len(sizes) and ix["metric"] == 1
    assert np.array_equal(ix["centroids"], cent) and np.array_equal(ix["sizes"], sizes)
    assert np.array_equal(ix["vectors"], stored) and np.array_equal(ix["ids"], ids)
    assert np.array_equal(retrieval.read_faiss_vectors(p), by_label)      # = index.reconstruct_n(0, ntotal)
    idx = retrieval.load_index(p, dev.device)
    assert idx.ivf and idx.nprobe == 1 and idx.ntotal == len(by_label)
    feats = (by_label[:20] + 0.3).astype(np.float32)
    D, I = faiss_ivf.ivf_search(feats, cent, sizes, stored, ids, 1)
    _, lab = idx.search(dev.t(torch.from_numpy(feats)))
    assert (lab.cpu().numpy() == I).mean() > 0.99
    # flat file + bare vectors
    pf = str(tmp_path / "flat.index")
    open(pf, "wb").write(b"IxF2" + _hdr(16, len(by_label)) + struct.pack("<Q", by_label.size) + by_label.tobytes())
    assert np.array_equal(retrieval.read_faiss_vectors(pf), by_label) and not retrieval.load_index(pf, dev.device).ivf
    np.save(tmp_path / "total_fea.npy", by_label)
    flat = retrieval.load_index(str(tmp_path / "total_fea.npy"), dev.device)
    assert flat.ntotal == len(by_label) and flat.dim == 16 and not flat.ivf
    (tmp_path / "bad.index").write_bytes(b"IxPQ" + b"\0" * 64)
    with pytest.raises(ValueError):
        retrieval.read_faiss_index(str(tmp_path / "bad.index"))
    open(str(tmp_path / "ip.index"), "wb").write(b"IxFI" + _hdr(16, 1, metric=0) + struct.pack("<Q", 16) + by_label[:1].tobytes())
    with pytest.raises(ValueError):
        retrieval.load_index(str(tmp_path / "ip.index"), dev.device)      # inner-product metric: not searched here


def test_reader_and_search_against_faiss_itself(tmp_path, dev):
    """THE PIN (skipped wherever faiss cannot be imported -- it cannot be installed in the build container): an index built the way
    RVC's trainer builds it, written by faiss, read by retrieval.read_faiss_index, searched on the device, compared with
    index.search."""
    faiss = pytest.importorskip("faiss")
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((16, 64))[rng.integers(0, 16, 4000)] * 2 + rng.standard_normal((4000, 64))).astype(np.float32)
    index = faiss.index_factory(64, "IVF16,Flat")
    faiss.extract_index_ivf(index).nprobe = 1
    index.train(x)
    for i in range(0, len(x), 1000):
        index.add(x[i:i + 1000])
    p = str(tmp_path / "added.index")
    faiss.write_index(index, p)
    ix = retrieval.read_faiss_index(p)
    assert ix["kind"] == "ivf" and ix["nprobe"] == 1 and ix["nlist"] == 16
    assert np.array_equal(retrieval.read_faiss_vectors(p), index.reconstruct_n(0, index.ntotal))
    q = (x[:200] + 0.5 * rng.standard_normal((200, 64))).astype(np.float32)
    score, want = index.search(q, 8)
    d, lab = retrieval.load_index(p, dev.device).search(dev.t(torch.from_numpy(q)))
    assert (lab.cpu().numpy() == want).mean() > 0.995
    assert np.allclose(d.cpu().numpy(), score, rtol=1e-5, atol=1e-5)
