"""Feature retrieval of VC.vc on the device (reference src/vc_infer_pipeline.py:409-431, index load :497-512).

The reference searches a faiss index on the host for the k = 8 nearest training features of every HuBERT frame and blends their
inverse-square-distance average into the features -- a device -> host -> device round trip inside every chunk.  Here the index
lives in HBM and the search runs on the device WITH FAISS' SEMANTICS for the index type the file holds:

  * IndexIVFFlat (what RVC trainers write: `index_factory(dim, "IVF<n>,Flat")`, `nprobe = 1`, file "added_IVF<n>_Flat_nprobe_1_*.index"):
    1. coarse quantizer = IndexFlatL2 over the nlist centroids: |q|^2 + |c|^2 - 2 q.c from one GEMM (the conv kernel as a 1x1
       convolution; faiss' own BLAS path for >= 20 queries uses the same expansion), the `nprobe` nearest centroids per query
       (aicg_knn8, nearest first);
    2. aicg_ivf_scan8 scans only those inverted lists, distances computed directly as sum (q - x)^2 (faiss fvec_L2sqr), 8 smallest;
    3. aicg_index_mix blends.  Neighbour SETS equal faiss' wherever distances differ by more than fp32 summation noise (faiss' SIMD
       accumulation order is not reproduced; tests measure against a float64 restatement of IndexIVFFlat::search).
  * IndexFlatL2 / a bare `.npy` of vectors (total_fea.npy): exhaustive search, as faiss does for a flat index: distance GEMM per
    column chunk + aicg_knn8, weights from directly recomputed distances.
  AICG_KNN=exact forces the exhaustive search on an IVF file too (better neighbours than the reference's, different from them).

File format (`read_faiss_index`), restated from faiss 1.7.x impl/index_write.cpp / index_read.cpp -- faiss is not installable here, so
the reader is pinned only by tests/test_retrieval.py's skip-unless-faiss round trip: PARITY UNPINNED until that test has run.
All integers little-endian; `idx_t` = int64, `size_t` = uint64:

  write_index_header      int32 d | int64 ntotal | int64 dummy (1 << 20) | int64 dummy | uint8 is_trained | int32 metric_type
                          (0 = inner product, 1 = L2) | [float32 metric_arg only when metric_type > 1]
  IndexFlat               fourcc "IxF2" (L2) / "IxFI" (IP) / "IxFl" | header | uint64 n_floats | n_floats x float32 (row-major
                          (ntotal, d); newer writers emit the same bytes through WRITEXBVECTOR: size = bytes / 4)
  IndexIVFFlat            fourcc "IwFl" | header | uint64 nlist | uint64 nprobe | <quantizer: a complete index, here IndexFlat>
                          | direct map: uint8 type (0 none, 1 array, 2 hashtable) | uint64 n | n x int64 [type 2: + uint64 m |
                          m x (int64, int64)] | inverted lists
  ArrayInvertedLists      fourcc "ilar" | uint64 nlist | uint64 code_size (= 4 d) | fourcc "full": uint64 nlist | nlist x uint64
                          sizes, or "sprs": uint64 2 m | m x (uint64 list, uint64 size) | then for every non-empty list in order:
                          size x code_size bytes of codes (float32 vectors) | size x int64 ids
"""
import os
import struct

import numpy as np
import torch
from . import _env

from . import ops

CHUNK = 16384  # index vectors per distance GEMM (exhaustive search)


class FeatureIndex:
    """Index vectors resident in HBM.  `vectors` (N, dim) in STORAGE order; with `lists` = (centroids (nlist, dim), sizes (nlist,),
    ids (N,), nprobe) the storage order is list by list and searches follow IndexIVFFlat; without, exhaustive (IndexFlatL2)."""

    def __init__(self, vectors, device, lists=None, exact=None):
        v = torch.as_tensor(np.ascontiguousarray(vectors, dtype=np.float32))
        assert v.dim() == 2
        self.big = v.to(device).contiguous()                       # (N, dim): rows scanned / gathered by the mix
        self.ntotal, self.dim = self.big.shape
        self.device = device
        if exact is None:
            exact = _env.dev("AICG_KNN", "").lower() == "exact"
        self.ivf = lists is not None and not exact
        self.labels = None                                         # storage position -> faiss label (None: identity)
        if lists is not None:
            cent, sizes, ids, nprobe = lists
            self.labels = torch.as_tensor(np.ascontiguousarray(ids, dtype=np.int64)).to(device)
        if self.ivf:
            self.nlist = int(len(sizes))
            self.nprobe = int(min(max(int(nprobe), 1), self.nlist))
            if self.nprobe > 8:
                raise NotImplementedError("IVF search with nprobe = %d: the device scan visits at most 8 lists per query (RVC index "
                                          "files carry nprobe = 1); set AICG_KNN=exact for an exhaustive search" % self.nprobe)
            off = np.zeros(self.nlist + 1, np.int64)
            off[1:] = np.cumsum(np.asarray(sizes, dtype=np.int64))
            assert off[-1] == self.ntotal
            self.list_off = torch.from_numpy(off).to(device)
            c = torch.as_tensor(np.ascontiguousarray(cent, dtype=np.float32)).to(device).contiguous()
            self.cent_t = c.t().contiguous().unsqueeze(0)          # (1, dim, nlist): channel-major input of the coarse GEMM
            self.cnorm = ops.row_sqnorm(c)
        else:
            self.big_t = self.big.t().contiguous().unsqueeze(0)   # (1, dim, N): channel-major input of the distance GEMM
            self.xnorm = ops.row_sqnorm(self.big)

    def _nearest_columns(self, feats, mat_t, norms):
        """8 nearest columns of (1, dim, n) `mat_t` for every row of feats, by the GEMM expansion (ascending; ties: lower column)."""
        t = feats.shape[0]
        with ops.fp32_layers():   # neighbour ids are an index selection: always the fp32 kernels
            pc = ops.PackedConv(feats.unsqueeze(-1), None, device=feats.device)   # queries as the GEMM's rows
        qnorm = ops.row_sqnorm(feats)
        best_d = torch.empty((t, 8), dtype=torch.float32, device=feats.device)
        best_i = torch.empty((t, 8), dtype=torch.int64, device=feats.device)
        n = mat_t.shape[2]
        for c0 in range(0, n, CHUNK):
            c1 = min(c0 + CHUNK, n)
            dots = ops.conv(mat_t[:, :, c0:c1], pc)[0]                   # (T, c1 - c0)
            ops.knn8_update(dots, norms[c0:c1], qnorm, c0, best_d, best_i, merge=c0 > 0)
        return best_d, best_i

    def search_positions(self, feats):
        """feats (T, dim) float32 device -> (squared distances (T, 8) ascending, storage positions (T, 8) int64; -1 / inf when an
        IVF list holds fewer than 8 vectors)."""
        feats = feats.contiguous().float()
        if not self.ivf:
            d, pos = self._nearest_columns(feats, self.big_t, self.xnorm)
            return d, torch.where(pos < self.ntotal, pos, torch.full_like(pos, -1))   # fewer than 8 vectors in total
        _, probe = self._nearest_columns(feats, self.cent_t, self.cnorm)
        return ops.ivf_scan8(feats, self.big, self.list_off, probe, self.nprobe)

    def search(self, feats):
        """-> (squared distances (T, 8), faiss labels (T, 8)) like `index.search(feats, 8)` (missing neighbours: label -1)."""
        d, pos = self.search_positions(feats)
        if self.labels is None:
            return d, pos
        return d, torch.where(pos >= 0, self.labels[pos.clamp_min(0)], pos)

    def mix_(self, feats, index_rate):
        """In-place blend of (T, dim) features (reference :415-431)."""
        feats = feats.contiguous()
        best_d, pos = self.search_positions(feats)
        return ops.index_mix_(feats, self.big, best_d, pos, index_rate, recompute=not self.ivf)


# ---- faiss file reader (IndexFlat / IndexIVFFlat with array inverted lists) --------------------------------------------------
class _Reader:
    def __init__(self, data):
        self.d, self.o = data, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.d, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def fourcc(self):
        s = self.d[self.o:self.o + 4].decode("latin1")
        self.o += 4
        return s

    def vector(self, dtype):
        n = self.take("Q")
        a = np.frombuffer(self.d, dtype=dtype, count=n, offset=self.o)
        self.o += n * np.dtype(dtype).itemsize
        return a

    def header(self):
        d, ntotal = self.take("i"), self.take("q")
        self.take("q"); self.take("q")                    # two dummies
        self.take("B")                                    # is_trained
        metric = self.take("i")
        if metric > 1:
            self.take("f")                                # metric_arg
        return d, ntotal, metric

    def flat(self, kind):
        if kind not in ("IxF2", "IxFI", "IxFl"):
            raise ValueError("unsupported faiss index type %r (IndexFlat / IndexIVFFlat are read here)" % kind)
        d, ntotal, metric = self.header()
        return self.vector(np.float32).reshape(ntotal, d), metric


def read_faiss_index(path):
    """-> dict(kind 'flat' | 'ivf', d, ntotal, metric (0 IP, 1 L2), vectors (ntotal, d) in STORAGE order, ids (ntotal,) faiss
    labels of the stored rows [, nlist, nprobe, centroids (nlist, d), sizes (nlist,)])."""
    r = _Reader(open(path, "rb").read())
    kind = r.fourcc()
    if kind != "IwFl":
        vec, metric = r.flat(kind)
        return {"kind": "flat", "d": vec.shape[1], "ntotal": vec.shape[0], "metric": metric, "vectors": vec.copy(),
                "ids": np.arange(vec.shape[0], dtype=np.int64)}
    d, ntotal, metric = r.header()
    nlist, nprobe = r.take("Q"), r.take("Q")
    cent, _ = r.flat(r.fourcc())                          # coarse quantizer: a flat index of the nlist centroids
    if cent.shape != (nlist, d):
        raise ValueError("coarse quantizer holds %s centroids, header says (%d, %d)" % (cent.shape, nlist, d))
    dm = r.take("B")                                      # direct map
    r.vector(np.int64)
    if dm == 2:
        r.o += 16 * r.take("Q")
    if r.fourcc() != "ilar":
        raise ValueError("only array inverted lists are supported")
    nl, code_size = r.take("Q"), r.take("Q")
    if nl != nlist or code_size != 4 * d:
        raise ValueError("inverted lists do not match the header (nlist %d/%d, code size %d for d = %d)" % (nl, nlist, code_size, d))
    lt = r.fourcc()
    if lt == "full":
        sizes = r.vector(np.uint64).astype(np.int64)
    elif lt == "sprs":
        sp = r.vector(np.uint64).astype(np.int64)
        sizes = np.zeros(nlist, np.int64)
        sizes[sp[0::2]] = sp[1::2]
    else:
        raise ValueError("unknown inverted-list layout %r" % lt)
    if int(sizes.sum()) != ntotal:
        raise ValueError("inverted lists hold %d vectors, header says %d" % (int(sizes.sum()), ntotal))
    vecs = np.empty((ntotal, d), np.float32)
    ids = np.empty(ntotal, np.int64)
    at = 0
    for n in sizes:
        n = int(n)
        vecs[at:at + n] = np.frombuffer(r.d, dtype=np.float32, count=n * d, offset=r.o).reshape(n, d)
        r.o += n * code_size
        ids[at:at + n] = np.frombuffer(r.d, dtype=np.int64, count=n, offset=r.o)
        r.o += 8 * n
        at += n
    return {"kind": "ivf", "d": d, "ntotal": ntotal, "metric": metric, "nlist": int(nlist), "nprobe": int(nprobe),
            "centroids": cent.copy(), "sizes": sizes, "vectors": vecs, "ids": ids}


def read_faiss_vectors(path):
    """All stored vectors in label order (= index.reconstruct_n(0, ntotal), the reference's `big_npy`)."""
    ix = read_faiss_index(path)
    out = np.empty_like(ix["vectors"])
    out[ix["ids"]] = ix["vectors"]
    return out


def load_index(path, device):
    """FeatureIndex from `path` (.npy of vectors or a faiss .index), or None when there is nothing to read.  Unsupported index
    types / metrics raise (pipeline() then runs without retrieval, like the reference after a failed read_index)."""
    if not path or not os.path.exists(path):
        return None
    if path.endswith(".npy"):
        return FeatureIndex(np.load(path), device)
    ix = read_faiss_index(path)
    if ix["metric"] != 1:
        raise ValueError("faiss index with metric_type %d: only L2 (1) is searched here" % ix["metric"])
    if ix["kind"] == "ivf":
        return FeatureIndex(ix["vectors"], device, lists=(ix["centroids"], ix["sizes"], ix["ids"], ix["nprobe"]))
    return FeatureIndex(ix["vectors"], device)
