"""Feature retrieval of VC.vc on the device (reference src/vc_infer_pipeline.py:409-431, index load :497-512).

The reference searches a faiss IVF-Flat index on the host for the k = 8 nearest training features of every HuBERT frame and
blends their inverse-square-distance average into the features -- a device -> host -> device round trip inside every chunk.
Here the index vectors live in HBM; the inner products are one 1x1 "convolution" on the MFMA kernel per column chunk
(queries = packed weights, index vectors = channel-major input), aicg_knn8 keeps the running 8 smallest squared L2 distances
and aicg_index_mix writes the blend.  The search is EXACT (brute force); faiss' IVF search with its default nprobe = 1 only
visits the nearest inverted list, so results can differ from the reference's where a neighbour sits in another list.

Index sources: a `.npy` of the vectors (RVC trainers write total_fea.npy next to the index), a faiss `.index` file read by faiss
when that package is installed, or by the small IndexIVFFlat / IndexFlat reader below (format restated from faiss'
impl/index_write.cpp -- no faiss here to cross-check: PARITY UNPINNED for the reader)."""
import os
import struct

import numpy as np
import torch

from . import ops

CHUNK = 16384  # index vectors per distance GEMM


class FeatureIndex:
    def __init__(self, vectors, device):
        v = torch.as_tensor(np.ascontiguousarray(vectors, dtype=np.float32))
        assert v.dim() == 2
        self.big = v.to(device).contiguous()                       # (N, dim): rows gathered by the mix
        self.big_t = self.big.t().contiguous().unsqueeze(0)       # (1, dim, N): channel-major input of the distance GEMM
        self.xnorm = ops.row_sqnorm(self.big)
        self.ntotal, self.dim = self.big.shape

    def search(self, feats):
        """feats (T, dim) float32 device -> (squared distances (T, 8) ascending, ids (T, 8) int64)."""
        t = feats.shape[0]
        feats = feats.contiguous().float()
        with ops.fp32_layers():   # neighbour ids are an index selection: always the fp32 kernels
            pc = ops.PackedConv(feats.unsqueeze(-1), None, device=feats.device)   # queries as the GEMM's rows
        qnorm = ops.row_sqnorm(feats)
        best_d = torch.empty((t, 8), dtype=torch.float32, device=feats.device)
        best_i = torch.empty((t, 8), dtype=torch.int64, device=feats.device)
        for c0 in range(0, self.ntotal, CHUNK):
            c1 = min(c0 + CHUNK, self.ntotal)
            dots = ops.conv(self.big_t[:, :, c0:c1], pc)[0]                   # (T, c1 - c0)
            ops.knn8_update(dots, self.xnorm[c0:c1], qnorm, c0, best_d, best_i, merge=c0 > 0)
        return best_d, best_i

    def mix_(self, feats, index_rate):
        """In-place blend of (T, dim) features (reference :415-431)."""
        best_d, best_i = self.search(feats)
        return ops.index_mix_(feats, self.big, best_d, best_i, index_rate)


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
        return d, ntotal


def read_faiss_vectors(path):
    """All stored vectors of a faiss IndexFlatL2/IP or IndexIVFFlat file in id order (= index.reconstruct_n(0, ntotal))."""
    r = _Reader(open(path, "rb").read())
    kind = r.fourcc()
    if kind in ("IxF2", "IxFI", "IxFl"):
        d, ntotal = r.header()
        return r.vector(np.float32).reshape(ntotal, d).copy()
    if kind != "IwFl":
        raise ValueError("unsupported faiss index type %r (IndexFlat / IndexIVFFlat are read here)" % kind)
    d, ntotal = r.header()
    nlist, _nprobe = r.take("Q"), r.take("Q")
    q = r.fourcc()                                        # coarse quantizer: a flat index of nlist centroids
    if q not in ("IxF2", "IxFI", "IxFl"):
        raise ValueError("unsupported coarse quantizer %r" % q)
    r.header()
    r.vector(np.float32)
    r.take("B")                                           # direct map type
    r.vector(np.int64)                                    # direct map array
    if r.fourcc() != "ilar":
        raise ValueError("only array inverted lists are supported")
    nl, code_size = r.take("Q"), r.take("Q")
    if nl != nlist or code_size != 4 * d:
        raise ValueError("inverted lists do not match the header (nlist %d/%d, code size %d for d = %d)" % (nl, nlist, code_size, d))
    lt = r.fourcc()
    if lt == "full":
        sizes = r.vector(np.uint64)
    elif lt == "sprs":
        sp = r.vector(np.uint64)
        sizes = np.zeros(nlist, np.uint64)
        sizes[sp[0::2].astype(np.int64)] = sp[1::2]
    else:
        raise ValueError("unknown inverted-list layout %r" % lt)
    out = np.zeros((ntotal, d), np.float32)
    seen = 0
    for n in sizes.astype(np.int64):
        codes = np.frombuffer(r.d, dtype=np.float32, count=n * d, offset=r.o).reshape(n, d)
        r.o += n * code_size
        ids = np.frombuffer(r.d, dtype=np.int64, count=n, offset=r.o)
        r.o += 8 * n
        out[ids] = codes
        seen += n
    if seen != ntotal:
        raise ValueError("inverted lists hold %d vectors, header says %d" % (seen, ntotal))
    return out


def load_index(path, device):
    """FeatureIndex from `path` (.npy of vectors or a faiss .index), or None when it cannot be read."""
    if not path or not os.path.exists(path):
        return None
    if path.endswith(".npy"):
        return FeatureIndex(np.load(path), device)
    try:
        import faiss  # the reference's own loader, when installed
        index = faiss.read_index(path)
        return FeatureIndex(index.reconstruct_n(0, index.ntotal), device)
    except ImportError:
        pass
    return FeatureIndex(read_faiss_vectors(path), device)
