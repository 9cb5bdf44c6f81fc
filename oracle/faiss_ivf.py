"""TEST INFRASTRUCTURE (never imported by the product): float64 restatement of the two faiss searches the reference can hit at
src/vc_infer_pipeline.py:421 `score, ix = index.search(npy, k=8)` -- faiss 1.7.x, not vendored in /root/reference
(requirements.txt:3 `faiss-cpu==1.7.3`) and not installable here: PARITY UNPINNED against faiss itself; tests/test_retrieval.py holds the
skip-unless-faiss pin.

IndexIVFFlat::search (faiss/IndexIVF.cpp search -> quantizer->search(n, x, nprobe) -> search_preassigned; IndexIVFFlat.cpp
IVFFlatScanner<METRIC_L2>::scan_codes):
  1. coarse: the nprobe centroids with the smallest squared L2 distance to the query (IndexFlatL2), nearest first;
  2. for each probed list in that order, for each stored vector in list order: dis = fvec_L2sqr(q, x) = sum (q - x)^2;
     `if (dis < heap_top) replace_top` -- a max-heap of size k: a later candidate must be STRICTLY closer;
  3. heap_reorder: ascending distance; unfilled slots keep label -1 (distance FLT_MAX).
IndexFlatL2::search: the same heap over all vectors in label order."""
import numpy as np


def _top8(dist, labels):
    """Emulates the size-8 max-heap with strict replacement over candidates in scan order -> ascending (distance, label) lists."""
    order = np.lexsort((np.arange(len(dist)), dist))[:8]      # smallest distances, ties: earlier scanned
    d = np.full(8, np.inf)
    i = np.full(8, -1, np.int64)
    d[:len(order)] = dist[order]
    i[:len(order)] = labels[order]
    return d, i


def ivf_search(queries, centroids, sizes, vectors, ids, nprobe):
    """queries (T, d); centroids (nlist, d); vectors (N, d) stored list by list with `sizes`; ids (N,) labels.
    -> (squared distances (T, 8) float64 ascending, labels (T, 8) int64, -1 where the probed lists hold fewer than 8)."""
    q = np.asarray(queries, np.float64)
    c = np.asarray(centroids, np.float64)
    v = np.asarray(vectors, np.float64)
    off = np.concatenate([[0], np.cumsum(np.asarray(sizes, np.int64))])
    D = np.empty((len(q), 8))
    I = np.empty((len(q), 8), np.int64)
    for t in range(len(q)):
        cd = ((c - q[t]) ** 2).sum(1)
        probe = np.lexsort((np.arange(len(cd)), cd))[:nprobe]
        rows = np.concatenate([np.arange(off[l], off[l + 1]) for l in probe]) if len(probe) else np.zeros(0, np.int64)
        dist = ((v[rows] - q[t]) ** 2).sum(1)
        D[t], I[t] = _top8(dist, np.asarray(ids)[rows])
    return D, I


def flat_search(queries, vectors):
    q = np.asarray(queries, np.float64)
    v = np.asarray(vectors, np.float64)
    D = np.empty((len(q), 8))
    I = np.empty((len(q), 8), np.int64)
    for t in range(len(q)):
        D[t], I[t] = _top8(((v - q[t]) ** 2).sum(1), np.arange(len(v)))
    return D, I


def mix(feats, big_by_label, D, I, rate):
    """reference :417-431 in float64 (missing neighbours: faiss returns FLT_MAX -> weight 0, label -1 -> numpy's last row x 0)."""
    f = np.asarray(feats, np.float64)
    with np.errstate(divide="ignore"):
        w = np.square(1.0 / D)
    w = w / w.sum(1, keepdims=True)
    npy = (np.asarray(big_by_label, np.float64)[I] * w[:, :, None]).sum(1)
    return npy * rate + (1 - rate) * f
