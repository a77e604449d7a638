"""Which float64 evaluation order reproduces the E-step of the reference's k-means
(sklearn 0.24.2 _k_means_lloyd.pyx: row_norms + BLAS dgemm(alpha=-2, beta=1)) bit for bit?

    gcc -O2 -mfma -ffp-contract=off -shared -fPIC -o /tmp/libfitprobe.so oracle/refgen/fit_probe_variants.c -lm
    /opt/conda/bin/python3.9 oracle/refgen/fit_probe.py [ncases]

Three experiments against the oracle stack (numpy 1.26.4, sklearn 0.24.2, MKL 2021.4 behind scipy's
cython BLAS), results as of this round in results/fit_probe.txt:
  1. the E-step of the fuzz cases (centred samples, three kinds of centres) under six candidate
     orders: only "dot product accumulated from zero by fma in band order, then |c|^2 + (-2 dot)"
     (variants 1 = 2 = 4, the scaling by -2 being exact) has no mismatch for nBands >= 2; for one
     band it is fma(x, -2c, |c|^2) (variant 0);
  2. KMeans.predict on tie-heavy data for 1..17 bands: same answer;
  3. |c|^2 = row_norms = einsum('ij,ij->i'): numpy's baseline-SSE2 two-lane loop (cn_sse2), 0
     mismatches over 1..33 bands; plain sequential orders (with or without fma) all differ.
Build container only (refenv.py)."""

# ---- experiment 1 ----
import sys, ctypes, numpy as np
import refenv
from refenv import shepseg
from oracle import oracle
import fuzz_vs_reference as fz
import sklearn
from sklearn.cluster import _k_means_lloyd as L
from sklearn.utils.extmath import row_norms
V = ctypes.CDLL('/tmp/libfitprobe.so')
def variant(X, C, cn, v):
    lab = np.empty(X.shape[0], np.int32)
    V.assign_variant(X.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(X.shape[0]), X.shape[1], C.ctypes.data_as(ctypes.c_void_p), C.shape[0], cn.ctypes.data_as(ctypes.c_void_p), v, lab.ctypes.data_as(ctypes.c_void_p))
    return lab
def sk_estep(X, C):
    n = X.shape[0]
    labels = np.full(n, -1, dtype=np.int32)
    cnew = np.zeros_like(C); w = np.zeros(C.shape[0]); shift = np.zeros(C.shape[0])
    L.lloyd_iter_chunked_dense(X, np.ones(n), row_norms(X, squared=True), C, cnew, w, labels, shift, 8, update_centers=False)
    return labels
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.RandomState(12345)
tot = np.zeros(6, np.int64); nrows = 0
for case in range(ncases):
    img, null_val = fz.make_img(rng, case)
    nb, nr, nc = img.shape
    k = int(rng.choice([2, 5, 10, 60])); min_seg = int(rng.choice([2, 5, 20, 50])); four = bool(rng.rand() < 0.6); pcnt = int(rng.choice([1, 10, 50, 100]))
    x = np.transpose(img, (1, 2, 0)).reshape(nr * nc, nb)
    if null_val is not None: x = x[(x != null_val).all(axis=1)]
    xs = x[::int(round(100. / pcnt))]
    if xs.shape[0] < k: continue
    X = xs.astype(np.float64); X = X - X.mean(axis=0)
    init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64) - xs.astype(np.float64).mean(axis=0)
    # a few E-steps on perturbed centres (means of random subsets, like real Lloyd centres)
    for rep in range(3):
        if rep == 0: C = np.ascontiguousarray(init)
        else:
            lab0 = rng.randint(0, k, X.shape[0])
            C = np.array([X[lab0 == j].mean(axis=0) if (lab0 == j).any() else init[j] for j in range(k)])
            if rep == 2: C = np.round(C * 4) / 4      # lattice-ish centres: ties
        C = np.ascontiguousarray(C)
        ref = sk_estep(np.ascontiguousarray(X), C)
        cn = row_norms(C, squared=True)
        mism = [int((variant(np.ascontiguousarray(X), C, cn, v) != ref).sum()) for v in range(6)]
        tot += np.array(mism); nrows += X.shape[0]
        if any(mism): print('case', case, 'rep', rep, img.dtype, 'nb', nb, 'k', k, 'n', X.shape[0], 'mismatches per variant', mism)
print('rows', nrows, 'total mismatches per variant', tot.tolist())

# ---- experiment 2 ----
import sys, ctypes, numpy as np
import refenv
from sklearn.cluster import _k_means_lloyd as L
from sklearn.cluster import KMeans
from sklearn.utils.extmath import row_norms
V = ctypes.CDLL('/tmp/libfitprobe.so')
def variant(X, C, cn, v):
    lab = np.empty(X.shape[0], np.int32)
    V.assign_variant(X.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(X.shape[0]), X.shape[1], C.ctypes.data_as(ctypes.c_void_p), C.shape[0], cn.ctypes.data_as(ctypes.c_void_p), v, lab.ctypes.data_as(ctypes.c_void_p))
    return lab
def cnv(C, v):
    out = np.empty(C.shape[0]); V.cn_variant(C.ctypes.data_as(ctypes.c_void_p), C.shape[0], C.shape[1], v, out.ctypes.data_as(ctypes.c_void_p)); return out
rng = np.random.RandomState(7)
cnm = np.zeros(4, np.int64); am = {}
for nb in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 17):
    for trial in range(6):
        k = int(rng.choice([2, 10, 60])); n = int(rng.randint(300, 5000))
        Xi = rng.randint(0, 40, size=(n, nb)) * (1 if trial % 2 else 50)      # few levels: ties
        C = np.ascontiguousarray(np.array([Xi[rng.randint(0, n, 7)].mean(axis=0) for _ in range(k)]))
        if trial >= 4: C = np.ascontiguousarray(C - C.mean(axis=0))       # non-integer-ish
        ref_cn = row_norms(C, squared=True)
        for v in range(4): cnm[v] += int((cnv(C, v) != ref_cn).sum())
        X = np.ascontiguousarray(Xi.astype(np.float64))
        km = KMeans(n_clusters=k, init=C, n_init=1, max_iter=1)
        km.cluster_centers_ = C; km._n_threads = 8
        ref = km.predict(Xi.astype(np.float64))
        for v in (0, 1):
            am[(nb, v)] = am.get((nb, v), 0) + int((variant(X, C, ref_cn, v) != ref).sum())
print('cn mismatches per variant (seq fma, seq mul+add, rev mul+add, rev fma):', cnm.tolist())
for nb in sorted(set(a for a, _ in am)): print('nb', nb, 'predict mismatches: variant0', am[(nb, 0)], 'variant1', am[(nb, 1)])

# ---- experiment 3 ----
import sys, ctypes, numpy as np
import refenv
from sklearn.utils.extmath import row_norms
V = ctypes.CDLL('/tmp/libfitprobe.so')
rng = np.random.RandomState(9)
bad = {}
for nb in list(range(1, 21)) + [33]:
    for trial in range(20):
        k = 60
        C = np.ascontiguousarray(rng.randn(k, nb) * 1000 + rng.randint(0, 5000, size=(1, nb)))
        out = np.empty(k); V.cn_sse2(C.ctypes.data_as(ctypes.c_void_p), k, nb, out.ctypes.data_as(ctypes.c_void_p))
        bad[nb] = bad.get(nb, 0) + int((out != row_norms(C, squared=True)).sum())
print(bad)
print(np.__version__, np.show_config.__module__)
