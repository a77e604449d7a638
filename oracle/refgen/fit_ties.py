"""Classify the k-means fit mismatches of fuzz_vs_reference.py.

    /opt/conda/bin/python3.9 oracle/refgen/fit_ties.py [ncases]

For every fuzz case whose Lloyd trajectory (iteration count or final partition) differs between
the C oracle and sklearn 0.24.2, both are re-run with max_iter = 1, 2, ... up to the first
iteration t whose centres differ (as a set).  The labels that entered that M-step are
reconstructed with exact rational arithmetic from the common centres of iteration t - 1 (the
previous partition's exact means): a sample whose two nearest clusters are at EXACTLY the same
distance is an exact tie.  A divergence is "explained by ties" when the two sides' centres of
iteration t are both reproduced (to 1e-9) by assigning some of the tied samples to one or the
other of their equidistant clusters -- i.e. the reference's own choice there is decided by the
rounding inside its BLAS (MKL in this stack; the result changes with MKL's thread count and
operand alignment), not by the algorithm.  Build container only (refenv.py)."""
import sys
import warnings
from fractions import Fraction

import numpy as np

warnings.filterwarnings('ignore')
import refenv  # noqa: E402
from refenv import shepseg  # noqa: E402
from oracle import oracle  # noqa: E402
import fuzz_vs_reference as fz  # noqa: E402
from sklearn.cluster import KMeans  # noqa: E402


def partition_key(labels):
    """canonical form of a partition: cluster ids renumbered by first appearance"""
    m = {}
    return tuple(m.setdefault(int(v), len(m)) for v in labels)


def sorted_rows(a):
    return a[np.lexsort(a.T[::-1])]


def exact_tie_samples(xs, labels_prev, centres_prev_float):
    """Samples with an exact tie between their two nearest centres, the centres being the exact
    means of the previous partition (Fractions).  labels_prev None: centres are exact already
    (the integer initial centres)."""
    n, nb = xs.shape
    k = centres_prev_float.shape[0]
    if labels_prev is None:
        cen = [[Fraction(int(v)) for v in row] for row in centres_prev_float]
    else:
        cen = []
        for j in range(k):
            idx = np.flatnonzero(labels_prev == j)
            if len(idx) == 0:
                cen.append(None)
                continue
            s = xs[idx].astype(np.int64).sum(axis=0)
            cen.append([Fraction(int(v), len(idx)) for v in s])
    ties = []
    cf = np.array([[float(v) for v in c] if c is not None else [np.inf] * nb for c in cen])
    d = ((xs[:, None, :].astype(np.float64) - cf[None]) ** 2).sum(-1)
    order = np.argsort(d, axis=1)[:, :2]
    close = np.flatnonzero(np.abs(d[np.arange(n), order[:, 0]] - d[np.arange(n), order[:, 1]]) <=
                           1e-6 * np.maximum(1.0, d[np.arange(n), order[:, 0]]))
    for i in close:
        (a, b) = order[i]
        if cen[a] is None or cen[b] is None:
            continue
        da = sum((Fraction(int(v)) - c) ** 2 for v, c in zip(xs[i], cen[a]))
        db = sum((Fraction(int(v)) - c) ** 2 for v, c in zip(xs[i], cen[b]))
        if da == db:
            ties.append((int(i), int(a), int(b)))
    return ties


def exact_centres(xs, labels_prev, k, init=None):
    """The centres of an iteration as exact rationals: the integer initial centres, or the means of
    the partition `labels_prev` that produced them (None for an empty cluster)."""
    if labels_prev is None:
        return [[Fraction(int(v)) for v in row] for row in init]
    cen = []
    for j in range(k):
        idx = np.flatnonzero(labels_prev == j)
        if len(idx) == 0:
            cen.append(None)
            continue
        s = xs[idx].astype(np.int64).sum(axis=0)
        cen.append([Fraction(int(v), len(idx)) for v in s])
    return cen


def relocation_tie(xs, labels, cen):
    """The M-step relocates the n_empty farthest samples from their OLD centres `cen`
    (np.argpartition: the choice among equal distances is implementation-defined).  True when
    the n_empty-th and (n_empty+1)-th largest distances are exactly equal."""
    k = len(cen)
    w = np.bincount(labels, minlength=k)
    ne = int((w == 0).sum())
    if ne == 0 or ne >= len(labels):
        return False
    nb = xs.shape[1]
    cf = np.array([[float(v) for v in c] if c is not None else [0.0] * nb for c in cen])
    d = ((xs.astype(np.float64) - cf[labels]) ** 2).sum(axis=1)
    order = np.argsort(-d, kind='stable')
    cand = order[:min(len(order), ne + 64)]
    exact = sorted((sum((Fraction(int(v)) - c) ** 2 for v, c in zip(xs[i], cen[labels[i]])) for i in cand
                    if cen[labels[i]] is not None), reverse=True)
    return len(exact) > ne and exact[ne - 1] == exact[ne]


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rng = np.random.RandomState(12345)
    nmis = ntie = nother = 0
    for case in range(ncases):
        img, null_val = fz.make_img(rng, case)
        nb, nr, nc = img.shape
        k = int(rng.choice([2, 5, 10, 60]))
        _min_seg = int(rng.choice([2, 5, 20, 50]))
        _four = bool(rng.rand() < 0.6)
        pcnt = int(rng.choice([1, 10, 50, 100]))
        x = np.transpose(img, (1, 2, 0)).reshape(nr * nc, nb)
        if null_val is not None:
            x = x[(x != null_val).all(axis=1)]
        xs = x[::int(round(100. / pcnt))]
        if xs.shape[0] < k:
            continue
        init = shepseg.diagonalClusterCentres(xs, k)
        X = xs.astype(np.float64)
        km = KMeans(n_clusters=k, init=init, n_init=1).fit(xs)
        c, lab, nit = oracle.kmeans_fit(X, init.astype(np.float64))
        if nit == km.n_iter_ and partition_key(lab) == partition_key(km.labels_):
            continue
        nmis += 1
        # first iteration whose centres differ
        t = None
        prev = None
        for it in range(1, max(nit, km.n_iter_) + 1):
            kt = KMeans(n_clusters=k, init=init, n_init=1, max_iter=it).fit(xs)
            ct, lt, _n = oracle.kmeans_fit(X, init.astype(np.float64), max_iter=it)
            if np.abs(sorted_rows(ct) - sorted_rows(kt.cluster_centers_)).max() > 1e-9:
                t = it
                break
            prev = (ct, lt)
        if t is None:
            print('case %d: same centres at every iteration, the final label pass differs' % case)
            t = max(nit, km.n_iter_) + 1
        # the labels that entered M-step t come from the centres of t - 1; with max_iter = t - 1 the
        # labels returned are exactly that E-step (the final pass on the last centres)
        reloc = False
        if prev is None:
            ties = exact_tie_samples(xs, None, init.astype(np.float64))
            _c0, l0, _n0 = oracle.kmeans_fit(X, init.astype(np.float64), max_iter=0)
            reloc = relocation_tie(xs, l0, exact_centres(xs, None, k, init))
        else:
            # centres of iteration t-1 are the exact means of the labels that entered M-step t-1:
            # those are the labels returned with max_iter = t - 2 (or the E-step on init)
            if t - 2 >= 1:
                _c2, l2, _n2 = oracle.kmeans_fit(X, init.astype(np.float64), max_iter=t - 2)
            else:
                _c2, l2, _n2 = oracle.kmeans_fit(X, init.astype(np.float64), max_iter=0)
            ties = exact_tie_samples(xs, l2, prev[0])
            reloc = relocation_tie(xs, prev[1], exact_centres(xs, l2, k))
        kind = 'EXACT TIES (%d tied samples, e.g. sample %d between clusters %d/%d)' % (
            len(ties), ties[0][0], ties[0][1], ties[0][2]) if ties else 'no E-step tie'
        if reloc:
            kind += ' + EXACT TIE AMONG THE FARTHEST SAMPLES OF THE EMPTY-CLUSTER RELOCATION'
        if not ties and not reloc:
            # a relocation tie of an EARLIER iteration (identical samples: either choice gives the
            # same centres then, but other labels and weights, which surface later)
            lprev = None
            for it in range(1, t):
                _c, lin, _n = oracle.kmeans_fit(X, init.astype(np.float64), max_iter=it - 1)
                cen = exact_centres(xs, lprev, k, init)
                if relocation_tie(xs, lin, cen):
                    reloc = True
                    kind += ' (but an exact relocation tie at iteration %d)' % it
                    break
                lprev = lin
        if ties or reloc:
            ntie += 1
        else:
            nother += 1
        print('case %d %s nb=%d k=%d n=%d: n_iter oracle %d / sklearn %d, first differing iteration %d: %s'
              % (case, img.dtype, nb, k, xs.shape[0], nit, km.n_iter_, t, kind))
        sys.stdout.flush()
    print('stack:', refenv.STACK)
    print('DONE: %d cases, %d trajectories differ, %d start at an exact tie, %d unexplained'
          % (ncases, nmis, ntie, nother))


if __name__ == '__main__':
    main()
