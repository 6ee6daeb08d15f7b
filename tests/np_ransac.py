"""Independent numpy restatement of the fundamental-matrix RANSAC the host adapter runs (pvio_amd/host/fundamental_ransac.cpp,
standing in for cv::findFundamentalMat(FM_RANSAC)): same sampler (cv::RNG multiply-with-carry, seed 2^64 - 1), 7-point
models from the SVD null space, same score, same acceptance and iteration-count rule.  Test infrastructure only."""
import numpy as np

MASK32 = 0xFFFFFFFF


class Mwc:
    def __init__(self, seed=(1 << 64) - 1):
        self.state = seed

    def next(self):
        self.state = ((self.state & MASK32) * 4164903690 + (self.state >> 32)) & ((1 << 64) - 1)
        return self.state & MASK32

    def uniform(self, a, b):
        return a if a == b else self.next() % (b - a) + a


def solve_cubic(c):
    a0 = c[0]
    if a0 == 0:
        return list(np.roots(c[1:]).real)
    a1, a2, a3 = c[1] / a0, c[2] / a0, c[3] / a0
    Q = (a1 * a1 - 3 * a2) / 9.0
    R = (2 * a1 ** 3 - 9 * a1 * a2 + 27 * a3) / 54.0
    Qc = Q ** 3
    d = Qc - R * R
    if d > 0:
        th = np.arccos(R / np.sqrt(Qc))
        t0, t1, t2 = -2 * np.sqrt(Q), th / 3, a1 / 3
        return [t0 * np.cos(t1) - t2, t0 * np.cos(t1 + 2 * np.pi / 3) - t2, t0 * np.cos(t1 + 4 * np.pi / 3) - t2]
    if d == 0:
        r = np.cbrt(abs(R))
        return [-2 * r - a1 / 3, r - a1 / 3] if R >= 0 else [2 * r - a1 / 3, -r - a1 / 3]
    d = np.sqrt(-d)
    e = np.cbrt(d + abs(R))
    if R > 0:
        e = -e
    return [(e + Q / e) - a1 / 3]


def seven_point(p, q):
    x1, y1, x2, y2 = p[:, 0].astype(float), p[:, 1].astype(float), q[:, 0].astype(float), q[:, 1].astype(float)
    A = np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, np.ones(7)], 1)
    _, _, vt = np.linalg.svd(A)
    f1, f2 = vt[7].reshape(3, 3), vt[8].reshape(3, 3)
    # det(f2 + l (f1 - f2)) through four samples of the cubic (independent of the C++'s row expansion)
    ls = np.array([-1.0, 0.0, 1.0, 2.0])
    dets = [np.linalg.det(f2 + l * (f1 - f2)) for l in ls]
    c = np.linalg.solve(np.vander(ls, 4), dets)
    out = []
    for l in solve_cubic(c):
        F = f2 + l * (f1 - f2)
        out.append(F / F[2, 2] if abs(F[2, 2]) > np.finfo(float).eps else F / np.linalg.norm(F))
    return out


def inliers(F, p, q, thr2):
    x1, y1, x2, y2 = p[:, 0].astype(float), p[:, 1].astype(float), q[:, 0].astype(float), q[:, 1].astype(float)
    a, b, c = F[0, 0] * x1 + F[0, 1] * y1 + F[0, 2], F[1, 0] * x1 + F[1, 1] * y1 + F[1, 2], F[2, 0] * x1 + F[2, 1] * y1 + F[2, 2]
    e2 = (x2 * a + y2 * b + c) ** 2 / (a * a + b * b)
    a, b, c = F[0, 0] * x2 + F[1, 0] * y2 + F[2, 0], F[0, 1] * x2 + F[1, 1] * y2 + F[2, 1], F[0, 2] * x2 + F[1, 2] * y2 + F[2, 2]
    e1 = (x1 * a + y1 * b + c) ** 2 / (a * a + b * b)
    return np.maximum(e1, e2).astype(np.float32) <= thr2


def collinear(m, count):
    i = count - 1
    for j in range(i):
        dx1, dy1 = float(m[j][0]) - float(m[i][0]), float(m[j][1]) - float(m[i][1])
        for k in range(j):
            dx2, dy2 = float(m[k][0]) - float(m[i][0]), float(m[k][1]) - float(m[i][1])
            if abs(dx2 * dy1 - dy2 * dx1) <= np.finfo(np.float32).eps * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                return True
    return False


def update_iters(p, ep, m, max_iters):
    p, ep = min(max(p, 0.0), 1.0), min(max(ep, 0.0), 1.0)
    num, den = max(1.0 - p, np.finfo(float).tiny), 1.0 - (1.0 - ep) ** m
    if den < np.finfo(float).tiny:
        return 0
    num, den = np.log(num), np.log(den)
    return max_iters if den >= 0 or -num >= max_iters * (-den) else int(np.rint(num / den))


def ransac(p, q, threshold=1.0, confidence=0.99, max_iters=1000):
    n = len(p)
    mask = np.zeros(n, bool)
    if n < 7:
        return mask, None
    rng, thr2 = Mwc(), threshold * threshold
    niters, best, bestF = (1 if n == 7 else max_iters), 0, None
    it = 0
    while it < niters:
        if n > 7:
            attempts, ok = 0, False
            while attempts < 10000:
                idx, i = [0] * 7, 0
                while i < 7 and attempts < 10000:
                    c = rng.uniform(0, n)
                    idx[i] = c
                    if c in idx[:i]:
                        continue
                    i += 1
                if i == 7 and (collinear(p[idx], 7) or collinear(q[idx], 7)):
                    attempts += 1
                    continue
                ok = i == 7
                break
            if not ok:
                break
            sp, sq = p[idx], q[idx]
        else:
            sp, sq = p, q
        for F in seven_point(sp, sq):
            m = inliers(F, p, q, thr2)
            good = int(m.sum())
            if good > max(best, 6):
                best, mask, bestF = good, m, F
                niters = update_iters(confidence, (n - good) / n, 7, niters)
        it += 1
    return mask, bestF
