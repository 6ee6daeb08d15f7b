"""CPU restatement (numpy) of the image-undistortion step in front of the feature tracker.  TEST INFRASTRUCTURE ONLY: imported
by tests/ (and nothing else); the product path (pvio_amd/host/undistort_maps.cpp + k_remap in pvio_amd/csrc/klt.hip) never
calls it.

Follows, line by line where the code is in /root/reference and from the published algorithm where it is OpenCV's:
  cv_undistort_fixed_maps   cv::undistort(img, K, dist) of pvio-pc/src/euroc_dataset_reader.cpp:72-75 =
                            initUndistortRectifyMap(K, dist, I, K', stripe size, CV_16SC2) per row stripe (K' = K with the
                            principal point shifted by the stripe origin) -- OpenCV (unpinned, not in the tree): PARITY UNPINNED
  image_undistorter_maps    pvio-extra/include/pvio/extra/image_undistorter.h:27-42,48-96 (distort_pixel in double,
                            float32 maps) + cv::convertMaps(CV_16SC2)
  remap_bilinear            cv::remap(INTER_LINEAR, BORDER_CONSTANT, 0) on CV_16SC2 + CV_16UC1 maps (image_undistorter.h:44-46):
                            5-bit fractions, 15-bit weights from the INTER_LINEAR table, (sum + 2^14) >> 15, outside taps = 0
This file is written independently of the C++ host code (vectorized, different evaluation structure) so that agreement of
the two is a check of both.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB = 1 << INTER_BITS


def _invert3_cofactor(S):
    s = S
    d = s[0, 0] * (s[1, 1] * s[2, 2] - s[1, 2] * s[2, 1]) - s[0, 1] * (s[1, 0] * s[2, 2] - s[1, 2] * s[2, 0]) + s[0, 2] * (s[1, 0] * s[2, 1] - s[1, 1] * s[2, 0])
    d = 1.0 / d
    return np.array([
        (s[1, 1] * s[2, 2] - s[1, 2] * s[2, 1]) * d, (s[0, 2] * s[2, 1] - s[0, 1] * s[2, 2]) * d, (s[0, 1] * s[1, 2] - s[0, 2] * s[1, 1]) * d,
        (s[1, 2] * s[2, 0] - s[1, 0] * s[2, 2]) * d, (s[0, 0] * s[2, 2] - s[0, 2] * s[2, 0]) * d, (s[0, 2] * s[1, 0] - s[0, 0] * s[1, 2]) * d,
        (s[1, 0] * s[2, 1] - s[1, 1] * s[2, 0]) * d, (s[0, 1] * s[2, 0] - s[0, 0] * s[2, 1]) * d, (s[0, 0] * s[1, 1] - s[0, 1] * s[1, 0]) * d])


def _fixed(u, v):
    """double positions -> (int16 xy, uint16 frac): cvRound(u * 32) split into integer part and 5-bit fraction."""
    iu = np.rint(u * INTER_TAB).astype(np.int64)
    iv = np.rint(v * INTER_TAB).astype(np.int64)
    xy = np.stack([(iu >> INTER_BITS), (iv >> INTER_BITS)], axis=-1)
    frac = ((iv & (INTER_TAB - 1)) * INTER_TAB + (iu & (INTER_TAB - 1))).astype(np.uint16)
    return xy, frac


def cv_undistort_fixed_maps(K32, dist32, width, height):
    A = np.asarray(K32, np.float32).reshape(3, 3).astype(np.float64)
    dc = np.asarray(dist32, np.float32).astype(np.float64)
    k1, k2, p1, p2 = dc[:4]
    k3 = dc[4] if dc.size > 4 else 0.0
    fx, fy, u0, v0 = A[0, 0], A[1, 1], A[0, 2], A[1, 2]
    stripe0 = min(max(1, (1 << 12) // max(width, 1)), height)
    xy = np.zeros((height, width, 2), np.int16)
    frac = np.zeros((height, width), np.uint16)
    for y0 in range(0, height, stripe0):
        n = min(stripe0, height - y0)
        Ar = A.copy()
        Ar[1, 2] = v0 - y0
        ir = _invert3_cofactor(Ar)
        i = np.arange(n, dtype=np.float64)[:, None]
        # x_{j+1} = x_j + ir[0]: a sequential accumulation (np.add.accumulate adds left to right)
        def run(first, inc):
            steps = np.full((n, width), inc)
            steps[:, 0:1] = first
            return np.add.accumulate(steps, axis=1)
        _x, _y, _w = run(i * ir[1] + ir[2], ir[0]), run(i * ir[4] + ir[5], ir[3]), run(i * ir[7] + ir[8], ir[6])
        w = 1.0 / _w
        x, y = _x * w, _y * w
        x2, y2 = x * x, y * y
        r2, _2xy = x2 + y2, 2 * x * y
        kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2)
        xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + 0.0 * r2 + 0.0 * r2 * r2
        yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + 0.0 * r2 + 0.0 * r2 * r2
        u = fx * 1.0 * xd + u0
        v = fy * 1.0 * yd + v0
        a, b = _fixed(u, v)
        xy[y0:y0 + n] = a.astype(np.int16)  # plain (short) cast in initUndistortRectifyMap
        frac[y0:y0 + n] = b
    return xy, frac


def _eigen_inverse3(m):
    def cof(i, j):
        i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
        return m[i1, j1] * m[i2, j2] - m[i1, j2] * m[i2, j1]
    det = cof(0, 0) * m[0, 0] + (cof(1, 0) * m[1, 0] + cof(2, 0) * m[2, 0])
    inv = 1.0 / det
    r = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            r[j, i] = cof(i, j) * inv
    return r


def distort_pixels(u, v, K, dist, model):
    """image_undistorter.h:48-96 for arrays of destination pixel coordinates (double)."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    Ki = _eigen_inverse3(K)
    D = [float(c) for c in dist]
    one = np.ones_like(u)
    mv = lambda M, a, b, c: [M[r, 0] * a + (M[r, 1] * b + M[r, 2] * c) for r in range(3)]
    x, y, z = mv(Ki, u, v, one)
    if model == "radtan":
        k1, k2, p1, p2 = D[:4]
        k3 = D[4] if len(D) > 4 else 0.0
        r2 = x * x + y * y
        r4 = r2 * r2
        r6 = r4 * r2
        kr = 1.0 + k1 * r2 + k2 * r4 + k3 * r6
        xd = x * kr + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        yd = y * kr + 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y)
        ox, oy, oz = mv(K, xd, yd, z)
        return ox / oz, oy / oz
    if model == "equidistant":
        k1, k2, k3, k4 = D[:4]
        r = np.sqrt(x * x + y * y)
        theta = np.arctan(r)
        t2 = theta * theta
        t4 = t2 * t2
        t6 = t2 * t4
        t8 = t4 * t4
        thetad = theta * (1 + k1 * t2 + k2 * t4 + k3 * t6 + k4 * t8)
        with np.errstate(divide="ignore", invalid="ignore"):
            scaling = np.where(r > 1e-8, thetad / r, 1.0)
        ox, oy, oz = mv(K, x * scaling, y * scaling, z)
        return np.where(r < 1e-10, u, ox / oz), np.where(r < 1e-10, v, oy / oz)
    raise ValueError("unknown model: " + model)


def image_undistorter_maps(width, height, K, dist, model):
    v, u = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    dx, dy = distort_pixels(u, v, K, dist, model)
    mx, my = dx.astype(np.float32), dy.astype(np.float32)
    # cv::convertMaps(CV_32FC1 x2 -> CV_16SC2 + CV_16UC1): float multiply, cvRound, saturate_cast<short>
    ix = np.rint(mx * np.float32(INTER_TAB)).astype(np.int64)
    iy = np.rint(my * np.float32(INTER_TAB)).astype(np.int64)
    xy = np.clip(np.stack([ix >> INTER_BITS, iy >> INTER_BITS], axis=-1), -32768, 32767).astype(np.int16)
    frac = ((iy & (INTER_TAB - 1)) * INTER_TAB + (ix & (INTER_TAB - 1))).astype(np.uint16)
    return xy, frac


def bilinear_table():
    """OpenCV's INTER_LINEAR fixed-point table: tab[fy*32+fx] = 2x2 int weights summing to 1 << 15.  Built the way
    initInterTab2D does: float weights, saturate_cast<short>(w * 32768), then the correction that restores the sum (only the
    (0, 0) entry needs it: 32768 saturates to 32767 and the missing 1 goes to the [1][1] slot)."""
    tab = np.zeros((INTER_TAB * INTER_TAB, 4), np.int64)
    for fy in range(INTER_TAB):
        for fx in range(INTER_TAB):
            ax, ay = np.float32(fx) / np.float32(INTER_TAB), np.float32(fy) / np.float32(INTER_TAB)
            wx, wy = [np.float32(1) - ax, ax], [np.float32(1) - ay, ay]
            w = [int(min(max(int(np.rint(np.float32(wy[a] * wx[b]) * np.float32(32768))), -32768), 32767)) for a in range(2) for b in range(2)]
            if sum(w) != 32768:
                w[3] -= sum(w) - 32768
            tab[fy * INTER_TAB + fx] = w
    return tab


def remap_bilinear(src, xy, frac):
    src = np.asarray(src, np.uint8)
    sh, sw = src.shape
    tab = bilinear_table()
    sx, sy = xy[..., 0].astype(np.int64), xy[..., 1].astype(np.int64)
    w = tab[frac.astype(np.int64) & (INTER_TAB * INTER_TAB - 1)]
    pad = np.zeros((sh + 2, sw + 2), np.int64)  # constant border 0 around the source; everything further out is 0 too
    pad[1:-1, 1:-1] = src

    def tap(yy, xx):
        inside = (xx >= -1) & (xx <= sw) & (yy >= -1) & (yy <= sh)
        return np.where(inside, pad[np.clip(yy + 1, 0, sh + 1), np.clip(xx + 1, 0, sw + 1)], 0)
    acc = tap(sy, sx) * w[..., 0] + tap(sy, sx + 1) * w[..., 1] + tap(sy + 1, sx) * w[..., 2] + tap(sy + 1, sx + 1) * w[..., 3]
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)
