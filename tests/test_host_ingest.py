"""Data formats in front of the hot path: undistortion maps, the device remap, image decode, the
EuRoC / TUM-VI sequence readers and the TUM trajectory writer (pvio_amd/host/{undistort_maps,image_io,dataset_reader}.*),
against the independent numpy restatement in oracle/oracle_undistort.py.  CPU tests run the same kernel sources through the
fiber emulator; the `gpu` tests run the product library."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import host_compare
from oracle import oracle_py
from oracle import oracle_undistort as U

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "hipemu")
u8p, i16p, u16p, f32p, f64p, i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_int16, C.c_uint16, C.c_float, C.c_double, C.c_int32))

EUROC_K = [458.654, 0, 367.215, 0, 457.296, 248.375, 0, 0, 1]             # euroc_dataset_reader.cpp:74
EUROC_D = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]          # :73
TUM_K = [190.97847715128717, 0, 254.93170605935475, 0, 190.9733070521226, 256.8974428996504, 0, 0, 1]  # tum_dataset_reader.cpp:75-77
TUM_D = [0.0034003170790442797, 0.001766278153469831, -0.00266312569781606, 0.0003299517423931039]   # :78


def _p(a, t):
    return a.ctypes.data_as(t)


@pytest.fixture(scope="module")
def host():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    return host_compare.load("libpvio_host_emu.so")


def _host_cv_maps(lib, K, D, w, h):
    xy, fr = np.zeros((h, w, 2), np.int16), np.zeros((h, w), np.uint16)
    K32, D32 = np.asarray(K, np.float32), np.asarray(D, np.float32)
    assert lib.host_cv_undistort_maps(_p(K32, f32p), _p(D32, f32p), C.c_int(len(D)), C.c_int(w), C.c_int(h), _p(xy, i16p), _p(fr, u16p)) == 0
    return xy, fr


def _host_iu_maps(lib, w, h, K, D, model, probes=None):
    xy, fr = np.zeros((h, w, 2), np.int16), np.zeros((h, w), np.uint16)
    Kd, Dd = np.asarray(K, np.float64), np.asarray(D, np.float64)
    pr = np.zeros((0, 2)) if probes is None else np.ascontiguousarray(probes, np.float64)
    out = np.zeros_like(pr)
    rc = lib.host_image_undistorter_maps(C.c_int(w), C.c_int(h), _p(Kd, f64p), _p(Dd, f64p), C.c_int(len(D)), model.encode(), _p(xy, i16p), _p(fr, u16p),
                                         C.c_int(len(pr)), _p(pr, f64p), _p(out, f64p))
    return rc, xy, fr, out


def test_euroc_undistort_maps_match_restatement(host):
    """cv::undistort's fixed-point map for the EuRoC camera constants, full 752 x 480, entry by entry."""
    xy, fr = _host_cv_maps(host, EUROC_K, EUROC_D, 752, 480)
    xy0, fr0 = U.cv_undistort_fixed_maps(EUROC_K, EUROC_D, 752, 480)
    assert (xy == xy0).all() and (fr == fr0).all()
    # sanity of the restatement itself: the principal point maps to itself, the map is a smooth outward warp
    cx, cy = 367, 248
    assert abs(int(xy[cy, cx, 0]) - cx) <= 1 and abs(int(xy[cy, cx, 1]) - cy) <= 1
    assert xy[0, 0, 0] > 0 and xy[0, 0, 1] > 0 and xy[-1, -1, 0] < 751 and xy[-1, -1, 1] < 479  # barrel distortion: corners pull inwards
    # odd sizes / a stripe height that does not divide the image
    xy, fr = _host_cv_maps(host, EUROC_K, EUROC_D + [0.01], 331, 77)
    xy0, fr0 = U.cv_undistort_fixed_maps(EUROC_K, EUROC_D + [0.01], 331, 77)
    assert (xy == xy0).all() and (fr == fr0).all()


@pytest.mark.parametrize("model,K,D,size", [("equidistant", TUM_K, TUM_D, (512, 512)),
                                            ("radtan", EUROC_K, EUROC_D + [0.0], (376, 240)),
                                            ("radtan", EUROC_K, EUROC_D, (100, 60))])
def test_image_undistorter_maps_match_restatement(host, model, K, D, size):
    w, h = size
    probes = np.array([[0, 0], [w - 1, h - 1], [K[2], K[5]], [w / 3, h / 5]], np.float64)
    rc, xy, fr, out = _host_iu_maps(host, w, h, K, D, model, probes)
    assert rc == 0
    xy0, fr0 = U.image_undistorter_maps(w, h, K, D, model)
    assert (xy == xy0).all() and (fr == fr0).all()
    dx, dy = U.distort_pixels(probes[:, 0].copy(), probes[:, 1].copy(), K, D, model)
    np.testing.assert_array_equal(out[:, 0], dx)
    np.testing.assert_array_equal(out[:, 1], dy)
    # the principal point is a fixed point of both models
    np.testing.assert_allclose(out[2], [K[2], K[5]], atol=1e-9)


def test_image_undistorter_rejects_unknown_model(host):
    rc, *_ = _host_iu_maps(host, 64, 64, TUM_K, TUM_D, "fov")
    assert rc != 0


def test_bilinear_table_properties():
    tab = U.bilinear_table()
    assert (tab.sum(axis=1) == 1 << 15).all() and (tab >= 0).all()
    assert list(tab[0]) == [32767, 0, 0, 1]
    fy, fx = np.divmod(np.arange(1, 1024), 32)
    np.testing.assert_array_equal(tab[1:, 0], (32 - fx) * (32 - fy) * 32)
    np.testing.assert_array_equal(tab[1:, 3], fx * fy * 32)
    # identity map = identity image; integer shift = shifted image with a zero border
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    yy, xx = np.meshgrid(np.arange(40), np.arange(50), indexing="ij")
    ident = np.stack([xx, yy], -1).astype(np.int16)
    assert (U.remap_bilinear(src, ident, np.zeros((40, 50), np.uint16)) == src).all()
    sh = U.remap_bilinear(src, ident + np.array([3, -2], np.int16), np.zeros((40, 50), np.uint16))
    assert (sh[2:, :-3] == src[:-2, 3:]).all() and (sh[:2] == 0).all() and (sh[:, -3:] == 0).all()


def _random_maps(rng, w, h, sw, sh):
    xy = np.stack([rng.integers(-3, sw + 2, (h, w)), rng.integers(-3, sh + 2, (h, w))], -1).astype(np.int16)
    fr = rng.integers(0, 1024, (h, w)).astype(np.uint16)
    fr[rng.random((h, w)) < 0.1] = 0
    return xy, fr


def _check_remap(ctx, oracle, rng, w, h, sw, sh, xy=None, fr=None, clahe=False):
    from pvio_amd.solver import HipImage, HipUndistort
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    if xy is None:
        xy, fr = _random_maps(rng, w, h, sw, sh)
    ud = HipUndistort(ctx, xy, fr)
    im = HipImage(ctx, src, clahe=clahe, undistort=ud)
    got, _ = im.level(0)
    ref = U.remap_bilinear(src, xy, fr)
    if clahe:
        ref = oracle.clahe(ref)
    assert got.shape == ref.shape == (h, w)
    assert (got == ref).all()
    # the rest of the pyramid is built from the undistorted image
    P = oracle.build_pyramid(ref)
    for l in range(1, len(P)):
        a, d = im.level(l)
        assert (a == P[l][0]).all() and (d == P[l][1]).all()
    im.release()
    ud.release()


@pytest.fixture(scope="module")
def emu_ctx():
    from pvio_amd import capi
    from pvio_amd.solver import HipContext
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    ctx = HipContext(lib=capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so")))
    yield ctx
    ctx.close()


def test_emulated_remap_matches_restatement(emu_ctx, oracle):
    rng = np.random.default_rng(11)
    _check_remap(emu_ctx, oracle, rng, 96, 64, 96, 64)              # random maps incl. positions outside the source, fraction 0
    _check_remap(emu_ctx, oracle, rng, 70, 50, 90, 61)              # map (= output) size differs from the source size
    xy, fr = U.image_undistorter_maps(128, 96, [60.0, 0, 63.5, 0, 60.0, 47.5, 0, 0, 1], TUM_D, "equidistant")
    _check_remap(emu_ctx, oracle, rng, 128, 96, 128, 96, xy, fr, clahe=True)


@pytest.mark.gpu
def test_gpu_remap_matches_restatement(oracle):
    from pvio_amd.solver import HipContext
    ctx = HipContext(device=0)
    rng = np.random.default_rng(12)
    _check_remap(ctx, oracle, rng, 300, 200, 320, 240)
    xy, fr = U.cv_undistort_fixed_maps(EUROC_K, EUROC_D, 752, 480)
    _check_remap(ctx, oracle, rng, 752, 480, 752, 480, xy, fr, clahe=True)           # the EuRoC camera
    xy, fr = U.image_undistorter_maps(512, 512, TUM_K, TUM_D, "equidistant")
    _check_remap(ctx, oracle, rng, 512, 512, 512, 512, xy, fr, clahe=True)           # the TUM-VI camera
    ctx.close()


# ---- image files ------------------------------------------------------------------------------------------------------
def _png(arr, depth=8, ctype=0, filters=None, idat_split=1):
    """Minimal PNG writer for tests: arr [h][w][channels] of uint8 / uint16, one filter type per row (cycled)."""
    h, w = arr.shape[:2]
    a = arr.reshape(h, w, -1)
    ch = a.shape[2]
    bpp = ch * depth // 8
    rows = (a.astype(">u2").tobytes() if depth == 16 else a.astype(np.uint8).tobytes())
    stride = w * bpp
    raw = bytearray()
    prev = bytearray(stride)
    for y in range(h):
        cur = bytearray(rows[y * stride:(y + 1) * stride])
        ft = (filters or [0])[y % len(filters or [0])]
        out = bytearray(stride)
        for i in range(stride):
            A = cur[i - bpp] if i >= bpp else 0
            B = prev[i]
            Cc = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = A
            elif ft == 2:
                pred = B
            elif ft == 3:
                pred = (A + B) >> 1
            else:
                p = A + B - Cc
                pa, pb, pc = abs(p - A), abs(p - B), abs(p - Cc)
                pred = A if (pa <= pb and pa <= pc) else (B if pb <= pc else Cc)
            out[i] = (cur[i] - pred) & 255
        raw.append(ft)
        raw += out
        prev = cur
    comp = zlib.compress(bytes(raw), 6)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    parts = [comp[i * len(comp) // idat_split:(i + 1) * len(comp) // idat_split] for i in range(idat_split)]
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"tEXt", b"Comment\0test")
            + b"".join(chunk(b"IDAT", p) for p in parts) + chunk(b"IEND", b""))


def _read(lib, path, cap=1 << 22):
    w, h = C.c_int(0), C.c_int(0)
    buf = np.zeros(cap, np.uint8)
    err = C.create_string_buffer(256)
    rc = lib.host_read_gray_image(str(path).encode(), C.byref(w), C.byref(h), _p(buf, u8p), C.c_int(cap), err, C.c_int(256))
    if rc != 0:
        return None, err.value.decode()
    return buf[:w.value * h.value].reshape(h.value, w.value).copy(), ""


def test_png_and_pgm_decode(host, tmp_path):
    rng = np.random.default_rng(5)
    g = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    (tmp_path / "g8.png").write_bytes(_png(g, 8, 0, filters=[0, 1, 2, 3, 4], idat_split=3))
    img, err = _read(host, tmp_path / "g8.png")
    assert err == "" and (img == g).all()
    g16 = rng.integers(0, 65536, (20, 31)).astype(np.uint16)
    (tmp_path / "g16.png").write_bytes(_png(g16, 16, 0, filters=[4, 3, 1]))
    img, err = _read(host, tmp_path / "g16.png")
    assert err == "" and (img == (g16 >> 8)).all()                      # high byte, like png_set_strip_16
    rgb = rng.integers(0, 256, (16, 24, 3), dtype=np.uint8)
    rgb[:4] = rgb[:4, :, :1]                                              # gray rows pass through unchanged
    (tmp_path / "rgb.png").write_bytes(_png(rgb, 8, 2, filters=[4, 2]))
    img, err = _read(host, tmp_path / "rgb.png")
    r, gg, b = (rgb[..., k].astype(np.int64) for k in range(3))
    assert err == "" and (img == ((9798 * r + 19235 * gg + 3735 * b + 16384) >> 15)).all() and (img[:4] == rgb[:4, :, 0]).all()
    ga = rng.integers(0, 256, (9, 11, 2), dtype=np.uint8)
    (tmp_path / "ga.png").write_bytes(_png(ga, 8, 4, filters=[3]))
    img, err = _read(host, tmp_path / "ga.png")
    assert err == "" and (img == ga[..., 0]).all()
    rgba = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    rgba[..., 1] = rgba[..., 0]
    rgba[..., 2] = rgba[..., 0]
    (tmp_path / "rgba.png").write_bytes(_png(rgba, 8, 6, filters=[1, 4]))
    img, err = _read(host, tmp_path / "rgba.png")
    assert err == "" and (img == rgba[..., 0]).all()
    (tmp_path / "p.pgm").write_bytes(b"P5\n# a comment\n53 37\n255\n" + g.tobytes())
    img, err = _read(host, tmp_path / "p.pgm")
    assert err == "" and (img == g).all()
    # failures are loud: corrupted CRC, truncated file, interlaced, missing file
    bad = bytearray(_png(g, 8, 0))
    bad[40] ^= 1
    (tmp_path / "bad.png").write_bytes(bytes(bad))
    assert _read(host, tmp_path / "bad.png")[0] is None
    (tmp_path / "trunc.png").write_bytes(_png(g, 8, 0)[:-20])
    assert _read(host, tmp_path / "trunc.png")[0] is None
    il = bytearray(_png(g, 8, 0))
    il[28] = 1                                                             # IHDR interlace byte (the CRC now fails too)
    (tmp_path / "il.png").write_bytes(bytes(il))
    assert _read(host, tmp_path / "il.png")[0] is None
    assert _read(host, tmp_path / "missing.png")[0] is None


# ---- sequences ----------------------------------------------------------------------------------------------------------
def _write_sequence(root, euroc, images, cam_t_ns, imu_rows):
    eol = "\r\n" if euroc else "\n"
    os.makedirs(root / "cam0" / "data")
    os.makedirs(root / "imu0")
    with open(root / "cam0" / "data.csv", "w", newline="") as f:
        f.write("#timestamp [ns],filename" + eol)
        for t, img in zip(cam_t_ns, images):
            name = "%d.png" % t
            f.write("%d,%s%s" % (t, name, eol))
            (root / "cam0" / "data" / name).write_bytes(_png(img, 8, 0, filters=[0, 4]))
    with open(root / "imu0" / "data.csv", "w", newline="") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_y,w_z,a_RS_S_x [m s^-2],a_y,a_z" + eol)
        for r in imu_rows:
            f.write("%d,%s%s" % (r[0], ",".join(repr(float(v)) for v in r[1:]), eol))


def _walk(lib, uri, max_events, img_cap):
    types, times, vals = np.zeros(max_events, np.int32), np.zeros(max_events), np.zeros((max_events, 3))
    imgs = np.zeros(img_cap, np.uint8)
    wh, nimg = np.zeros(2, np.int32), C.c_int32(0)
    err = C.create_string_buffer(512)
    lib.host_dataset_walk.restype = C.c_int
    n = lib.host_dataset_walk(uri.encode(), C.c_int(max_events), _p(types, i32p), _p(times, f64p), _p(vals, f64p), _p(imgs, u8p), C.c_int64(img_cap),
                              _p(wh, i32p), C.byref(nimg), err, C.c_int(512))
    assert n >= 0, err.value.decode()
    return types[:n], times[:n], vals[:n], imgs, wh, nimg.value


@pytest.mark.parametrize("kind", ["euroc", "tum"])
def test_sequence_reader_event_order_and_images(host, oracle, tmp_path, kind):
    """A synthetic sequence in the dataset's own layout: events come out merged by time (camera / gyroscope / accelerometer),
    timestamps are ns * 1e-9 of the value parsed as a double, and every image reaches the tracker undistorted with the
    camera constants of the reference reader, CLAHE'd, as level 0 of the device pyramid."""
    rng = np.random.default_rng(21)
    w, h = (188, 120) if kind == "euroc" else (128, 128)
    images = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(3)]
    cam_t = [1403636579763555584, 1403636579813555456, 1403636579863555584]
    imu = [(1403636579758555392 + 5000000 * k, *rng.normal(size=6)) for k in range(24)]
    root = tmp_path / kind
    _write_sequence(root, kind == "euroc", images, cam_t, imu)
    types, times, vals, imgs, wh, nimg = _walk(host, kind + "://" + str(root), 100, 3 * w * h)
    ev = [(float(t) * 1e-9, 1, None) for t in cam_t]
    for r in imu:
        ev.append((float(r[0]) * 1e-9, 2, r[1:4]))
        ev.append((float(r[0]) * 1e-9, 3, r[4:7]))
    ev.sort(key=lambda e: e[0])  # stable: camera, gyroscope, accelerometer at equal times
    assert len(types) == len(ev) and nimg == 3 and tuple(wh) == (w, h)
    for k, (t, ty, v) in enumerate(ev):
        assert types[k] == ty and times[k] == t
        if v is not None:
            np.testing.assert_array_equal(vals[k], np.array(v, np.float64))
    if kind == "euroc":
        xy, fr = U.cv_undistort_fixed_maps(EUROC_K, EUROC_D, w, h)
    else:
        xy, fr = U.image_undistorter_maps(w, h, TUM_K, TUM_D, "equidistant")
    for k in range(3):
        ref = oracle.clahe(U.remap_bilinear(images[k], xy, fr))
        assert (imgs[k * w * h:(k + 1) * w * h].reshape(h, w) == ref).all()


def test_unknown_scheme_and_missing_directory(host, tmp_path):
    err = C.create_string_buffer(256)
    z = np.zeros(8)
    host.host_dataset_walk.restype = C.c_int
    args = (C.c_int(4), _p(np.zeros(4, np.int32), i32p), _p(z, f64p), _p(np.zeros(12), f64p), _p(np.zeros(4, np.uint8), u8p), C.c_int64(4),
            _p(np.zeros(2, np.int32), i32p), C.byref(C.c_int32(0)), err, C.c_int(256))
    assert host.host_dataset_walk(b"sensors:///nowhere", *args) == -1           # not provided
    assert host.host_dataset_walk(("euroc://" + str(tmp_path / "empty")).encode(), *args) == 0  # like the reference: an empty sequence


def test_tum_writer_format(host, tmp_path):
    t = np.array([1403636579.7635555, 0.1, 12345.678901234567])
    p = np.array([[1.0, -2.5, 1e-9], [0.1, 0.2, 0.3], [123456.789012345678, -1e20, 0.0]])
    q = np.array([[0, 0, 0, 1.0], [0.5, -0.5, 0.5, 0.5], [1 / 3, 2 / 3, 0.0, 2 / 3]])
    path = tmp_path / "trajectory.tum"
    assert host.host_tum_write(str(path).encode(), C.c_int(3), _p(t, f64p), _p(p, f64p), _p(q, f64p)) == 0
    lines = path.read_text().split("\n")
    assert lines[-1] == "" and len(lines) == 4
    for k in range(3):
        want = " ".join("%.15g" % v for v in [t[k], *p[k], *q[k]])   # ostream << double with precision(15), default float format
        assert lines[k] == want
    assert host.host_tum_write(str(tmp_path / "no" / "dir" / "x.tum").encode(), C.c_int(0), _p(t, f64p), _p(p, f64p), _p(q, f64p)) == -1


# ---- end to end: reader -> device undistortion / pyramid -> LK + RANSAC -> detection, frame after frame ---------------------
def _moving_sequence(n_frames, w, h, shift):
    """Frames cut out of one band-limited texture, the window moving by `shift` pixels per frame (a pure image translation)."""
    from pvio_amd import synth
    big, *_ = synth.make_image_pair(w + 64, h + 64, 8)
    return [np.ascontiguousarray(big[16 + k * shift[1]:16 + k * shift[1] + h, 16 + k * shift[0]:16 + k * shift[0] + w]) for k in range(n_frames)]


def _replay(lib, uri, max_frames, distance):
    per = np.zeros((max_frames, 3), np.int32)
    ms, flow = np.zeros(max_frames), np.zeros((max_frames, 2))
    err = C.create_string_buffer(512)
    lib.host_replay_front_end.restype = C.c_int
    n = lib.host_replay_front_end(uri.encode(), C.c_int(max_frames), C.c_double(distance), _p(per, i32p), _p(ms, f64p), _p(flow, f64p), err, C.c_int(512))
    assert n >= 0, err.value.decode()
    return n, per[:n], ms[:n], flow[:n]


def _check_replay(lib, tmp_path, w, h, n_frames, shift):
    frames = _moving_sequence(n_frames, w, h, shift)
    t0 = 1520530308199447626
    root = tmp_path / "seq"
    _write_sequence(root, False, frames, [t0 + 50000000 * k for k in range(n_frames)], [(t0 - 1000000 + 5000000 * k, 0, 0, 0, 0, 0, 9.8) for k in range(10 * n_frames)])
    n, per, ms, flow = _replay(lib, "tum://" + str(root), n_frames, 20.0)
    assert n == n_frames
    assert per[0, 0] == 0 and per[0, 2] >= 10                       # the first frame only detects
    for k in range(1, n):
        assert per[k, 0] == per[k - 1, 2]                            # everything known goes into the tracker
        assert per[k, 1] >= 0.6 * per[k, 0]                          # most of it survives LK, the border gate and RANSAC
        assert per[k, 2] >= per[k, 1]                                # detection only adds
    return per, ms, flow


def test_emulated_front_end_replay(host, tmp_path):
    per, ms, flow = _check_replay(host, tmp_path, 320, 320, 3, (2, 1))  # large enough to hold the TUM-VI principal point
    assert per[-1, 1] > 0


@pytest.mark.gpu
def test_gpu_front_end_replay(tmp_path):
    """A TUM-VI-sized sequence through the whole front end on the device: the window moves by (-3, -2) px per frame in image
    coordinates of the undistorted centre region; reports the per-frame time."""
    lib = host_compare.load("libpvio_host.so")
    per, ms, flow = _check_replay(lib, tmp_path, 512, 512, 8, (3, 2))
    # the scene content moves against the window by (-3, -2) px per frame of the DISTORTED image; undistorting the fisheye
    # model magnifies the image centre (by about 1.5 for the TUM-VI constants), so the measured flow is that vector scaled:
    # same direction, the same from frame to frame
    assert np.all(flow[1:, 0] < -3) and np.all(flow[1:, 0] > -6) and np.all(flow[1:, 1] < -2) and np.all(flow[1:, 1] > -4), flow
    assert np.all(np.abs(flow[1:, 0] / flow[1:, 1] - 1.5) < 0.1) and np.ptp(flow[1:, 0]) < 0.3, flow
    print("front end replay 512x512: tracks", per[:, 2].tolist(), "ms/frame (decode + upload + undistort + pyramid + LK + RANSAC + detect)", np.round(ms, 2).tolist())
