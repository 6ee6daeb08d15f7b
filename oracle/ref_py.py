"""ctypes loader for oracle/_ref/libpvio_ref.so: the REFERENCE'S OWN sources (bundle_adjustor.cpp, the estimation/ceres cost
functions, preintegrator.cpp, lie_algebra.cpp, map/*.cpp, pnp.cpp ...) compiled unedited by oracle/ref/Makefile against the
mini-Eigen / mini-Ceres stand-ins of oracle/ref/.

TEST INFRASTRUCTURE ONLY (tests/, never the product).  The library is built where /root/reference exists (this container) and
travels prebuilt to the GPU box; `available()` is False when neither is there and the tests that need it skip.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from pvio_amd import capi  # noqa: E402  (struct layouts only)

LIB = os.path.join(HERE, "_ref", "libpvio_ref.so")
REF = os.environ.get("PVIO_REFERENCE", "/root/reference")
dp = capi.c_double_p
u8p, i32p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class RefTracksC(C.Structure):
    _fields_ = [("n_tracks", C.c_int32), ("n_planes", C.c_int32), ("obs_ptr", i32p), ("obs_frame", i32p), ("obs_z", dp),
                ("inv_depth", dp), ("valid", u8p), ("plane", u8p), ("life", i64p), ("best_plane", i32p), ("quality", dp),
                ("plane_normal", dp), ("plane_distance", dp), ("membership", u8p), ("pad_small_planes", C.c_int32), ("reserved", C.c_int32),
                ("keep_small", u8p)]


class RefImuC(C.Structure):
    _fields_ = [("frame_t", dp), ("ptr", i32p), ("t", dp), ("w", dp), ("a", dp), ("noise", C.POINTER(capi.ImuNoiseC))]


def build():
    """(Re)builds the library when the reference tree is present; a prebuilt library is used as it is otherwise."""
    if os.path.isdir(os.path.join(REF, "pvio", "src")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "ref"), "REF=" + REF])
    return LIB if os.path.exists(LIB) else None


def available():
    return build() is not None


_lib = None


def lib():
    global _lib
    if _lib is None:
        if build() is None:
            raise RuntimeError("oracle/_ref/libpvio_ref.so is not built and %s is absent" % REF)
        L = _lib = C.CDLL(LIB)
        L.ref_version.restype = C.c_char_p
        for n in ("ref_expmap", "ref_logmap", "ref_right_jacobian"):
            getattr(L, n).argtypes = [dp, dp]
            getattr(L, n).restype = None
        L.ref_plus.argtypes = [dp, dp, dp]
        L.ref_plus.restype = None
        L.ref_preintegrate.argtypes = [C.c_int32, dp, dp, dp, C.c_double, dp, dp, C.POINTER(capi.ImuNoiseC), dp, dp, dp, dp]
        L.ref_eval_reprojection.argtypes = [dp, dp, C.c_double, dp, dp, dp, dp, dp, dp, dp]
        L.ref_eval_reprojection.restype = None
        L.ref_eval_preintegration.argtypes = [dp] * 10
        L.ref_eval_preintegration.restype = None
        L.ref_eval_prior.argtypes = [C.c_int32, dp, dp, dp, dp, dp, dp]
        L.ref_eval_prior.restype = None
        L.ref_eval_plane.argtypes = [C.c_int32, dp, dp, dp, dp, C.c_double, C.c_double, dp, dp]
        L.ref_eval_plane.restype = None
        L.ref_fault_injection.argtypes = [C.c_int32, C.c_int32]
        L.ref_fault_injection.restype = None
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(dp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def noise_c(nd):
    nz = capi.ImuNoiseC()
    for k in ("cov_w", "cov_a", "cov_bg", "cov_ba"):
        getattr(nz, k)[:] = list(np.asarray(nd[k], float).ravel())
    return nz


def preintegrate(t, w, a, t_end, bg, ba, noise):
    """PreIntegrator::integrate(t_end, bg, ba, true, true); same return as oracle_py.preintegrate"""
    t, w, a, bg, ba = f64(t), f64(w), f64(a), f64(bg), f64(ba)
    delta, cov, U, jac = np.zeros(11), np.zeros(225), np.zeros(225), np.zeros(45)
    nz = noise_c(noise)
    rc = lib().ref_preintegrate(len(t), _d(t), _d(w), _d(a), float(t_end), _d(bg), _d(ba), C.byref(nz), _d(delta), _d(cov), _d(U), _d(jac))
    assert rc == 0
    return delta, cov, U, jac


class Tracks:
    """The track table of a window (all observations of every track, anchor first)."""

    def __init__(self, ptr, frame, z, inv_depth, valid, plane, life=None, best_plane=None, normal=None, distance=None, membership=None,
                 pad_small_planes=0, keep_small=None):
        T = len(ptr) - 1
        self.ptr = np.ascontiguousarray(ptr, np.int32)
        self.frame = np.ascontiguousarray(frame, np.int32)
        self.z = f64(z).reshape(-1, 2)
        self.inv_depth = f64(inv_depth).copy()
        self.valid = np.ascontiguousarray(valid, np.uint8).copy()
        self.plane = np.ascontiguousarray(plane, np.uint8).copy()
        self.life = np.ascontiguousarray(life if life is not None else np.diff(self.ptr), np.int64)
        self.best_plane = np.ascontiguousarray(best_plane if best_plane is not None else np.full(T, -1), np.int32)
        self.quality = np.zeros(T)
        self.normal = f64(normal if normal is not None else np.zeros((0, 3))).reshape(-1, 3)
        self.distance = f64(distance if distance is not None else np.zeros(0))
        P = len(self.distance)
        self.membership = np.ascontiguousarray(membership if membership is not None else np.zeros((P, T)), np.uint8).reshape(P, T).copy()
        self.pad_small_planes = int(pad_small_planes)
        self.keep_small = np.ascontiguousarray(keep_small if keep_small is not None else np.zeros(P), np.uint8)

    def as_c(self):
        c = RefTracksC()
        c.n_tracks, c.n_planes = len(self.ptr) - 1, len(self.distance)
        c.obs_ptr, c.obs_frame, c.obs_z = self.ptr.ctypes.data_as(i32p), self.frame.ctypes.data_as(i32p), _d(self.z)
        c.inv_depth, c.valid, c.plane = _d(self.inv_depth), self.valid.ctypes.data_as(u8p), self.plane.ctypes.data_as(u8p)
        c.life, c.best_plane, c.quality = self.life.ctypes.data_as(i64p), self.best_plane.ctypes.data_as(i32p), _d(self.quality)
        c.plane_normal, c.plane_distance = _d(self.normal), _d(self.distance)
        c.membership = self.membership.ctypes.data_as(u8p)
        c.pad_small_planes = self.pad_small_planes
        c.keep_small = self.keep_small.ctypes.data_as(u8p)
        return c


def tracks_of_problem(pb, inv_depth=None):
    """The flat pvio_ba_problem as a track table: landmarks -> VALID tracks (anchor + observations), plane factors -> PLANE tracks of
    planes grouped by (normal, distance) and padded to 20 members (the reference only builds AugmentedPlaneDistanceErrorCost for planes
    with >= 20 tracks, bundle_adjustor.cpp:180).  Plane tracks get life 0: the re-validation pass (:251-275) belongs to the adapter
    level (oracle_post.cpp), the flat solve does not run it.  Returns (Tracks, index of the landmarks in the table)."""
    pb._canon()
    M, Pn = pb.n_landmarks, pb.n_plane_factors
    ptr, frame, z = [0], [], []
    for l in range(M):
        b, e = pb.lm_obs_ptr[l], pb.lm_obs_ptr[l + 1]
        frame.append(pb.lm_anchor_frame[l])
        z.append(pb.lm_anchor_z[l])
        assert np.all(pb.obs_frame[b:e] > pb.lm_anchor_frame[l]), "the anchor must be the first observation"
        frame.extend(pb.obs_frame[b:e])
        z.extend(pb.obs_z[b:e])
        ptr.append(len(frame))
    planes = {}
    best = [-1] * M
    for k in range(Pn):
        b, e = pb.plane_obs_ptr[k], pb.plane_obs_ptr[k + 1]
        frame.extend(pb.plane_obs_frame[b:e])
        z.extend(pb.plane_obs_z[b:e])
        ptr.append(len(frame))
        key = (tuple(pb.plane_normal[k]), float(pb.plane_distance[k]))
        best.append(planes.setdefault(key, len(planes)))
    T = M + Pn
    rho = np.concatenate([pb.lm_inv_depth if inv_depth is None else inv_depth, np.full(Pn, 0.5)])
    valid = np.concatenate([np.ones(M, np.uint8), np.zeros(Pn, np.uint8)])
    plane = 1 - valid
    life = np.concatenate([np.diff(np.asarray(ptr))[:M], np.zeros(Pn)]).astype(np.int64)
    keys = list(planes.keys())
    normals, dists = [k[0] for k in keys], [k[1] for k in keys]
    rows = [np.zeros(T, np.uint8) for _ in keys]
    for k in range(Pn):
        rows[best[M + k]][M + k] = 1
    keep_small = [0] * len(keys)
    # lm_multiplicity (duplicate residual blocks, bundle_adjustor.cpp:165-179): a landmark listed m times is a VALID non-PLANE track
    # that sits in m - 1 planes of fewer than 20 tracks -- groups of at most 19 such tracks per plane; where the plane lies does not
    # enter the solve (a small plane has no plane-distance factor; its tracks are VALID, so the re-validation pass skips them)
    mult = getattr(pb, "lm_multiplicity", None)
    if mult is not None:
        for level in range(2, int(np.max(mult)) + 1 if M else 2):
            idx = np.nonzero(np.asarray(mult) >= level)[0]
            for g in range(0, len(idx), 19):
                row = np.zeros(T, np.uint8)
                row[idx[g:g + 19]] = 1
                rows.append(row), normals.append((0.0, 0.0, 1.0)), dists.append(100.0 + len(rows)), keep_small.append(1)
    membership = np.stack(rows) if rows else np.zeros((0, T), np.uint8)
    t = Tracks(ptr, frame, np.array(z).reshape(-1, 2), rho, valid, plane, life, best, np.array(normals).reshape(-1, 3),
               np.array(dists), membership, pad_small_planes=20, keep_small=keep_small)
    return t, np.arange(M)


def imu_of_problem(pb, kf_dt=0.25):
    """Raw IMU samples of a synthetic window (pb.meta["imu"], pvio_amd/synth.py) as a ref_imu; the owner list keeps the arrays alive."""
    N = pb.n_frames
    ptr, ts, ws, as_ = [0, 0], [], [], []
    frame_t = np.zeros(N)
    for j, (t, w, a, t_end) in enumerate(pb.meta["imu"], start=1):
        ts.append(t), ws.append(w), as_.append(a)
        ptr.append(ptr[-1] + len(t))
        frame_t[j] = t_end
    keep = dict(frame_t=f64(frame_t), ptr=np.ascontiguousarray(ptr, np.int32), t=f64(np.concatenate(ts)), w=f64(np.concatenate(ws)),
                a=f64(np.concatenate(as_)), noise=noise_c(pb.meta["imu_noise"]))
    c = RefImuC()
    c.frame_t, c.ptr, c.t, c.w, c.a = _d(keep["frame_t"]), keep["ptr"].ctypes.data_as(i32p), _d(keep["t"]), _d(keep["w"]), _d(keep["a"])
    c.noise = C.pointer(keep["noise"])
    return c, keep


class Harness:
    """The window-level entry points of one of the two libraries oracle/ref/Makefile builds on the reference's REAL object graph:
    libpvio_ref.so (prefix ref_: the reference's own BundleAdjustor / visual_inertial_pnp) or libpvio_dropin[_emu].so (prefix dropin_: the
    product's pvio_amd/host adapter linked in their place above the HIP C ABI / the kernel emulator).  Same flat layout on both sides."""

    def __init__(self, L, prefix):
        self.L, self.prefix = L, prefix
        f = lambda n: getattr(L, prefix + n)  # noqa: E731
        f("ba_solve").argtypes = [C.POINTER(capi.BAProblemC), dp, C.POINTER(RefTracksC), C.POINTER(RefImuC), C.POINTER(capi.BASummaryC)]
        f("ba_marginalize").argtypes = [C.POINTER(capi.BAProblemC), dp, C.POINTER(RefTracksC), C.POINTER(RefImuC), C.c_int32, C.POINTER(capi.BAPriorC)]
        f("ba_reprojection_error").argtypes = [C.POINTER(capi.BAProblemC), dp, C.POINTER(RefTracksC), dp]
        f("marginalize_then_solve").argtypes = [C.POINTER(capi.BAProblemC), dp, C.POINTER(RefTracksC), C.POINTER(RefImuC), C.c_int32, dp, i32p]
        f("pnp").argtypes = [C.POINTER(capi.BAProblemC), dp, C.POINTER(RefTracksC), dp, dp, dp, dp, dp, C.c_int32, i32p, dp, dp, dp, dp, C.c_int32, i32p]
        self._f = f

    def solve(self, pb, tracks=None, use_raw_imu=True, trace=True):
        """BundleAdjustor::solve on the window `pb` describes.  Returns (frame_state, Tracks, summary)."""
        from pvio_amd.problem import BASummary
        if tracks is None:
            tracks, _ = tracks_of_problem(pb)
        pbc = pb.as_c()
        fs = f64(pb.frame_state).copy()
        tc = tracks.as_c()
        imu_c, keep = (None, None)
        if pb.use_inertial:
            assert use_raw_imu and "imu" in pb.meta, "solve() re-integrates the raw IMU samples (bundle_adjustor.cpp:224)"
            imu_c, keep = imu_of_problem(pb)
        T = len(tracks.ptr) - 1

        class _Shape:  # BASummary sizes its trace buffer from these
            max_iterations = pb.max_iterations

            @staticmethod
            def state_dim():
                return pb.n_frames * 16 + T
        sm = BASummary(_Shape, trace=trace)
        rc = self._f("ba_solve")(C.byref(pbc), _d(fs), C.byref(tc), C.byref(imu_c) if imu_c is not None else None, C.byref(sm.c))
        assert rc == 0, rc
        return fs, tracks, sm

    def marginalize(self, pb, frame_state, tracks, victim):
        """BundleAdjustor::marginalize_frame; the pre-integration blocks are the ones in `pb` (what the last solve integrated)."""
        pbc = pb.as_c()
        fs = f64(frame_state)
        tc = tracks.as_c()
        n = pb.n_frames - 1
        S, s = np.zeros((15 * n, 15 * n)), np.zeros(15 * n)
        IM, iv = np.zeros((15 * n, 15 * n)), np.zeros(15 * n)
        pr = capi.BAPriorC()
        pr.S, pr.s, pr.info_matrix, pr.info_vector = _d(S), _d(s), _d(IM), _d(iv)
        rc = self._f("ba_marginalize")(C.byref(pbc), _d(fs), C.byref(tc), None, int(victim), C.byref(pr))
        assert rc == 0, rc
        return S, s, IM, iv

    def marginalize_then_solve(self, pb, frame_state, tracks, victim):
        """The keyframe cycle on ONE Map: Map::marginalize_frame(victim) (map.cpp:73-88) then BundleAdjustor::solve of the N - 1 frames left,
        the new prior included (sliding_window_tracker.cpp:91-113).  Returns (states [(N-1)][16], usable); `tracks` is updated in place."""
        pbc = pb.as_c()
        fs = f64(frame_state)
        tc = tracks.as_c()
        imu_c, keep = (None, None)
        if pb.use_inertial:
            imu_c, keep = imu_of_problem(pb)
        out = np.zeros((pb.n_frames - 1, 16))
        usable = C.c_int32(0)
        rc = self._f("marginalize_then_solve")(C.byref(pbc), _d(fs), C.byref(tc), C.byref(imu_c) if imu_c is not None else None, int(victim), _d(out), C.byref(usable))
        assert rc == 0, rc
        return out, bool(usable.value)

    def reprojection_error(self, pb, frame_state, tracks):
        pbc = pb.as_c()
        fs = f64(frame_state)
        tc = tracks.as_c()
        out = np.zeros(1)
        rc = self._f("ba_reprojection_error")(C.byref(pbc), _d(fs), C.byref(tc), _d(out))
        assert rc == 0
        return out[0]

    def pnp(self, pb_window, frame_state, tracks, new_state, new_cam, new_imu, new_W, new_K, obs_track, obs_z, delta=None, U=None, jac=None, use_inertial=False):
        """visual_inertial_pnp (pnp.cpp:32-100) of a NEW frame against the window; returns (state16, iterations)"""
        pbc = pb_window.as_c()
        fs = f64(frame_state)
        tc = tracks.as_c()
        x = f64(new_state).copy()
        ot = np.ascontiguousarray(obs_track, np.int32)
        oz = f64(obs_z).reshape(-1, 2)
        z = np.zeros(225)
        it = C.c_int32(0)
        rc = self._f("pnp")(C.byref(pbc), _d(fs), C.byref(tc), _d(x), _d(f64(new_cam)), _d(f64(new_imu)), _d(f64(new_W)), _d(f64(new_K)), len(ot),
                            ot.ctypes.data_as(i32p), _d(oz), _d(f64(delta) if delta is not None else z), _d(f64(U) if U is not None else z),
                            _d(f64(jac) if jac is not None else z), 1 if use_inertial else 0, C.byref(it))
        assert rc == 0, rc
        return x, it.value


_ref_harness = None


def reference():
    """the reference's own code (libpvio_ref.so)"""
    global _ref_harness
    if _ref_harness is None:
        _ref_harness = Harness(lib(), "ref_")
    return _ref_harness


DROPIN_LIB = {"gpu": os.path.join(HERE, "_ref", "libpvio_dropin.so"), "emu": os.path.join(HERE, "_ref", "libpvio_dropin_emu.so")}
_dropin = {}


def build_dropin():
    if os.path.isdir(os.path.join(REF, "pvio", "src")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "ref"), "REF=" + REF, "dropin"])
    return all(os.path.exists(p) for p in DROPIN_LIB.values())


def dropin(kind):
    """the PRODUCT's host adapter on the reference's Map: kind = "gpu" (libpvio_hip.so below it; needs a GPU: the adapter reports every solve
    as unusable without one) or "emu" (the kernel sources in the fiber emulator).  One GPU context per process (process_ctx())."""
    if kind not in _dropin:
        if not build_dropin():
            raise RuntimeError("oracle/_ref/libpvio_dropin*.so are not built and %s is absent" % REF)
        _dropin[kind] = Harness(C.CDLL(DROPIN_LIB[kind]), "dropin_")
    return _dropin[kind]


def solve(pb, tracks=None, use_raw_imu=True, trace=True):
    return reference().solve(pb, tracks, use_raw_imu, trace)


def marginalize(pb, frame_state, tracks, victim):
    return reference().marginalize(pb, frame_state, tracks, victim)


def reprojection_error(pb, frame_state, tracks):
    return reference().reprojection_error(pb, frame_state, tracks)


def pnp(*a, **kw):
    return reference().pnp(*a, **kw)
