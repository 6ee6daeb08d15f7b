"""Comparison of chain record streams (tests/host/chain_log.h): the product chain against the oracle, record by record, in two ways.

REPLAY (records 4 / 5 / 7 of the product chain's own log): every window solve, marginalization and PnP solve of the product chain is
run again by the oracle ON THE SAME INPUTS.  This is the north_star's bar as it stands: identical accept / reject trace, iteration
counts and termination, states after EVERY iteration within 1e-6 (observed ~1e-9), S^T S / S^T s of the new prior equal.

FREE RUNNING (the product chain's log against the oracle chain's log): nothing is re-synchronized between the two.  (Rounds 2-3: the LK
tracker's float accumulators were reduced in another order on the device than in the oracle, keypoints differed by up to 1e-3 px per call and
the chains parted at the first threshold that met such a difference -- frame 48 of 60.  Round 4: the oracle DEFINES the order the kernel
implements (oracle/oracle_klt.cpp header), keypoints are bit-identical given identical inputs, and the chains run all 60 frames with
identical decisions; the tolerances below are kept as the bars they were, the observed values are in the reports.)  Required here:
  per camera frame   IDENTICAL frame ids, track id and track length of every keypoint (= the same tracks survived LK, the 20 px border,
                     the F-RANSAC and the Poisson-disk selection, the same corners were added, in the same order); new corners (integer
                     pixels) bit-identical; tracked keypoints within FREE_KLT_PX
  per window solve   IDENTICAL shapes (frames, landmarks, factors), termination, iteration counts, accept / reject trace, depth-gate mask;
                     states after every iteration within FREE_STATE (inverse depths triangulated over a few pixels of parallax move by
                     1e-5 when a keypoint moves by 1e-3 px: that, not the solver, sets this number)
  per PnP / marginalization  identical shapes and iteration counts, results within the same tolerance
  trajectory.tum     same poses at the same times within FREE_POSE.
compare_free still takes `min_identical_frames` (strict record-by-record identity up to the first differing decision, which must not come
earlier than that; None = never) for experiments with another summation order; the reported poses are held together over the WHOLE
sequence regardless."""
import numpy as np

from chain_run import parse_log

REPLAY_STATE = 1.0e-6    # north_star: pose / landmark states after every trust-region iteration
FREE_KLT_PX = 2.0e-2     # observed 4e-3 over 60 frames
FREE_STATE = 5.0e-4      # observed 2e-5 .. 1e-4 (inverse depths)
FREE_POSE = 5.0e-5       # observed 5e-6
FREE_COST_RTOL, FREE_COST_ATOL = 2.0e-2, 1.0e-3  # a sum of squared residuals of a few hundredths of a pixel
MARG_FLOOR = 1.0e-7      # absolute floor under the entries of S^T S / S^T s (the reference's own eigenvalue cut is 1e-8)


def _solve(tag, k, Ia, Da, Ib, Db, state_tol, cost_rtol, cost_atol, info):
    assert (Ia == Ib).all(), "%s window solve %d: shapes / termination / trace flags / depth gate differ:\n%s\n%s" % (tag, k, Ia[:10], Ib[:10])
    N, M, ln = int(Ia[0]), int(Ia[1]), int(Ia[9])
    S = 16 * N + M
    np.testing.assert_allclose(Da[:2], Db[:2], rtol=cost_rtol, atol=cost_atol)
    ta, tb = Da[2:2 + 7 * ln].reshape(ln, 7), Db[2:2 + 7 * ln].reshape(ln, 7)
    np.testing.assert_allclose(ta[:, 0], tb[:, 0], rtol=cost_rtol, atol=cost_atol)  # cost
    np.testing.assert_allclose(ta[:, 6], tb[:, 6], rtol=1e-12)                      # mu: powers of the same constants
    if state_tol <= REPLAY_STATE:
        np.testing.assert_allclose(ta[:, 5], tb[:, 5], rtol=1e-5)                   # trust-region radius
    o = 2 + 7 * ln
    sa, sb = Da[o:o + ln * S].reshape(ln, S), Db[o:o + ln * S].reshape(ln, S)
    d = float(np.abs(sa - sb).max())
    info["max_state_" + tag] = max(info.get("max_state_" + tag, 0.0), d)
    assert d <= state_tol, "%s window solve %d: states differ by %.3g after iteration %d" % (tag, k, d, int(np.abs(sa - sb).max(1).argmax()))
    o += ln * S
    assert float(np.abs(Da[o:o + S] - Db[o:o + S]).max()) <= state_tol
    qa, qb = Da[o + S:], Db[o + S:]
    np.testing.assert_allclose(qa, qb, rtol=0, atol=1e-6 if state_tol <= REPLAY_STATE else 2e-2)  # mean pixel error of each track
    flags = Ia[10:10 + 3 * ln].reshape(ln, 3)
    return int(Ia[7]), int((flags[1:, 2] == 0).sum())


def _marg(tag, k, Ia, Da, Ib, Db, rtol):
    assert (Ia == Ib).all() and Ia[3] == 0, "%s marginalization %d: %s %s" % (tag, k, Ia, Ib)
    D = 15 * int(Ia[2])
    Sa, sa, Sb, sb = Da[:D * D].reshape(D, D), Da[D * D:], Db[:D * D].reshape(D, D), Db[D * D:]
    Ha, Hb = Sa.T @ Sa, Sb.T @ Sb
    # entry (i, j) against sqrt(H_ii H_jj); marginalize_frame zeroes eigenvalues below 1e-8 (bundle_adjustor.cpp:586-588), so entries of
    # that size are whatever the eigen-solver's rounding left of them on either side: an absolute floor of a few 1e-8 goes with the ratio
    # The floor also scales with the matrix: a symmetric eigen-solver is backward stable to a few eps |H| and no better -- a coordinate
    # without any information (an exactly zero row of the Schur complement: velocity / bias of a frame no IMU factor reaches) comes back
    # with an eigenvalue of +-(a few) eps |H|, which the 1e-8 cut keeps or drops as the rounding falls (measured on one 45 x 45 matrix with
    # |H| = 7.4e8, eighteen zero rows: 4.7e-8 with round 2's solver, 1.2e-8 / 5.0e-7 with round 3's AVX2 / baseline builds, -3.5e-8 LAPACK,
    # the oracle's Jacobi sweeps keep exact zeros).  Eigen's tridiagonal QR in the reference is in the same position.
    floor = max(MARG_FLOOR, 32 * np.finfo(float).eps * float(np.abs(Ha).max()))
    scale = np.sqrt(np.outer(np.diag(Ha), np.diag(Ha)))
    excess = np.abs(Ha - Hb) - floor
    d = float((excess / (scale + 1e-300)).max())
    i, j = np.unravel_index(int((excess / (scale + 1e-300)).argmax()), Ha.shape)
    assert d <= rtol, "%s marginalization %d: information matrix differs by %.3g of sqrt(H_ii H_jj) at (%d, %d): %.6e vs %.6e, diagonal %.3e %.3e" % (
        tag, k, d, i, j, Ha[i, j], Hb[i, j], Ha[i, i], Ha[j, j])
    ga, gb = Sa.T @ sa, Sb.T @ sb
    gscale = np.sqrt(np.diag(Ha)) * max(1.0, float(np.linalg.norm(sa)))
    dg = float(((np.abs(ga - gb) - floor) / (gscale + 1e-300)).max())
    assert dg <= rtol, "%s marginalization %d: information vector differs by %.3g" % (tag, k, dg)
    return max(d, 0.0)


def _pnp(tag, k, Ia, Da, Ib, Db, tol, same_inputs):
    if same_inputs:
        assert (Ia == Ib).all(), "%s PnP %d: %s %s" % (tag, k, Ia, Ib)
        assert (Da[:16] == Db[:16]).all()
    else:
        assert (Ia[:3] == Ib[:3]).all(), "%s PnP %d: factor counts differ %s %s" % (tag, k, Ia, Ib)
    d = float(np.abs(Da[16:32] - Db[16:32]).max())
    assert d <= tol, "%s PnP %d: states differ by %.3g" % (tag, k, d)
    return d


def compare_replay(log):
    """records 2/3/6 of the product chain against the oracle's replay 4/5/7 that follows each of them"""
    R = parse_log(log)
    info = dict(solves=0, iterations=0, rejected_steps=0, margs=0, pnps=0, max_marg=0.0, max_pnp=0.0)
    pair = {2: 4, 3: 5, 6: 7}
    i = 0
    while i < len(R):
        tag, Ia, Da = R[i]
        assert tag > 0, "a call failed in the chain (tag %d)" % tag
        if tag in pair:
            assert i + 1 < len(R) and R[i + 1][0] == pair[tag], "record %d (tag %d) is not followed by its replay" % (i, tag)
            _, Ib, Db = R[i + 1]
            if tag == 2:
                it, rej = _solve("replay", info["solves"], Ia, Da, Ib, Db, REPLAY_STATE, 1e-7, 1e-12, info)
                info["solves"] += 1
                info["iterations"] += it
                info["rejected_steps"] += rej
            elif tag == 3:
                info["max_marg"] = max(info["max_marg"], _marg("replay", info["margs"], Ia, Da, Ib, Db, 1e-7))
                info["margs"] += 1
            else:
                info["max_pnp"] = max(info["max_pnp"], _pnp("replay", info["pnps"], Ia, Da, Ib, Db, REPLAY_STATE, True))
                info["pnps"] += 1
            i += 2
        else:
            i += 1
    return info


def compare_free(log_product, log_oracle, fx, tum_product=None, tum_oracle=None, min_identical_frames=None):
    """The two chains side by side.  They are compared record by record for as long as every discrete decision is the same; the first
    record whose integers differ (a track that survived on one side only, another factor count, another accept / reject) ends the
    strict part -- from there on the chains are different experiments -- and only the reported poses are held together (FREE_POSE) to
    the end of the sequence.  min_identical_frames: how many camera frames must pass before that may happen (None: never)."""
    A = [r for r in parse_log(log_product) if abs(r[0]) not in (4, 5, 7)]
    B = parse_log(log_oracle)
    info = dict(frames=0, identical_frames=0, first_divergence=None, solves=0, margs=0, pnps=0, iterations=0, rejected_steps=0, tracked=0, new=0,
                max_kp_px=0.0, max_pose=0.0, max_marg=0.0, max_pnp=0.0, max_window_inv_depth=0.0)
    frame = -1
    for (tag, Ia, Da), (tagb, Ib, Db) in zip(A, B):
        assert tag > 0 and tagb > 0, "a call failed in the chains (tags %d %d)" % (tag, tagb)
        same = tag == tagb and Ia.shape == Ib.shape and bool((Ia == Ib).all())
        if tag == 6 and tagb == 6:
            same = bool((Ia[:3] == Ib[:3]).all())  # factor counts; the iteration count of a converged PnP may move by one
        if not same:
            what = {1: "surviving tracks / new corners", 2: "window solve (shape, trace or depth gate)", 3: "marginalization", 6: "PnP factor count", 8: "window track flags"}
            info["first_divergence"] = dict(after_frame=frame, record=what.get(tag, str(tag)))
            break
        if tag == 1:
            frame = int(Ia[0])
            n = int(Ia[4])
            ka, kb = Da[:2 * n].reshape(n, 2), Db[:2 * n].reshape(n, 2)
            fresh = Ia[5:].reshape(n, 2)[:, 1] == 0  # a corner detected in this frame: no track yet (Frame::detect_keypoints only appends keypoints)
            assert (ka[fresh] == kb[fresh]).all(), "frame %d: new corners differ" % frame
            if (~fresh).any():
                d = float(np.abs(ka[~fresh] - kb[~fresh]).max() * fx)
                info["max_kp_px"] = max(info["max_kp_px"], d)
                assert d <= FREE_KLT_PX, "frame %d: tracked keypoints differ by %.3g px" % (frame, d)
            info["identical_frames"] += 1
            info["tracked"] += int((~fresh).sum())
            info["new"] += int(fresh.sum())
        elif tag == 2:
            it, rej = _solve("free", info["solves"], Ia, Da, Ib, Db, FREE_STATE, FREE_COST_RTOL, FREE_COST_ATOL, info)
            info["solves"] += 1
            info["iterations"] += it
            info["rejected_steps"] += rej
        elif tag == 3:
            info["max_marg"] = max(info["max_marg"], _marg("free", info["margs"], Ia, Da, Ib, Db, 1e-2))
            info["margs"] += 1
        elif tag == 6:
            info["max_pnp"] = max(info["max_pnp"], _pnp("free", info["pnps"], Ia, Da, Ib, Db, FREE_STATE, False))
            info["pnps"] += 1
        elif tag == 8:
            n = int(Ia[2])
            valid = Ia[3:].reshape(n, 3)[:, 1] == 1
            if valid.any():
                d = float(np.abs(Da.reshape(n, 2)[valid, 0] - Db.reshape(n, 2)[valid, 0]).max())
                info["max_window_inv_depth"] = max(info["max_window_inv_depth"], d)
                assert d <= FREE_STATE, "frame %d: inverse depths of the window's valid tracks differ by %.3g" % (frame, d)
    # the reported pose of EVERY frame, divergence or not
    pa = {int(I[0]): D[-8:] for t, I, D in A if t == 1}
    pb = {int(I[0]): D[-8:] for t, I, D in B if t == 1}
    assert sorted(pa) == sorted(pb)
    info["frames"] = len(pa)
    for f in sorted(pa):
        d = float(np.abs(pa[f] - pb[f]).max())
        info["max_pose"] = max(info["max_pose"], d)
        assert d <= FREE_POSE, "frame %d: reported pose differs by %.3g" % (f, d)
    if min_identical_frames is not None:
        assert info["identical_frames"] >= min_identical_frames, "the chains part after %d identical frames: %s" % (info["identical_frames"], info["first_divergence"])
    else:
        assert info["first_divergence"] is None, info["first_divergence"]
    if tum_product and tum_oracle:
        ta, tb = np.loadtxt(tum_product, ndmin=2), np.loadtxt(tum_oracle, ndmin=2)
        assert ta.shape == tb.shape and ta.shape[0] > 0 and (ta[:, 0] == tb[:, 0]).all()
        info["tum_max_diff"] = float(np.abs(ta - tb).max())
        info["tum_poses"] = int(ta.shape[0])
        assert info["tum_max_diff"] <= FREE_POSE, "trajectory.tum files differ by %.3g" % info["tum_max_diff"]
    return info


SEQ_STATE = 1.0e-6       # north_star's bar on every window state, while the two runs have made the same discrete choices
SEQ_POSE_AFTER_FLIP = 5.0e-2  # [m] reported poses once the reference's own best-plane coin flip has made the two runs different experiments


def compare_seq(log_ref, log_dropin, fx, kp_px=2.0e-3, allow_divergence=False, pose_after_flip=SEQ_POSE_AFTER_FLIP):
    """The reference's own pvio::PVIO over a sequence, twice (oracle/ref/seq_capi.cpp): with the reference's BundleAdjustor / visual_inertial_pnp
    (libpvio_ref.so) and with the product's linked in their place (libpvio_dropin*.so).  Records 1 / 8 / 9 after EVERY camera frame, compared strictly:
      integers   frame ids, track id + length of every keypoint, window frame ids / keyframe / fix flags, TF_VALID / TF_PLANE of every window track,
                 plane ids and sizes: IDENTICAL;
      keypoints  within kp_px pixels (bit-identical when the same front end -- or the product's, whose LK sums are in the oracle's defined order -- sees
                 the same inputs; its initial guesses are gyro predictions from the window's biases, frame.cpp:97-103, which carry the back-ends'
                 1e-10 into a float32 cast);
      window     Frame::pose / motion of every window frame, inverse depth of every VALID track, plane parameters within SEQ_STATE;
      poses      reported by PVIO::track_camera within FREE_POSE.
    The strict comparison ENDS at the reference's best-plane coin flip, reported as `strict_frames` / `coin_flip_frame`: PlaneExtractor::
    extend_planes_and_cast_plane_points (plane_extractor.cpp:112-156) picks the plane of a track by `rpe_after_project < min_rpe`; for a track with ONE
    observation the reprojection error of the point cast on ANY plane is 0 up to 1e-14, and the extractor reports the same wall once per frame between
    keyframes (plane_extractor.cpp:39-81, update_map only at keyframes), so the map holds several planes a millimetre apart until merge_planes joins
    them: which of them a one-observation track is cast on, and which plane its factor then attaches to, is decided by the last bit.  A dozen tracks
    landing on the other plane move the next window solve by millimetres (measured on the GPU run of the wall scene: 12 of 14 such tracks, states
    3.7e-3 apart at the first solve with planes, with plane parameters agreeing to 2e-10).  From there on the two runs -- or two runs of the REFERENCE
    whose inputs differ in the last bit -- are different experiments; only the reported poses are still held together (SEQ_POSE_AFTER_FLIP).  The relief
    scene (tests/test_host_headless._relief) has no planes: there the strict comparison covers the whole sequence."""
    A, B = parse_log(log_ref), parse_log(log_dropin)
    assert len(A) == len(B), (len(A), len(B))
    info = dict(frames=0, strict_frames=0, coin_flip_frame=None, first_divergence=None, window_records=0, max_state=0.0, max_inv_depth=0.0, max_plane=0.0,
                max_kp_px=0.0, max_pose=0.0, max_pose_after_flip=0.0, coin_flips=0, planes_seen=0, plane_tracks_seen=0, keyframes=0)
    flipped, frame = False, -1
    for (tag, Ia, Da), (tagb, Ib, Db) in zip(A, B):
        if tag == 1:
            info["frames"] += 1
            dpose = float(np.abs(Da[-8:] - Db[-8:]).max())
            if flipped:
                info["max_pose_after_flip"] = max(info["max_pose_after_flip"], dpose)
                assert dpose <= pose_after_flip, "frame %d: reported pose differs by %.3g after the coin flip" % (int(Ia[0]), dpose)
        if flipped:
            continue
        same = tag == tagb and Ia.shape == Ib.shape and bool((Ia == Ib).all())
        if not same:
            info["first_divergence"] = dict(after_frame=frame, record=tag, ints_ref=int(Ia.size), ints_other=int(Ib.size),
                                            differing=int((Ia != Ib).sum()) if Ia.shape == Ib.shape else None)
            if not allow_divergence:
                break
            # long sequences: a discrete choice (a track surviving the F-matrix RANSAC, a corner passing the distance filter) has come out differently;
            # from here on the two runs are different experiments, like after the coin flip: only the reported poses are still held together
            flipped = True
            continue
        if tag == 1:
            frame = int(Ia[0])
            n = int(Ia[4])
            if n:
                info["max_kp_px"] = max(info["max_kp_px"], float(np.abs(Da[:2 * n] - Db[:2 * n]).max() * fx))
            info["max_pose"] = max(info["max_pose"], dpose)
            assert dpose <= FREE_POSE, "frame %d: reported pose differs by %.3g" % (frame, dpose)
        elif tag == 8:
            n = int(Ia[2])
            fl = Ia[3:].reshape(n, 3)
            valid, plane, one = (fl[:, 1] & 1) == 1, (fl[:, 1] & 2) == 2, fl[:, 2] <= 1
            da, db = Da.reshape(n, 2), Db.reshape(n, 2)
            coin = valid & plane & one
            if coin.any() and float(np.abs(da[coin, 0] - db[coin, 0]).max()) > SEQ_STATE:
                info["coin_flips"] = int((np.abs(da[coin, 0] - db[coin, 0]) > SEQ_STATE).sum())
                info["coin_flip_frame"] = frame
                flipped = True
                continue
            if valid.any():
                d = float(np.abs(da[valid, 0] - db[valid, 0]).max())
                info["max_inv_depth"] = max(info["max_inv_depth"], d)
                assert d <= SEQ_STATE, "frame %d: inverse depths of the window's valid tracks differ by %.3g" % (frame, d)
            info["plane_tracks_seen"] = max(info["plane_tracks_seen"], int(plane.sum()))
        elif tag == 9:
            N = int(Ia[1])
            P = int(Ia[2 + 4 * N])
            d = float(np.abs(Da[:16 * N] - Db[:16 * N]).max())
            info["max_state"] = max(info["max_state"], d)
            assert d <= SEQ_STATE, "frame %d: window frame states differ by %.3g" % (frame, d)
            if P:
                dp = float(np.abs(Da[16 * N:] - Db[16 * N:]).max())
                info["max_plane"] = max(info["max_plane"], dp)
                assert dp <= SEQ_STATE, "frame %d: plane parameters differ by %.3g" % (frame, dp)
            info["planes_seen"] = max(info["planes_seen"], P)
            info["window_records"] += 1
            is_kf = int(Ia[2 + 4 * (N - 1) + 1])
            info["keyframes"] += is_kf
            # keyframe windows that carry plane-distance factors: a plane with >= 20 tracks (bundle_adjustor.cpp:180-195); identical on both sides (ints are compared above)
            sizes = [int(Ia[2 + 4 * N + 1 + 2 * k + 1]) for k in range(P)]
            if is_kf and any(t >= 20 for t in sizes):
                info["keyframes_with_plane_factors"] = info.get("keyframes_with_plane_factors", 0) + 1
            info["largest_plane_tracks"] = max(info.get("largest_plane_tracks", 0), max(sizes, default=0))
            info["strict_frames"] = frame + 1
    assert info["max_kp_px"] <= kp_px, "tracked keypoints differ by %.3g px" % info["max_kp_px"]
    assert allow_divergence or info["first_divergence"] is None, info["first_divergence"]
    return info


def ate_rmse(tum_path, gt):
    """Absolute trajectory error the way the TUM / EuRoC benchmarks define it: positions of trajectory.tum (t px py pz q) matched to the ground truth
    [t p q] by time, aligned by the best rigid transform (Horn / Umeyama without scale), RMSE of what is left.  Returns (rmse [m], poses used)."""
    T = np.loadtxt(tum_path, ndmin=2)
    idx = [int(np.argmin(np.abs(gt[:, 0] - t))) for t in T[:, 0]]
    assert np.abs(gt[idx, 0] - T[:, 0]).max() < 1e-6
    A, B = T[:, 1:4], gt[idx, 1:4]
    ca, cb = A.mean(0), B.mean(0)
    U, _, Vt = np.linalg.svd((B - cb).T @ (A - ca))
    S = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ S @ Vt
    res = (A - ca) @ R.T + cb - B
    return float(np.sqrt((res ** 2).sum(1).mean())), int(len(T))
