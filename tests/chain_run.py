"""Runs ONE chain (tests/host/libpvio_chain_*.so) over the rendered sequence of tests/test_host_headless.py in its own process (track
and frame ids come from process-wide counters; the emulated and the real kernels cannot share a process) and leaves the record stream of
tests/host/chain_log.h plus the reported trajectory in TUM format.
usage: python tests/chain_run.py <library> <out prefix> <n_frames> <window> <gap> <distance> <small|full>[_relief][_sweep][_b]
(the libraries of oracle/ref/Makefile export the same entry point: the reference's own pvio::PVIO over the sequence, oracle/ref/seq_capi.cpp;
PVIO_SEQ_IMAGE=oracle|hip picks their pvio::Image)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(dp)


def parse_log(path):
    """-> list of (tag, ints int64 array, doubles float64 array)"""
    buf = open(path, "rb").read()
    out, o = [], 0
    while o < len(buf):
        tag, ni, nd = np.frombuffer(buf, np.int32, 3, o)
        o += 12
        I = np.frombuffer(buf, np.int64, ni, o)
        o += 8 * int(ni)
        D = np.frombuffer(buf, np.float64, nd, o)
        o += 8 * int(nd)
        out.append((int(tag), I, D))
    return out


def main():
    lib_path, prefix, n_frames, window, gap, distance, size = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), sys.argv[7]
    import test_host_headless as hh
    tokens = size.split("_")           # small|full [relief] [sweep]
    relief = "relief" in tokens        # the scene without planes (test_host_headless._relief)
    sweep = "sweep" in tokens          # the back-and-forth trajectory for long sequences (test_host_headless._pose_sweep)
    variant = 1 if "b" in tokens else 0  # a second family: other texture, relief and sweep (round 6)
    sz = hh.SMALL if tokens[0] == "small" else None
    W, H, K4 = sz if sz is not None else (hh.W, hh.H, hh.K4)
    images, times, imu_t, imu_w, imu_a, gt, q_bc, p_bc = hh.render_sequence(n_frames, size=sz, relief=relief, sweep=sweep, variant=variant)
    lib = C.CDLL(lib_path, mode=C.RTLD_GLOBAL)
    out, stats = np.zeros((n_frames, 8)), np.zeros(4, np.int32)
    err = C.create_string_buffer(512)
    lib.host_chain_run.restype = C.c_int
    rc = lib.host_chain_run(C.c_int(n_frames), C.c_int(W), C.c_int(H), images.ctypes.data_as(C.POINTER(C.c_uint8)), _d(times), C.c_int(len(imu_t)), _d(imu_t),
                            _d(imu_w), _d(imu_a), _d(K4), _d(np.ascontiguousarray(q_bc)), _d(np.ascontiguousarray(p_bc)), C.c_int(len(gt)), _d(gt), C.c_int(window),
                            C.c_int(gap), C.c_double(distance), (prefix + ".log").encode(), _d(out), stats.ctypes.data_as(C.POINTER(C.c_int32)), err, C.c_int(512))
    if rc != 0:
        print("chain failed:", err.value.decode())
        sys.exit(1)
    with open(prefix + ".tum", "w") as f:  # pvio-pc's trajectory.tum: t px py pz qx qy qz qw, full precision
        for r in out:
            if np.abs(r[4:8]).sum() > 0:
                f.write(" ".join(repr(float(v)) for v in r) + "\n")
    np.save(prefix + ".gt.npy", gt)
    print("chain ok: initialized %d window %d solves %d tracks %d" % tuple(stats))


if __name__ == "__main__":
    main()
