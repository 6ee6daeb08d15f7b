"""Compares the LK calls two runs of a sequence dumped with PVIO_KLT_DUMP=<prefix>: <prefix>_hip.bin (HipImage::track_keypoints,
pvio_amd/host/feature_front.cpp) against <prefix>_oracle.bin (OracleImage, tests/host/oracle_image.h), call by call -- the record is
[int32 n][prev_xy 2n f32][initial next_xy 2n f32][tracked next_xy 2n f32][status n u8].  This is how round 5 showed that every LK call of
the 360-frame sequences was bit-identical in inputs AND outputs and that the two sides parted at the F-RANSAC behind it.
usage: python tests/probe_klt_dump.py <prefix>     (or two file names)"""
import sys

import numpy as np


def read_calls(path):
    raw = open(path, "rb").read()
    calls, o = [], 0
    while o < len(raw):
        n = int(np.frombuffer(raw, np.int32, 1, o)[0])
        o += 4
        f = np.frombuffer(raw, np.float32, 6 * n, o).reshape(3, n, 2)
        o += 24 * n
        st = np.frombuffer(raw, np.uint8, n, o)
        o += n
        calls.append((f[0], f[1], f[2], st))
    return calls


def compare(a_path, b_path, out=sys.stdout):
    A, B = read_calls(a_path), read_calls(b_path)
    print("%d / %d calls" % (len(A), len(B)), file=out)
    first = None
    for i, (a, b) in enumerate(zip(A, B)):
        if a[0].shape != b[0].shape:
            print("call %d: %d against %d tracks -- the runs have parted before this call" % (i, a[0].shape[0], b[0].shape[0]), file=out)
            first = first if first is not None else i
            break
        same_in = a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
        same_st = (a[3] == b[3]).all()
        ok = a[3] > 0
        dpos = float(np.abs(a[2] - b[2])[ok & (b[3] > 0)].max()) if (ok & (b[3] > 0)).any() else 0.0
        if not (same_in and same_st and dpos == 0.0):
            print("call %d: inputs %s, status bytes %s (%d differ), largest position difference of tracked points %.3g px"
                  % (i, "identical" if same_in else "DIFFER", "identical" if same_st else "DIFFER", int((a[3] != b[3]).sum()), dpos), file=out)
            first = first if first is not None else i
    if first is None and len(A) == len(B):
        print("all calls bit-identical in inputs and outputs", file=out)
    return first


if __name__ == "__main__":
    a, b = (sys.argv[1] + "_hip.bin", sys.argv[1] + "_oracle.bin") if len(sys.argv) == 2 else sys.argv[1:3]
    sys.exit(0 if compare(a, b) is None else 1)
