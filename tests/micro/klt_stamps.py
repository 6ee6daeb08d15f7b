"""Where one LK track's time goes (the makespan of k_lk_track): builds VARIANTS of libpvio_hip.so whose k_lk_track
returns shader-clock counts in place of the tracked position (s_memtime at the level boundaries), and prints their distribution.
The product kernel is untouched; the variants are textual substitutions on a copy of klt.hip (tests/micro/variants/, git-ignored).
usage:  python tests/micro/klt_stamps.py build        (CPU box: cross-compiles the variants)
        python tests/micro/klt_stamps.py run [n ...]  (GPU box: one process per variant and track count)"""
import os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "pvio_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "micro", "variants")

# (pair written to next_xy) per variant
MODES = {
    "total_iters": ("(float)(clock64() - pv_t0)", "(float)pv_n"),
    "tpl_iter": ("(float)pv_tpl", "(float)pv_it"),
    "start_reloads": ("(float)pv_start", "(float)pv_rl"),
    "tplload_tplform": ("(float)pv_ld", "(float)(pv_tpl - pv_ld)"),
}

SUBS = [
    ("    LkTplRaw raw;\n    LkTpl T;\n    int skip = lk_template_load(I, level, pxf, pyf, wy, wx, raw); // wave-uniform\n    if (!skip) skip = lk_template_form(raw, live, T);\n",
     "    LkTplRaw raw;\n    LkTpl T;\n    const long long pv_l0 = clock64();\n    if (pv_start == 0) pv_start = pv_l0 - pv_t0;\n"
     "    int skip = lk_template_load(I, level, pxf, pyf, wy, wx, raw); // wave-uniform\n"
     "    if (!skip) { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); }\n    const long long pv_l05 = clock64();\n    pv_ld += pv_l05 - pv_l0;\n"
     "    if (!skip) skip = lk_template_form(raw, live, T);\n    const long long pv_l1 = clock64();\n    pv_tpl += pv_l1 - pv_l0;\n"),
    ("            lk_load_tap_pairs(s0, r0), lk_load_tap_pairs(s1, r1);\n            cinx = inx, ciny = iny;\n",
     "            lk_load_tap_pairs(s0, r0), lk_load_tap_pairs(s1, r1);\n            cinx = inx, ciny = iny;\n            ++pv_rl;\n"),
    ("        const float b1 = wave_sum_f(sb1) * FLT_SCALE, b2 = wave_sum_f(sb2) * FLT_SCALE;\n        const float dx = (a12 * b2 - a22 * b1) * D",
     "        ++pv_n;\n        const float b1 = wave_sum_f(sb1) * FLT_SCALE, b2 = wave_sum_f(sb2) * FLT_SCALE;\n        const float dx = (a12 * b2 - a22 * b1) * D"),
    ("    if (st && level == 0) {\n        const int ix = (int)floorf(outx - half)", "    pv_it += clock64() - pv_l1;\n    if (st && level == 0) {\n        const int ix = (int)floorf(outx - half)"),
    # the counters live in the kernel and reach lk_level / lk_finish by reference
    ("int lane, float &outx, float &outy, int &st) {", "int lane, float &outx, float &outy, int &st, const long long pv_t0, long long &pv_tpl, long long &pv_it, long long &pv_start, long long &pv_ld, int &pv_n, int &pv_rl) {"),
    ("__device__ __forceinline__ void lk_finish(const TrackArgs &a, int p, int lane, float outx, float outy, int st) {",
     "__device__ __forceinline__ void lk_finish(const TrackArgs &a, int p, int lane, float outx, float outy, int st, const long long pv_t0, long long pv_tpl, long long pv_it, long long pv_start, long long pv_ld, int pv_n, int pv_rl) {"),
    ("    int st = 1;\n#pragma unroll\n    for (int li = 0; li < kLevels; ++li) {", "    int st = 1;\n    const long long pv_t0 = clock64();\n    long long pv_tpl = 0, pv_it = 0, pv_start = 0, pv_ld = 0;\n    int pv_n = 0, pv_rl = 0;\n#pragma unroll\n    for (int li = 0; li < kLevels; ++li) {"),
    ("        lk_level(a.prev[level], a.next[level], level, level == a.n_levels - 1, pxf, pyf, lane, outx, outy, st);\n    }\n    lk_finish(a, p, lane, outx, outy, st);",
     "        lk_level(a.prev[level], a.next[level], level, level == a.n_levels - 1, pxf, pyf, lane, outx, outy, st, pv_t0, pv_tpl, pv_it, pv_start, pv_ld, pv_n, pv_rl);\n    }\n    lk_finish(a, p, lane, outx, outy, st, pv_t0, pv_tpl, pv_it, pv_start, pv_ld, pv_n, pv_rl);"),
    # the unit kernel is not stamped: its calls get dummies
    ("        lk_level(I, J, level, li == 0, pxf, pyf, lane, outx, outy, st);\n        if (level == 0) lk_finish(a, p, lane, outx, outy, st);",
     "        long long pv_a = 0, pv_b = 0, pv_c = 0, pv_d = 0;\n        int pv_e = 0, pv_f = 0;\n        lk_level(I, J, level, li == 0, pxf, pyf, lane, outx, outy, st, 0, pv_a, pv_b, pv_c, pv_d, pv_e, pv_f);\n        if (level == 0) lk_finish(a, p, lane, outx, outy, st, 0, 0, 0, 0, 0, 0, 0);"),
]
TAIL = "        a.next_xy[2 * p] = outx, a.next_xy[2 * p + 1] = outy;\n        a.status[p] = (uint8_t)st;\n"


def build():
    base = open(os.path.join(CSRC, "klt.hip")).read()
    for old, new in SUBS:
        assert base.count(old) == 1, "substitution target must occur exactly once: %r (%d)" % (old[:70], base.count(old))
        base = base.replace(old, new)
    assert base.count(TAIL) == 1
    os.makedirs(OUT, exist_ok=True)
    for mode, (e0, e1) in MODES.items():
        src = base.replace(TAIL, "        a.next_xy[2 * p] = %s, a.next_xy[2 * p + 1] = %s;\n        a.status[p] = (uint8_t)st;\n" % (e0, e1))
        with tempfile.TemporaryDirectory() as td:
            p, obj = os.path.join(td, "klt.hip"), os.path.join(td, "klt.o")
            open(p, "w").write(src)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable",
                                   "-Wno-unused-but-set-variable", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-c", p, "-o", obj])
            others = [os.path.join(CSRC, o) for o in ("ba_kernels.o", "ba_solver.o", "ba_comm.o", "capi.o", "preintegrator.o", "sym_eig.o", "sym_eig_avx2.o")]
            out = os.path.join(OUT, "klt_stamps_%s.so" % mode)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out, obj] + others + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
            print("built", out)


def child(mode, n, truth_guess):
    sys.path.insert(0, ROOT)
    import numpy as np
    from pvio_amd import synth, capi
    from pvio_amd.solver import HipContext, HipImage, klt_track
    os.environ["PVIO_HIP_LK_FORM"] = "1"  # the stamped kernel is the one with a wave per track
    ctx = HipContext(lib=capi.load(os.path.join(OUT, "klt_stamps_%s.so" % mode)), device=0)
    img0, img1, p, truth, init = synth.make_image_pair(512, 512, 6000)
    A, B = HipImage(ctx, img0), HipImage(ctx, img1)
    guess = truth[:n].astype(np.float32) if truth_guess else init[:n]
    for _ in range(3):
        q, st, t = klt_track(ctx, A, B, p[:n], guess)
    f = lambda v: "min %8.0f  p10 %8.0f  median %8.0f  p90 %8.0f  max %8.0f  mean %8.0f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max(), v.mean())
    print("%-16s n %5d %s launch %.1f us\n    [0] %s\n    [1] %s" % (mode, n, "guess=truth" if truth_guess else "           ", 1e3 * t, f(q[:, 0]), f(q[:, 1])), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        ns = [int(x) for x in sys.argv[2:]] or [64, 1024, 1500, 6000]
        print("shader-clock counts (s_memtime; the launch time by HIP events beside them)")
        for n in ns:
            for mode in MODES:
                for tg in ((0, 1) if n == 1500 else (0,)):
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, str(n), str(tg)], capture_output=True, text=True, timeout=300)
                    print(r.stdout.strip() if r.returncode == 0 else "%s %d failed: %s" % (mode, n, r.stderr[-300:]), flush=True)
