"""Builds a VARIANT of libpvio_hip.so for same-box experiments (tests/prof_ab.py, tests/micro/order_probe.py): ba_kernels.hip with a
list of textual substitutions applied, compiled with optional extra flags, linked with the product's other objects.
usage: python tests/micro/build_variant.py NAME [--flag=...]* [--sub OLD NEW]* [--patch file.py]
The output tests/micro/variants/NAME.so is git-ignored (it travels to the GPU box with the snapshot)."""
import os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "pvio_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "micro", "variants")


def build(name, subs=(), flags=(), defines=()):
    src = open(os.path.join(CSRC, "ba_kernels.hip")).read()
    for old, new in subs:
        assert src.count(old) == 1, "substitution target must occur exactly once: %r (%d)" % (old[:60], src.count(old))
        src = src.replace(old, new)
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "ba_kernels.hip")
        open(p, "w").write(src)
        obj = os.path.join(td, "ba_kernels.o")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable",
               "-Wno-unused-but-set-variable", "-I" + CSRC, "-I" + os.path.join(ROOT, "include")] + list(flags) + ["-D" + d for d in defines] + ["-c", p, "-o", obj]
        subprocess.check_call(cmd)
        others = [os.path.join(CSRC, o) for o in ("klt.o", "ba_solver.o", "ba_comm.o", "capi.o", "preintegrator.o", "sym_eig.o", "sym_eig_avx2.o")]
        out = os.path.join(OUT, name + ".so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out, obj] + others +
                              ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    return out


# the operand request of k_dense's rank-8 update as shipped (round 3: unconditional) and the round-2 forms it replaced
OPB_SHIPPED = """                for (int g = 0; g < kDenseCols; ++g)
                    opB[g] = *reinterpret_cast<const lds_d2 *>(Lpan + 128 * (g < R ? nbk - 1 - g : b0));"""
OPB_R2_ASC = """                for (int g = 0; g < kDenseCols; ++g)
                    if (g < R) opB[g] = *reinterpret_cast<const lds_d2 *>(Lpan + 128 * (nbk - 1 - g)); // live columns only"""
OPB_R2_DESC = """                for (int g = kDenseCols - 1; g >= 0; --g)
                    if (g < R) opB[g] = *reinterpret_cast<const lds_d2 *>(Lpan + 128 * (nbk - 1 - g)); // live columns only"""
OPB_DESC_UNCOND = """                for (int g = kDenseCols - 1; g >= 0; --g)
                    opB[g] = *reinterpret_cast<const lds_d2 *>(Lpan + 128 * (g < R ? nbk - 1 - g : b0));"""

FIN_REF = "        Ctrl &cc = *c;\n        const int lr = cc.lin_result;\n        if (cc.it_success) cc.num_success++;"
FIN_COPY = "        Ctrl cc = *c;\n        const int lr = cc.lin_result;\n        if (cc.it_success) cc.num_success++;"

RECIPES = {
    # round 3, second finding: a register copy of the control block in k_dense's Finalize section on top of the one in its control
    # section -> the first factorization of every solve fails on the GPU (tests/micro/dbg_case.py shows the control block)
    "fin_copy": dict(subs=[(FIN_REF, FIN_COPY)]),
    "shipped": dict(),
    # scheduler settings tried on the whole file at the end of round 3 (profiles/r3_ab_sched_flags.txt)
    "sched_max_ilp": dict(flags=["-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
    "sched_max_clause": dict(flags=["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]),
    "sched_no_cluster": dict(flags=["-mllvm", "-misched-cluster=false"]),
    "sched_bias0": dict(flags=["-mllvm", "-amdgpu-schedule-metric-bias=0"]),
    "sched_postra": dict(flags=["-mllvm", "-misched-postra"]),
    # round 3, clean-up of the emulator conditionals: the no-op wave barrier in front of lane 0's release moves 218 lines of k_dense<true, true>;
    # this is the build without it (the state up to commit 216133f)
    "no_wave_barrier": dict(subs=[("""    __builtin_amdgcn_wave_barrier();
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(""", """    if ((threadIdx.x & 63) == 0) __hip_atomic_store("""),
                                  ("""    __builtin_amdgcn_wave_barrier();
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(""", """    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(""")]),
    "r2_asc": dict(subs=[(OPB_SHIPPED, OPB_R2_ASC)]),                      # round 2's product: passes
    "desc": dict(subs=[(OPB_SHIPPED, OPB_R2_DESC)]),                       # THE REPRODUCER: wrong results on the GPU
    "desc_nospill": dict(subs=[(OPB_SHIPPED, OPB_R2_DESC)], flags=["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=false"]),  # passes
    "desc_wc0": dict(subs=[(OPB_SHIPPED, OPB_R2_DESC)], flags=["-mllvm", "-amdgpu-waitcnt-forcezero=1"]),           # fails, same numbers
    "desc_mfmapad": dict(subs=[(OPB_SHIPPED, OPB_R2_DESC)], flags=["-mllvm", "-amdgpu-mfma-padding-ratio=100"]),    # fails, same numbers
    "desc_uncond": dict(subs=[(OPB_SHIPPED, OPB_DESC_UNCOND)]),            # passes
    # round 4, the factor wave's instruction diet (profiles/r4_ab_panel_loop_diet.txt): the loop of round 3 piece by piece
    "mask_upper": dict(subs=[("constexpr bool kDenseMaskUpper = kDenseConservative;", "constexpr bool kDenseMaskUpper = true;")]),
    "operands_all": dict(subs=[("constexpr bool kDenseOperandGroups = !kDenseConservative;", "constexpr bool kDenseOperandGroups = false;")]),
    "rsqrt_newton": dict(subs=[("constexpr bool kRsqrtCubic = true;", "constexpr bool kRsqrtCubic = false;")]),
    "fail_per_pivot": dict(subs=[("constexpr bool kDenseFailAtEnd = !kDenseConservative;", "constexpr bool kDenseFailAtEnd = false;")]),
    "conservative": dict(defines=["PVIO_DENSE_CONSERVATIVE"]),  # what build() ships when hipcc is not csrc/KNOWN_GOOD_TOOLCHAIN
    "loop_stamps": dict(defines=["PVIO_DENSE_LOOP_STAMPS"]),
    # round 5, TIMING ABLATIONS of the look-ahead panel loop (results are WRONG by construction; only tests/prof_phases.py's `factorization done - first panel`
    # of the first factoring launch is read off them: profiles/r5_ablation_panel_loop.txt).  No failure test, so that garbage pivots do not end the loop early.
    "abl_norows": dict(subs=[('#pragma unroll\n                        for (int t = 0; t < kPass; ++t)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 < kPanel; ++c2) x[t][c2] -= x[t][cc] * Ls[c2][cc];\n                    }\n                    // A pivot that is negative, zero, infinite or NaN', '                    }\n                    // A pivot that is negative, zero, infinite or NaN'), ('                    if (kDenseFailAtEnd) fail |= (inv[kPanel - 1] * 0.0 == 0.0) ? 0 : 1;', '                    /* ablation: no failure test */')]),
    "abl_noblock": dict(subs=[('#pragma unroll\n                        for (int r = cc + 1; r < kPanel; ++r)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 <= r; ++c2) Ld[r][c2] -= Ld[r][cc] * Ls[c2][cc];\n#pragma unroll\n                        for (int t = 0; t < kPass; ++t)', '#pragma unroll\n                        for (int t = 0; t < kPass; ++t)'), ('                    if (kDenseFailAtEnd) fail |= (inv[kPanel - 1] * 0.0 == 0.0) ? 0 : 1;', '                    /* ablation: no failure test */')]),
    "abl_nomfma": dict(subs=[('        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][0], opB[g][0], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \\\n        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][1], opB[g][1], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \\', '        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q][0] += opA[q][0] * opB[g][0] + opA[q][1] * opB[g][1]; /* ablation: one FMA pair instead of two MFMAs */ \\'), ('                    if (kDenseFailAtEnd) fail |= (inv[kPanel - 1] * 0.0 == 0.0) ? 0 : 1;', '                    /* ablation: no failure test */')]),
    "abl_noarith": dict(subs=[('#pragma unroll\n                        for (int r = cc + 1; r < kPanel; ++r)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 <= r; ++c2) Ld[r][c2] -= Ld[r][cc] * Ls[c2][cc];\n#pragma unroll\n                        for (int t = 0; t < kPass; ++t)', '#pragma unroll\n                        for (int t = 0; t < kPass; ++t)'), ('#pragma unroll\n                        for (int t = 0; t < kPass; ++t)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 < kPanel; ++c2) x[t][c2] -= x[t][cc] * Ls[c2][cc];\n                    }\n                    // A pivot that is negative, zero, infinite or NaN', '                    }\n                    // A pivot that is negative, zero, infinite or NaN'), ('                    if (kDenseFailAtEnd) fail |= (inv[kPanel - 1] * 0.0 == 0.0) ? 0 : 1;', '                    /* ablation: no failure test */')]),
    "abl_skeleton": dict(subs=[('#pragma unroll\n                        for (int r = cc + 1; r < kPanel; ++r)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 <= r; ++c2) Ld[r][c2] -= Ld[r][cc] * Ls[c2][cc];\n#pragma unroll\n                        for (int t = 0; t < kPass; ++t)', '#pragma unroll\n                        for (int t = 0; t < kPass; ++t)'), ('#pragma unroll\n                        for (int t = 0; t < kPass; ++t)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 < kPanel; ++c2) x[t][c2] -= x[t][cc] * Ls[c2][cc];\n                    }\n                    // A pivot that is negative, zero, infinite or NaN', '                    }\n                    // A pivot that is negative, zero, infinite or NaN'), ('        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][0], opB[g][0], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \\\n        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][1], opB[g][1], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \\', '        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q][0] += opA[q][0] * opB[g][0] + opA[q][1] * opB[g][1]; /* ablation: one FMA pair instead of two MFMAs */ \\'), ('                    if (kDenseFailAtEnd) fail |= (inv[kPanel - 1] * 0.0 == 0.0) ? 0 : 1;', '                    /* ablation: no failure test */')]),
    "relaxed_nosleep": dict(subs=[("""    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);""",
                                   """    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);"""),
                                  ("""    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);""",
                                   """    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);"""),
                                  ("val < target) __builtin_amdgcn_s_sleep(1);", "val < target) __builtin_amdgcn_s_sleep(0);")]),
    "abl_skel_norsq": dict(subs=[('#pragma unroll\n                        for (int r = cc + 1; r < kPanel; ++r)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 <= r; ++c2) Ld[r][c2] -= Ld[r][cc] * Ls[c2][cc];\n#pragma unroll\n                        for (int t = 0; t < kPass; ++t)', '#pragma unroll\n                        for (int t = 0; t < kPass; ++t)'), ('#pragma unroll\n                        for (int t = 0; t < kPass; ++t)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 < kPanel; ++c2) x[t][c2] -= x[t][cc] * Ls[c2][cc];\n                    }\n                    // A pivot that is negative, zero, infinite or NaN', '                    }\n                    // A pivot that is negative, zero, infinite or NaN'), ('        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][0], opB[g][0], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \\\n        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][1], opB[g][1], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \\', '        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q][0] += opA[q][0] * opB[g][0] + opA[q][1] * opB[g][1]; /* ablation: one FMA pair instead of two MFMAs */ \\'), ('                    if (kDenseFailAtEnd) fail |= (inv[kPanel - 1] * 0.0 == 0.0) ? 0 : 1;', '                    /* ablation: no failure test */'), ('                        inv[cc] = fast_rsqrt(dd);\n                        const double inv2 = inv[cc] * inv[cc];', '                        inv[cc] = 1.0;\n                        const double inv2 = dd;')]),
    "abl_skel_norowio": dict(subs=[('#pragma unroll\n                        for (int r = cc + 1; r < kPanel; ++r)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 <= r; ++c2) Ld[r][c2] -= Ld[r][cc] * Ls[c2][cc];\n#pragma unroll\n                        for (int t = 0; t < kPass; ++t)', '#pragma unroll\n                        for (int t = 0; t < kPass; ++t)'), ('#pragma unroll\n                        for (int t = 0; t < kPass; ++t)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 < kPanel; ++c2) x[t][c2] -= x[t][cc] * Ls[c2][cc];\n                    }\n                    // A pivot that is negative, zero, infinite or NaN', '                    }\n                    // A pivot that is negative, zero, infinite or NaN'), ('        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][0], opB[g][0], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \\\n        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][1], opB[g][1], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \\', '        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q][0] += opA[q][0] * opB[g][0] + opA[q][1] * opB[g][1]; /* ablation: one FMA pair instead of two MFMAs */ \\'), ('                    if (kDenseFailAtEnd) fail |= (inv[kPanel - 1] * 0.0 == 0.0) ? 0 : 1;', '                    /* ablation: no failure test */'), ('                            const lds_d2 g2 = *reinterpret_cast<const lds_d2 *>(xrow[t] + 2 * h);', '                            const lds_d2 g2 = {dg2[h][0], dg2[h][1]}; /* ablation: no row reads */'), ('                                Lrow[h] = pr;', '                                if (h == 7) Lrow[h] = pr; /* ablation: no row stores */')]),
    "abl_skel_bare": dict(subs=[('#pragma unroll\n                        for (int r = cc + 1; r < kPanel; ++r)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 <= r; ++c2) Ld[r][c2] -= Ld[r][cc] * Ls[c2][cc];\n#pragma unroll\n                        for (int t = 0; t < kPass; ++t)', '#pragma unroll\n                        for (int t = 0; t < kPass; ++t)'), ('#pragma unroll\n                        for (int t = 0; t < kPass; ++t)\n#pragma unroll\n                            for (int c2 = cc + 1; c2 < kPanel; ++c2) x[t][c2] -= x[t][cc] * Ls[c2][cc];\n                    }\n                    // A pivot that is negative, zero, infinite or NaN', '                    }\n                    // A pivot that is negative, zero, infinite or NaN'), ('        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][0], opB[g][0], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \\\n        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[q][1], opB[g][1], acc[dt_col_slot<LA>(g) + q], 0, 0, 0); \\', '        _Pragma("unroll") for (int q = 0; q < dt_col_slot<LA>((g) + 1) - dt_col_slot<LA>(g); ++q)                      \\\n            acc[dt_col_slot<LA>(g) + q][0] += opA[q][0] * opB[g][0] + opA[q][1] * opB[g][1]; /* ablation: one FMA pair instead of two MFMAs */ \\'), ('                    if (kDenseFailAtEnd) fail |= (inv[kPanel - 1] * 0.0 == 0.0) ? 0 : 1;', '                    /* ablation: no failure test */'), ('                        inv[cc] = fast_rsqrt(dd);\n                        const double inv2 = inv[cc] * inv[cc];', '                        inv[cc] = 1.0;\n                        const double inv2 = dd;'), ('                            const lds_d2 g2 = *reinterpret_cast<const lds_d2 *>(xrow[t] + 2 * h);', '                            const lds_d2 g2 = {dg2[h][0], dg2[h][1]}; /* ablation: no row reads */'), ('                                Lrow[h] = pr;', '                                if (h == 7) Lrow[h] = pr; /* ablation: no row stores */')]),
    # round 5: the look-ahead form with the hardware barrier in place of the two LDS counters: two s_barrier per panel executed by all four waves (A: the
    # L rows are stored; B: the next columns are published); the factor wave still runs one panel ahead of the update waves' trailing work
    "la_barrier": dict(subs=[('                    dense_wait(flag_pub, 3 * (pidx + 1));', "                    if (pidx > 0) __syncthreads(); /* barrier B(p - 1): the update waves have published this panel's columns */"), ('                    if (fail) { // uniform\n                        if (lane == 0) sh_fail = 1;\n                        dense_signal_set(flag_L, -1);\n                        break;\n                    }', '                    if (fail && lane == 0) sh_fail = 1; /* barrier form: every wave walks all panels (a failed factorization finishes on NaNs), no early exit */'), ('                    dense_signal_set(flag_L, pidx + 1); // (release: the rows above are in LDS before the counter moves)', "                    __syncthreads(); /* barrier A(p): the panel's L rows are in LDS */"), ('                    lfo += LS * (LDV - j0);\n                }\n            } else {\n                int pidx = 0;', '                    lfo += LS * (LDV - j0);\n                }\n                __syncthreads(); /* barrier B(last) */\n            } else {\n                int pidx = 0;'), ('                    if (dense_wait(flag_L, pidx + 1) < 0) break;', '                    __syncthreads(); /* barrier A(p) */'), ('                                dense_signal_add(flag_pub); // (release; one count per wave and panel, whether it owns a row here or not)', "                                __syncthreads(); /* barrier B(p): this panel's next columns are published */")]),
    "abl_nofailtest": dict(subs=[('                    if (kDenseFailAtEnd) fail |= (inv[kPanel - 1] * 0.0 == 0.0) ? 0 : 1;', '                    /* ablation: no failure test */')]),
    # round 5: the two LDS counters of the look-ahead form moved by RELAXED stores / adds behind a compiler barrier instead of release operations: the
    # LDS executes one wave's instructions in order, so the counter still becomes visible after the data, and the producer does not wait for its writes
    # to drain (s_waitcnt lgkmcnt(0)) before it moves the counter
    "relaxed_signals": dict(subs=[("""    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);""",
                                   """    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);"""),
                                  ("""    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);""",
                                   """    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);""")]),  # the per-panel stamp sites 8-17 (tests/prof_phases.py reads them)
}

if __name__ == "__main__":
    for n in sys.argv[1:]:
        print(build(n, **RECIPES[n]))
