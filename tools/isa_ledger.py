#!/usr/bin/env python3
"""The ledger of GPU-verified kernel ISA (tests/golden/isa_verified.json).

k_dense sits at 256 VGPRs + 244 AGPRs and has been miscompiled by equivalent source forms twice; k_lk_track_units hung the GPU in
its first build because of how the compiler laid out a loop (profiles/r5_ab_klt_units_hang.txt).  Neither shows on a box without a GPU: the
fiber emulator runs the SOURCE's semantics.  What a box without a GPU can do is notice that the code the compiler emits for a kernel is no
longer the code that passed the GPU suite.  This tool compiles the kernel sources to ISA text with the product's flags (hipcc cross-compiles
gfx950 anywhere), cuts out every kernel's body (its label .. its end label, comments and blank lines dropped: symbolic, position-independent) and
hashes it.

    python tools/isa_ledger.py            compare with the ledger, list what differs (exit status 1 if anything does)
    python tools/isa_ledger.py --update "<what verified it>"
                                          AFTER `pytest -m gpu` and smoke() passed on a GPU with a library built from this very tree:
                                          record the hashes, the toolchain and the evidence named on the command line

tests/test_isa_guards.py::test_kernel_isa_is_the_gpu_verified_one runs the comparison in the CPU suite (skipped under another toolchain than
the ledger's: __graft_entry__.build() then builds the conservative k_dense and smoke()'s gate is what stands guard)."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pvio_amd", "csrc")
LEDGER = os.path.join(ROOT, "tests", "golden", "isa_verified.json")
HIPCC = "/opt/rocm/bin/hipcc"
SOURCES = ["ba_kernels.hip", "klt.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950"]  # pvio_amd/csrc/Makefile CXXFLAGS (the warnings switches do not reach the code)


def toolchain():
    out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    return next((l.strip() for l in out.splitlines() if l.startswith("HIP version")), "unknown")


def kernel_bodies(source, defines=()):
    """{mangled name: (sha256 of the normalized body, {vgpr, agpr, sgpr, scratch bytes})} of every kernel of one source file"""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([HIPCC] + FLAGS + list(defines) + ["-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, source)],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    kernels = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, re.M))
    res, lines, i = {}, text.splitlines(), 0
    while i < len(lines):
        m = re.match(r"^(\S+):", lines[i])
        if m and m.group(1) in kernels:
            name, body = m.group(1), []
            i += 1
            while i < len(lines):
                code = lines[i].split(";")[0].strip()
                if code.startswith(".Lfunc_end"):  # (a kernel may hold several s_endpgm: its end is the function's end label)
                    break
                if code:
                    body.append(re.sub(r"\.LBB\d+_", ".LBB_", code))  # block labels carry the function's ordinal in the file: a kernel added in front of this one is not a change of this one
                i += 1
            res[name] = [hashlib.sha256("\n".join(body).encode()).hexdigest(), {"instructions": sum(1 for b in body if not b.endswith(":") and not b.startswith("."))}]
        i += 1
    # register budgets from the kernel descriptors
    for name, blk in re.findall(r"\.amdhsa_kernel\s+(\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        if name in res:
            g = lambda key: int((re.search(r"\.amdhsa_%s\s+(\d+)" % key, blk) or [0, 0])[1])
            res[name][1].update(next_free_vgpr=g("next_free_vgpr"), accum_offset=g("accum_offset"), next_free_sgpr=g("next_free_sgpr"), scratch_bytes=g("private_segment_fixed_size"))
    return res


def current():
    out = {}
    for s in SOURCES:
        for k, (h, info) in kernel_bodies(s).items():
            out[k] = {"source": s, "sha256": h, **info}
    return out


def compare(verbose=True):
    led = json.load(open(LEDGER))
    if led["toolchain"] != toolchain():
        return None, "another toolchain (%s; the ledger's is %s)" % (toolchain(), led["toolchain"])
    cur = current()
    diff = sorted(k for k in set(cur) | set(led["kernels"]) if cur.get(k, {}).get("sha256") != led["kernels"].get(k, {}).get("sha256"))
    if verbose:
        for k in diff:
            a, b = led["kernels"].get(k), cur.get(k)
            print("%s: %s" % (k, "new kernel" if a is None else "kernel gone" if b is None else "ISA differs from the GPU-verified one (%d -> %d instructions)" % (a["instructions"], b["instructions"])))
    return diff, None


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "--update":
        if len(sys.argv) < 3:
            sys.exit("name the evidence: --update \"<GPU run that verified this tree>\"")
        json.dump({"what": "sha256 of every kernel's compiled body (tools/isa_ledger.py) as verified on an MI355X", "toolchain": toolchain(), "flags": FLAGS,
                   "verified_by": sys.argv[2], "kernels": current()}, open(LEDGER, "w"), indent=1, sort_keys=True)
        print("ledger written:", LEDGER)
        sys.exit(0)
    diff, why = compare()
    if diff is None:
        print("not compared:", why)
        sys.exit(0)
    print("%d kernel(s) differ from the ledger" % len(diff))
    sys.exit(1 if diff else 0)
