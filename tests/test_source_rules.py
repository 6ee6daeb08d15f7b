"""Rules of the tree that are cheap to hold mechanically (no GPU needed).

* the product (pvio_amd/) never reaches into oracle/ -- the oracle is test infrastructure, the product has no CPU fallback;
* the kernel sources do not branch on the fiber emulator of tests/hipemu (its header supplies the builtins they use);
* no CUDA / hipify / Triton compatibility layer in the product sources."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = os.path.join(ROOT, "pvio_amd")


def _sources(top, exts):
    for d, _, files in os.walk(top):
        if os.path.basename(d) in ("__pycache__", "lib"):
            continue
        for f in files:
            if f.endswith(exts):
                yield os.path.join(d, f)


def test_product_never_reaches_into_the_oracle():
    bad = []
    for p in _sources(PRODUCT, (".py", ".cpp", ".h", ".hip", "Makefile")):
        text = open(p, errors="replace").read()
        for m in re.finditer(r'^\s*(#\s*include\s*[<"][^>"]*oracle[^>"]*[>"]|from\s+oracle\b|import\s+oracle\b)|-loracle|liboracle', text, re.M):
            bad.append((os.path.relpath(p, ROOT), m.group(0).strip()))
    assert not bad, bad


def test_kernel_sources_do_not_branch_on_the_emulator():
    hits = []
    for p in _sources(os.path.join(PRODUCT, "csrc"), (".hip",)):
        for n, line in enumerate(open(p), 1):
            if re.match(r"\s*#\s*if.*PV_HIPEMU", line):
                hits.append("%s:%d" % (os.path.relpath(p, ROOT), n))
    assert not hits, hits


def test_no_cuda_or_multi_backend_layer_in_the_product():
    pat = re.compile(r"__HIP_PLATFORM_AMD__|__CUDACC__|cuda_runtime|hipify|\btriton\b|#\s*include\s*<cuda")
    hits = []
    for p in _sources(PRODUCT, (".py", ".cpp", ".h", ".hip")):
        for n, line in enumerate(open(p, errors="replace"), 1):
            if pat.search(line):
                hits.append("%s:%d: %s" % (os.path.relpath(p, ROOT), n, line.strip()[:80]))
    assert not hits, hits
