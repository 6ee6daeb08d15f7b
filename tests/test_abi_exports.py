"""The drop-in boundary without a GPU: libpvio_hip.so loads, exports every entry point include/pvio_hip.h declares (and the ctypes mirror lists
exactly those), reports the header's ABI version, and refuses loudly to create a context on a box without a GPU -- there is no CPU path."""
import ctypes as C
import os
import re

import pytest

from pvio_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "pvio_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", "", hdr)
    return sorted(set(re.findall(r"\b(pvio_(?:hip_)?[a-z_0-9]+)\s*\(", hdr))), hdr


def test_library_exports_every_declared_entry_point():
    names, _ = _declared()
    assert len(names) >= 29
    lib = capi.load()
    assert [n for n in names if not hasattr(lib, n)] == []
    assert sorted(capi.EXPORTS) == names  # the ctypes mirror neither misses one nor names one the header does not declare


def test_abi_version_of_the_library_is_the_headers():
    _, hdr = _declared()
    declared = int(re.search(r"#define\s+PVIO_HIP_ABI_VERSION\s+(\d+)", hdr).group(1))
    lib = capi.load()
    lib.pvio_hip_abi_version.restype = C.c_int32
    assert lib.pvio_hip_abi_version() == declared == capi.ABI_VERSION
    lib.pvio_hip_version.restype = C.c_char_p
    assert ("ABI %d" % declared).encode() in lib.pvio_hip_version()


def test_no_context_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    from pvio_amd.solver import HipContext, HipError
    with pytest.raises(HipError, match="no CPU fallback"):
        HipContext(device=0)
