"""The C-ABI library builds for gfx950 here (no GPU) and exports every symbol include/*.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qz[dA-Z]\w*)\s*\(", txt)))


def test_library_builds_and_exports_device_abi():
    import qatzip_amd
    so = qatzip_amd.build.build()
    L = ctypes.CDLL(so)
    names = _declared("qzamd_device.h")
    assert "qzd_deflate_raw" in names and len(names) >= 12
    for n in names:
        assert hasattr(L, n), "missing export: " + n
    assert L.qzd_device_count() >= 0
