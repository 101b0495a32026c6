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


def test_cli_builds_and_prints_usage_without_a_gpu():
    """the qzip-style front end is plain C against include/qatzip.h; -h needs neither a session nor a device"""
    import subprocess
    import qatzip_amd.build as B
    B.build()
    r = subprocess.run([B.CLI, "-h"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "-O <fmt>" in r.stdout
    r = subprocess.run([B.CLI, "-O", "zstd", "x"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "not offered" in r.stderr
