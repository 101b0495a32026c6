"""Build the gfx950 shared library in-tree (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libqatzip_amd.so")


def _sources():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".cpp", ".c", ".h")):
            out.append(os.path.join(CSRC, f))
    out.append(os.path.join(HERE, "cli", "qzip_amd.c"))
    out.append(os.path.join(os.path.dirname(HERE), "include", "qzamd_device.h"))
    q = os.path.join(os.path.dirname(HERE), "include", "qatzip.h")
    if os.path.exists(q):
        out.append(q)
    return out


def needs_build():
    if not os.path.exists(SO) or not os.path.exists(os.path.join(HERE, "qzip-amd")):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> qatzip_amd/libqatzip_amd.so"""
    if not force and not needs_build():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the MI355X backend (no CPU fallback exists)")
    units = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", CSRC, "-x", "hip"] + units + \
          ["-o", SO, "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_cli(verbose)
    return SO


CLI = os.path.join(HERE, "qzip-amd")


def build_cli(verbose=False):
    """the qzip-style file front end: plain C against include/qatzip.h, linked like any application"""
    cmd = ["gcc", "-O2", "-std=gnu99", "-Wall", "-I", os.path.join(os.path.dirname(HERE), "include"),
           os.path.join(HERE, "cli", "qzip_amd.c"), "-o", CLI, "-L", HERE, "-lqatzip_amd", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force=True, verbose=True))
