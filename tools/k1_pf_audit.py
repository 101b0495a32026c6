#!/usr/bin/env python3
"""Developer tool (round 6): K1 asks for the next window's table entries with a load the compiler does not count
(qzk_ld_bkt_ahead, qzk_deflate_lz77.h) and waits for it itself at the next window's top.  Between the two statements nothing
may read or write the destination registers - the compiler believes they were written when the load was issued, so a copy,
a spill or a reuse would take whatever the registers held before the data landed.  This script compiles the device code to
assembly and checks exactly that for every kernel that carries the load: on every path of the control-flow graph from the load
to a full wait no instruction names the destination registers.
usage: k1_pf_audit.py [extra -D flags]   exit code 0 = clean"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_of(tok):
    """v5 -> {5}; v[4:7] -> {4,5,6,7}"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1) is not None:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit(asm, name):
    """every path from the load to a full wait (`s_waitcnt vmcnt(0)`, the kernel's own or one the compiler placed): no
    instruction on it may name the destination registers.  The control-flow graph comes from the labels and branches."""
    lines = asm.split("\n")
    label = {}
    for k, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label[m.group(1)] = k

    def succ(k):
        l = lines[k].split(";")[0].strip()
        m = re.match(r"s_(c?branch\w*)\s+(\.LBB\d+_\d+)", l)
        out = []
        if m:
            out.append(label[m.group(2)])
            if m.group(1) == "branch":
                return out
        if l.startswith("s_endpgm") or l.startswith("s_setpc"):
            return out
        if k + 1 < len(lines):
            out.append(k + 1)
        return out

    bad = 0
    issues = [i for i, l in enumerate(lines) if "global_load_dwordx4" in l and "sc1" in l and i > 0 and "ASMSTART" in lines[i - 1]
              and "s_waitcnt" not in lines[i + 1]]
    for i in issues:
        dst = regs_of(lines[i].split(",")[0])
        seen, todo, checked, waits = set(), [i + 1], 0, set()
        while todo:
            k = todo.pop()
            if k in seen:
                continue
            seen.add(k)
            l = lines[k].split(";")[0].strip()
            if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
                waits.add(k + 1)
                continue                    # everything asked for has landed behind this
            if l and not l.startswith(".") and not l.endswith(":"):
                checked += 1
                undef_read = "v_readfirstlane_b32" in l and "implicit-def" in lines[k + 1]
                if k != i and not undef_read and regs_of(l) & dst:   # (the compiler's read of an UNDEFINED value names the lowest register: nothing depends on it)
                    print("%s: line %d touches v%s with the load of line %d still on its way: %s" % (name, k + 1, sorted(dst), i + 1, l))
                    bad += 1
            todo.extend(succ(k))
        print("%s: load at line %d into v%s: %d instructions on the paths to the waits at lines %s" % (name, i + 1, sorted(dst), checked, sorted(waits)))
    return bad, len(issues)


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "dev.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value"] + sys.argv[1:] +
                              ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "qatzip_amd", "csrc"), "-x", "hip",
                               os.path.join(ROOT, "qatzip_amd", "csrc", "qzd_device.hip"), "--cuda-device-only", "-S", "-o", out],
                              stderr=subprocess.DEVNULL)
        asm = open(out).read()
    bad = found = 0
    for m in re.finditer(r"^(_Z\w*qzk_lz77_pull\w*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
        b, f = audit(m.group(2), m.group(1)[:40])
        bad += b; found += f
    print("%d untracked loads, %d violations" % (found, bad))
    return 1 if bad or not found else 0


if __name__ == "__main__":
    sys.exit(main())
