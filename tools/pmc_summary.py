#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are collected in SEPARATE runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/<round>_pmc.json.

usage: tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> <K1 launch chunks> "<profiled command>"
The summary carries the SHA-256 of the library's sources (bench.src_sha256) at the time it is written - run it before the
sources change again: bench.py quotes a traffic figure only for the code it was measured on.
Units: rocprofv3 reports both counters in KiB.  gfx950 correction (guide, §HBM): FETCH_SIZE counts 128-B requests
as 64 B for wide coalesced streaming reads => x2; calibrated here on qzk_crc_kernel, a pure 16-B/lane streaming read
whose byte count is known.  For gather-heavy kernels the factor is between 1 and 2; both figures are kept."""
import collections
import csv
import glob
import json
import os
import sys


def load(d, name):
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)     # gpurun merges into a directory that may hold an older pass
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    fd, wd, out = sys.argv[1:4]
    k1_chunks = int(sys.argv[4]); cmd = sys.argv[5]
    bench_mb = int(cmd.split("--mb")[1].split()[0]) if "--mb" in cmd else 4096
    fetch, nf = load(fd, "FETCH_SIZE")
    write, nw = load(wd, "WRITE_SIZE")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- " + cmd,
           "src_sha256": bench.src_sha256(),
           "unit": "bytes per launch (average over the launches of the run)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k[0].startswith("qzk_"):
            continue
        f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
        res["kernels"]["%s[grid=%d]" % k] = {
            "launches": nf.get(k, nw.get(k, 0)), "fetch_size_kib": round(f_kib, 1), "write_size_kib": round(w_kib, 1),
            "hbm_bytes_raw": int((f_kib + w_kib) * 1024), "hbm_bytes_fetch_x2": int((2 * f_kib + w_kib) * 1024)}
    # the full-batch launches of K1 (largest grid of the pull kernel): what bench.py's roofline.traffic quotes
    k1 = [k for k in res["kernels"] if k.startswith("qzk_lz77_pull_kernel")]
    if k1:
        res["k1_key"] = max(k1, key=lambda k: int(k.split("grid=")[1].rstrip("]")))
        res["k1_launch_chunks"] = k1_chunks
        res["bench_mb"] = bench_mb
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
