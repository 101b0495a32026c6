#!/usr/bin/env python3
"""Developer tool: qzCompress / qzDecompress host to host (bench.py's api leg alone), for piece counts of the decode
(QATZIP_AMD_PIPE) and cuts (QATZIP_AMD_PIPE_CUTS).  usage: api_h2h.py [MiB] [pieces[:cut%,cut%...] ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import datagen  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 2047
base = datagen.gen("silesia", 128 << 20, 20250523)
tile = len(base) - 4099
for pieces in (sys.argv[2:] or ["default"]):          # "default", "3", or "3:8,40" = three pieces cut at 8 % and 40 % of the member
    os.environ.pop("QATZIP_AMD_PIPE", None); os.environ.pop("QATZIP_AMD_PIPE_CUTS", None)
    if pieces != "default":
        np_, _, cuts = pieces.partition(":")
        os.environ["QATZIP_AMD_PIPE"] = np_
        if cuts:
            os.environ["QATZIP_AMD_PIPE_CUTS"] = cuts
    r = bench.api_leg(base, tile, mb, timed=int(os.environ.get("API_PASSES", "5")))
    print("pieces %-10s" % pieces, json.dumps({k: r[k] for k in ("api_bytes_MiB", "api_compress_GBps", "api_decompress_GBps", "api_decompress_ms")}), flush=True)
