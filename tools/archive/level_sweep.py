import sys, time, zlib
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, qatzip_amd
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
src = datagen.gen("silesia", 64 << 20, 20250523)
import numpy as np
src = np.tile(src, mb // 64).tobytes()
c = qatzip_amd.Context(0)
d_src = c.alloc(len(src)); d_src.upload(src)
d_dst = c.alloc(qatzip_amd.max_deflate_len(len(src), 65536))
import os
for lvl in [int(x) for x in os.environ.get('LEVELS', '1,2,3,4,6,9').split(',')]:
    best = 1e9
    for rep in range(2):
        t = time.time(); n, _ = c.deflate_raw(d_src, len(src), 65536, lvl, 1, d_dst); best = min(best, time.time() - t)
    print("level %d  %7.1f ms  %6.2f GB/s  ratio %.3f" % (lvl, best * 1e3, len(src) / best / 1e9, n / len(src)), flush=True)
