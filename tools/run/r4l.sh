cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 60 python - <<'PY'
import ctypes as C, qatzip_amd
ctx = qatzip_amd.Context(0)
o = (C.c_int * 4)()
ctx.L.qzd_inflate_occupancy(o)
print("resident workgroups per CU: tok<16,2> %d  spec<4> %d  spec<8> %d  resolve %d" % tuple(o))
PY
