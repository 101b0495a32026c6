#!/bin/bash
# developer tool (round 6): build K1 variants into build/var/ - waves per workgroup, entries of the LDS entry cache (log2; -1 none),
# slot tables.  usage: tools/k1_variants.sh name waves cnblog nslot nslot2 [extra flags] ...   (one variant per call; run several with &)
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/var
hipcc --offload-arch=gfx950 ${QZ_OPT:--O3} -std=c++17 -fPIC -shared -Wno-unused-value -DQZK_K1_WAVES=$2 -DQZK_CNBLOG=$3 -DQZK_NSLOT=$4 -DQZK_NSLOT2=$5 $6 \
  -I $R/include -I $R/qatzip_amd/csrc -x hip $R/qatzip_amd/csrc/qz_api.cpp $R/qatzip_amd/csrc/qzd_device.hip $R/qatzip_amd/csrc/qzd_inflate.hip $R/qatzip_amd/csrc/qzd_shard.hip \
  -o $R/build/var/lib_$1.so -lpthread -ldl
