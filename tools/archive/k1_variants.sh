#!/bin/bash
# developer tool: build K1 variants (waves per workgroup / LDS ring / slot table / workgroups per CU) into build/var/
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/var
mk() { # name waves ring nslot occ [extra flags]
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DQZK_K1_WAVES=$2 -DQZK_RING=$3 -DQZK_NSLOT=$4 -DQZK_K1_OCC=$5 $6 \
    -I $R/include -I $R/qatzip_amd/csrc -x hip $R/qatzip_amd/csrc/qz_api.cpp $R/qatzip_amd/csrc/qzd_device.hip $R/qatzip_amd/csrc/qzd_inflate.hip $R/qatzip_amd/csrc/qzd_shard.hip \
    -o $R/build/var/lib_$1.so -lpthread &
}
mk w24 12 4096 256 2
mk w20 10 4096 256 2
mk w16s 16 4096 256 1
wait
ls -la $R/build/var
