#!/bin/bash
# developer tool: build phase-A (inflate) variants into build/var/: the round-3 kernels from git (r3), the present default,
# and -D variants of it.  usage: tools/inflate_variants.sh [name:flags ...]   (default set below)
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/var
mk() { # name srcdir flags...
  local name=$1 src=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "$@" \
    -I $R/include -I $src -x hip $src/qz_api.cpp $src/qzd_device.hip $src/qzd_inflate.hip $src/qzd_shard.hip \
    -o $R/build/var/lib_$name.so -lpthread -ldl &
}
if [ "$1" = "r3" ]; then
  T=$(mktemp -d); mkdir -p $T/qatzip_amd/csrc; ln -s $R/include $T/include      # the sources include "../../include/qatzip.h"
  for f in $(git -C $R ls-tree --name-only ${2:-3061084} qatzip_amd/csrc/); do git -C $R show ${2:-3061084}:$f > $T/qatzip_amd/csrc/$(basename $f); done
  mk r3 $T/qatzip_amd/csrc; wait; rm -rf $T; ls -la $R/build/var/lib_r3.so; exit 0
fi
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  mk $name $R/qatzip_amd/csrc $flags
done
wait
ls -la $R/build/var
