#!/bin/bash
# developer tool: the reference's fleet shape on one GPU - P independent PROCESSES, each one thread making synchronous
# qzCompress (then qzDecompress) calls of the harness' default 512 KB buffer, -l loops passes (run_perf_test.sh:100-123:
# `qatzip-test -m 4 -l 1000 -t 1 -D comp` per process, rates summed).  Prints, per P: summed Gbit/s per direction, the
# wall time of the slowest process, the time until the first process reported, and the device memory in use while they ran.
# usage: tools/fleet.sh [loops] [P ...]
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
L=${1:-1000}; shift
PS=${@:-8 24 48}
EXE=$R/build/var/bt_sweep
[ -x $EXE ] || gcc -O2 -std=gnu99 -I $R/include $R/tests/c/bt_sweep.c -o $EXE -L $R/qatzip_amd -lqatzip_amd -lpthread -Wl,-rpath,$R/qatzip_amd
vram() { rocm-smi --showmeminfo vram 2>/dev/null | awk '/Used Memory/ {print $NF; exit}'; }
idle=$(vram)
for P in $PS; do
  for D in comp decomp; do
    T=$(mktemp -d); t0=$(date +%s.%N); peak=0
    for ((p = 0; p < P; p++)); do
      ( taskset -c $p timeout 600 $EXE run -l $L -t 1 -D $D > $T/out.$p 2> $T/err.$p; date +%s.%N > $T/end.$p ) &
    done
    while [ $(ls $T/end.* 2>/dev/null | wc -l) -lt $P ]; do v=$(vram); [ -n "$v" ] && [ "$v" -gt "$peak" ] && peak=$v; sleep 0.2; done
    wait
    sum=$(cat $T/out.* | awk '/Gbps/ {for (i = 1; i <= NF; i++) if ($i == "Gbps") s += $(i - 1)} END {printf "%.3f", s}')
    ok=$(cat $T/out.* | grep -c Gbps)
    first=$(cat $T/end.* | sort -n | head -1); last=$(cat $T/end.* | sort -n | tail -1)
    awk -v P=$P -v D=$D -v L=$L -v sum="$sum" -v ok=$ok -v f=$first -v l=$last -v t0=$t0 -v pk=$peak -v id=${idle:-0} 'BEGIN {
      printf "P %2d %-6s loops %d: %8s Gbit/s summed (%d of %d processes reported)  first done %.2f s  last done %.2f s  vram in use %.2f GiB (idle %.2f)\n",
             P, D, L, sum, ok, P, f - t0, l - t0, pk / 1073741824, id / 1073741824 }'
    grep -h -m1 . $T/err.* 2>/dev/null | sort | uniq -c | head -3
    rm -rf $T
  done
done
