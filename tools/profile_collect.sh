#!/bin/bash
# build box, after tools/profile_round.sh ran through gpurun: the summaries the judge reads go to profiles/ (tag = round)
R=$(cd $(dirname $0)/.. && pwd); T=${1:-r4}; MB=${2:-4096}
cd $R
cp $(ls -t gpurun_out/prof_stats/*/*kernel_stats.csv | head -1) profiles/${T}_bench_${MB}MiB_kernel_stats.csv
grep '^{' gpurun_out/prof_stats.log | tail -1 > profiles/${T}_bench_${MB}MiB_line.json
python tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write profiles/${T}_pmc.json $((MB * 16)) "python bench.py --mb $MB --steps 1 --warmup 1 --no-cpu --no-extra --no-probe" > /dev/null
for leg in lz4 raw16 raw64 raw128; do
  [ -d gpurun_out/prof_${leg}_stats ] || continue
  cp $(ls -t gpurun_out/prof_${leg}_stats/*/*kernel_stats.csv | head -1) profiles/${T}_${leg}_kernel_stats.csv
  python tools/pmc_summary.py gpurun_out/prof_${leg}_fetch gpurun_out/prof_${leg}_write profiles/${T}_${leg}_pmc.json 0 "python tools/legs_run.py $leg 1024" > /dev/null
done
python - <<PY
import collections, csv, glob, os
f = max(glob.glob("gpurun_out/prof_tcc/*/*counter_collection.csv"), key=os.path.getmtime)     # (gpurun merges into a directory that may hold older passes)
agg = collections.defaultdict(lambda: [0.0, 0.0])
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if r["Counter_Name"] == "TCC_HIT_sum": agg[k][0] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_MISS_sum": agg[k][1] += float(r["Counter_Value"])
with open("profiles/${T}_tcc_hit_miss.txt", "w") as o:
    o.write("# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -- python bench.py --mb $MB --steps 1 --warmup 1 --no-cpu --no-extra --no-probe\n")
    for k, (h, m) in sorted(agg.items()):
        if k.startswith("qzk_") and h + m > 0:
            o.write("%-44s hits %.4g misses %.4g hit rate %.1f %%\n" % (k[:44], h, m, 100 * h / (h + m)))
PY
ls -la profiles | grep "${T}_" | awk '{print $5, $9}'
