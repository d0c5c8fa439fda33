#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench command, then separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE, TCC_MISS_sum) as MI355X_MICROARCH.md prescribes.  Output under gpurun_out/prof_<tag>/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r3}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
echo "kernel-trace pass done"
PMCARGS="--steps 50 --warmup 0 --repeats 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $PMCARGS --dump-launch-log $OUT/launch_log.json > /dev/null 2> $OUT/pmc_fetch.err
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $PMCARGS > /dev/null 2> $OUT/pmc_write.err
echo "WRITE_SIZE pass done"
rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_HIT_sum --output-format csv -d $OUT/pmc_miss -- python3 bench.py $PMCARGS > /dev/null 2> $OUT/pmc_miss.err
echo "TCC_MISS pass done"
python3 scripts/summarise_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/launch_log.json $OUT/pmc_traffic_model.json 1000519 $OUT/pmc_miss
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv
