#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench command, then two
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.  Output under gpurun_out/prof_<tag>/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 50 --warmup 0 --repeats 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 50 --warmup 0 --repeats 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/pmc_write.err
python3 scripts/summarise_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic_model.json 50 1000519
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv
