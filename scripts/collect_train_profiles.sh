#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of one training-step bench per model family.  Output: gpurun_out/prof_train/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_train
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/psignn -- python3 scripts/train_bench.py 50 13 5 0 1.0 > $OUT/psignn.json 2> $OUT/psignn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/baselines -- python3 scripts/baseline_train_bench.py 50 13 5 0 > $OUT/baselines.json 2> $OUT/baselines.err
cp $(ls $OUT/psignn/*/*kernel_stats.csv | head -1) $OUT/psignn_kernel_stats.csv
cp $(ls $OUT/baselines/*/*kernel_stats.csv | head -1) $OUT/baselines_kernel_stats.csv
head -14 $OUT/psignn_kernel_stats.csv | cut -c1-70,200-330
head -14 $OUT/baselines_kernel_stats.csv | cut -c1-70,200-330
