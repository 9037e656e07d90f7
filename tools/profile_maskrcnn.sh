#!/bin/bash
# Kernel statistics of the Mask R-CNN step (bench.py --config maskrcnn, BASELINE configs[2]) -> gpurun_out/prof_mr/
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_mr
rm -rf $O; mkdir -p $O
python3 bench.py --config maskrcnn --steps 20 --warmup 5 > $O/bench_maskrcnn.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --config maskrcnn --steps 5 --warmup 2 > $O/bench_under_trace.json 2> $O/kt.err
python3 tools/prof_summary.py $(ls $O/kt/*/*kernel_stats.csv | head -1) 7 60 > $O/kernel_stats.txt
rm -rf $O/kt
head -30 $O/kernel_stats.txt
cat $O/bench_maskrcnn.json | cut -c1-600
