#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 summaries of the bench command.
#   pass 1: kernel trace + stats          -> gpurun_out/prof/stats
#   pass 2: PMC FETCH_SIZE (own run)      -> gpurun_out/prof/fetch
#   pass 3: PMC WRITE_SIZE (own run)      -> gpurun_out/prof/write
# (counters are never combined with sys/hip/hsa traces; see MI355X_MICROARCH.md, rocprofv3 PMC slots)
set -u
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/bench_write.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.csv" | head -40
tail -1 "$OUT/bench_stats.log" | cut -c1-300
