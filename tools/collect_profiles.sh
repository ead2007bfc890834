#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 summaries of the bench command, for the metric's workload (kodak24) and for the
# chip-filling one (256 streams in one batch: bench.py --scaling throughput on one GPU), and for the rate model (tools/prof_rate.py).
# clic41 / uhd4k (legs of bench.py) run through tools/prof_workload.py so that their kernels do not mix with kodak24's.
#   pass 1: kernel trace + stats          -> gpurun_out/prof/<tag>/stats
#   pass 2: PMC FETCH_SIZE (own run)      -> gpurun_out/prof/<tag>/fetch
#   pass 3: PMC WRITE_SIZE (own run)      -> gpurun_out/prof/<tag>/write
# (counters are never combined with sys/hip/hsa traces; see MI355X_MICROARCH.md, rocprofv3 PMC slots)
# Afterwards (here or in the build container):  python tools/summarise_pmc.py gpurun_out/prof/<tag> profiles/r06 <tag>
set -u
REPO=$(pwd)
export TMPDIR=/tmp
for TAG in kodak24 kodak256 clic41 uhd4k kodak24_hq rate; do
  OUT=$REPO/gpurun_out/prof/$TAG
  rm -rf "$OUT"; mkdir -p "$OUT"
  EXTRA=""; [ "$TAG" = kodak256 ] && EXTRA="--scaling throughput"
  CMD="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic --legs none $EXTRA"
  case "$TAG" in clic41|uhd4k|kodak24_hq) CMD="python $REPO/tools/prof_workload.py $TAG 3";; rate) CMD="python $REPO/tools/prof_rate.py";; esac
  # r06: the counters of the metric's workload come from the SAME command as bench.py's live pass (measure_traffic_live):
  # the metric's batch alone, f32 output kept - r05's tracked summary profiled the whole bench command, whose from-bytes steps
  # (integer planes only) pulled the float path's mean from 427 to 364 MB per step
  PMC_CMD="$CMD"; [ "$TAG" = kodak24 ] && PMC_CMD="python $REPO/tools/prof_workload.py kodak24 2 keep_float"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/bench_stats.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $PMC_CMD > "$OUT/bench_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $PMC_CMD > "$OUT/bench_write.log" 2>&1
  cd "$REPO"
  python tools/summarise_pmc.py "$OUT" "$REPO/gpurun_out/prof/summary" $TAG > /dev/null
  cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$REPO/gpurun_out/prof/summary/${TAG}_kernel_stats.csv"
  grep '^{' "$OUT/bench_stats.log" | tail -1 | cut -c1-400 > "$REPO/gpurun_out/prof/summary/${TAG}_bench_under_rocprof.json"
  # the counter CSVs are large: keep the summaries only
  rm -rf "$OUT/fetch" "$OUT/write"
done
ls -la "$REPO/gpurun_out/prof/summary"
