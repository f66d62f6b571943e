#!/bin/bash
# Copy the judged artefacts of a tools/round_bundle.sh run from gpurun_out/ (scratch) to profiles/ (tracked).
TAG=${1:-r04}
cd "$(dirname "$0")/.."
for f in gpurun_out/${TAG}_bench_n1.json gpurun_out/${TAG}_bench_detail.json gpurun_out/${TAG}_bench_models.jsonl gpurun_out/${TAG}_bench_sets.jsonl gpurun_out/${TAG}_driver.jsonl gpurun_out/${TAG}_driver_long.jsonl gpurun_out/${TAG}_lengths.jsonl gpurun_out/${TAG}_ragged.jsonl gpurun_out/${TAG}_kernel_stats_*.md gpurun_out/${TAG}_pmc_dominant_kernel_*.json; do
  [ -s "$f" ] && cp "$f" profiles/
done
ls profiles | grep "^${TAG}_"
