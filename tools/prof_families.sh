#!/bin/bash
# rocprofv3 kernel tables of single families (one gpurun call): bash tools/prof_families.sh <tag> <name:bench args>...
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
TAG=${1:-r04a}; shift
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  rm -rf "$OUT/prof_$name"
  rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o p -- python "$REPO/bench.py" --headline-only --windows 1 --lanes 1 $args > "$OUT/prof_$name.log" 2>&1
  python "$REPO/tools/rocprof_summary.py" "$(ls $OUT/prof_$name/*.db 2>/dev/null | head -1)" > "$OUT/${TAG}_kernel_stats_$name.md" 2>/dev/null
  rm -rf "$OUT/prof_$name"
  head -12 "$OUT/${TAG}_kernel_stats_$name.md" | cut -c1-180
done
