#!/bin/bash
# Same-box A/B of environment switches on the headline-only bench, one JSON line per point:
#   gpurun --timeout 900 -- 'bash tools/ab_bench.sh r05_ab "WS_STREAM_DYN=0|--lanes 2" "WS_STREAM_DYN=1|--lanes 2" ...'
# Every point is "<ENV=VAL ...>|<bench.py args>"; the points run in the order given, then once more in reverse (the
# chip's clock and the box's noise drift over a call: a point is only better if it is better in both passes).
TAG=${1:-ab}; shift
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/${TAG}.jsonl
mkdir -p gpurun_out; rm -f "$OUT"
point() {
  local envs="${1%%|*}" args="${1#*|}" line
  line=$(env $envs timeout 300 python bench.py $args --headline-only --no-cpu-baseline --sustain-s 0 2> /dev/null | tail -1)
  python - "$envs" "$args" "$line" >> "$OUT" <<'PY'
import json, sys
envs, args, line = sys.argv[1:4]
try:
    d = json.loads(line)
    print(json.dumps({"env": envs, "args": args, "value": round(d["value"], 1), "ms_per_step": round(d["ms_per_step"], 4),
                      "median": round(d.get("value_median_over_windows") or 0, 1),
                      "one_lane": round(d.get("value_one_batch_in_flight") or 0, 1),
                      "checksum": d.get("embedding_checksum")}))
except Exception as e:
    print(json.dumps({"env": envs, "args": args, "error": str(e)[:100], "line": line[:200]}))
PY
  tail -1 "$OUT"
}
pts=("$@")
for p in "${pts[@]}"; do point "$p"; done
for ((i=${#pts[@]}-1; i>=0; i--)); do point "${pts[$i]}"; done
