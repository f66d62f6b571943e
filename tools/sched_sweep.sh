#!/bin/bash
# (utterances per step) x (batches in flight) sweep of the headline-only bench, one line per point:
#   gpurun --timeout 600 -- 'bash tools/sched_sweep.sh r04'
# The kernels are the same in every point; what changes is how many workgroups a launch has (one or two per CU for the
# one-workgroup-per-utterance kernels) and how many launches of different streams share the chip.
TAG=${1:-r04}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/${TAG}_sched_sweep.jsonl
mkdir -p gpurun_out; rm -f "$OUT"
point() {  # model precision batch lanes steps
  local line
  line=$(timeout 240 python bench.py --model $1 --precision $2 --batch $3 --lanes $4 --steps $5 --warmup 2 --windows 2 \
         --headline-only --no-cpu-baseline --sustain-s 0 2> /dev/null | tail -1)
  python - "$1" "$2" "$3" "$4" "$line" >> "$OUT" <<'PY'
import json, sys
m, prec, b, l, line = sys.argv[1:6]
try:
    d = json.loads(line)
    print(json.dumps({"model": m, "precision": prec, "batch": int(b), "lanes": int(l), "value": round(d["value"], 1),
                      "ms_per_step": round(d["ms_per_step"], 4), "median": round(d.get("value_median_over_windows") or 0, 1),
                      "one_lane": round(d.get("value_one_batch_in_flight") or 0, 1)}))
except Exception as e:
    print(json.dumps({"model": m, "precision": prec, "batch": int(b), "lanes": int(l), "error": str(e)[:100]}))
PY
  tail -1 "$OUT"
}
point CAMPPlus fp32 512 2 8
point CAMPPlus fp32 256 2 16
point CAMPPlus fp32 256 4 16
point CAMPPlus fp32 512 3 8
point CAMPPlus fp32 128 4 32
point ResNet34 fp32 512 2 4
point ResNet34 fp32 256 4 8
point ResNet221 fp32 256 2 3
point ResNet221 fp32 128 4 6
point ECAPA_TDNN_GLOB_c512 fp32 256 2 20
point ECAPA_TDNN_GLOB_c512 fp32 256 3 20
point ECAPA_TDNN_GLOB_c512 fp32 128 4 40
point ECAPA_TDNN_GLOB_c512 f16 256 3 20
