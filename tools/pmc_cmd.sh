#!/bin/bash
# rocprofv3 PMC passes (kernel-trace only, one counter group per pass, as gpurun requires) of any command of this repo,
# per-kernel JSON with the counter values of every dispatch in dispatch order:
#   bash tools/pmc_cmd.sh gpurun_out/r05_pmc_plda.json "bench.py --plda-only --no-cpu-baseline --steps 3" \
#        "plda_llr_pairs_kernel|plda_gemm_f64" FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$(pwd); mkdir -p "$REPO/gpurun_out"
OUTJSON=$REPO/$1; CMD=$2; NEEDLES=$3; shift 3
cd /tmp && export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmcc_$i
  ( cd "$REPO" && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcc_$i -- python $CMD > /tmp/pmcc_$i.log 2>&1 ) || tail -3 /tmp/pmcc_$i.log
done
python - "$NEEDLES" "$CMD" "$OUTJSON" /tmp/pmcc_* <<'PY'
import sys, csv, glob, os, json, collections
needles, cmd, outp = sys.argv[1].split("|"), sys.argv[2], sys.argv[3]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[4:]:
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if any(n in r["Kernel_Name"] for n in needles)]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id") or 0))
        for r in rows:
            name = next(n for n in needles if n in r["Kernel_Name"])
            per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
json.dump({"command": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python " + cmd,
           "per_kernel_dispatch_values": per}, open(outp, "w"))
for k, c in per.items():
    for cn, v in sorted(c.items()):
        print("%-28s %-16s n=%3d avg=%.5g min=%.5g max=%.5g" % (k, cn, len(v), sum(v) / len(v), min(v), max(v)))
PY
