#!/bin/bash
# PMC counters of one kernel (separate passes, kernel-trace only): bash tools/pmc_kernel.sh <needle> "<bench args>" "<counters pass 1>" ["<pass 2>" ...]
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
NEEDLE=$1; ARGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$i -- python "$REPO/bench.py" $ARGS --steps 2 --warmup 1 --windows 1 --headline-only --lanes 1 > /tmp/pmc_$i.log 2>&1 || tail -3 /tmp/pmc_$i.log
done
python - "$NEEDLE" /tmp/pmc_* <<'PY'
import sys, csv, glob, os, collections
needle = sys.argv[1]
vals = collections.defaultdict(list)
for d in sys.argv[2:]:
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(vals.items()):
    print("%-40s n=%4d avg=%.4g max=%.4g" % (k, len(v), sum(v) / len(v), max(v)))
PY
