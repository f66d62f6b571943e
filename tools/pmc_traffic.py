#!/usr/bin/env python
"""Aggregate rocprofv3 PMC passes of `bench.py` into profiles/r01_pmc_dominant_kernel_<prec>.json.

Run on the GPU box (counters in SEPARATE passes, kernel-trace only, as gpurun requires):

    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" ; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$i -- \
          python bench.py --precision f16 --steps 3 --warmup 1 --windows 1 --headline-only [--model M] ; done
    python tools/pmc_traffic.py f16 "gemm_f16_dma_kernel<128, 128|gemm_f16_p8_kernel" $OUT/pmc_* \
        > profiles/r01_pmc_dominant_kernel_f16.json

The needle may name several kernels ("a|b"): the dominant class of the f16 back-end is the 128x128 LDS-DMA
GEMM plus the phase-staggered 256x256 GEMM of the wide layer.  Counters are averaged per DISPATCH of any of
them (bench.py's roofline launches are launch_conv_gemm calls: 10 per step, 11 dispatches).

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced read (MI355X_MICROARCH.md, HBM section) and is doubled here; WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    prec, needles = sys.argv[1], sys.argv[2].split("|")
    vals = collections.defaultdict(list)
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    names = set()
    # whole forward: every wsamd:: dispatch, divided by the number of forwards (= fbank_kernel dispatches)
    whole = collections.defaultdict(float)
    forwards = collections.defaultdict(int)
    launches = collections.defaultdict(int)
    for d in sys.argv[3:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "wsamd::" in r["Kernel_Name"]:
                    whole[r["Counter_Name"]] += float(r["Counter_Value"])
                    launches[r["Counter_Name"]] += 1
                    if "fbank_kernel" in r["Kernel_Name"]:
                        forwards[r["Counter_Name"]] += 1
                if any(n in r["Kernel_Name"] for n in needles):
                    names.add(r["Kernel_Name"])
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    per_kernel[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {k: sum(v) / len(v) for k, v in vals.items()}
    out = {"command": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python bench.py --precision %s "
                      "--steps 3 --warmup 1 --windows 1 --headline-only [--model M]" % prec,
           "kernel": sorted(names), "launches_profiled": max(len(v) for v in vals.values()),
           "counters_avg_per_launch": avg,
           "per_kernel": {k: {"dispatches": max(len(x) for x in c.values()),
                              "counters_avg": {cn: sum(x) / len(x) for cn, x in c.items()}}
                          for k, c in per_kernel.items()}}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        out["fetch_bytes_corrected_x2"] = 2 * 1024 * avg["FETCH_SIZE"]
        out["write_bytes"] = 1024 * avg["WRITE_SIZE"]
        out["traffic_bytes_per_launch"] = out["fetch_bytes_corrected_x2"] + out["write_bytes"]
        out["note"] = ("gfx950: FETCH_SIZE reports half the bytes of wide coalesced streaming reads "
                       "(MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE uncalibrated, taken as is")
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals and forwards.get("FETCH_SIZE") and forwards.get("WRITE_SIZE"):
        # the dominant kernels' bytes per forward: bench.py divides by ITS launches per step (the kernel names
        # also match the two split-K dispatches of the embedding layers: bytes negligible, count not)
        out["dominant_bytes_per_forward"] = (2 * 1024 * sum(vals["FETCH_SIZE"]) / forwards["FETCH_SIZE"] +
                                             1024 * sum(vals["WRITE_SIZE"]) / forwards["WRITE_SIZE"])
    if whole.get("FETCH_SIZE") and whole.get("WRITE_SIZE") and forwards.get("FETCH_SIZE"):
        nf = forwards["FETCH_SIZE"]
        fb, wb = 2 * 1024 * whole["FETCH_SIZE"] / nf, 1024 * whole["WRITE_SIZE"] / forwards["WRITE_SIZE"]
        out["whole_forward"] = {"forwards_profiled": nf, "kernel_launches_per_forward": launches["FETCH_SIZE"] / nf,
                                "fetch_bytes_corrected_x2": fb, "write_bytes": wb, "traffic_bytes": fb + wb,
                                "note": "all wsamd:: kernels of one fbank -> embedding pass at the bench batch"}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg:
        # busy cycles are summed over 1024 SIMDs, GRBM_GUI_ACTIVE over 8 XCDs
        out["mfma_busy_fraction_of_cycles"] = (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (avg["GRBM_GUI_ACTIVE"] / 8.0)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
