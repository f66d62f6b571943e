#!/usr/bin/env python
"""gpurun_out/r05_pmc_plda_raw.json (tools/pmc_cmd.sh on `bench.py --plda-only --no-cpu-baseline --steps 3`) ->
profiles/r05_pmc_plda.json: fabric-side traffic and L2 hit rate per launch of the PLDA kernels.

Dispatch order of that command (k = 3 timed launches behind 2 warm-up launches, twice per list): plda_llr_pairs_kernel
dispatches 0..9 are the RANDOM-order list, 10..19 the list GROUPED by enrollment model; ws_debug_row_gather's kernel is
the bench's gather yardstick; plda_gemm_f64_big: the 10 000 x 10 000 dense legs (lowest FETCH_SIZE = D 192, highest =
D 512).  FETCH_SIZE (KB) counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM section) and is doubled;
WRITE_SIZE (KB) is taken as is -- the dense kernel's 781 250 KB = its 8e8 output bytes exactly, which calibrates it.
Infinity-Cache hits are counted by these memory-side counters, not excluded: `fabric` below means "left the L2"."""
import json
import sys


def stats(per, name, sl=slice(None)):
    c = per[name]
    out = {}
    for k, v in c.items():
        v = v[sl]
        out[k] = sum(v) / len(v)
    fetch = 2 * 1024 * out["FETCH_SIZE"]
    write = 1024 * out["WRITE_SIZE"]
    return {"dispatches": len(c["FETCH_SIZE"][sl]), "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
            "traffic_bytes_per_launch": fetch + write,
            "l2_requests_x128B_bytes": 128 * out["TCC_REQ_sum"], "l2_hit_rate": out["TCC_HIT_sum"] /
            (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]), "l2_miss_x128B_bytes": 128 * out["TCC_MISS_sum"],
            "grbm_gui_active_cycles": out["GRBM_GUI_ACTIVE"]}


def main():
    raw = json.load(open(sys.argv[1]))
    per = raw["per_kernel_dispatch_values"]
    n = len(per["plda_llr_pairs_kernel"]["FETCH_SIZE"])
    out = {"command": raw["command"],
           "counters": "separate passes: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | TCC_REQ_sum GRBM_GUI_ACTIVE",
           "plda_llr_pairs_random_list": stats(per, "plda_llr_pairs_kernel", slice(0, n // 2)),
           "plda_llr_pairs_grouped_list": stats(per, "plda_llr_pairs_kernel", slice(n // 2, n)),
           "row_gather_yardstick": stats(per, "row_gather"),
           "plda_gemm_f64_big": stats(per, "plda_gemm_f64_big")}
    big = per["plda_gemm_f64_big"]["FETCH_SIZE"]
    out["plda_gemm_f64_big"]["fetch_bytes_corrected_x2_min_max"] = [2 * 1024 * min(big), 2 * 1024 * max(big)]
    out["reading"] = ("1e6 random trials over two 10 000 x 192 float64 tables (30.7 MB): the L2 (4 MB per XCD) serves "
                      "only ~1/5 of the row gathers, the rest leave it -- to the 256-MB Infinity Cache, which holds both "
                      "tables whole, and behind it HBM (these counters cannot tell the two apart)")
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
