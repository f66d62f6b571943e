#!/usr/bin/env python
"""Per-kernel resource table of the built library (no GPU needed): VGPRs, SGPRs, LDS, scratch, spills and the MFMA /
packed-fp32 instruction counts of every gfx950 kernel in wespeaker_amd/lib/libwespeaker_amd.so.

    python tools/isa_report.py > profiles/r04_isa_report.md
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wespeaker_amd import build  # noqa: E402


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:  # noqa: BLE001
        return {n: n for n in names}


def main():
    rocm = os.path.dirname(os.path.dirname(os.path.realpath(build._hipcc())))
    bindir = os.path.join(rocm, "lib", "llvm", "bin")
    rows = {}
    for co in build.gfx950_code_objects(build.LIB):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([os.path.join(bindir, "llvm-readelf"), "--notes", f.name], stdout=subprocess.PIPE,
                                   text=True).stdout
            dis = subprocess.run([os.path.join(bindir, "llvm-objdump"), "-d", f.name], stdout=subprocess.PIPE,
                                 text=True).stdout
        # amdhsa.kernels: one YAML list entry per kernel, keys in alphabetical order (.name sits in the middle)
        entry = None

        def flush(e):
            if e and e.get("name", "").startswith("_Z"):
                rows.setdefault(e["name"], {}).update({k: v for k, v in e.items() if k != "name"})
        for line in notes.splitlines():
            if re.match(r"^\s*- \.(agpr_count|args):", line):
                flush(entry)
                entry = {}
            if entry is None:
                continue
            m = re.match(r"^\s*(?:- )?\.name:\s+(\S+)\s*$", line)
            if m and m.group(1).startswith("_Z") and "name" not in entry:
                entry["name"] = m.group(1)
            for key in ("vgpr_count", "sgpr_count", "agpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                        "vgpr_spill_count", "max_flat_workgroup_size"):
                m = re.match(r"^\s*(?:- )?\.%s:\s+(\d+)" % key, line)
                if m:
                    entry[key] = int(m.group(1))
        flush(entry)
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
            if m:
                cur = rows.setdefault(m.group(1), {})
                continue
            if cur is None:
                continue
            if "v_mfma" in line:
                cur["mfma"] = cur.get("mfma", 0) + 1
            if re.search(r"\bv_pk_\w+_f32\b", line):
                cur["pk_f32"] = cur.get("pk_f32", 0) + 1
            if "global_load_lds" in line or re.search(r"buffer_load.*\blds\b", line):
                cur["lds_dma"] = cur.get("lds_dma", 0) + 1
    names = demangle(sorted(rows))
    print("| kernel | VGPR | AGPR | SGPR | LDS static (B) | scratch (B) | spilled VGPRs | MFMA instr | packed-fp32 instr | "
          "LDS-DMA instr | threads |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k in sorted(rows, key=lambda n: names[n]):
        r = rows[k]
        if "vgpr_count" not in r:
            continue
        nm = names[k]
        nm = nm.replace("(anonymous namespace)::", "").replace("wsamd::", "")
        if nm.endswith(")") and "(" in nm:                   # drop the parameter list, keep the template arguments
            nm = nm[:nm.rfind("(")]
        nm = nm.replace("void ", "")
        print("| `%s` | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (
            nm[:90], r.get("vgpr_count", 0), r.get("agpr_count", 0), r.get("sgpr_count", 0),
            r.get("group_segment_fixed_size", 0), r.get("private_segment_fixed_size", 0), r.get("vgpr_spill_count", 0),
            r.get("mfma", 0), r.get("pk_f32", 0), r.get("lds_dma", 0), r.get("max_flat_workgroup_size", 0)))


if __name__ == "__main__":
    main()
