"""Build the gfx950 C-ABI library in-tree with hipcc (cross-compiles without a GPU).

    python -m wespeaker_amd.build            # -> wespeaker_amd/lib/libwespeaker_amd.so

The .so is git-ignored but travels to the GPU box with the repo snapshot.  Objects are rebuilt
only when their source (or a header) is newer.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libwespeaker_amd.so")
MAIN_BIN = os.path.join(LIBDIR, "extract_emb_main")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _extra_flags(src):
    """Per-file compiler flags: a `// WS_BUILD_FLAGS: ...` line among the first lines of the source."""
    with open(src) as f:
        for _ in range(5):
            line = f.readline()
            if line.startswith("// WS_BUILD_FLAGS:"):
                return line.split(":", 1)[1].split()
    return []


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "wespeaker_amd.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_time = _newest_header()
    jobs, objs = [], []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s[:-4] + ".o")
        objs.append(obj)
        if (force or not os.path.exists(obj)
                or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time)):
            jobs.append([hipcc] + FLAGS + _extra_flags(src) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return r.returncode, r.stdout

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        results = list(ex.map(run, jobs))
    for (rc, out), cmd in zip(results, jobs):
        if out.strip() and verbose:
            print(out)
        if rc != 0:
            raise RuntimeError("compile failed: %s\n%s" % (" ".join(cmd), out))
    # (after a build whose ISA check failed the objects are up to date and the OLD library is still in
    # place -- "no jobs" must not mean "nothing to link": any object newer than the library forces the link + check)
    need_link = (bool(jobs) or not os.path.exists(LIB) or force
                 or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs))
    if need_link:
        # only the C-ABI of include/wespeaker_amd.h leaves the library (kernel handles and every internal
        # C++ symbol stay local)
        vmap = os.path.join(OBJDIR, "exports.map")
        with open(vmap, "w") as f:
            f.write("{ global: ws_*; local: *; };\n")
        # linked under a temporary name, ISA-checked there, and only then moved into place: a library that fails the
        # check never sits at LIB, where the next build() (no jobs, LIB present) would have returned it unchecked
        tmp_lib = LIB + ".tmp"
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + vmap,
               "-o", tmp_lib] + objs
        rc, out = run(cmd)
        if rc != 0:
            raise RuntimeError("link failed:\n" + out)
        bad = check_isa(tmp_lib)
        if bad:
            os.remove(tmp_lib)
            raise RuntimeError("forbidden instruction forms / pairs in the library (DESIGN.md 6.0):\n" +
                               "\n".join("  %s: %d x %s" % b for b in bad))
        os.replace(tmp_lib, LIB)
    # the C++ caller of the C-ABI (native twin of runtime/core/bin/extract_emb_main.cc)
    main_src = os.path.join(CSRC, "bin", "extract_emb_main.cc")
    if (need_link or not os.path.exists(MAIN_BIN)
            or os.path.getmtime(MAIN_BIN) < max(os.path.getmtime(main_src), hdr_time)):
        # a plain host compiler on purpose: the boundary is C, the caller needs no hipcc
        rocm = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
               "-I" + os.path.join(rocm, "include"), main_src, "-L" + LIBDIR, "-lwespeaker_amd",
               "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath,$ORIGIN",
               "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", MAIN_BIN]
        rc, out = run(cmd)
        if rc != 0:
            raise RuntimeError("extract_emb_main build failed:\n" + out)
    return LIB


# Kernels that may hold the forbidden form: the reproducer build of the fbank kernel (ws_debug_fbank_mode(1)).
ISA_CHECK_EXEMPT = ("fbank_kernel_packed",)


def gfx950_code_objects(lib_path):
    """The gfx950 code objects of a hipcc-linked library: every translation unit leaves one clang offload bundle
    (magic, entry count, then (offset, size, triple) per entry) in .hip_fatbin."""
    import struct
    data = open(lib_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos, out = 0, []
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return out
        cnt, = struct.unpack_from("<Q", data, i + 24)
        off = i + 32
        for _ in range(cnt):
            o, sz, ts = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + ts].decode("ascii", "replace")
            off += ts
            if ARCH in triple and sz:
                out.append(data[i + o:i + o + sz])
        pos = i + len(magic)


def check_isa(lib_path=LIB):
    """on MI355X a packed-fp32 instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) whose op_sel
    starts [0,1 -- low result from src0's low half and src1's HIGH half -- returns wrong values in lanes 48..63 while
    binary16 GEMM kernels of another stream share the CU (tools/pk_probe.py: every such form, no other of the 33 forms
    tried).  hipcc's SLP vectoriser emits it freely, so the linked library is disassembled and searched.  Returns
    [(kernel, count, mnemonic)] of the offenders outside ISA_CHECK_EXEMPT."""
    import re
    import tempfile
    rocm = os.path.dirname(os.path.dirname(os.path.realpath(_hipcc())))
    objdump = os.path.join(rocm, "lib", "llvm", "bin", "llvm-objdump")
    if not os.path.exists(objdump):
        objdump = shutil.which("llvm-objdump") or objdump
    cos = gfx950_code_objects(lib_path)
    if not cos:
        raise RuntimeError("no %s code object found in %s" % (ARCH, lib_path))
    hits = {}
    pat = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\b.*op_sel:\[0,1")
    sym = re.compile(r"^[0-9a-f]+ <(.*)>:")
    for co in cos:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([objdump, "-d", f.name], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                 text=True, check=True).stdout
        cur = "?"
        lines = txt.splitlines()
        for li, line in enumerate(lines):
            m = sym.match(line)
            if m:
                cur = m.group(1)
                continue
            m = pat.search(line)
            if m and not any(e in cur for e in ISA_CHECK_EXEMPT):
                hits[(cur, m.group(1))] = hits.get((cur, m.group(1)), 0) + 1
            if _store_data_overwritten(lines, li):
                key = (cur, "wide store + VALU write of its data registers")
                hits[key] = hits.get(key, 0) + 1
    return sorted((k, n, op) for (k, op), n in hits.items())


_WIDE_STORE = None


def _store_data_overwritten(lines, li):
    """A second rule (round 5): a store of more than 64 bits whose data VGPRs a VALU instruction
    overwrites within the next two wait states.  LLVM pads that pair with s_nop only when the store has no scalar
    offset register; `buffer_store_dwordx4 v[12:15], v185, s[28:31], s0 offen` directly in front of
    `v_pk_add_f32 v[12:13], ..` went out unpadded and MI355X stored the new values of some lanes
    (tools/conv_stream_probe: the last tile of every workgroup of the persistent kernel's CONV form)."""
    import re
    global _WIDE_STORE
    if _WIDE_STORE is None:
        _WIDE_STORE = (re.compile(r"^\s*((?:buffer|global|flat|scratch)_store_dwordx[34])\s+([^/]*)"),
                       re.compile(r"v\[(\d+):(\d+)\]"), re.compile(r"^v(\d+)$"))
    st, vrange, vone = _WIDE_STORE
    m = st.match(lines[li])
    if not m:
        return False
    ops = [x.strip() for x in m.group(2).split(",")]
    data = ops[0] if m.group(1).startswith("buffer") else (ops[1] if len(ops) > 1 else "")
    dm = vrange.match(data)
    if not dm:
        return False
    lo, hi = int(dm.group(1)), int(dm.group(2))
    waited = 0
    for nxt in lines[li + 1:li + 4]:
        ins = nxt.split("//")[0].strip()
        if not ins:
            continue
        if ins.startswith("s_nop"):
            waited += int(ins.split()[1], 0) + 1
        elif ins.startswith("v_") and not ins.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            dst = ins.split(None, 1)[1].split(",")[0].strip() if " " in ins else ""
            wm = vrange.match(dst) or vone.match(dst)
            if wm:
                wlo = int(wm.group(1))
                whi = int(wm.group(2)) if wm.re is vrange else wlo
                if wlo <= hi and whi >= lo:
                    return waited < 2
            waited += 1
        else:
            waited += 1
        if waited >= 2:
            return False
    return False


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
