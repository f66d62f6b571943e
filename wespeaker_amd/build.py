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


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
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
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

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
    need_link = bool(jobs) or not os.path.exists(LIB) or force
    if need_link:
        # only the C-ABI of include/wespeaker_amd.h leaves the library (kernel handles and every internal
        # C++ symbol stay local)
        vmap = os.path.join(OBJDIR, "exports.map")
        with open(vmap, "w") as f:
            f.write("{ global: ws_*; local: *; };\n")
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + vmap,
               "-o", LIB] + objs
        rc, out = run(cmd)
        if rc != 0:
            raise RuntimeError("link failed:\n" + out)
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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
