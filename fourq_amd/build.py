"""Builds libfourq_amd.so (HIP, gfx950 only) in-tree with hipcc.

    python -m fourq_amd.build            # or: from fourq_amd.build import build_library

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libfourq_amd.so")
# four translation units: FQ_CHAIN=0 / 1 (kernels.hip.h), and the constant-time-selection builds of both flavours
SOURCES = ["fourq_amd.hip", "fourq_chain.hip", "fourq_ct_fused.hip", "fourq_ct_chain.hip"]
HEADERS = ["fp127.hip.h", "curve.hip.h", "recode.hip.h", "kernels.hip.h", "constants.inc", os.path.join("..", "..", "include", "fourq_amd.h")]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fvisibility=hidden"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the FourQ engine needs the ROCm toolchain to build")
    return exe


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(SRC_DIR, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(os.path.normpath(d)) > built for d in deps)


def build_library(force=False, verbose=False, extra_flags=(), out_path=None):
    """Compile the library if it is missing or older than its sources; returns its path.  `out_path` + `extra_flags`
    build an experiment variant beside the product library (tools/ab_bench.sh compares them on one GPU box)."""
    if out_path is None and not force and not is_stale():
        return LIB_PATH
    lib_path = out_path or LIB_PATH
    suffix = "" if out_path is None else "." + os.path.splitext(os.path.basename(out_path))[0]
    objs, procs = [], []
    for src in SOURCES:                                   # compile the translation units in parallel (~1 min each)
        obj = os.path.join(SRC_DIR, os.path.splitext(src)[0] + suffix + ".o")
        cmd = [_hipcc()] + HIPCC_FLAGS + list(extra_flags) + ["-c", "-o", obj, os.path.join(SRC_DIR, src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for cmd, proc in procs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(link), file=sys.stderr)
    proc = subprocess.run(link, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("link failed:\n" + proc.stdout + proc.stderr)
    return lib_path


if __name__ == "__main__":
    # python -m fourq_amd.build [--force] [--out variants/libX.so -DFLAG=1 ...]
    argv = sys.argv[1:]
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    flags = [a for a in argv if a.startswith("-D") or a.startswith("-m")]
    print(build_library(force="--force" in argv, verbose=True, extra_flags=flags, out_path=os.path.abspath(out) if out else None))
