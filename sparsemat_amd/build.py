"""Builds sparsemat_amd/libsparsemat_hip.so with hipcc for gfx950 (in-tree, no JIT cache).

    python -m sparsemat_amd.build [--force]

hipcc cross-compiles without a GPU.  The library links against the HIP runtime and RCCL (librccl.so.1: the
multi-GPU exchange of csrc/par.hip).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsparsemat_hip.so")
SOURCES = ["capi.hip", "spmv_vector.hip", "spmv_ring.hip", "spmv_ring2.hip", "spmv_stream.hip", "spmv_stream_xd.hip", "spmv_colblock.hip", "spmv_colfused.hip", "spmv_colsplit.hip", "spmv_tiled.hip", "spmv_merge.hip", "blas1.hip", "cg.hip", "synth.hip", "assemble.hip", "matops.hip", "par.hip", "pcg.hip", "pool.hip", "transpose_bucket.hip"]
HEADERS = [os.path.join(CSRC, "internal.hpp"), os.path.join(HERE, "..", "include", "sparsemat_hip.h")]
# -ffp-contract=off: the element-wise / SEQ kernels must round a*b and (a*b)+c separately, like the
# reference; where an FMA is wanted (K1's accumulation) the kernels call __builtin_fma explicitly.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=off"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None, objsuffix=""):
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path.
    extra_flags/out/objsuffix build an A/B variant next to the product (development aid)."""
    global LIB
    lib = out or LIB
    if out is None and not force and not stale():
        return LIB
    objdir = os.path.join(HERE, "csrc", "_obj" + objsuffix)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc()] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for s, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            failed.append((s, log.decode(errors="replace")))
        elif verbose and log:
            print(log.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join("%s:\n%s" % f for f in failed))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
