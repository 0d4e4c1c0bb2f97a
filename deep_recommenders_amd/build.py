"""Builds the C-ABI shared library (include/dr_hotpath.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the resulting deep_recommenders_amd/lib/libdr_hotpath.so travels
to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libdr_hotpath.so")
HOST_SRC = os.path.join(HERE, "csrc_host")
HOST_SO = os.path.join(LIBDIR, "libdr_input.so")          # host-side input boundary (include/dr_input.h), plain g++
COLL_SRC = os.path.join(HERE, "csrc_coll")
COLL_SO = os.path.join(LIBDIR, "libdr_collectives.so")    # exchange steps over RCCL (include/dr_collectives.h), host C++
# -pragma-unroll-threshold: `#pragma unroll` means it (the register-split GEMM's epilogue over 2 x 8 accumulator tiles is larger than
# the default limit; a loop-indexed accumulator array that is NOT unrolled lives in scratch memory)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result", "-mllvm", "-pragma-unroll-threshold=131072"]
FLAGS += os.environ.get("DR_HIPCC_EXTRA", "").split()     # experiment builds only (tools/exp: -DDR_OCC_ABLATE, -DDR_BF3_ABLATE ...)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(SO, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_host(force=force, verbose=verbose)
    build_collectives(force=force, verbose=verbose)
    return SO


def build_collectives(force=False, verbose=False):
    """libdr_collectives.so: the C-ABI exchange steps over RCCL (host code only; linked against ROCm's librccl)."""
    srcs = sorted(glob.glob(os.path.join(COLL_SRC, "*.cpp")))
    headers = glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    if force or _stale(COLL_SO, srcs + headers):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I/opt/rocm/include"] + srcs + \
              ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-o", COLL_SO]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return COLL_SO


def build_host(force=False, verbose=False):
    """libdr_input.so: TFRecord / tf.Example reader (host C++, no GPU code)."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(HOST_SRC, "*.cpp")))
    headers = glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    if force or _stale(HOST_SO, srcs + headers):
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra",
               "-I" + os.path.join(HERE, "..", "include")] + srcs + ["-o", HOST_SO]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
