"""Builds libadmm_hip.so in-tree (admm-elastic-sca_amd/libadmm_hip.so).

  dense.cpp, factor.cpp                  : host-side ordering / multifrontal factorization, g++ -O3 -fopenmp
  host_setup.cpp, partition.cpp, comm.cpp : host-only parts of the library around the context (ctx.hpp): assembly + factor set-up,
                                           subtree sharding, RCCL / all-reduce hooks -- hipcc as a host compile (HIP runtime API only)
  admm_hip.hip                           : the device translation unit: kernels (kernels_*.hpp, factor_dev.hpp), device factorization,
                                           upload, launches, step loop, C ABI; hipcc --offload-arch=gfx950, -ffp-contract=off
                                           (operation order = reference's)

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only
container as well as on the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libadmm_hip.so")
OBJ = os.path.join(HERE, "_build")

HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fopenmp", "-Wall", "-Wno-unknown-pragmas"]
# -disable-promote-alloca-to-vector keeps the L-BFGS history (a run-time indexed
# private array, local_math.hpp) in scratch instead of 80 VGPRs: measured -15 % on
# the tet kernel together with __launch_bounds__(256, 3) (tools/ab_local.py).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
             "-Wno-unused-function", "-Wno-unused-result", "-Wno-comment", "-Wno-bitwise-instead-of-logical",
             "-mllvm", "-disable-promote-alloca-to-vector"]


# host-only units that see the HIP runtime API (hipStream_t & co. in ctx.hpp): hipcc with no offload target
HIPHOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-comment", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=False, extra_hip_flags=(), out=None, tag=""):
    """extra_hip_flags/out/tag build an experimental variant next to the default library."""
    global OUT
    out = out or OUT
    os.makedirs(OBJ, exist_ok=True)
    # one builder at a time (the ranks of a multi-process launch all come through here): the others wait on the lock and then
    # find everything up to date; the library is linked under a temporary name and renamed into place, so a process that is
    # dlopen'ing it never sees a half-written file
    import fcntl
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, extra_hip_flags, out, tag)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, extra_hip_flags, out, tag):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h", ".inc"))]
    headers += [os.path.join(HERE, "..", "include", f) for f in ("admm_hip.h", "admm_kinds.h")]
    jobs = [
        (["g++"] + HOST_FLAGS + ["-mavx2", "-mfma"], "dense.cpp", "dense.o"),
        (["g++"] + HOST_FLAGS, "factor.cpp", "factor.o"),
        ([hipcc] + HIPHOST_FLAGS, "host_setup.cpp", "host_setup.o"),
        ([hipcc] + HIPHOST_FLAGS, "partition.cpp", "partition.o"),
        ([hipcc] + HIPHOST_FLAGS, "comm.cpp", "comm.o"),
        ([hipcc] + HIP_FLAGS + list(extra_hip_flags), "admm_hip.hip", "admm_hip%s.o" % tag),
    ]
    objs = []
    rebuilt = False
    for cmd, src, obj in jobs:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, obj)
        objs.append(o)
        if force or _newer([s] + headers, o):
            full = cmd + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(full))
            subprocess.check_call(full, stderr=subprocess.DEVNULL if not verbose else None)
            rebuilt = True
    if rebuilt or not os.path.exists(out):
        tmp = "%s.tmp.%d" % (out, os.getpid())
        full = [hipcc, "--offload-arch=gfx950", "-shared", "-o", tmp] + objs + ["-lgomp"]
        if verbose:
            print(" ".join(full))
        subprocess.check_call(full, stderr=subprocess.DEVNULL if not verbose else None)
        os.replace(tmp, out)
    return out


def source_hash():
    """sha256 over the library's sources (csrc/*, include/*.h): stamps counter files (tools/pmc_collect.sh) so that bench.py can tell
    whether the committed PMC summary was collected on the kernels it is running."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h", ".hip", ".cpp", ".inc")))
    files += [os.path.join(HERE, "..", "include", f) for f in ("admm_hip.h", "admm_kinds.h")]
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
