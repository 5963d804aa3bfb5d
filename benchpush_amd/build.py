"""In-tree build of libbenchpush_hip.so (gfx950 only) with hipcc.  The .so is git-ignored but travels with gpurun."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libbenchpush_hip.so")
SOURCES = ["bp_capi.hip"]


def _headers():
    """Every header the translation unit can see: csrc/*.hpp plus the public C header."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(_HERE, "..", "include", "benchpush_amd.h")]

# -ffp-contract=off / -fno-fast-math: the physics must round exactly like the binary64 reference arithmetic.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-std=c++17",
               "-Wno-unused-value"]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for p in [os.path.join(CSRC, f) for f in SOURCES] + _headers():
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_hip(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


DBG_LIB_PATH = os.path.join(_HERE, "libbenchpush_hip_dbgpaths.so")


def build_debug_paths(force=False, verbose=False):
    """Diagnostic twin of the library (-DBP_DEBUG_PATHS): BP_DEBUG_PATHS=<mask> forces the narrow phase's fallback paths.  Only tests load it
    (tests/test_gpu_parity.py::test_rarely_taken_narrow_phase_paths_match_oracle); the product library compiles those tests away."""
    if not force and os.path.exists(DBG_LIB_PATH):
        t = os.path.getmtime(DBG_LIB_PATH)
        if all(os.path.getmtime(p) <= t for p in [os.path.join(CSRC, f) for f in SOURCES] + _headers() if os.path.exists(p)):
            return DBG_LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-DBP_DEBUG_PATHS=1", "-o", DBG_LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return DBG_LIB_PATH


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(build_debug_paths(force=True, verbose=True))
