"""Build recipe for libofl_hip.so (hipcc, gfx950 only, in-tree so the .so travels with the repo)."""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SOURCES = [os.path.join(_PKG, "csrc", "ofl_kernels.hip"), os.path.join(_PKG, "csrc", "ofl_aux_kernels.hip"),
           os.path.join(_PKG, "csrc", "ofl_warp_wide.hip")]
HEADERS = [os.path.join(_ROOT, "include", "oflib_hip.h")]
LIB_PATH = os.path.join(_PKG, "libofl_hip.so")

# -ffp-contract=off / -fno-fast-math: the kernels restate the reference's fp32 operation order;
# the only fused multiply-adds are explicit.
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-fvisibility=hidden"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("oflibpytorch_amd: hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP kernels + C ABI into oflibpytorch_amd/libofl_hip.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc_path()] + HIPCC_FLAGS + ["-I", os.path.join(_ROOT, "include"), "-o", LIB_PATH] + SOURCES
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("oflibpytorch_amd: hipcc failed\n" + res.stdout + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
