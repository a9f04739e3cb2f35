"""Build recipe for libofl_hip.so (hipcc, gfx950 only, in-tree so the .so travels with the repo)."""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SOURCES = [os.path.join(_PKG, "csrc", "ofl_kernels.hip"), os.path.join(_PKG, "csrc", "ofl_aux_kernels.hip"), os.path.join(_PKG, "csrc", "ofl_splat_gather.hip"),
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
    # the three translation units are compiled side by side (ofl_kernels.hip alone is ~80 s of device code generation: the splat
    # gather kernel has 16 instantiations), then linked
    import tempfile
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="ofl_build_") as tmp:
        objs, procs = [], []
        for src in SOURCES:
            obj = os.path.join(tmp, os.path.splitext(os.path.basename(src))[0] + ".o")
            cmd = [hipcc_path()] + compile_flags + ["-I", os.path.join(_ROOT, "include"), "-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            objs.append(obj)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        for src, pr in procs:
            out, err = pr.communicate()
            if pr.returncode != 0:
                raise RuntimeError("oflibpytorch_amd: hipcc failed on %s\n%s%s" % (src, out, err))
        link = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(link), flush=True)
        res = subprocess.run(link, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("oflibpytorch_amd: hipcc (link) failed\n" + res.stdout + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
