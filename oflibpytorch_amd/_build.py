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
    # (the link line is derived from the same list: what is not a code-generation or language option of the compile step)
    link_flags = [f for f in HIPCC_FLAGS if f.startswith("--offload-arch") or f in ("-shared", "-fPIC", "-fvisibility=hidden")]
    with tempfile.TemporaryDirectory(prefix="ofl_build_") as tmp:
        objs, procs = [], []
        try:
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
        finally:
            # a failed translation unit must not leave its siblings writing into a directory that is being deleted
            for _, pr in procs:
                if pr.poll() is None:
                    pr.kill()
                    pr.communicate()
        # linked next to the library and moved over it in ONE step: ranks that start together with a stale library may each rebuild
        # it, but none of them ever dlopens a half-written file (os.replace is atomic within a directory)
        tmp_lib = "%s.%d.tmp" % (LIB_PATH, os.getpid())
        link = [hipcc_path()] + link_flags + ["-o", tmp_lib] + objs
        if verbose:
            print(" ".join(link), flush=True)
        res = subprocess.run(link, capture_output=True, text=True)
        if res.returncode != 0:
            if os.path.exists(tmp_lib):
                os.remove(tmp_lib)
            raise RuntimeError("oflibpytorch_amd: hipcc (link) failed\n" + res.stdout + res.stderr)
        os.replace(tmp_lib, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
