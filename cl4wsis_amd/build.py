"""Build libcl4wsis_hip.so (gfx950) from cl4wsis_amd/csrc/*.hip with hipcc.

The library is built IN-TREE (cl4wsis_amd/libcl4wsis_hip.so) so that it travels with a
snapshot of the repository to a GPU box; objects are cached under cl4wsis_amd/csrc/_obj.
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libcl4wsis_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wno-unused-result"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libcl4wsis_hip.so")
    return exe


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def _compile(src, obj, hipcc):
    cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build_library(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "cl4wsis.h"))
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj, headers):
            jobs.append((src, obj))
    if jobs:
        if verbose:
            print(f"[cl4wsis_amd.build] compiling {len(jobs)} HIP source(s) for {ARCH}", file=sys.stderr)
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(lambda j: _compile(j[0], j[1], hipcc), jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def build_variant(name, defines, sources=("conv_igemm.hip",)):
    """Experiments only: libcl4wsis_<name>.so = the product objects with `sources` recompiled under extra -D flags
    (loaded through CL4_LIB by the probes under tools/)."""
    hipcc = _hipcc()
    build_library(verbose=False)
    vdir = os.path.join(OBJ, name)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for s in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        if s in sources:
            obj = os.path.join(vdir, s[:-4] + ".o")
            cmd = [hipcc] + FLAGS + [f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, s), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        else:
            obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
    lib = os.path.join(HERE, f"libcl4wsis_{name}.so")
    r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":      # build.py --variant tl2 [--src conv_u.hip] CL4_TL2=1 [more defines]
        rest = sys.argv[3:]
        srcs = ("conv_igemm.hip",)
        if rest and rest[0] == "--src":
            srcs, rest = tuple(rest[1].split(",")), rest[2:]
        print(build_variant(sys.argv[2], rest, srcs))
    else:
        print(build_library(force="--force" in sys.argv))
