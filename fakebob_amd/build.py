"""Builds fakebob_amd/lib/libfakebob_hip.so with hipcc for gfx950 (in-tree).

The HIP sources are compiled with -ffp-contract=off: the float64 NES arithmetic
and the float32 Box-Muller must round exactly like NumPy / the CPU oracle, and
every fused multiply-add in the kernels is written explicitly.
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libfakebob_hip.so")
SOURCES = ["nes_kernels.hip", "frontend_kernels.hip", "frontend_f32_kernels.hip", "gmm_kernels.hip", "gmm_wide_kernel.hip", "ivector_kernels.hip", "ivector_solve.hip",
           "fb_engine.hip"]
# per-source flags.  k_gmm_fx2w keeps its MFMA accumulators in vector registers (the logsumexp update reads them in
# place) and its parked frame operands in accumulation registers: hipcc picks that form of the MFMA with this option
SOURCE_FLAGS = {"gmm_wide_kernel.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libfakebob_hip.so)")


def _deps():
    out = []
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip", ".cpp")):
                out.append(os.path.join(root, f))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _deps())


# variants of the library next to the product build: name -> (extra flags, sources they change; the other objects are the
# product build's).  "fenced": every cross-workgroup exchange with release / acquire semantics (csrc/fb_device.h, FB_FENCED) --
# tests/test_gpu_fenced.py runs it against the product build and asserts identical bits.
VARIANTS = {"fenced": (["-DFB_FENCED"], ["nes_kernels.hip", "frontend_kernels.hip", "gmm_kernels.hip", "ivector_solve.hip", "ivector_kernels.hip"])}


def variant_path(name):
    return os.path.join(LIBDIR, "libfakebob_hip_%s.so" % name)


def build_variant(name, force=False):
    """the product build first (its objects are reused), then the variant's own objects and library"""
    build()
    flags, own = VARIANTS[name]
    lib = variant_path(name)
    if not force and os.path.exists(lib) and not any(os.path.getmtime(f) > os.path.getmtime(lib) for f in _deps()):
        return lib
    hipcc = _hipcc()
    odir = os.path.join(OBJDIR, name)
    os.makedirs(odir, exist_ok=True)

    def comp(src):
        obj = os.path.join(odir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + SOURCE_FLAGS.get(src, []) + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s (%s):\n%s" % (src, name, r.stdout))
        return obj

    with cf.ThreadPoolExecutor(max_workers=len(own)) as ex:
        objs = list(ex.map(comp, own))
    objs += [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in SOURCES if s not in own]
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed (%s):\n%s" % (name, r.stdout))
    return lib


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def comp(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        extra = os.environ.get("FB_EXTRA_HIPCC_FLAGS", "").split()
        cmd = [hipcc] + FLAGS + SOURCE_FLAGS.get(src, []) + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return src, obj, r.returncode, r.stdout

    objs = []
    with cf.ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        for src, obj, rc, out in ex.map(comp, srcs):
            if verbose and out.strip():
                print(out)
            if rc != 0:
                raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
            objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
