"""Builds the two native libraries of the package for gfx950:

* ``libnfft_hip.so`` -- the C-ABI library of include/nfft_hip.h (kernels + drivers).  Plain hipcc, no torch
  headers: the library's ABI is C.  Sources are compiled in parallel and relinked only when something changed.
* ``core.so`` -- the torch operator registry (csrc/core.cpp: ``TORCH_LIBRARY(torch_nfft, ...)``) on top of it,
  the counterpart of the reference's ``torch_nfft/core.so`` (setup.py:14-29, no ABI suffix).  Host code only
  (g++ against the torch headers); it finds libnfft_hip.so next to itself (``$ORIGIN``).

Usage: ``python torch_nfft_amd/build.py [--force]`` (run as a script so that a stale or missing library cannot
block its own rebuild through the package import).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libnfft_hip.so")
CORE = os.path.join(HERE, "core.so")
CORE_SRC = os.path.join(CSRC, "core.cpp")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
SOURCES = ["api.hip", "binning.hip", "spread.hip", "spread_reg.hip", "spread_mfma.hip", "interp.hip", "interp_mfma.hip", "interp_cols.hip", "interp_stream.hip", "smallgrid.hip", "spectral.hip", "colfft.hip", "coeffs.hip", "selftest.hip", "fft.cpp"]
# -ffp-contract=on: multiply-adds are fused only where one expression says so.  Until round 4 the library was built with
# =fast (fusion across statements), which silently broke an error-free transformation twice (the f16 split in round 2,
# split_cell in round 3: DESIGN.md section 5); the transformations are inline asm now and pinned bit for bit by
# tests/test_gpu_eft.py, and =on costs nothing measurable (profiles/r04_experiments.md).
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=on", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROCM, "include")]


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "nfft_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force, hdr_mtime):
    obj = os.path.join(OBJ, src + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_mtime)):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", path, "-o", obj]
    subprocess.check_call(cmd)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr = _headers_mtime()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda s: _compile(s, force, hdr), SOURCES))
    objs = [r[0] for r in results]
    if force or any(r[1] for r in results) or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-Wl,-soname,libnfft_hip.so", "-o", LIB] + objs + \
              ["-L" + os.path.join(ROCM, "lib"), "-lrocfft", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        subprocess.check_call(cmd)
        if verbose:
            print("built", LIB)
    build_core(force, hdr, verbose)
    build_variants(objs, force, hdr, verbose)
    return LIB


# Variant libraries for tests that must provoke a device-side failure path: the same objects with ONE source
# recompiled under a test macro.  Same soname as the product library, so a process that loads the variant first
# (NFFT_HIP_LIB) binds core.so to it.
VARIANT_DIR = os.path.join(os.path.dirname(HERE), "tests", "variants")
VARIANTS = {"spin0": ("interp_stream.hip", ["-DNFFT_HIP_SPIN_LIMIT=0"])}


def build_variants(objs, force=False, hdr_mtime=0.0, verbose=True):
    os.makedirs(VARIANT_DIR, exist_ok=True)
    for name, (src, defs) in VARIANTS.items():
        lib = os.path.join(VARIANT_DIR, "libnfft_hip_%s.so" % name)
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, "%s.%s.o" % (src, name))
        if not force and os.path.exists(lib) and os.path.getmtime(lib) >= max(os.path.getmtime(LIB), os.path.getmtime(path), hdr_mtime):
            continue
        subprocess.check_call([HIPCC] + FLAGS + defs + ["-x", "hip", "-c", path, "-o", obj])
        others = [o for o in objs if os.path.basename(o) != src + ".o"]
        subprocess.check_call([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-Wl,-soname,libnfft_hip.so", "-o", lib] +
                              others + [obj, "-L" + os.path.join(ROCM, "lib"), "-lrocfft", "-Wl,-rpath," + os.path.join(ROCM, "lib")])
        if verbose:
            print("built", lib)


def build_core(force=False, hdr_mtime=0.0, verbose=True):
    """core.so: the TORCH_LIBRARY operator registry (host C++ only) linked against libnfft_hip.so."""
    newest = max(os.path.getmtime(CORE_SRC), os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "nfft_hip.h")))
    if not force and os.path.exists(CORE) and os.path.getmtime(CORE) >= newest:
        return CORE
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-isystem" + p for p in ce.include_paths()] + ["-isystem" + os.path.join(ROCM, "include")]
    cmd += [CORE_SRC, "-o", CORE, "-L" + HERE, "-lnfft_hip", "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu",
            "-ltorch_hip", "-ltorch", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    subprocess.check_call(cmd)
    if verbose:
        print("built", CORE)
    return CORE


if __name__ == "__main__":
    build(force="--force" in sys.argv)
