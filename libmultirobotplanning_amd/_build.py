"""Build recipe for the in-tree native libraries (hipcc, gfx950 only).

``libmrp_ll.so``  — HIP kernels + C-ABI of the low-level search engine (include/mrp_ll.h)
``libmrp_hl.so``  — host-side C++ conflict-tree drivers (CBS / ECBS) that call the C-ABI, instance generator, YAML I/O

Both are built into ``libmultirobotplanning_amd/lib/`` so they travel with the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIBDIR = os.path.join(_PKG, "lib")
INCLUDE = os.path.join(os.path.dirname(_PKG), "include")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", INCLUDE]

TARGETS = {
    "libmrp_ll.so": dict(srcs=["ll_kernel.hip", "conflict_kernel.hip", "mrp_ll_host.cpp"],
                         deps=["ll_device.h", "ll_compact.h", "wave_dev.h", "../../include/mrp_ll.h"],
                         extra=[]),
    "libmrp_hl.so": dict(srcs=["hl/mrp_hl.cpp"], deps=["hl/exact_heap.hpp", "hl/grid_mapf.hpp", "hl/ct_solver.hpp",
                                                       "hl/instance_io.hpp", "../../include/mrp_ll.h",
                                                       "../../include/mrp_hl.h"],
                         extra=["-pthread", "-L", LIBDIR, "-lmrp_ll", "-Wl,-rpath,$ORIGIN"]),
}


def _stale(out, paths):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in paths)


def build_trace(verbose=False):
    """Diagnostic variant of libmrp_ll.so (-DMRP_LL_TRACE): the kernel writes progress words to a host-mapped buffer
    that mrp_ll_wait dumps if a launch does not finish within 10 s (run with MRP_LL_DEBUG=1 MRP_LL_LIB=<this file>)."""
    os.makedirs(LIBDIR, exist_ok=True)
    spec = TARGETS["libmrp_ll.so"]
    out = os.path.join(LIBDIR, "libmrp_ll_trace.so")
    cmd = [HIPCC] + COMMON + ["-DMRP_LL_TRACE", "-o", out] + [os.path.join(CSRC, s) for s in spec["srcs"]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return out


def build_variant(name, defines, verbose=False):
    """A/B build of libmrp_ll.so with extra -D flags (e.g. build_variant("ldsparams", ["-DMRP_LL_PARAMS_IN_LDS"]));
    use it under the host drivers with LD_PRELOAD=<returned path>."""
    os.makedirs(LIBDIR, exist_ok=True)
    spec = TARGETS["libmrp_ll.so"]
    out = os.path.join(LIBDIR, "libmrp_ll_%s.so" % name)
    cmd = [HIPCC] + COMMON + list(defines) + ["-o", out] + [os.path.join(CSRC, s) for s in spec["srcs"]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return out


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    built = []
    for name, spec in TARGETS.items():
        srcs = [os.path.join(CSRC, s) for s in spec["srcs"]]
        if not all(os.path.exists(s) for s in srcs):
            continue
        deps = srcs + [os.path.normpath(os.path.join(CSRC, d)) for d in spec["deps"]]
        out = os.path.join(LIBDIR, name)
        if force or _stale(out, deps):
            cmd = [HIPCC] + COMMON + spec["extra"] + ["-o", out] + srcs
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=CSRC)
        built.append(out)
    return built


if __name__ == "__main__":
    import sys
    for p in build(force="--force" in sys.argv, verbose=True):
        print("built", p)
    if "--trace" in sys.argv:
        print("built", build_trace(verbose=True))
