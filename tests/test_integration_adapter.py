"""INTEGRATION.md §1's binding (include/gpu_low_level.hpp) through a compiler: against the REFERENCE's own headers
(planresult.hpp, neighbor.hpp — neither needs Boost) and linked with libmrp_ll.so.  Build-container only: the reference
does not travel to the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INCLUDE = "/root/reference/include"


@pytest.mark.skipif(not os.path.isdir(REF_INCLUDE), reason="the reference's headers exist in the build container only")
def test_adapter_compiles_against_the_reference_headers_and_links():
    from libmultirobotplanning_amd import _build
    _build.build()
    out = os.path.join(ROOT, "tests", "_build", "integration_adapter_check")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", REF_INCLUDE, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "support", "integration_adapter_check.cpp"), "-o", out,
                           "-L", _build.LIBDIR, "-lmrp_ll", "-Wl,-rpath," + _build.LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib"])
    r = subprocess.run([out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "adapter built and linked" in r.stdout or "ecbs ok=1" in r.stdout
