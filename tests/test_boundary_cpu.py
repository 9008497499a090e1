"""CPU: the drop-in boundary compiles.  (a) include/jackalope_hip.h is C99 and a plain C program that fills its structs
by hand links against the library; (b) the four Rcpp shims of jackalope_amd/rcpp/ compile as C++11 against the header
(Rcpp itself is stood in for by tests/rcpp_stubs/, see its README); (c) every function the header declares is exported.
No compute call is made here; tests/test_gpu_boundary.py runs both programs on the GPU box."""
import ctypes as C
import os
import re
import subprocess

import boundary_build as bb


def test_header_is_c99_and_the_c_driver_links(built):
    hdr = os.path.join(bb.ROOT, "include", "jackalope_hip.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    exe = bb.abi_driver()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_rcpp_shims_compile_against_the_header(built):
    so = bb.shim_driver()
    L = C.CDLL(so)
    for name in ("drv_illumina_ref", "drv_illumina_hap", "drv_pacbio_ref", "drv_pacbio_hap"):
        assert hasattr(L, name)
    # the shims define exactly the four entry points of src/RcppExports.cpp:111-244
    syms = subprocess.check_output(["nm", "-D", "--defined-only", "-C", so], text=True)
    for fn in ("illumina_ref_cpp(", "illumina_hap_cpp(", "pacbio_ref_cpp(", "pacbio_hap_cpp("):
        assert fn in syms


def test_every_declared_function_is_exported(built, ja):
    hdr = open(os.path.join(bb.ROOT, "include", "jackalope_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(jk_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"jk_seed_fn"}
    L = ja.lib()
    missing = [f for f in sorted(declared) if not hasattr(L, f)]
    assert not missing, missing
    from jackalope_amd import _abi
    assert declared == set(_abi.EXPORTS), (declared ^ set(_abi.EXPORTS))
