"""pytest configuration: registers the `gpu` marker and provides the package / oracle loaders.

  -m "not gpu": oracle vs golden vectors, host logic, C-ABI symbol export (no compute calls).
  -m gpu      : parity tests proper -- HIP path through the C-ABI vs the oracle.
"""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def load_package():
    """Import the hyphen-named package directory as module `vspg_pbrt_v4_amd`."""
    name = "vspg_pbrt_v4_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "vspg-pbrt-v4_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    """The package with the HIP library loaded; fails loudly if the extension is missing.
    torch bundles a HIP runtime of its own: it opens the device only if it does so BEFORE the system runtime behind
    libvspg_hip.so has (two GPU tests hand device pointers to torch) -- so it goes first, whatever subset of the suite runs."""
    try:
        import torch
        torch.cuda.is_available()
    except ImportError:
        pass
    pkg.load()
    return pkg


@pytest.fixture(scope="session")
def libm_shim():
    """tests/libm_model_shim.cpp built for the host: the product's vspg_libm.h functions next to
    the running libm's logf/sinf/cosf."""
    import ctypes as C
    import subprocess

    src = os.path.join(ROOT, "tests", "libm_model_shim.cpp")
    out = os.path.join(ROOT, "tests", "_build", "libm_model_shim.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hdr = os.path.join(ROOT, "vspg-pbrt-v4_amd", "csrc", "vspg_libm.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        flags = ["-O2", "-ffp-contract=off", "-fno-builtin", "-shared", "-fPIC"]
        if " fma " in open("/proc/cpuinfo").read():
            flags.append("-mfma")
        subprocess.check_call(["g++"] + flags + ["-o", out, src, "-lm"])
    lib = C.CDLL(out)
    fp = C.POINTER(C.c_float)
    for n in ("model_logf", "model_sinf", "model_cosf", "libm_logf", "libm_sinf", "libm_cosf", "model_atanhf", "libm_atanhf", "model_log1pf", "libm_log1pf"):
        getattr(lib, n).argtypes = [C.c_int, fp, fp]
    dp = C.POINTER(C.c_double)
    for n in ("model_log", "libm_log"):
        getattr(lib, n).argtypes = [C.c_int, dp, dp]
    lib.libm_neg_log1m.argtypes = [C.c_int, fp, fp]
    return lib


@pytest.fixture(autouse=True)
def _debug_build_checks(request):
    """With a diagnostic build of the library (csrc built with -DVSPG_WF_DEBUG, selected through VSPG_LIB) every GPU test
    also asserts that the kernels' own index checks never fired (path-pool slots, list entries: vspg_dbg_read)."""
    yield
    if os.environ.get("VSPG_DBG_CHECK") and request.node.get_closest_marker("gpu"):
        import ctypes as C
        pkg = load_package()
        lib = pkg.load()
        if hasattr(lib, "vspg_dbg_read"):
            out = (C.c_uint * 8)()
            assert lib.vspg_dbg_read(out) == 0
            assert list(out)[:2] == [0, 0], "device-side index check fired: %s" % list(out)
