"""The drop-in boundary: libmgx.so loads on a CPU-only host, exports every symbol include/mgx.h
declares, and its compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from magics_amd import MgxError, World, hostlib, scenarios

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "mgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgx_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    declared = _declared_functions()
    assert len(declared) >= 25
    assert sorted(hostlib.SYMBOLS) == declared


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(hostlib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(L, name), name


def test_no_torch_types_or_oracle_in_the_product():
    # the product never imports / links the oracle, and the ABI has no torch types
    for dirpath, _, files in os.walk(os.path.join(ROOT, "magics_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "gbp_oracle" not in txt, f
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mgx.h")).read(), flags=re.S)
    assert "torch" not in hdr and "at::" not in hdr and "Tensor" not in hdr


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU behaviour")
def test_compute_fails_loudly_without_gpu():
    with pytest.raises(MgxError, match="no usable HIP device"):
        World(scenarios.JUNCTION_PARAMS)


def test_argument_validation_needs_no_device():
    L = hostlib.lib()
    assert L.mgx_world_create(None, None) == -1
    assert L.mgx_robot_add(None, None, None) == -1
    assert L.mgx_iterate(None, None, 0) == -1
    assert b"null" in L.mgx_last_error()


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: include/mgx.h must compile as C99 (-pedantic) and a C program must
    link against libmgx.so and run (host-only entry points work without a GPU)."""
    import subprocess
    root = ROOT
    src = os.path.join(ROOT, "tests", "c_abi", "c_client.c")
    exe = str(tmp_path / "c_client")
    libdir = os.path.join(root, "magics_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), src, "-o", exe,
                    "-L", libdir, "-lmgx", f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout.decode())
    assert b"c client ok" in r.stdout


@pytest.mark.gpu
def test_c_client_on_the_gpu(tmp_path):
    """The same C program on a machine with a GPU: it creates a world, rasterises an environment
    through mgx_env_* / mgx_world_set_environment and checks the error path of the rasteriser."""
    import subprocess
    src = os.path.join(ROOT, "tests", "c_abi", "c_client.c")
    exe = str(tmp_path / "c_client")
    libdir = os.path.join(ROOT, "magics_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                    "-L", libdir, "-lmgx", f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout.decode())
    assert b"gpu: world created" in r.stdout and b"c client ok" in r.stdout
