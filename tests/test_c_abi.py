"""The drop-in boundary from plain C: examples/abi_smoke.c is compiled with gcc against include/pnpx.h + libpnpx.so
(CPU test: it builds and links) and executed on the GPU box (gpu test: no Python, no torch in that process)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    lib = os.path.join(ROOT, "tfpnp_amd")
    if not os.path.exists(os.path.join(lib, "libpnpx.so")):
        import __graft_entry__ as g
        g.build()
    cmd = ["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", f"-I{ROCM}/include", f"-I{ROOT}/include",
           os.path.join(ROOT, "examples", "abi_smoke.c"), f"-L{lib}", "-lpnpx", f"-L{ROCM}/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{lib}", f"-Wl,-rpath,{ROCM}/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_consumer_builds_and_links(tmp_path):
    exe = build(tmp_path)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libpnpx.so" in out and "not found" not in out.split("libpnpx.so")[1].splitlines()[0]
    # the header is valid C99 and C++11 on its own
    for lang, std in (("c", "-std=c99"), ("c++", "-std=c++11")):
        r = subprocess.run(["gcc", std, "-fsyntax-only", "-x", lang, os.path.join(ROOT, "include", "pnpx.h")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_c_consumer_runs(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke OK" in r.stdout
