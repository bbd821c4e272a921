"""A plain-C host of the C ABI (examples/c_host/abi_client.c): gcc + include/f5e_abi.h + libf5e_hip.so, no Python or torch on
the data path.  CPU: it compiles as ISO C11 and links against exactly the exported symbols.  GPU: it runs and checks the fp32
GEMM, the bf16 GEMM and a captured, replayed CFG + Euler update against loops on the host."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "c_host", "abi_client.c")
PKG = os.path.join(ROOT, "f5e-tts_amd")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build(out):
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(ROCM, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc or the HIP runtime headers are not installed")
    if not os.path.exists(os.path.join(PKG, "libf5e_hip.so")):
        import __graft_entry__ as g
        g.build()
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROCM, "include"), SRC, "-L", PKG, "-lf5e_hip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
           "-lm", f"-Wl,-rpath,{PKG}", f"-Wl,-rpath,{os.path.join(ROCM, 'lib')}", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_c_client_compiles_as_iso_c_and_links_against_the_exported_symbols(tmp_path):
    exe = str(tmp_path / "abi_client")
    build(exe)
    nm = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    used = sorted({ln.split()[-1].split("@")[0] for ln in nm.splitlines() if " f5e_" in ln})
    assert {"f5e_abi_version", "f5e_check_device", "f5e_gemm_f32", "f5e_gemm_bf16_bias", "f5e_ode_update", "f5e_graph_begin",
            "f5e_graph_end", "f5e_graph_launch", "f5e_graph_destroy", "f5e_last_error"} <= set(used), used


@pytest.mark.gpu
def test_c_client_runs_on_the_gpu(tmp_path):
    exe = str(tmp_path / "abi_client")
    build(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK")
