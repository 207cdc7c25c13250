"""CPU: the index maps of nerf_pl_amd/csrc/mlp_layout.h — which input column of which weight matrix a B-operand slot multiplies,
where a ReLU gate bit lives, how the packed weight stream is laid out (bias block, the two looped layer triples), which saved
sections a weight-gradient job reads — checked on the host by a small g++ program (tests/host/layout_check.cpp).  The pack
kernels and the MLP kernels share these constexpr functions, so what holds here holds on the device."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_maps_are_consistent(tmp_path):
    exe = str(tmp_path / "layout_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "nerf_pl_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "layout_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "layout ok" in r.stdout, r.stdout + r.stderr
