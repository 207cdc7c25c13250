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


def test_ahead_of_step_weight_images_are_trusted_only_under_a_live_announcing_optimizer():
    """`_serial_tracked` used to be a flag FlatAdam set and nobody cleared.  It is a weak reference to the optimizer now (a
    model later stepped by something else is not vouched for by a dead FlatAdam), and load_state_dict — the reference's load_ckpt,
    utils/__init__.py:55-76 — bumps the weights' serial like an announced update."""
    import gc
    import weakref

    from nerf_pl_amd.models import NeRF
    from nerf_pl_amd.models.train_step import _tracked

    class Opt:                      # stands in for optim.FlatAdam (which needs a GPU): what matters is the weak reference
        pass
    m = NeRF()
    assert not _tracked(m)
    opt = Opt()
    m._serial_tracked = weakref.ref(opt)
    assert _tracked(m)
    del opt
    gc.collect()
    assert not _tracked(m)
    before = getattr(m, "_weights_serial", 0)
    m.load_state_dict(NeRF().state_dict())
    assert m._weights_serial == before + 1
