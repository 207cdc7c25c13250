"""CPU: the weight-gradient launch's split plan (host logic of csrc/mlp_bwd.hip `dw_plan`, exported as nerfhip_mlp_dw_plan).
The launch computes dW_l = dY_l^T X_l for the 12 parameter tensors of each model (reference nerf.py:42-81 / autograd of
nerf.py:100-124); a workgroup = (job, point range).  No compute calls: the plan is plain host arithmetic."""
import ctypes

import pytest

F32, BF16, BF16_F8 = 0, 1, 2
STAGE_KIB = [20, 32, 32, 32, 36, 32, 32, 32, 0, 28, 0, 10]          # dY + X slabs of a 32-point tile, per job (mlp_layout.h kDwJobs)
FINAL, DIR, SIGMA = 8, 9, 10                                        # jobs without workgroups: the final layer (derived), the sigma head (folded)


@pytest.fixture(scope="module")
def lib():
    from nerf_pl_amd import build
    build.build(verbose=False)
    from nerf_pl_amd import _lib
    return _lib.load()


def _plan(lib, points, dtype):
    n = (ctypes.c_int64 * len(points))(*points)
    sp = (ctypes.c_int * (12 * len(points)))()
    kb = (ctypes.c_int * (12 * len(points)))()
    total = lib.nerfhip_mlp_dw_plan(n, len(points), dtype, sp, kb)
    return total, list(sp), list(kb)


def test_benchmark_step_is_one_round_of_the_256_cus(lib):
    """configs[2]: fine 1024 x 192 + coarse 1024 x 64 points in ONE launch."""
    # Round 6: xyz_encoding_final (job 8) has no workgroups at all — it is a linear layer without activation, its gradients are
    # finished from the dir job's G = dY_dir^T h8 by mlp_bwd_fold_kernel (mlp_layout.h kDwJobs) — and the sigma head (job 10) none of
    # its own: the dir layer's workgroups (job 9) form its gradient from the h8 stage they now hold, which grows by the 2 dY_sigma slabs.
    for dtype, want, unit in ((BF16_F8, 256, 1), (BF16, 256, 1), (F32, 512, 2)):
        total, sp, kb = _plan(lib, [1024 * 192, 1024 * 64], dtype)
        assert total == want == sum(sp)
        assert all(sp[12 * m + j] == 0 for m in (0, 1) for j in (FINAL, SIGMA))
        assert min(s for j, s in enumerate(sp) if j % 12 not in (FINAL, SIGMA)) >= 1
        assert kb == [unit * k for k in STAGE_KIB * 2]


def iter_cost(kib):
    """Round 6 cost model of one ring iteration of the bf16 kernel, in ns: 150 + 45 per KiB of its stage (csrc/mlp_bwd.hip
    NERFHIP_DW_COST_A/B; sweep in profiles/r06_dw_plan_cost_ab.txt).  Round 4's kernel measured 0.7-1.8 us per iteration
    (tools/dw_probe.py) and planned with 300 + 35 per KiB; the 2 x 4 wave split, dot2 bias sums and register-major epilogue of
    round 6 shrank the fixed part, and the sweep then preferred a plan nearer to bytes-proportional."""
    return 150 + 45 * kib


def test_bf16_plan_equalises_time_not_iterations(lib):
    """An iteration of the bf16 kernel costs more the more bytes its stage holds, so equal iteration counts leave the skip-layer
    workgroups behind the 256 x 256 layers and the small heads idle for part of the launch.  The plan balances
    iterations x iter_cost(stage KiB): the slowest workgroup within 8 % of the mean under that model."""
    pts = [1024 * 192, 1024 * 64]
    total, sp, kb = _plan(lib, pts, BF16)
    t = []
    for j, (s, k) in enumerate(zip(sp, kb)):
        tiles = pts[j // 12] // 32
        t.append(-(-tiles // s) * iter_cost(k) if s else 0)
    mean = sum(ti * s for ti, s in zip(t, sp)) / total
    assert max(t) <= 1.08 * mean, (max(t), mean, sp)
    assert sp[4] > sp[1] > sp[0] > sp[11]            # skip layer (36 KiB) > 256 x 256 (32) > first (20) > rgb head (10)
    # equal iteration counts would be 18 % off
    eq = [384 * iter_cost(k) for k in kb[:12] if k] + [-(-2048 // s) * iter_cost(k) for s, k in zip([5, 6, 6, 6, 6, 6, 6, 6, 6, 5, 5, 5], kb[12:]) if k]
    assert max(eq) > 1.15 * mean
    # the e4m3 kernel keeps equal iteration counts (measured: 254 us against 336 us with a byte-weighted plan)
    total, sp8, _ = _plan(lib, pts, BF16_F8)
    it = [-(-(pts[j // 12] // 64) // s) for j, s in enumerate(sp8) if s]
    assert max(it) <= 1.25 * min(it), it


def test_small_and_ragged_sizes(lib):
    # fewer than 48 ring iterations per workgroup are never planned: a tiny batch gets one workgroup per job
    total, sp, _ = _plan(lib, [100], BF16)
    assert total == 10 and sp == [1] * 8 + [0, 1, 0, 1]
    total, sp, _ = _plan(lib, [100], BF16_F8)
    assert total == 10 and sp == [1] * 8 + [0, 1, 0, 1]
    total, sp, _ = _plan(lib, [32 * 48 * 3 + 5], BF16)       # (padded to whole 256-point blocks)
    assert all(1 <= s <= 3 for j, s in enumerate(sp) if j not in (FINAL, SIGMA)) and sp[FINAL] == 0 == sp[SIGMA], sp
    # one model with the benchmark's fine pass alone
    total, sp, _ = _plan(lib, [1024 * 192], BF16_F8)
    live = [s for j, s in enumerate(sp) if j not in (FINAL, SIGMA)]
    assert total == 256 and sp[FINAL] == 0 == sp[SIGMA] and max(live) - min(live) <= 1


def test_bad_arguments(lib):
    sp = (ctypes.c_int * 24)()
    n = (ctypes.c_int64 * 2)(1024, 0)
    assert lib.nerfhip_mlp_dw_plan(n, 2, BF16, sp, None) < 0          # empty model
    assert lib.nerfhip_mlp_dw_plan(n, 3, BF16, sp, None) < 0          # more models than one launch serves
    assert lib.nerfhip_mlp_dw_plan(n, 1, 7, sp, None) < 0             # unknown dtype
    assert lib.nerfhip_mlp_dw_plan(None, 1, BF16, sp, None) < 0
    assert lib.nerfhip_mlp_dw_plan(n, 1, BF16, sp, None) == sum(sp[:12])
