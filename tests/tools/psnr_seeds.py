"""Multi-seed PSNR@step statistics on the analytic scene: for each jitter/batch seed train fp32, bf16 and bf16_f8 from the same
init with the compressed recipe of tests/test_gpu_bf16.py (900 steps, lr 5e-4 -> 5e-5 at 600) and report, per dtype, the
mean and spread of the final PSNR over the seeds and the paired differences to fp32.  (Training is chaotic: single
trajectory pairs differ by 0.1-0.4 dB already between two fp32 runs; the mean over seeds is the meaningful comparison.)
    python tests/tools/psnr_seeds.py [--seeds 0,1,3] [--out gpurun_out/psnr_seeds.json]"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_gpu_bf16 as T  # noqa: E402
from tests.helpers import analytic_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0,1,3", help="comma-separated init/jitter seeds (seed 2 is a dead-ReLU init: 7.3 dB in every precision)")
    ap.add_argument("--dtypes", default="fp32,bf16,bf16_f8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "psnr_seeds.json"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from nerf_pl_amd.models import NeRF
    rays, rgbs = analytic_scene(200000, 1, dev)
    rays_val, rgb_val = analytic_scene(8192, 2, dev)
    res = {dt: [] for dt in a.dtypes.split(",")}
    for seed in [int(x) for x in a.seeds.split(",")]:
        torch.manual_seed(seed)
        init = [NeRF().state_dict(), NeRF().state_dict()]
        for dt in res:
            curve, _ = T._train(dt, dev, rays, rgbs, rays_val, rgb_val, init, jitter_seed=1000 + seed)
            res[dt].append(curve)
            print(seed, dt, {k: round(v, 3) for k, v in curve.items()}, flush=True)
    summary = {}
    for dt, curves in res.items():
        finals = [sum(c[s] for s in T.EVAL_AT) / len(T.EVAL_AT) for c in curves]
        summary[dt] = {"mean_psnr_700_900": round(statistics.mean(finals), 3),
                       "stdev_over_seeds": round(statistics.stdev(finals), 3) if len(finals) > 1 else None,
                       "per_seed": [round(f, 3) for f in finals]}
    for dt in res:
        if dt != "fp32" and "fp32" in res:
            d = [a_ - b_ for a_, b_ in zip(summary[dt]["per_seed"], summary["fp32"]["per_seed"])]
            summary[dt]["paired_diff_to_fp32"] = {"mean": round(statistics.mean(d), 3),
                                                  "stderr": round(statistics.stdev(d) / len(d) ** 0.5, 3) if len(d) > 1 else None,
                                                  "per_seed": [round(x, 3) for x in d]}
    print(json.dumps(summary, indent=1))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump({"recipe": "analytic scene, 1024 rays/step, 64+64 samples, Adam 5e-4 -> 5e-5 at step 600, 900 steps; PSNR on 8192 "
                             "held-out rays, mean of steps 700/800/900", "summary": summary, "curves": res}, fh, indent=1)


if __name__ == "__main__":
    main()
