"""PSNR@step gate of the reduced-precision modes on the lego-like procedural scene (tests/test_gpu_psnr_gate.py), as a tool:

    python tests/tools/psnr_gate.py --sweep "freq=6,9,14 amp=0.3,0.42"     # fp32 only, 2 seeds per scene: where does PSNR@step end?
    python tests/tools/psnr_gate.py --gate [--scene "freq=9 amp=0.42"] [--seeds 12] [--out gpurun_out/psnr_gate.json]
    python tests/tools/psnr_gate.py --sweep ... --gate --auto 31.0          # sweep, pick the scene closest to 31 dB, run the gate on it

--gate runs fp32 / bf16 / bf16_f8 from the same init, batches and draws for >= N live seeds and reports the paired
differences to fp32 (mean, standard error).  The JSON goes to profiles/ by hand."""
import argparse
import itertools
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_gpu_psnr_gate as G  # noqa: E402


def parse_scene(s):
    out = {}
    for tok in (s or "").split():
        k, v = tok.split("=")
        out[k] = [float(x) for x in v.split(",")]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", default=None)
    ap.add_argument("--sweep-list", default=None)
    ap.add_argument("--gate", action="store_true")
    ap.add_argument("--scene", default="")
    ap.add_argument("--auto", type=float, default=None, help="with --sweep --gate: gate on the swept scene whose fp32 PSNR is closest to this")
    ap.add_argument("--seeds", type=int, default=G.N_LIVE)
    ap.add_argument("--dtypes", default="bf16,bf16_f8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "psnr_gate.json"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from nerf_pl_amd.models import NeRF
    report = {}
    scene = {k: v[0] for k, v in parse_scene(a.scene).items()}
    if a.sweep or a.sweep_list:
        grid = parse_scene(a.sweep)
        keys = sorted(grid)
        rows = []
        combos = [dict(zip(keys, c)) for c in itertools.product(*[grid[k] for k in keys])]
        if a.sweep_list:                       # explicit scenes instead of a grid: "freq=6 amp=0.2; freq=10 amp=0.3"
            combos = [{k: v[0] for k, v in parse_scene(t).items()} for t in a.sweep_list.split(";")]
        for sc in combos:
            data = G.make_data(dev, **sc)
            vals = []
            for seed in (0, 1, 3):
                torch.manual_seed(seed)
                init = [NeRF().state_dict(), NeRF().state_dict()]
                t0 = time.time()
                c = G.train_curve("fp32", dev, data, init, 1000 + seed, dead_check=(G.DEAD_CHECK_AT, G.DEAD_BELOW_DB))
                if c is None:
                    continue
                f = sum(c[s] for s in G.EVAL_AT) / len(G.EVAL_AT)
                vals.append(f)
                print("sweep", sc, "seed", seed, "fp32 %.3f dB" % f, {k: round(v, 2) for k, v in c.items()}, "%.1f s" % (time.time() - t0), flush=True)
                if len(vals) == 2:
                    break
            rows.append({"scene": sc, "fp32_psnr": vals})
        report["sweep"] = rows
        if a.auto is not None:
            live = [r for r in rows if r["fp32_psnr"]]
            best = min(live, key=lambda r: abs(sum(r["fp32_psnr"]) / len(r["fp32_psnr"]) - a.auto))
            scene = best["scene"]
            print("auto-selected scene", scene, best["fp32_psnr"], flush=True)
    if a.gate:
        t0 = time.time()
        res = G.paired_statistics(dev, dtypes=tuple(a.dtypes.split(",")), n_live=a.seeds, scene=scene,
                                  log=lambda m: print(m, flush=True))
        res["wall_s"] = round(time.time() - t0, 1)
        res["recipe"] = ("brick scene, %d rays/step, %d+%d samples, Adam lr %s, %d steps; PSNR on %d held-out rays, mean of steps %s; "
                         "paired differences to the HIP fp32-MFMA path (same init, batches, draws)"
                         % (G.B, G.S, G.N, G.LR_AT, G.STEPS, G.N_VAL_RAYS, list(G.EVAL_AT)))
        report["gate"] = res
        print(json.dumps({k: res[k] for k in ("scene", "seeds", "dead_seeds", "mean_psnr", "paired", "wall_s")}, indent=1), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()
