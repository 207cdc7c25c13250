"""PSNR@step of the HIP training path against the minted curves of the REAL reference (tests/golden/reference_psnr_curves.json,
oracle/make_psnr_curves.py): the statistic of tests/test_gpu_psnr_gate.py::test_psnr_at_equal_steps_within_0p1_db_of_the_reference
for every arithmetic mode and both roundings of sample_pdf's row total.  Run ON THE GPU BOX:

    python tests/tools/psnr_vs_reference.py [--dtypes fp32,bf16,bf16_f8] [--row-totals aten,exact] [--out gpurun_out/psnr_vs_reference.json]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_psnr_gate import reference_paired_statistics  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="fp32,bf16,bf16_f8")
    ap.add_argument("--row-totals", default="aten,exact")
    ap.add_argument("--max-seeds", type=int, default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "psnr_vs_reference.json"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    doc = {}
    for rt in a.row_totals.split(","):
        res = reference_paired_statistics(dev, dtypes=tuple(a.dtypes.split(",")), row_total=rt, max_seeds=a.max_seeds)
        doc[rt] = res
        print("row total %s:" % rt, json.dumps({k: res[k] for k in ("seeds", "dead_seeds", "window", "reference_mean_psnr_in_window", "per_seed", "paired")}), flush=True)
        print("row total %s, mean paired difference by checkpoint:" % rt, json.dumps(res["by_checkpoint"]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1)


if __name__ == "__main__":
    main()
