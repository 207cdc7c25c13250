"""Whole-image inference: the reference's `eval.py:batched_inference` (eval.py:58-86) and an
MI355X-specific variant that replays the fixed-shape 32768-ray chunk as one hipGraph and shards the
ray list across ranks (BASELINE.json configs[4]).

`batched_inference` keeps the reference's signature and behaviour (chunk forced to 32768 rays,
perturb=0, noise_std=0, test_time=True => sigma-only coarse pass, keys opacity_coarse / rgb_fine /
depth_fine / opacity_fine concatenated over chunks).
"""
from collections import defaultdict

import torch

from . import ops
from .models.rendering import _fusable, render_rays

EVAL_CHUNK = 1024 * 32          # eval.py:65 hard-codes this, ignoring --chunk


@torch.no_grad()
def batched_inference(models, embeddings, rays, N_samples, N_importance, use_disp, chunk, white_back):
    """Do batched inference on rays using chunk (eval.py:58-86; `white_back` is the argument the
    reference declares but then shadows with the global `dataset.white_back`, eval.py:78)."""
    B = rays.shape[0]
    chunk = EVAL_CHUNK
    results = defaultdict(list)
    for i in range(0, B, chunk):
        rendered_ray_chunks = render_rays(models, embeddings, rays[i:i + chunk], N_samples, use_disp, 0, 0,
                                          N_importance, chunk, white_back, test_time=True)
        for k, v in rendered_ray_chunks.items():
            results[k] += [v]
    for k, v in results.items():
        results[k] = torch.cat(v, 0)
    return results


def _render_test_time(models, rays, N_samples, N_importance, use_disp, white_back):
    """The test_time launch sequence of render_rays without the (unused, noise_std=0) RNG draws:
    sample_coarse_z -> mlp(coarse, sigma only) -> composite -> fine_z -> mlp(fine) -> composite.
    Allocation-free apart from torch.empty outputs => capturable.
    Where the single-launch kernel takes the shape (nerfhip_render_test_fwd: sigma-only coarse sub-passes, round 6) the whole
    sequence is that ONE launch — the same bits (tests/test_gpu_render_fused.py)."""
    from .models import rendering as R
    dtype = models[0].mlp_dtype
    if (R.FUSE_TEST_TIME and N_importance > 0 and models[1].mlp_dtype == dtype
            and ops.render_supported(rays.shape[0], N_samples, N_importance, dtype)):
        o = ops.render_fwd(rays, N_samples, N_importance, models[0].packed_weights(), models[1].packed_weights(), dtype, use_disp, 0.0, None,
                           None, None, 0.0, white_back, None, want_coarse=False, test_time=True)
        return {k: o[k] for k in ("opacity_coarse", "rgb_fine", "depth_fine", "opacity_fine")}
    z = ops.sample_coarse_z(rays, N_samples, use_disp, 0.0, None)
    sig = ops.mlp_fwd_rays(rays, z, models[0].packed_weights(), True, models[0].mlp_dtype)
    w, opac_c = ops.composite(sig, z, rays, None, 0.0, white_back)
    out = {"opacity_coarse": opac_c}
    if N_importance > 0:
        zf = ops.fine_z(z, w, N_importance, u=None)
        raw = ops.mlp_fwd_rays(rays, zf, models[1].packed_weights(), False, models[1].mlp_dtype)
        _, opac_f, rgb_f, depth_f = ops.composite(raw, zf, rays, None, 0.0, white_back)
        out.update(rgb_fine=rgb_f, depth_fine=depth_f, opacity_fine=opac_f)
    return out


class GraphRenderer:
    """hipGraph-captured inference of one fixed-shape ray chunk (default 32768 rays).

    The six launches of a test-time `render_rays` are captured once on static buffers
    (`torch.cuda.CUDAGraph` == hipGraph on ROCm; libnerfhip launches on torch's current stream, which
    is the capturing stream, allocates nothing and never synchronises) and replayed per chunk; a
    ragged tail chunk is padded with copies of its last ray and the outputs sliced.  Weights are
    re-packed (outside the graph, into the same packed buffers the graph reads) when they change."""

    def __init__(self, models, embeddings, N_samples=64, N_importance=128, use_disp=False, white_back=True,
                 chunk=EVAL_CHUNK, device=None):
        if not _fusable(models, embeddings):
            raise NotImplementedError("GraphRenderer needs the reference's default NeRF/Embedding configuration")
        self.models = list(models)
        self.cfg = (int(N_samples), int(N_importance), bool(use_disp), bool(white_back))
        self.chunk = int(chunk)
        dev = device or next(models[0].parameters()).device
        self.rays = torch.zeros(self.chunk, 8, device=dev, dtype=torch.float32)
        self.rays[:, 5] = 1.0
        self.rays[:, 6] = 2.0
        self.rays[:, 7] = 6.0
        self.graph = None
        self.out = None
        self._capture()

    def _pack(self):
        for m in self.models:
            m.packed_weights()          # repack into the same device buffers the captured graph reads

    def _capture(self):
        with torch.cuda.device(self.rays.device):       # streams and the graph belong to the GPU that owns the buffers
            self._capture_on_device()

    def _capture_on_device(self):
        self._pack()
        S, N, disp, wb = self.cfg
        side = torch.cuda.Stream(device=self.rays.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(2):           # warm-up outside capture (module load, allocator pools)
                _render_test_time(self.models, self.rays, S, N, disp, wb)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # (thread_local: an RCCL watchdog thread polling events elsewhere in the process must not abort the capture)
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = _render_test_time(self.models, self.rays, S, N, disp, wb)

    @torch.no_grad()
    def render_chunk(self, rays):
        n = rays.shape[0]
        if n > self.chunk:
            raise ValueError("chunk larger than the captured shape")
        with torch.cuda.device(self.rays.device):
            self._pack()
            self.rays[:n].copy_(rays)
            if n < self.chunk:
                self.rays[n:].copy_(rays[n - 1:n].expand(self.chunk - n, 8))
            self.graph.replay()
            return {k: v[:n].clone() for k, v in self.out.items()}

    @torch.no_grad()
    def render_to_host(self, rays, keys=("rgb_fine", "depth_fine")):
        """Whole ray list -> PINNED host tensors, with the device-to-host copy of every finished chunk overlapped with
        the replay of the next one (eval.py:123-131 does `.cpu()` on the whole image after the last chunk).

        Compute stream: replay chunk i, then copy its outputs into slice i of a device-side image buffer (the graph's
        static outputs are overwritten by the next replay; the slice is not) and record an event.  Copy stream: wait for
        that event, DMA the slice into the pinned buffer.  Only the last chunk's copy is exposed."""
        with torch.cuda.device(self.rays.device):       # current stream / copy stream of the GPU that owns the buffers
            return self._render_to_host(rays, keys)

    def _render_to_host(self, rays, keys):
        n = rays.shape[0]
        dev = self.rays.device
        cur = torch.cuda.current_stream(dev)
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        img = {k: torch.empty((n,) + tuple(self.out[k].shape[1:]), device=dev, dtype=torch.float32) for k in keys}
        host = {k: torch.empty(v.shape, dtype=torch.float32, pin_memory=True) for k, v in img.items()}
        self._pack()
        for i in range(0, n, self.chunk):
            m = min(self.chunk, n - i)
            self.rays[:m].copy_(rays[i:i + m])
            if m < self.chunk:
                self.rays[m:].copy_(rays[i + m - 1:i + m].expand(self.chunk - m, 8))
            self.graph.replay()
            for k in keys:
                img[k][i:i + m].copy_(self.out[k][:m])
            done = torch.cuda.Event()
            done.record(cur)
            self._copy_stream.wait_event(done)
            with torch.cuda.stream(self._copy_stream):
                for k in keys:
                    host[k][i:i + m].copy_(img[k][i:i + m], non_blocking=True)
        cur.wait_stream(self._copy_stream)
        self._copy_stream.synchronize()
        return host

    @torch.no_grad()
    def __call__(self, rays):
        """Same result dict as batched_inference for a whole ray list."""
        results = defaultdict(list)
        for i in range(0, rays.shape[0], self.chunk):
            for k, v in self.render_chunk(rays[i:i + self.chunk]).items():
                results[k] += [v]
        return {k: (torch.cat(v, 0) if len(v) > 1 else v[0]) for k, v in results.items()}


def save_image_outputs(results, h, w, dir_name, index, save_depth=False, depth_format="pfm"):
    """The per-image tail of the reference's eval loop (eval.py:123-141): `results` holds host (or device) tensors
    rgb_fine (h*w, 3) and depth_fine (h*w,); writes `{index:03d}.png` and, when asked, `depth_{index:03d}.pfm` or the raw
    float32 bytes.  Returns the uint8 image (the reference collects them for the gif, eval.py:140,149)."""
    import os

    import numpy as np

    from .imageio_min import depth_bytes, save_pfm, write_png
    img_pred = results["rgb_fine"].reshape(h, w, 3).cpu().numpy()
    if save_depth:
        depth_pred = np.nan_to_num(results["depth_fine"].reshape(h, w).cpu().numpy())
        if depth_format == "pfm":
            save_pfm(os.path.join(dir_name, "depth_%03d.pfm" % index), depth_pred)
        else:
            with open(os.path.join(dir_name, "depth_%03d" % index), "wb") as f:
                f.write(depth_bytes(depth_pred))
    img_pred_ = (img_pred * 255).astype(np.uint8)
    write_png(os.path.join(dir_name, "%03d.png" % index), img_pred_)
    return img_pred_
