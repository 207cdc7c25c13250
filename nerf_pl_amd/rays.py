"""`datasets/ray_utils.py` of the reference on the device, plus a device-resident ray/pixel store.

`get_ray_directions`, `get_rays`, `get_ndc_rays` keep the reference signatures (ray_utils.py:5-94) and run as single HIP
launches; `RayStore` replaces the reference's CPU-side "precompute every ray of the dataset + DataLoader" pattern
(blender.py:42-69,81-84, train.py:89-94): it keeps only the camera poses and the pixel colours in HBM and regenerates
each batch's rays from 8-byte pixel ids (`nerfhip_gen_rays`), so a training batch never crosses PCIe."""
import torch

from . import _lib
from ._lib import check, device_guard, ptr, require_gpu, stream_ptr


def get_ray_directions(H, W, focal, device="cuda"):
    """(H, W, 3) camera-space ray directions.  Reference: datasets/ray_utils.py:5-24."""
    dirs = torch.empty(H, W, 3, device=device, dtype=torch.float32)
    with torch.cuda.device(dirs.device):
        require_gpu(dirs)
        check(_lib.load().nerfhip_ray_directions(ptr(dirs), int(H), int(W), float(focal), stream_ptr()), "nerfhip_ray_directions")
    return dirs


@device_guard
def get_rays(directions, c2w):
    """rays_o, rays_d (H*W, 3) in world coordinates.  Reference: datasets/ray_utils.py:27-52."""
    require_gpu(directions)
    directions = directions.contiguous()
    c2w = c2w.to(directions.device, torch.float32).contiguous()
    if c2w.shape != (3, 4):
        raise ValueError("c2w must be (3, 4)")
    n = directions.numel() // 3
    rays_o = torch.empty(n, 3, device=directions.device, dtype=torch.float32)
    rays_d = torch.empty(n, 3, device=directions.device, dtype=torch.float32)
    check(_lib.load().nerfhip_get_rays(ptr(directions), ptr(c2w), ptr(rays_o), ptr(rays_d), n, stream_ptr()), "nerfhip_get_rays")
    return rays_o, rays_d


@device_guard
def get_ndc_rays(H, W, focal, near, rays_o, rays_d):
    """World rays -> NDC rays.  Reference: datasets/ray_utils.py:55-94."""
    require_gpu(rays_o, rays_d)
    rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
    n = rays_o.numel() // 3
    out_o, out_d = torch.empty_like(rays_o), torch.empty_like(rays_d)
    check(_lib.load().nerfhip_ndc_rays(int(H), int(W), float(focal), float(near), ptr(rays_o), ptr(rays_d), ptr(out_o), ptr(out_d),
                                       n, stream_ptr()), "nerfhip_ndc_rays")
    return out_o, out_d


@device_guard
def gen_rays(c2w, H, W, focal, near, far, pixel_ids=None, first_pixel=0, n=None, use_ndc=False, ndc_near_plane=1.0):
    """rays (n, 8) = [o d near far] for global pixel ids (image*H*W + row*W + col) under poses c2w (n_images, 3, 4)."""
    require_gpu(c2w)
    c2w = c2w.contiguous()
    if c2w.dim() == 2:
        c2w = c2w[None]
    if pixel_ids is not None:
        if pixel_ids.dtype != torch.int64 or not pixel_ids.is_cuda:
            raise ValueError("pixel_ids must be an int64 device tensor")
        pixel_ids = pixel_ids.contiguous()
        n = pixel_ids.numel()
    elif n is None:
        n = c2w.shape[0] * H * W - first_pixel
    rays = torch.empty(n, 8, device=c2w.device, dtype=torch.float32)
    check(_lib.load().nerfhip_gen_rays(ptr(c2w), ptr(pixel_ids), int(first_pixel), int(n), int(H), int(W), float(focal),
                                       float(near), float(far), int(bool(use_ndc)), float(ndc_near_plane), ptr(rays), stream_ptr()),
          "nerfhip_gen_rays")
    return rays


class RayStore:
    """Device-resident training set: poses (n_img,3,4) + pixel colours (n_img*H*W, 3) in HBM; batches are drawn and
    their rays generated on the GPU.  `sample(B)` returns the reference's batch dict {'rays': (B,8), 'rgbs': (B,3)}
    (blender.py:81-84); `image_rays(i)` the (H*W, 8) rays of one image (validation / eval)."""

    def __init__(self, poses, rgbs, H, W, focal, near, far, use_ndc=False, ndc_near_plane=1.0):
        if not poses.is_cuda:
            require_gpu(poses)                          # raises: no CPU fallback
        with torch.cuda.device(poses.device):
            require_gpu(poses, rgbs)
        self.poses = poses.contiguous()
        self.rgbs = rgbs.reshape(-1, 3).contiguous()
        self.H, self.W, self.focal, self.near, self.far = int(H), int(W), float(focal), float(near), float(far)
        self.use_ndc, self.ndc_near_plane = bool(use_ndc), float(ndc_near_plane)
        self.n_pixels = self.poses.shape[0] * self.H * self.W
        if self.rgbs.shape[0] != self.n_pixels:
            raise ValueError("rgbs must hold n_images*H*W pixels")

    def __len__(self):
        return self.n_pixels

    def sample(self, batch_size, generator=None, step_draws=None, return_ids=False, pack_models=None):
        """One training batch in ONE launch (nerfhip_torch_draws with a ray batch): the pixel ids torch.randint(0, n_pixels,
        (batch_size,)) would draw from `generator` (default: the device's default generator, advanced identically), their rays
        and their colours.
        step_draws = (N_samples, N_importance, perturb, noise_std): the same launch also makes the draws the training step's
        render_rays will need (draws.step_specs: the reference's rand / randn calls, in its order, right after the batch's
        randint on the generator's stream) and returns them under batch['draws'] for NeRFSystem.training_step.
        pack_models = (models, mlp_dtype): the launch also packs the weight images the step's MLP kernels stream
        (nerfhip_train_prologue); batch['packed'] then names the models, and the fused training node skips its own pack launch
        while the weights are still the ones packed here."""
        from . import draws as D
        dev = self.poses.device
        B = int(batch_size)
        rays = torch.empty(B, 8, device=dev, dtype=torch.float32)
        rgbs = torch.empty(B, 3, device=dev, dtype=torch.float32)
        rb = _lib.RayBatch()
        rb.c2w, rb.rgbs_all, rb.rays, rb.rgbs = self.poses.data_ptr(), self.rgbs.data_ptr(), rays.data_ptr(), rgbs.data_ptr()
        rb.H, rb.W, rb.focal, rb.near, rb.far = self.H, self.W, self.focal, self.near, self.far
        rb.use_ndc, rb.ndc_near_plane = int(self.use_ndc), self.ndc_near_plane
        specs, keys = [("randint", (B,), self.n_pixels, bool(return_ids))], []
        if step_draws is not None:
            S, N, perturb, noise_std = step_draws
            keys, more = D.step_specs(B, int(S), int(N), float(perturb), float(noise_std))
            specs += more
        outs = D.draws(specs, dev, generator, batch=rb, pack=pack_models)
        batch = {"rays": rays, "rgbs": rgbs}
        if pack_models is not None:
            batch["packed"] = (tuple(id(m) for m in pack_models[0]), pack_models[1], tuple(m._packed_serial for m in pack_models[0]))
        if return_ids:
            batch["ids"] = outs[0]
        if step_draws is not None:
            batch["draws"] = {k: t for k, t in zip(keys, outs[1:]) if t is not None}
        return batch

    def image_rays(self, i):
        hw = self.H * self.W
        return gen_rays(self.poses, self.H, self.W, self.focal, self.near, self.far, first_pixel=i * hw, n=hw,
                        use_ndc=self.use_ndc, ndc_near_plane=self.ndc_near_plane)
