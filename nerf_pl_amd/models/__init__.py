"""Mirror of the reference's `models` package (the drop-in boundary)."""
from .nerf import Embedding, NeRF  # noqa: F401
from .rendering import render_rays, sample_pdf  # noqa: F401
