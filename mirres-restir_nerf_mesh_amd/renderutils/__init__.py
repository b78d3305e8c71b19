"""Host-side mirror of the part of nerf/renderutils the hot path uses (renderer_restir.py:11): the bilateral denoiser."""
from .ops import bilateral_denoiser, bilateral_denoiser_no_di, prepare_shading_normal  # noqa: F401
