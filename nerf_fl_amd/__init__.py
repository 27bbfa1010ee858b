"""nerf_fl_amd -- MI355X-native renderer for the render_rays hot path of nmerty/nerf-fl.

Public surface mirrors the reference modules it replaces:
    nerf_fl_amd.rendering.render_rays   <- models/rendering.py
    nerf_fl_amd.nerf.{NeRF,PosEmbedding} <- models/nerf.py (parameter containers)
"""
from .nerf import NeRF, PosEmbedding, BarfPosEmbedding          # noqa: F401
from .rendering import render_rays, set_precision, get_precision, set_rounding_seed, check_status, CameraRays   # noqa: F401

__version__ = "0.1"
