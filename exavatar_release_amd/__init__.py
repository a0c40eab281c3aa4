"""MI355X-native differentiable 3D-Gaussian rasterizer for ExAvatar's ``GaussianRenderer``.

Public surface (drop-in for ``diff_gaussian_rasterization_depth`` as used at reference
``avatar/common/nets/module.py:11,609-640``):

    from exavatar_release_amd import GaussianRasterizationSettings, GaussianRasterizer
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, config,
                         rasterize_gaussians, rasterize_gaussians_batch)
from .densify import track_densify_stats
from .losses import SSIM, PhotometricLoss, RGBLoss
from .renderer import ITERATION_RENDERS, GaussianRenderer, GraphedRenderer, render_iteration, render_many, render_views
from .graphed import GraphedIteration
from .static import StaticRender, required_capacity

__all__ = ['GaussianRasterizationSettings', 'GaussianRasterizer', 'GaussianRenderer', 'rasterize_gaussians',
           'rasterize_gaussians_batch', 'config', 'track_densify_stats', 'render_many', 'render_views',
           'render_iteration', 'ITERATION_RENDERS', 'GraphedRenderer', 'GraphedIteration', 'StaticRender', 'required_capacity',
           'SSIM', 'RGBLoss', 'PhotometricLoss']
