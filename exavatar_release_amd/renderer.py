"""Host-side mirror of the reference's render plugin ``GaussianRenderer``
(``avatar/common/nets/module.py:588-647``): same ``forward(gaussian_assets, img_shape, cam_param, bg)``
signature, same output dict ``{img, depthmap, mask, mean_2d, is_vis, radius}``, bound to the gfx950
rasterizer instead of the third-party CUDA extension.

The only intentional differences: tensors are created on the device of the inputs instead of a
hard-coded ``.cuda()``, and the default background is created per call (the reference evaluates
``torch.ones(3).cuda()`` at import time, module.py:592).
"""
import torch
import torch.nn as nn

from .camera import get_fov, get_proj_matrix, get_view_matrix
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


class GaussianRenderer(nn.Module):
    def __init__(self):
        super(GaussianRenderer, self).__init__()

    def forward(self, gaussian_assets, img_shape, cam_param, bg=None):
        # assets for the rendering (module.py:594-598)
        mean_3d = gaussian_assets['mean_3d']
        opacity = gaussian_assets['opacity']
        scale = gaussian_assets['scale']
        rotation = gaussian_assets['rotation']
        rgb = gaussian_assets['rgb']
        device = mean_3d.device
        if bg is None:
            bg = torch.ones((3), dtype=torch.float32, device=device)

        # camera matrices in the rasterizer's row-vector convention (module.py:604-608)
        fov = get_fov(cam_param['focal'], cam_param['princpt'], img_shape)
        view_matrix = get_view_matrix(cam_param['R'], cam_param['t']).permute(1, 0).to(device)
        proj_matrix = get_proj_matrix(cam_param['focal'], cam_param['princpt'], img_shape, 0.01, 100, 1.0)
        proj_matrix = proj_matrix.permute(1, 0).to(device)
        full_proj_matrix = torch.mm(view_matrix, proj_matrix)
        cam_pos = view_matrix.inverse()[3, :3]
        raster_settings = GaussianRasterizationSettings(
            image_height=img_shape[0],
            image_width=img_shape[1],
            tanfovx=float(torch.tan(fov[0] / 2)),
            tanfovy=float(torch.tan(fov[1] / 2)),
            bg=bg,
            scale_modifier=1.0,
            viewmatrix=view_matrix,
            projmatrix=full_proj_matrix,
            sh_degree=0,  # dummy: rgb is already computed (module.py:618)
            campos=cam_pos,
            prefiltered=False,
            debug=False,
        )
        rasterizer = GaussianRasterizer(raster_settings=raster_settings)

        # screen-space position probe for the densification gradient (module.py:626-629)
        point_num = mean_3d.shape[0]
        mean_2d = torch.zeros((point_num, 3), dtype=torch.float32, device=device)
        mean_2d.requires_grad = True
        mean_2d.retain_grad()

        render_img, radius, render_depthmap, render_mask = rasterizer(
            means3D=mean_3d,
            means2D=mean_2d,
            shs=None,
            colors_precomp=rgb,
            opacities=opacity,
            scales=scale,
            rotations=rotation,
            cov3D_precomp=None)

        return {'img': render_img,
                'depthmap': render_depthmap,
                'mask': render_mask,
                'mean_2d': mean_2d,
                'is_vis': radius > 0,
                'radius': radius}
