"""CPU oracle for the image losses in front of the rasterizer's backward.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (only tests/, smoke() and bench's cpu_baseline may import oracle/).

PARITY PINNED: unlike the rasterizer, these losses are in the reference tree (``avatar/common/nets/loss.py``).  The
functions below restate ``RGBLoss.forward`` (loss.py:15-29) and ``SSIM.forward`` (loss.py:45-74, window from
loss.py:35-43) statement by statement without the ``.cuda()`` calls; ``tests/golden/ref_ssim.npz`` holds outputs and
autograd gradients produced by EXECUTING the reference's own class source (tests/golden/make_golden_ssim.py), and
tests/test_oracle.py checks this restatement against them.
"""
import math

import torch
from torch.nn import functional as F


def _crop(img_out, img_target, bbox):
    img_height, img_width = img_out.shape[2:]
    xmin, ymin, width, height = [int(x) for x in bbox[0]]
    xmin = max(xmin, 0)
    ymin = max(ymin, 0)
    xmax = min(xmin + width, img_width)
    ymax = min(ymin + height, img_height)
    return img_out[:, :, ymin:ymax, xmin:xmax], img_target[:, :, ymin:ymax, xmin:xmax]


def rgb_loss(img_out, img_target, bbox=None, mask=None, bg=None):
    """reference loss.py:15-29"""
    if (mask is not None) and (bg is not None):
        img_target = img_target * mask + (1 - mask) * bg[:, :, None, None]
    if bbox is not None:
        img_out, img_target = _crop(img_out, img_target, bbox)
    return torch.abs(img_out - img_target)


def ssim_window(window_size, feat_dim, dtype=torch.float32):
    """reference loss.py:35-43: 1-D Gaussian (sigma 1.5) from math.exp, normalised in float32, outer product"""
    gauss = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)],
                         dtype=torch.float32)
    window_1d = (gauss / gauss.sum())[:, None]
    window_2d = torch.mm(window_1d, window_1d.permute(1, 0))[None, None, :, :]
    return window_2d.repeat(feat_dim, 1, 1, 1).to(dtype)


def ssim_map(img_out, img_target, bbox=None, mask=None, window_size=11):
    """reference loss.py:45-74"""
    batch_size, feat_dim, img_height, img_width = img_out.shape
    if mask is not None:
        img_out = img_out * mask
        img_target = img_target * mask
    if bbox is not None:
        img_out, img_target = _crop(img_out, img_target, bbox)
    window = ssim_window(window_size, feat_dim, img_out.dtype).to(img_out.device)
    pad = window_size // 2
    mu1 = F.conv2d(img_out, window, padding=pad, groups=feat_dim)
    mu2 = F.conv2d(img_target, window, padding=pad, groups=feat_dim)
    mu1_sq = mu1 ** 2
    mu2_sq = mu2 ** 2
    mu1_mu2 = mu1 * mu2
    sigma1_sq = F.conv2d(img_out * img_out, window, padding=pad, groups=feat_dim) - mu1_sq
    sigma2_sq = F.conv2d(img_target * img_target, window, padding=pad, groups=feat_dim) - mu2_sq
    sigma1_sigma2 = F.conv2d(img_out * img_target, window, padding=pad, groups=feat_dim) - mu1_mu2
    C1 = 0.01 ** 2
    C2 = 0.03 ** 2
    return ((2 * mu1_mu2 + C1) * (2 * sigma1_sigma2 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))


def photometric_loss(img_out, img_target, bbox=None, l1_weight=None, ssim_mask=None, w_l1=0.8, w_ssim=0.2):
    """The per-render objective the reference assembles from the two classes above (avatar/main/model.py:197-198 for the
    human renders, :214-215 for the scene render with l1_weight = ssim_mask = 1 - mask; weights config.py:35-36; the
    ``.mean()`` is avatar/main/train.py:43):  (rgb_loss * w_l1 [* l1_weight]).mean() + ((1 - ssim) * w_ssim).mean()."""
    l1 = rgb_loss(img_out, img_target, bbox=bbox)
    if l1_weight is not None:
        l1 = l1 * (l1_weight if bbox is None else _crop(l1_weight, l1_weight, bbox)[0])
    return (l1 * w_l1).mean() + ((1 - ssim_map(img_out, img_target, bbox=bbox, mask=ssim_mask)) * w_ssim).mean()
