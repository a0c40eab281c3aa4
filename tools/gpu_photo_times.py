"""Kernel times of the fused photometric loss (photo_stats_kernel + photo_grad_kernel, csrc/ssim.hip) on a 3 x H x W render, full
image and a bbox crop, against their algorithmic bytes (stats: read 8 B + write 12 B per pixel and channel; grad: read 20, write 4).
python tools/gpu_photo_times.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exavatar_release_amd import _lib
from exavatar_release_amd.rasterizer import _ptr, _stream_ptr
dev = torch.device('cuda:0'); lib = _lib.load()
for (H, W, crop) in ((1024, 1024, (0, 0, 1024, 1024)), (1024, 1024, (256, 128, 512, 768)), (540, 960, (0, 0, 960, 540))):
    B, C = 1, 3
    x = torch.rand(B, C, H, W, device=dev); y = torch.rand(B, C, H, W, device=dev)
    cw, ch = crop[2], crop[3]
    n = B * C * cw * ch
    maps = torch.empty(3 * n, device=dev); nblk = int(lib.exa_photo_loss_blocks(B, C, cw, ch))
    partials = torch.empty(nblk, 2, device=dev); dimg = torch.zeros_like(x)
    cc = (ctypes.c_int32 * 4)(*crop); st = _stream_ptr(dev)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for rep in range(2):
        e[0].record()
        for _ in range(50):
            _lib.check(lib.exa_photo_loss_forward(B, C, H, W, cc, _ptr(x), _ptr(y), None, None, _ptr(maps), _ptr(partials), st))
        e[1].record()
        for _ in range(50):
            _lib.check(lib.exa_photo_loss_grad(B, C, H, W, cc, _ptr(x), _ptr(y), None, None, 0.8, 0.2, _ptr(maps), _ptr(dimg), st))
        e[2].record(); torch.cuda.synchronize()
    tf, tb = e[0].elapsed_time(e[1]) / 50 * 1e3, e[1].elapsed_time(e[2]) / 50 * 1e3
    print('%dx%d crop %s: stats %.1f us (%.2f TB/s of 20 B/px/ch), grad %.1f us (%.2f TB/s of 24 B/px/ch)' % (
        W, H, crop, tf, n * 20 / tf / 1e6, tb, n * 24 / tb / 1e6))
