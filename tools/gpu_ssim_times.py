"""Measures SURVEY 8f-4: fused SSIM (forward + backward) vs the reference's conv2d formulation in PyTorch on the same GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exavatar_release_amd as exa
from oracle import loss_oracle as lo      # the reference's statements (conv2d), used here as the PyTorch-on-GPU comparison

dev = torch.device('cuda:0')
for shape in ((1, 3, 1024, 1024), (1, 3, 540, 960)):
    x = torch.rand(*shape, device=dev, requires_grad=True)
    y = torch.rand(*shape, device=dev)
    fused = exa.SSIM()

    def run(fn):
        x.grad = None
        (1 - fn(x, y)).mean().backward()
    for name, fn in (('fused HIP kernels', fused), ('reference conv2d formulation (PyTorch-ROCm)', lambda a, b: lo.ssim_map(a, b))):
        for _ in range(5):
            run(fn)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            run(fn)
        torch.cuda.synchronize()
        print('%s  %-45s fwd+bwd incl. mean(): %.3f ms' % (shape, name, (time.perf_counter() - t0) / n * 1e3))
    # kernel-only timing: 50 back-to-back launches through the C ABI between two HIP events
    from exavatar_release_amd import _lib
    from exavatar_release_amd.rasterizer import _ptr, _stream_ptr
    lib = _lib.load()
    B, C, H, W = shape
    xd = x.detach().contiguous(); m = torch.empty_like(xd); maps = [torch.empty_like(xd) for _ in range(3)]
    g = torch.ones_like(xd); dx = torch.empty_like(xd)
    e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    st = _stream_ptr(dev)
    for rep in range(2):
        e0.record()
        for _ in range(50):
            _lib.check(lib.exa_ssim_forward(B * C, H, W, _ptr(xd), _ptr(y), _ptr(m), _ptr(maps[0]), _ptr(maps[1]), _ptr(maps[2]), st))
        e1.record()
        for _ in range(50):
            _lib.check(lib.exa_ssim_backward(B * C, H, W, _ptr(xd), _ptr(y), _ptr(g), _ptr(maps[0]), _ptr(maps[1]), _ptr(maps[2]), _ptr(dx), st))
        e2.record(); torch.cuda.synchronize()
    tf, tb = e0.elapsed_time(e1) / 50, e1.elapsed_time(e2) / 50
    px = B * C * H * W
    print('%s  kernels: forward %.1f us (%.2f TB/s of 24 B/px algorithmic), backward %.1f us (%.2f TB/s of 28 B/px)' % (
        shape, tf * 1e3, px * 24 / (tf * 1e-3) / 1e12, tb * 1e3, px * 28 / (tb * 1e-3) / 1e12))
