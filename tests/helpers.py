"""Shared helpers of the parity tests: tolerances are BASELINE.json's (image L-inf 1e-4, grads 1e-3 rel)."""
import torch

IMG_TOL = 1e-4        # per-pixel L-infinity on colour / depth / alpha (north_star)
GRAD_REL_TOL = 1e-3   # relative gradient error (north_star)
AMBIGUOUS_MAX_FRACTION = 5e-3   # pixels whose discrete decisions sit within 1e-4 (relative) of a threshold
AMBIGUOUS_TOL = 2e-2            # such a pixel may flip one alpha >= 1/255 / T < 1e-4 decision


def grad_rel_err(got, ref):
    """max-norm and L2 relative errors of a gradient tensor."""
    ref = ref.double()
    got = got.double()
    scale = ref.abs().max().clamp_min(1e-30)
    return float((got - ref).abs().max() / scale), float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def assert_image_close(got, ref, ambiguous, name='img'):
    got = got.detach().cpu().float()
    ref = ref.detach().cpu().float()
    d = (got - ref).abs()
    amb = ambiguous.bool()
    frac = float(amb.float().mean())
    assert frac <= AMBIGUOUS_MAX_FRACTION, '%s: %.2e of the pixels are ambiguous' % (name, frac)
    strict = d.clone()
    strict[..., amb] = 0
    assert float(strict.max()) <= IMG_TOL, '%s: L-inf %.3e on unambiguous pixels' % (name, float(strict.max()))
    if amb.any():
        assert float(d[..., amb].max()) <= AMBIGUOUS_TOL, '%s: ambiguous pixel off by %.3e' % (name, float(d[..., amb].max()))


def assert_grads_close(got, ref, name):
    mx, l2 = grad_rel_err(got.detach().cpu(), ref.detach().cpu())
    assert mx <= GRAD_REL_TOL and l2 <= GRAD_REL_TOL, 'grad %s: max-rel %.3e, L2-rel %.3e' % (name, mx, l2)
