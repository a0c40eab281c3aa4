"""<dir>/parity_stats.jsonl (written by the GPU suite of ONE lease: tools/profile_round.sh) -> profiles/<tag>_parity.md, stamped
with the library build that lease ran.  Fails when any of the seven full-size workloads is missing (a stale or partial file
must not turn into a summary) or when the lease's build is not this tree's.
Usage: python tools/make_parity_profile.py <tag> [dir=gpurun_out/<tag>]   (the last record of every workload wins)"""
import json
import os
import sys

import subprocess

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, 'gpurun_out', tag)
sys.path.insert(0, root)
from exavatar_release_amd import build as _b      # noqa: E402
digest = open(os.path.join(src, 'digest.txt')).read().strip()
if digest != _b._digest()[:16]:
    sys.exit('make_parity_profile: %s was measured on library build %s, this tree builds %s' % (src, digest, _b._digest()[:16]))
head = subprocess.run(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
rows = {}
for line in open(os.path.join(src, 'parity_stats.jsonl')):
    r = json.loads(line)
    rows[r['tag']] = r
FULL_SIZE = ('c3_P150000_view0', 'c3_P150000_view37', 'c3_P167000_view113', 'c2', 'c2l', 'c3s', 'c5_fwd_sh3')
missing = [t for t in FULL_SIZE if t not in rows]
if missing:
    sys.exit('make_parity_profile: %s holds no record of %s -- the full-size tests of that lease did not run' % (src, missing))
order = list(FULL_SIZE)
order += [t for t in rows if t not in order and 'img' in rows[t]]
fuzz = [rows[t] for t in sorted((t for t in rows if t.startswith('fuzz_')), key=lambda t: int(t.split('_')[1]))]
two_rank = rows.get('two_rank_gradient')
out = ['# %s: HIP path vs CPU oracle at BASELINE.json\'s full sizes (MI355X, `pytest -m gpu tests/test_gpu_fullsize.py`)' % tag, '',
       'Library build %s (exavatar_release_amd.build._digest), sources at git %s; %d records of ONE GPU-suite run (tools/profile_round.sh).' % (digest, head, len(rows)), '',
       'Rows `*_vs_c_oracle` and `c5_fwd_sh3` are checked against the C restatement (oracle/c), the others against the PyTorch oracle; '
       'the two oracles agree with each other to 1e-6 on these workloads (tests/test_c_oracle.py).', '',
       'Written by the tests themselves (`tests/helpers.record_stats` -> gpurun_out/parity_stats.jsonl, turned into this file by '
       '`tools/make_parity_profile.py`), measured before the asserts.  "ambiguous" = pixels whose alpha >= 1/255, T < 1e-4 or '
       'power > 0 decision lies within 1e-4 (relative) of its threshold in the ORACLE (a property of the scene, identical on any '
       'machine); "off" = how many of them actually differ by > 1e-4 between the two fp32 implementations.  Gradients: max-norm and '
       'L2 relative error over the whole tensor, and the absolute floor (in units of the mean per-Gaussian gradient magnitude) a '
       'per-Gaussian `|err| <= 1e-3 |ref| + floor` check needs -- for Gaussians away from / near an ambiguous pixel.', '',
       '| workload | P | image | ambiguous px | off (img / depth / alpha) | L-inf on unambiguous px (img / depth / alpha) | radii equal |',
       '|---|---|---|---|---|---|---|']
for t in order:
    r = rows[t]
    out.append('| %s | %d | %dx%d | %d | %d / %d / %d | %.1e / %.1e / %.1e | %s |' % (
        t, r['P'], r['H'], r['W'], r['img']['n_ambiguous'], r['img']['n_ambiguous_off'], r['depth']['n_ambiguous_off'],
        r['alpha']['n_ambiguous_off'], r['img']['linf_unambiguous'], r['depth']['linf_unambiguous'],
        r['alpha']['linf_unambiguous'], r.get('radii_equal', True)))
out += ['', '| workload | tensor | max-rel | L2-rel | per-Gaussian floor needed (clean) | (near ambiguous px) |', '|---|---|---|---|---|---|']
for t in order:
    r = rows[t]
    for k in ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb', 'mean_2d'):
        g = r.get('grad_' + k)
        if g:
            out.append('| %s | %s | %.2e | %.2e | %.2e | %.2e |' % (t, k, g['max_rel'], g['l2_rel'], g['per_gaussian_floor_needed'],
                                                                  g['per_gaussian_floor_needed_near_ambiguous']))
out += ['', 'Bars asserted by the tests: image L-inf 1e-4 on unambiguous pixels; ambiguous count <= ~1.3 x the oracle\'s own count per '
        'workload; at most max(3, 1 %) of the ambiguous pixels off; radii bit-equal; gradients 1e-3 relative globally and per Gaussian '
        '(floor 1e-3 clean, 1e-1 for Gaussians whose 3-sigma square covers an ambiguous pixel).']
if fuzz:
    out += ['', '## Edge-case fuzz through the HIP path (tests/test_gpu_edge_cases.py, 16 seeded trials)', '',
            'Opacity 0 / 1 / at the 1/255 bar, scales x1e-4 .. x300, centres at / around / behind the near plane and on the camera plane, '
            'unnormalised quaternions, ragged sizes, all three image gradients.  Radii / visibility bit-equal with the C oracle, culled '
            'Gaussians exactly zero, images 1e-4 off ambiguous pixels in every trial; gradients against the float64 oracle where no '
            'decision flips between float32 and float64 and no pixel is ambiguous (max-norm relative error, bar 1e-3):', '',
            '| trial | H x W | P | ambiguous px | arbiter usable (no f32 / f64 flip) | HIP image within 1e-4 on EVERY pixel | worst gradient error |', '|---|---|---|---|---|---|---|']
    for r in fuzz:
        errs = [v for k, v in r.items() if k.startswith('grad_')]
        out.append('| %d | %dx%d | %d | %d | %s | %s | %s |' % (r['trial'], r['H'], r['W'], r['P'], r['n_ambiguous'], r['same'], r.get('agree_everywhere', '-'),
                                                        ('%.1e' % max(errs)) if errs else '-'))
if two_rank:
    out += ['', '## Two ranks: all-reduced flat gradient vs the single-process sum over the same six views', '',
            '| tensor | max-norm relative difference (bar 1e-6) |', '|---|---|']
    out += ['| %s | %.1e |' % (k, v) for k, v in two_rank['rel_err'].items()]
path = os.path.join(root, 'profiles', '%s_parity.md' % tag)
open(path, 'w').write('\n'.join(out) + '\n')
print('wrote', path, 'with', len(order), 'workloads')
