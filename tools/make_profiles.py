"""Turns the raw output of tools/profile_round.sh (gpurun_out/<tag>/) into the committed summaries under profiles/:
   <prefix>_kernel_stats.md, <prefix>_hbm_traffic.{md,json}, <prefix>_pmc.md, <prefix>_bench.json.
   Usage: python tools/make_profiles.py gpurun_out/r01b r01_final"""
import sys, os, csv, glob, json, collections

import subprocess

src, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'profiles')
# Every summary is stamped with the build it measured: the digest the lease itself recorded (tools/profile_round.sh ->
# digest.txt) must be the digest of the sources in this tree -- a summary of another build is refused here, and bench.py quotes
# a summary only when its digest equals the loaded library's.
sys.path.insert(0, root)
from exavatar_release_amd import build as _b      # noqa: E402
DIGEST = open(os.path.join(src, 'digest.txt')).read().strip()
if DIGEST != _b._digest()[:16]:
    sys.exit('make_profiles: %s was measured on library build %s, this tree builds %s -- run tools/profile_round.sh on THIS build'
             % (src, DIGEST, _b._digest()[:16]))
HEAD = subprocess.run(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
DIRTY = bool(subprocess.run(['git', '-C', root, 'status', '--porcelain', '--', 'exavatar_release_amd/csrc', 'include'], capture_output=True, text=True).stdout.strip())
STAMP = 'library build %s (exavatar_release_amd.build._digest), sources at git %s%s' % (DIGEST, HEAD, ' + uncommitted changes' if DIRTY else '')
SHORT = {'zero_kernel': 'zero', 'preprocess_fwd_kernel': 'preprocess_fwd', 'col_scan_kernel': 'col_scan', 'cell_scan_kernel': 'cell_scan',
         'subtile_count_kernel': 'subtile_count',
         'cell_scatter_kernel': 'cell_scatter', 'subtile_bin_kernel': 'subtile_bin', 'sort_subtiles_kernel': 'sort_subtiles',
         'render_fwd_kernel': 'render_fwd', 'render_bwd_kernel': 'render_bwd', 'preprocess_bwd_kernel': 'preprocess_bwd'}


def short(name):
    for k, v in SHORT.items():
        if k in name:
            return v
    return None


def counters(d):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(src, d, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r['Kernel_Name'])
                if k:
                    rows[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in rows.items()}


bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
bench['profile_stamp'] = {'library_digest': DIGEST, 'git_head': HEAD}
with open(os.path.join(out, prefix + '_bench.json'), 'w') as f:
    json.dump(bench, f, indent=1)

# ---- kernel stats ----------------------------------------------------------------------------------------
stats = glob.glob(os.path.join(src, 'stats', '**', '*kernel_stats.csv'), recursive=True)
if len(stats) > 1:
    sys.exit('make_profiles: %s holds the output of more than one lease (%d kernel_stats files): gpurun MERGES into gpurun_out/ -- '
             'remove the directory before running tools/profile_round.sh again' % (src, len(stats)))
_ks = os.path.join(out, prefix + '_kernel_stats.md')
_tail = ''            # hand-written sections ("## ...") of an existing summary survive a regeneration
if os.path.exists(_ks):
    _old = open(_ks).read()
    if '\n## ' in _old:
        _tail = _old[_old.index('\n## '):]
with open(_ks, 'w') as f:
    f.write('# %s: rocprofv3 kernel statistics of the bench command\n\n%s.\n\n' % (prefix, STAMP))
    f.write('Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 100 --warmup 10 '
            '--no-cpu-baseline --no-concurrent` (C3, 1 x MI355X, launch protocol of that bench line (`config.launch`); includes the untimed calibration / warm-up '
            'launches of bench.py, hence more calls than steps).\n\n')
    f.write('Bench line of the same build (default `python bench.py`): %.1f it/s, %.4f ms/step; roofline kernel %s '
            'avg %.1f us (HIP events) .\n\n' % (bench['value'], bench['ms_per_step'], bench['roofline']['kernel'],
                                                  bench['roofline']['avg_launch_us']))
    f.write('| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|\n')
    if stats:
        with open(stats[0]) as g:
            for r in csv.DictReader(g):
                f.write('| %s | %s | %.1f | %.1f | %.1f | %s |\n' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3,
                                                                  float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Percentage']))
    f.write(_tail)

# machine-readable copy (bench.py quotes the dominant kernel's rocprof average next to its own HIP-event bracket)
if stats:
    kern = {}
    with open(stats[0]) as g:
        for r in csv.DictReader(g):
            k = short(r['Name'])
            if k and k not in kern:          # (first = the instantiation with the most GPU time)
                kern[k] = {'avg_us': float(r['AverageNs']) / 1e3, 'calls': int(r['Calls'])}
    with open(os.path.join(out, prefix + '_kernel_stats.json'), 'w') as f:
        json.dump({'what': 'rocprofv3 --kernel-trace --stats of `python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-concurrent` '
                           '(C3, the launch protocol of that bench line, ring average)', 'library_digest': DIGEST, 'git_head': HEAD,
                   'kernels': kern}, f, indent=1)

# ---- HBM traffic -----------------------------------------------------------------------------------------
fe, wr = counters('pmc_fetch'), counters('pmc_write')
alg = bench['roofline'].get('algorithmic_bytes_per_launch')
# calibration of the two counters on THIS box (tools/probe/fetch_calib.hip run under the same two PMC passes by
# tools/profile_round.sh): factor = known bytes / reported bytes
calib = {}
for d, c in (('calib_fetch', 'FETCH_SIZE'), ('calib_write', 'WRITE_SIZE')):
    rows = collections.defaultdict(list)
    for path in glob.glob(os.path.join(src, d, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r['Counter_Name'] == c:
                    rows[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    calib[c] = {k: sum(v) / len(v) for k, v in rows.items()}
GIB_KB = float(1 << 20)
known = {'stream_read': GIB_KB, 'stream_read4': GIB_KB, 'stream_read12': GIB_KB, 'gather64': 4 * (1 << 20) * 64 / 1024.0}
factors = {k: known[k] / calib['FETCH_SIZE'][k] for k in known if calib['FETCH_SIZE'].get(k)}
if calib['WRITE_SIZE'].get('stream_write'):
    factors['stream_write'] = GIB_KB / calib['WRITE_SIZE']['stream_write']
tr = {'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (two separate passes, counters only), python '
                'tools/gpu_kernel_times.py 0 (C3, ring view 0, eager, mean of the launches).  hbm_bytes_raw = (FETCH_SIZE + '
                'WRITE_SIZE) * 1024; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 = the gfx950 correction of '
                'MI355X_MICROARCH.md applied to ALL reads (an upper bound: the calibration probe shows factor 2 for coalesced '
                '4 / 12 / 16 B-per-lane streams but factor 1 for 64-byte record gathers, and these kernels mix both)',
      'calibration_factors': factors, 'library_digest': DIGEST, 'git_head': HEAD, 'kernels': {}}
with open(os.path.join(out, prefix + '_hbm_traffic.md'), 'w') as f:
    f.write('# %s: HBM-side traffic per kernel launch (PMC)\n\n%s.\n\n' % (prefix, STAMP))
    f.write('Two separate PMC passes (counters only, `--kernel-trace`, no other trace domains): `rocprofv3 --kernel-trace --pmc '
            'FETCH_SIZE --output-format csv -- python tools/gpu_kernel_times.py 0` and the same with `--pmc WRITE_SIZE`. Config '
            'C3, ring view 0 (exact mode: the two-stage protocol, so col_scan / cell_scan appear as launches).\n\n'
            'Calibration on the same box in the same two passes (`tools/probe/fetch_calib.hip`, 1 GiB buffer, known byte '
            'counts; factor = actual / reported): ' + ', '.join('%s x%.2f' % (k, v) for k, v in sorted(factors.items())) +
            '. I.e. FETCH_SIZE reports HALF of every coalesced streaming read (4, 12 and 16 bytes per lane alike, as '
            'MI355X_MICROARCH.md says for 16 B/lane) but the full 64 bytes of a gathered record; WRITE_SIZE is exact. The '
            'render kernels mix record gathers with streams (ids, checkpoints, pixel gradients), so both the raw sum and the '
            'all-reads-doubled upper bound are listed; `bench.py` reports the upper bound as `roofline.traffic`.\n\n')
    f.write('| kernel | FETCH_SIZE (KB) | WRITE_SIZE (KB) | raw bytes | upper bound (2 x FETCH + WRITE) |\n|---|---|---|---|---|\n')
    for k in SHORT.values():
        if k in fe and k in wr:
            a, b = fe[k].get('FETCH_SIZE', 0.0), wr[k].get('WRITE_SIZE', 0.0)
            tr['kernels'][k] = {'fetch_KB': a, 'write_KB': b, 'hbm_bytes_raw': (a + b) * 1024, 'hbm_bytes': (2 * a + b) * 1024}
            f.write('| %s | %.0f | %.0f | %.3g | %.3g |\n' % (k, a, b, (a + b) * 1024, (2 * a + b) * 1024))
with open(os.path.join(out, prefix + '_hbm_traffic.json'), 'w') as f:
    json.dump(tr, f, indent=1)

# ---- SQ counters -----------------------------------------------------------------------------------------
s1, s2 = counters('pmc_sq1'), counters('pmc_sq2')
with open(os.path.join(out, prefix + '_pmc.md'), 'w') as f:
    f.write('# %s: SQ counters per kernel (mean per dispatch; SQ_*_CYCLES / ACTIVE / WAIT in quad-cycles)\n\n%s.\n\n' % (prefix, STAMP))
    f.write('Two PMC passes of `python tools/gpu_kernel_times.py 0` (C3, ring view 0, eager), counters only with --kernel-trace.\n\n')
    for tab in (s1, s2):
        names = sorted({c for k in tab.values() for c in k})
        if not names:
            continue
        f.write('| kernel | ' + ' | '.join(names) + ' |\n|---|' + '---|' * len(names) + '\n')
        for k in SHORT.values():
            if k in tab:
                f.write('| %s | ' % k + ' | '.join('%.3g' % tab[k].get(c, float('nan')) for c in names) + ' |\n')
        f.write('\n')
    f.write('| kernel | VALU-active / wave-cycles | parked (WAIT_ANY) | issue-stalled (WAIT_INST_ANY) | LDS-active / wave-cycles |\n|---|---|---|---|---|\n')
    for k in SHORT.values():
        if k in s1 and s1[k].get('SQ_WAVE_CYCLES'):
            wc = s1[k]['SQ_WAVE_CYCLES']
            f.write('| %s | %.0f %% | %.0f %% | %.0f %% | %.0f %% |\n' % (
                k, 100 * s1[k].get('SQ_ACTIVE_INST_VALU', 0) / wc, 100 * s1[k].get('SQ_WAIT_ANY', 0) / wc,
                100 * s1[k].get('SQ_WAIT_INST_ANY', 0) / wc, 100 * s2.get(k, {}).get('SQ_ACTIVE_INST_LDS', 0) / wc))
# machine-readable copy of the SQ counters (bench.py: roofline.secondary prices the blends' VALU instruction counts)
with open(os.path.join(out, prefix + '_pmc.json'), 'w') as f:
    merged = {k: dict(s1.get(k, {}), **s2.get(k, {})) for k in SHORT.values() if k in s1 or k in s2}
    json.dump({'what': 'rocprofv3 --kernel-trace --pmc SQ_* passes of `python tools/gpu_kernel_times.py 0` (C3, ring view 0, eager), '
                       'mean per dispatch', 'library_digest': DIGEST, 'git_head': HEAD, 'kernels': merged}, f, indent=1)
print('wrote profiles/%s_*' % prefix)
