"""Turns the iteration / C5 / host parts of tools/profile_round.sh's raw output (gpurun_out/<tag>/) into the committed summaries
profiles/<prefix>_iteration.md, <prefix>_c5_kernel_stats.md and <prefix>_host.md (tools/make_profiles.py writes the others).
Usage: python tools/make_iteration_profiles.py gpurun_out/r04e r04"""
import csv
import glob
import json
import os
import sys

src, prefix = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')


def stats_table(d, top=30):
    paths = glob.glob(os.path.join(src, d, '**', '*kernel_stats.csv'), recursive=True)
    lines = ['| kernel | calls | avg us | min us | max us | % of GPU time |', '|---|---|---|---|---|---|']
    rows = list(csv.DictReader(open(paths[0]))) if paths else []
    for r in rows[:top]:
        lines.append('| %s | %s | %.1f | %.1f | %.1f | %s |' % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3,
                                                                float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Percentage']))
    return '\n'.join(lines), rows


def read(name):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return ''
    return ''.join(ln for ln in open(p) if 'amdgpu.ids' not in ln)


def per_iteration(rows, iters):
    """Sum of calls x average per kernel family, divided by the iterations of the run."""
    fam = {}
    for r in rows:
        n = r['Name']
        key = 'rasterizer' if 'exa::' in n or n.startswith('store_pointers') else 'PyTorch / runtime'
        fam[key] = fam.get(key, 0.0) + int(r['Calls']) * float(r['AverageNs']) / 1e3
    return {k: v / iters for k, v in fam.items()}


# ---- the five-render iteration ---------------------------------------------------------------------------------------------
g_tab, g_rows = stats_table('iter_graphed')
s_tab, s_rows = stats_table('iter_sets')
g_sum, s_sum = per_iteration(g_rows, 70), per_iteration(s_rows, 50)
bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
it = bench.get('extra_exavatar_iteration', {})
with open(os.path.join(out, prefix + '_iteration.md'), 'w') as f:
    f.write('# %s: the five-render ExAvatar iteration (100 k Dist-C scene + 50 k avatar, 1024 x 1024, fwd + bwd) per kernel\n\n' % prefix)
    f.write('`rocprofv3 --kernel-trace --stats -- python tools/gpu_iteration_profile.py graphed 60` (`GraphedIteration`: two hipGraphs per\n'
            'iteration; ~70 iterations incl. warm-up, so calls / 70 = launches per iteration) and `... sets 40` (eager `render_iteration`,\n'
            '~50 iterations).  Loss = `sum((img * G).sum())` in PyTorch (the `at::native` kernels below: ~0.19 ms of GPU time per\n'
            'iteration).  Round 4: composites copy the scene render\'s pixels where the human has no entry; the backward reads dL/dimg\n'
            'through a pointer table; `is_vis` comes from the forward kernel, the composites\' `radius` / `is_vis` are built on first access\n'
            '(no comparison / concatenation kernels); a composite\'s gradients for the human are added inside the per-Gaussian kernel of\n'
            'the human\'s own render (no `add` kernels from autograd); backward launches bounded by the batch slots in use; inside the\n'
            'captured graphs the composites\' list merges and backward run on a side stream next to their sources\' blend (so the kernel\n'
            'times below overlap and add up to more than the iteration).\n\n')
    f.write('Sum of calls x average per iteration: GraphedIteration %s; eager %s (us).\n\n'
            % (', '.join('%s %.0f' % kv for kv in sorted(g_sum.items())), ', '.join('%s %.0f' % kv for kv in sorted(s_sum.items()))))
    f.write('## `extra_exavatar_iteration` of the bench line of this run (ms per iteration, median of three windows)\n\n')
    f.write('| how | ms / iteration | host ms | windows |\n|---|---|---|---|\n')
    for k, v in it.items():
        if isinstance(v, dict):
            f.write('| %s | %.4f | %.4f | %s |\n' % (k, v['ms_per_iteration'], v['host_ms_per_iteration'], v['windows_ms']))
    f.write('\n`graphed_raster_only`: dL/dimg handed straight to `backward` (what the headline measures for one render);\n'
            '`*_in_graph`: `GraphedIteration(loss_fn=...)`, forward + loss + backward in ONE hipGraph per iteration.\n\n')
    f.write('## GraphedIteration\n\n' + g_tab + '\n\n## eager render_iteration (same build)\n\n' + s_tab + '\n\n')
    f.write('## host / device split (`tools/gpu_graphed_iter_profile.py`; "host only" = Python time per segment while the GPU runs behind,\n'
            '"device-inclusive" = with a synchronize after every segment)\n\n```\n' + read('graphed_iter_profile.log') + '```\n\n')
    f.write('## repeated timings (`tools/gpu_iter_repeat.py`: bench.iteration_throughput twice in one process; `tools/gpu_iteration_profile.py <how> 300`)\n\n```\n'
            + '\n'.join(ln[:900] for ln in read('iter_repeat.log').splitlines()[-2:]) + '\n' + read('iter_times.log') + '```\n')

# ---- C5 --------------------------------------------------------------------------------------------------------------------
c_tab, _ = stats_table('c5_stats', 16)
c5 = json.loads(read('c5_bench.json').strip().splitlines()[-1])
rf = c5.get('roofline', {})
with open(os.path.join(out, prefix + '_c5_kernel_stats.md'), 'w') as f:
    f.write('# %s: C5 (BASELINE configs[4]) per kernel -- 300 k Gaussians, SH degree 3 in-kernel, 2048 x 2048, forward only, hipGraph\n\n' % prefix)
    f.write('Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline\n'
            '--no-concurrent --no-other-configs` (1 x MI355X; includes the untimed calibration / settle launches of bench.py).\n\n')
    f.write('Bench line of that run: **%.1f frames/s, %.4f ms/frame** (launch protocol: `config.launch` of that line, no backward context stored).\n'
            % (c5['value'], c5['ms_per_step']))
    if rf:
        st = rf.get('step', {})
        f.write('Roofline of that line: dominant kernel `%s` %.1f MB / %.1f us (HIP events) = %.0f GB/s = **%.3f** of 8 TB/s; whole step '
                '(60 P + 88 V + 40 D + 28 W H + 192 P SH coefficients = %.1f MB with the run\'s own V and D) = %.0f GB/s = **%.3f** (the '
                'default bench line carries the step figure as `extra_c5_forward.roofline`).\n'
                % (rf.get('kernel'), rf['algorithmic_bytes_per_launch'] / 1e6, rf['avg_launch_us'], rf['achieved'], rf['frac'],
                   st.get('algorithmic_bytes', 0) / 1e6, st.get('achieved_GBs_at_measured_step', 0), st.get('frac_at_measured_step', 0)))
    f.write('\n' + c_tab + '\n')

# ---- host ------------------------------------------------------------------------------------------------------------------
with open(os.path.join(out, prefix + '_host.md'), 'w') as f:
    f.write('# %s: host side of the drop-in surface (1 x MI355X box of the round)\n\n' % prefix)
    hp = read('host_profile.log').splitlines()
    keep = [ln for ln in hp if ln.startswith(('eager render', 'host:'))]
    f.write('## `tools/gpu_host_profile.py` (eager `GaussianRenderer` fwd + bwd, C3, `config.mode = \'auto\'`)\n\n```\n'
            + '\n'.join(keep) + '\n```\n\n')
    f.write('## `tools/gpu_graphed_times.py` (`GraphedRenderer`, C5)\n\n```\n' + read('graphed.log') + '```\n')
print('wrote profiles/%s_{iteration,c5_kernel_stats,host}.md' % prefix)
