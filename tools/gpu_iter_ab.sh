#!/bin/bash
# Per-kernel time of the five-render iteration ('sets') for library variants: bash tools/gpu_iter_ab.sh <variant|intree> ...
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  O=$R/gpurun_out/iter_ab/$v; rm -rf $O; mkdir -p $O
  L=""; [ $v != intree ] && L="EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/$v.so"
  env $L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/gpu_iteration_profile.py sets 40 > $O/log 2>&1
  echo "== $v: $(tail -1 $O/log | cut -c1-120)"
  python - <<PY
import csv, glob
f = glob.glob('$O/**/*kernel_stats.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'exa::' in r['Name']]
print('   ' + ', '.join('%s %.1f' % (r['Name'].replace('void exa::', '').replace('exa::', '').split('(')[0][:26], float(r['TotalDurationNs']) / 1e3 / 50) for r in rows))
PY
done
