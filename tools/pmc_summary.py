"""Developer tool: summarise rocprofv3 --pmc csv output (mean counter value per dispatch, per kernel)."""
import sys, csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in rows.values() for c in k})
print('| kernel | ' + ' | '.join(names) + ' |')
print('|---|' + '---|' * len(names))
for k, cs in rows.items():
    if not k.startswith('exa::') and 'exa' not in k: continue
    print('| %s | ' % k[:48] + ' | '.join('%.3g' % (sum(cs[c]) / len(cs[c])) if cs.get(c) else '-' for c in names) + ' |')
