#!/bin/bash
# Round 5, run D: VALU probe, GPU suite (LBS tests, single protocol), full default bench line.
mkdir -p gpurun_out/r05d
./tools/probe/valu_probe > gpurun_out/r05d/valu_probe.txt 2>&1; cat gpurun_out/r05d/valu_probe.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r05d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05d/pytest.log
tail -4 gpurun_out/r05d/pytest.log
timeout 900 python bench.py > gpurun_out/r05d/bench.json 2> gpurun_out/r05d/bench.err; tail -c 400 gpurun_out/r05d/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05d/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d['roofline'].get('secondary'), indent=1)[:1500])
print(json.dumps(d['config'].get('list_length_histogram')))
print(json.dumps(d.get('extra_c3_lbs'))[:900])
PY
