#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line (file argument): step times, class tables of the extra configs."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print('C2 ms', d['ms_per_step'], 'xRT', d['value'], 'k7 frac', r.get('frac'), 'whole TF', r.get('whole_step_tflops'))
for k, v in r['all_classes'].items():
    print('   ', k, v)
print('cpu', json.dumps(d.get('cpu_baseline'))[:600])
for name, e in (d.get('extra_configs') or {}).items():
    if not isinstance(e, dict):
        print(name, e); continue
    print('==', name, 'ms', e['ms_per_step'], 'xRT', e['x_realtime'], 'TF', e['whole_step_tflops'], 'kernel ms', e['kernel_ms_per_step'], e.get('gpu_equals_oracle'))
    for k, v in e['classes'].items():
        print('   ', k, v)
