#!/usr/bin/env python3
"""Print the key figures of a closing pass (gpurun_out/r02_final/bench.json + gpurun_out/prof_r02/)."""
import json, sys, re
b = json.load(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r02_final/bench.json'))
print(b['metric']); print(round(b['value']), round(b['ms_per_step'], 3), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in b['roofline'].items() if k != 'traffic_source'})
print([(p['L_pq'], round(p['recall_at_10'], 4), round(p['pct_of_8000'], 1), round(p['qps'])) for p in b['L_pq_sweep']])
for k in ('cpu_baseline', 'cpu_baseline_1_thread', 'cpu_baseline_config1', 'roofline_worstcase', 'host_form_pcie_inclusive', 'non_parity_modes'):
    print(k, json.dumps(b.get(k))[:520])
print(json.dumps(b['gt_build'])[:330]); print(b['config']['setup_seconds'])
try:
    for e in json.load(open('gpurun_out/prof_r02/search_traffic.json')):
        print(e['workload']['graph'], e['workload']['L'], 'fetch %.1f write %.2f alg %.1f GB, %.2f ms' % (e['fetch_bytes_corrected'] / 1e9, e['write_bytes'] / 1e9, e['algorithmic_bytes_per_launch'] / 1e9, e['kernel_ms_avg_under_rocprof']), [re.sub(r'\(.*', '', k)[5:] for k in e['kernels']])
except Exception as ex: print(ex)
