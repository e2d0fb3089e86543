"""Condense a bench.py JSON line read from stdin: python bench.py ... | python tools/bench_line.py [tag]"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
tag = sys.argv[1] if len(sys.argv) > 1 else ''
k = d.get('kernel_ms_per_step') or {n: v['ms_per_step'] for n, v in d['roofline']['others'].items()}
print(tag, d['config']['per_gpu_batch'], round(d['value']), 'utt/s', round(d['ms_per_step'], 2), 'ms', {n[:28]: round(v, 2) for n, v in k.items()})
