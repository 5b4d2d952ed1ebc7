import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'],d['ms_per_step'])
print(d.get('pipeline_parts_alone'))
ks=d['kernels']
for k,v in sorted(ks.items(), key=lambda kv:-kv[1]['ms_per_step']): print(f"  {k:28s} {v['ms_per_step']:8.3f} ms  x{v['launches_per_step']}")
print(d['roofline']); print(d.get('roofline_hbm')); print(d.get('cpu_baseline'))
