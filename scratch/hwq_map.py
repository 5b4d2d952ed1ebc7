"""Which hardware queue does each kernel family of the pipelined step land on?  Parses an AMD_LOG_LEVEL=4 log:
counts (kernel family, HWq) pairs from the 'ShaderName' / 'SWq=..., HWq=...' line pairs."""
import collections, re, sys
fam = {"fps_kernel<48": "fps1", "fps_kernel<12": "fps2", "gcc_fwd": "feat", "kg_query": "geomB", "adamw": "update",
       "multi_tensor_apply": "update", "ccl": "rccl", "bn_stats": "feat-bn"}
cnt = collections.Counter(); last = None; queues = {}
hq = re.compile(r"SWq=(0x[0-9a-f]+), HWq=(0x[0-9a-f]+), id=(\d+)")
created = 0
for line in open(sys.argv[1], errors="replace"):
    if "ShaderName" in line:
        last = line.split("ShaderName :")[-1].strip()
        continue
    if "hsa_queue_create" in line or "acquireQueue" in line or "created hardware queue" in line.lower():
        created += 1
    m = hq.search(line)
    if m and last is not None:
        name = next((v for k, v in fam.items() if k in last), None)
        if name:
            cnt[(name, m.group(3), m.group(1)[-6:])] += 1
        last = None
print("queue-creation log lines:", created)
for (name, qid, sw), n in sorted(cnt.items()):
    print(f"{name:10s} HWq id={qid:3s} SWq=..{sw}  x{n}")
