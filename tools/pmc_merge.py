"""Merge the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh into one per-kernel table, the per-step total and the
per-operator bytes (amcontrast3d_amd/roofline.py OPERATOR_KERNELS; only kernels that belong to ONE operator are attributed):
python pmc_merge.py fetch.csv write.csv nsteps out.csv out.json [workload tag, e.g. S:192000]
Units: rocprofv3 reports both counters in KB.  Correction (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE tallies wide
coalesced reads at half their bytes -> doubled; WRITE_SIZE is exact for 16-byte stores and float atomics."""
import csv, json, sys
fetch, write, nsteps, out_csv, out_json = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
workload = sys.argv[6] if len(sys.argv) > 6 else None
rows = {}
for path, col in ((fetch, "fetch_kb"), (write, "write_kb")):
    for r in csv.DictReader(open(path)):
        d = rows.setdefault(r["kernel"], {"dispatches": int(r["dispatches"]), "fetch_kb": 0.0, "write_kb": 0.0})
        d[col] = float(r["sum"])
tot = 0.0
table = []
for k, d in rows.items():
    b = (2 * d["fetch_kb"] + d["write_kb"]) * 1024 / nsteps
    tot += b
    table.append((b, k, d))
table.sort(reverse=True)
with open(out_csv, "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches_per_step", "FETCH_SIZE_KB_per_step_raw", "WRITE_SIZE_KB_per_step", "HBM_MB_per_step_corrected", "share"])
    for b, k, d in table:
        w.writerow([k, round(d["dispatches"] / nsteps, 2), round(d["fetch_kb"] / nsteps, 1), round(d["write_kb"] / nsteps, 1),
                    round(b / 1e6, 2), round(b / tot, 4)])
amc = sum(b for b, k, d in table if "amc::" in k)
fam = {}
for b, k, d in table:
    for tag in ("bn_", "lagg_", "sat_", "gm_gemm", "pw_", "gb_gemm", "contrast_", "scatter_pm", "fps_", "kg_", "knn_", "nn3_", "Cijk_", "at::native", "rocclr"):
        if tag in k:
            fam[tag] = fam.get(tag, 0.0) + b
            break
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amcontrast3d_amd.roofline import OPERATOR_KERNELS
owners = {}
for op, pats in OPERATOR_KERNELS.items():
    for pat in pats:
        owners.setdefault(pat, []).append(op)
ops = {}
exclusive = {op for op, pats in OPERATOR_KERNELS.items() if all(len(owners[p]) == 1 for p in pats)}  # no kernel shared with another operator
for b, k, d in table:
    mine = {op for pat, os_ in owners.items() if pat in k and len(os_) == 1 for op in os_ if op in exclusive}
    if len(mine) == 1:
        op = mine.pop()
        e = ops.setdefault(op, {"bytes_per_step": 0.0, "dispatches_per_step": 0.0})
        e["bytes_per_step"] += b
        e["dispatches_per_step"] += d["dispatches"] / nsteps
out = {"_step": {"workload": workload, "bytes_per_step": tot, "amc_kernels_bytes_per_step": amc,
                 "by_family_GB": {k: round(v / 1e9, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])},
                 "note": "sum over every kernel of one eager train step of 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes, tools/pmc_passes.sh)"}}
out.update(ops)
json.dump(out, open(out_json, "w"), indent=1)
print(f"HBM traffic per step: {tot / 1e9:.2f} GB (amc kernels {amc / 1e9:.2f} GB); top kernels:")
for b, k, d in table[:14]: print(f"  {b / 1e6:8.1f} MB  x{d['dispatches'] / nsteps:5.1f}  {k[:100]}")
print({k: round(v / 1e9, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})
