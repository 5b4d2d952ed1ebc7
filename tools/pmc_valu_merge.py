"""merge the per-counter tables of tools/pmc_valu.sh into one per-kernel table: python pmc_valu_merge.py <prefix> <nsteps> out.csv"""
import csv, sys
prefix, nsteps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"]
tab = {}
for c in names:
    for r in csv.DictReader(open(f"{prefix}_pmc_{c}.csv")):
        d = tab.setdefault(r["kernel"], {"n": int(r["dispatches"])})
        d[c] = float(r["sum"])
rows = []
for k, d in tab.items():
    if any(c not in d for c in names) or d["GRBM_GUI_ACTIVE"] <= 0:
        continue
    cyc = d["GRBM_GUI_ACTIVE"] / 8.0            # shader cycles, summed over the dispatches
    rows.append({"kernel": k[:110], "dispatches_per_step": d["n"] / nsteps, "Mcycles_per_step": cyc / nsteps / 1e6,
                 "valu_busy": d["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), "inst_busy": d["SQ_ACTIVE_INST_ANY"] * 4 / (1024 * cyc),
                 "wait_share": d["SQ_WAIT_ANY"] / max(d["SQ_WAVE_CYCLES"], 1), "issue_stall_share": d["SQ_WAIT_INST_ANY"] / max(d["SQ_WAVE_CYCLES"], 1),
                 "valu_insts_per_wave": d["SQ_INSTS_VALU"] / max(d["SQ_WAVES"], 1), "waves_per_dispatch": d["SQ_WAVES"] / d["n"]})
rows.sort(key=lambda r: -r["Mcycles_per_step"])
with open(out, "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader()
    for r in rows:
        w.writerow({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
for r in rows[:60]:
    print(f"{r['Mcycles_per_step']:7.3f} Mcyc x{r['dispatches_per_step']:5.1f}  valu {r['valu_busy']:.2f} inst {r['inst_busy']:.2f} wait {r['wait_share']:.2f} stall {r['issue_stall_share']:.2f}  valu/wave {r['valu_insts_per_wave']:8.0f}  {r['kernel'][:70]}")
