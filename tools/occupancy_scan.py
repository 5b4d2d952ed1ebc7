"""Registers x workgroup size of every kernel in csrc/*.hip, from the compiler's own metadata (no GPU needed).

The occupancy hipcc prints per kernel ("; Occupancy: 3") is WAVES per SIMD by registers.  A workgroup of more than 256 threads
puts several waves on every SIMD, so the number of workgroups a CU holds is floor(waves_by_registers / waves_per_SIMD_per_group):
512 threads at 142 registers = 3 waves allowed, 2 needed per group -> ONE group per CU, 8 waves instead of 12.  That is what held
`csr_collapse_kernel` at a third of its speed for two rounds; this script lists every such case, every kernel
that spills, and (with --all) the whole table.

    python tools/occupancy_scan.py [--all] [name-substring ...]
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "amcontrast3d_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-fPIC", "-I" + CSRC,
         "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]


def kernels_of(asm):
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", asm, re.S):
        blk = m.group(0)
        get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "0"])[1]  # noqa: E731
        yield {"name": get("name"), "vgpr": int(get("vgpr_count")), "lds": int(get("group_segment_fixed_size")),
               "wg": int(get("max_flat_workgroup_size")), "spill": int(get("vgpr_spill_count"))}


def demangle(name):
    out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    return re.sub(r"\(.*", "", out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
            procs.append((src, out, subprocess.Popen([hipcc] + FLAGS + ["-o", out, src], stderr=subprocess.DEVNULL)))
        for src, out, p in procs:
            if p.wait() != 0:
                sys.exit(f"hipcc failed on {src}")
            for k in kernels_of(open(out).read()):
                k["file"] = os.path.basename(src)
                rows.append(k)
    flagged = 0
    for k in sorted(rows, key=lambda r: (r["file"], r["name"])):
        if "rocprim" in k["name"] or "hipcub" in k["name"]:
            continue
        if a.names and not any(s in k["name"] for s in a.names):
            continue
        regs = max(8, (k["vgpr"] + 7) // 8 * 8)             # allocation granule: 8
        by_regs = min(8, 512 // regs)                       # waves per SIMD the register file allows
        waves = (k["wg"] + 63) // 64
        per_simd = (waves + 3) // 4                         # waves one workgroup puts on a SIMD
        groups = min(by_regs // per_simd, (160 * 1024) // k["lds"] if k["lds"] else 99, 32 // waves)
        wasted = per_simd > 1 and by_regs % per_simd != 0 and groups * per_simd < by_regs
        if not (a.all or a.names or wasted or k["spill"]):
            continue
        flagged += wasted or k["spill"] > 0
        note = ("  <- %d waves per SIMD allowed, %d used" % (by_regs, groups * per_simd)) if wasted else ""
        note += ("  <- spills %d registers" % k["spill"]) if k["spill"] else ""
        print(f"{k['file']:14s} {demangle(k['name'])[:64]:64s} wg={k['wg']:4d} vgpr={k['vgpr']:3d} lds={k['lds']:6d} "
              f"groups/CU={groups} waves/CU={groups * waves:2d}{note}")
    print(f"{len(rows)} kernels, {flagged} flagged")


if __name__ == "__main__":
    main()
