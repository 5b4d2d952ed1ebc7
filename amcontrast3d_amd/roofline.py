"""Roofline accounting of the train step (what bench.py prints).

Three sources, kept apart:

  live        HIP-event time of every C-ABI operator, measured by bench.py on the stream it is launched on
              (amcontrast3d_amd/timing.py), with the operator's ALGORITHMIC bytes -- every input read once, every output
              written once -- and its dense FLOPs.  `achieved` = algorithmic bytes (FLOPs) / that time.
  profiles/   committed summaries of rocprofv3 runs of the same command on the same box:
                round3_kernel_stats.csv   per-KERNEL average durations of the steady-state step (--kernel-trace; tools/steady2.py)
                hbm_traffic.json          HBM bytes per kernel / operator / step from FETCH_SIZE and WRITE_SIZE, collected in
                                          separate --pmc passes and corrected as MI355X_MICROARCH.md prescribes (FETCH x 2)
              They let a reader redo `frac` by hand: rocprof's average duration of the operator's kernels must agree with the
              live figure, and `traffic` / algorithmic says how much of the memory system's work was not compulsory.
  peaks       MI355X_MICROARCH.md: HBM3E 8 TB/s, v_mfma_f32_32x32x2_f32 dense 157.3 TFLOP/s.

The dominant operator is chosen by live time among the operators of the stream that bounds the overlapped step (the
feature half); the FPS chain is a latency chain and is reported as one.
"""
import csv
import json
import os

HBM_PEAK_GBS = 8000.0
F32_MFMA_PEAK_TFLOPS = 157.3
# SURVEY.md section 8(d) / BASELINE.md section 2: algorithmic work per POINT of a train step (forward x 3), derived there
# for B=8 x N=24000 (S: 1.20 GB, 137.7 GF; L: 1.99 GB, 578.8 GF; XL: 4.26 GB, 3241 GF per 192000 points)
ALGORITHMIC_PER_POINT = {"S": (1.197e9 / 192000, 137.7e9 / 192000), "L": (1.992e9 / 192000, 578.8e9 / 192000),
                         "XL": (4.26e9 / 192000, 3241e9 / 192000)}
GEOMETRY_OPS = ("furthest_point_sampling", "ball_query", "three_nn", "knnquery", "posmask", "ambiguity", "vote_labels",
                "select_anchors", "group_csr", "group_moments", "group_points", "contrast_csr")
# C-ABI operator -> substrings of the kernels it launches (names as rocprofv3 prints them)
OPERATOR_KERNELS = {
    "contrast_backward": ("contrast_backward_kernel", "contrast_backward_mutual_kernel", "contrast_record_kernel",
                          "contrast_record_stats_kernel"),
    "contrast_backward_csr": ("contrast_backward_rows_kernel", "contrast_coef_kernel"),
    "contrast_forward": ("contrast_forward_unit_kernel", "contrast_forward_kernel", "row_norm_kernel", "row_unit_kernel",
                         "row_unit_cm_kernel", "masked_mean_kernel"),
    "pointwise_conv_forward": ("pw_gemm_kernel", "gm_gemm_kernel", "gb_gemm_kernel", "gm_split_reduce"),
    "pointwise_conv_backward": ("pw_gemm_kernel", "pw_wgrad_kernel", "gw_wgrad_kernel", "gm_gemm_kernel", "gcc_reduce", "gb_gemm_kernel"),
    "sa_tail_forward": ("sat_kernel", "sat_finalize"),
    "sa_tail_backward": ("sat_kernel", "sat_bwd", "sat_pool_grad", "sat_alg_reduce", "sat_alg_dw"),
    "grouped_conv_bn_forward": ("lagg_stats", "lagg_expand", "lagg_finalize"),
    "grouped_conv_bn_backward": ("csr_collapse", "lagg_collapse", "lagg_bwd"),
    "local_aggregation_forward": ("lagg_stats", "lagg_pool", "lagg_finalize"),
    "local_aggregation_backward": ("lagg_bwd_scatter", "lagg_bwd_apply"),
    "bn_act_forward": ("bn_stats", "bn_act_kernel", "bn_fwd_channel"), "bn_act_backward": ("bn_bwd",),
    "bn_max_forward": ("bn_stats", "bn_max"), "bn_max_backward": ("bn_max_bwd", "bn_bwd"),
    "three_interpolate": ("three_interpolate",), "three_interpolate_grad": ("three_interpolate_grad", "scatter_pm"),
    "sa_residual_forward": ("sa_res_fwd",), "sa_residual_backward": ("sa_res_mask", "sa_res_bwd"),
    "cross_entropy_forward": ("ce_forward",), "cross_entropy_backward": ("ce_backward",),
    "furthest_point_sampling": ("fps_kernel",), "knnquery": ("kg_", "knn_"),
}


def _profiles_dir():
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def load_profiles(variant="S"):
    """-> (hbm_traffic dict, [rocprof kernel rows], csv name) from profiles/ (empty when absent)"""
    d = _profiles_dir()
    traffic, rows, name = {}, [], None
    try:
        with open(os.path.join(d, "hbm_traffic.json")) as fh:
            traffic = json.load(fh)
    except OSError:
        pass
    name = "round3_kernel_stats.csv" if variant == "S" else f"round3_{variant}_kernel_stats.csv"
    try:
        with open(os.path.join(d, name)) as fh:
            rows = list(csv.DictReader(fh))
    except OSError:
        name = None
    return traffic, rows, name


def family_traffic(op, kernels, steps):
    """PMC bytes per step of every kernel `op` launches (profiles/round3_hbm_pmc.csv) against the algorithmic bytes per step of
    every operator that shares one of those kernels (a pointwise conv's forward and backward run the same GEMM kernels, so
    counters cannot be split between them): -> {'operators', 'traffic_GB_per_step', 'algorithmic_GB_per_step', 'ratio'} or None"""
    pats = OPERATOR_KERNELS.get(op)
    try:
        with open(os.path.join(_profiles_dir(), "round3_hbm_pmc.csv")) as fh:
            pmc = list(csv.DictReader(fh))
    except OSError:
        return None
    if not pats or not pmc:
        return None
    family = sorted(o for o, ps in OPERATOR_KERNELS.items() if set(ps) & set(pats) and o in kernels)
    allp = sorted({p for o in family for p in OPERATOR_KERNELS[o]})
    mb = sum(float(r["HBM_MB_per_step_corrected"]) for r in pmc if any(p in r["kernel"] for p in allp))
    alg = sum(kernels[o]["bytes"] for o in family) / steps
    if not mb or not alg:
        return None
    return {"operators": family, "traffic_GB_per_step": round(mb / 1e3, 3), "algorithmic_GB_per_step": round(alg / 1e9, 3),
            "ratio": round(mb * 1e6 / alg, 2), "file": "profiles/round3_hbm_pmc.csv"}


def rocprof_of(op, rows, csv_name):
    """the committed rocprofv3 rows of the kernels operator `op` launches: us per step and per-kernel averages"""
    pats = OPERATOR_KERNELS.get(op)
    if not pats or not rows:
        return None
    hit = [r for r in rows if any(p in r["Name"] for p in pats)]
    if not hit:
        return None
    return {"file": f"profiles/{csv_name}", "us_per_step": round(sum(float(r["UsPerStep"]) for r in hit), 1),
            "kernels": [{"name": r["Name"].split("(")[0][-60:], "calls_per_step": float(r["CallsPerStep"]),
                         "avg_us": float(r["AverageUs"])} for r in hit[:6]]}


def operator_roofline(name, v, steps, traffic, rows, csv_name):
    """v: one entry of timing.collect() scaled to `steps` steps.  Algorithmic bytes (or FLOPs) of all launches / their summed
    HIP-event time, against the roof the operator's arithmetic intensity puts it under (machine balance 157.3 TF / 8 TB/s
    ~ 20 flop/byte)."""
    nbytes, fl, tms, n = v["bytes"], v["flops"], v["total_ms"], max(v["launches"], 1)
    tr = traffic.get(name, {}).get("bytes_per_launch")
    if tr is None and traffic.get(name, {}).get("bytes_per_step"):  # per step in the PMC table -> per operator launch
        tr = traffic[name]["bytes_per_step"] / max(n / steps, 1e-9)
    out = {"kernel": name, "avg_launch_ms": round(tms / n, 4), "launches_per_step": round(n / steps, 2),
           "ms_per_step": round(tms / steps, 4), "algorithmic_bytes_per_launch": int(nbytes / n),
           "moved_bytes_per_launch": int(v.get("moved", nbytes) / n),
           "moved_note": "what the implementation requests incl. gathered neighbour rows / atomic rows / re-read passes; mostly L2 + Infinity Cache",
           "traffic": tr, "traffic_over_algorithmic": round(tr / (nbytes / n), 2) if tr and nbytes else None,
           "rocprof": rocprof_of(name, rows, csv_name)}
    if fl > 0 and fl / max(nbytes, 1) > F32_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
        ach = fl / (tms * 1e-3) / 1e12
        out.update(bound="mfma", achieved=round(ach, 2), peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                   frac=round(ach / F32_MFMA_PEAK_TFLOPS, 4), algorithmic_flops_per_launch=fl / n)
    else:
        ach = nbytes / (tms * 1e-3) / 1e9
        out.update(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4))
    return out


def report(kernels, steps, ms_per_step, points_per_step, variant, mm, overlapped, fps=None):
    """kernels: timing.collect() scaled to `steps` steps.  -> the roofline objects of the bench line."""
    traffic, rows, csv_name = load_profiles(variant)
    crit = {k: v for k, v in kernels.items() if not (overlapped and k in GEOMETRY_OPS)}
    out = {"roofline": None, "roofline_hbm": None, "roofline_mfma": None, "roofline_step": None, "latency_chain": None}
    if crit:
        name, v = max(crit.items(), key=lambda kv: kv[1]["total_ms"])
        out["roofline"] = operator_roofline(name, v, steps, traffic, rows, csv_name)
        if out["roofline"]["traffic"] is None:  # its kernels are shared with other operators: the family's counters instead
            out["roofline"]["traffic_family"] = family_traffic(name, kernels, steps)
        out["roofline"]["note"] = ("largest live HIP-event time among the operators of the feature half (the stream that bounds the "
                                   "overlapped step); achieved = SURVEY 8(d) algorithmic bytes or FLOPs / that time")
    hbm = {k: v for k, v in crit.items() if v["flops"] == 0 or v["flops"] / max(v["bytes"], 1) < 19.7}
    if hbm:
        name, v = max(hbm.items(), key=lambda kv: kv[1]["total_ms"])
        out["roofline_hbm"] = operator_roofline(name, v, steps, traffic, rows, csv_name)
    mf = [k for k in kernels if k.startswith(("grouped_conv", "pointwise_conv", "sa_tail", "local_aggregation"))]
    if mf:
        fl, tm = sum(kernels[k]["flops"] for k in mf), sum(kernels[k]["total_ms"] for k in mf)
        ach = fl / (tm * 1e-3) / 1e12
        out["roofline_mfma"] = {"bound": "mfma", "kernel": "+".join(sorted(mf)), "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                                "ms_per_step": round(tm / steps, 3),
                                "note": "launched FLOPs (recomputation included) of every MFMA operator / their summed time"}
    if variant in ALGORITHMIC_PER_POINT and not mm:
        bpp, fpp = ALGORITHMIC_PER_POINT[variant]
        ab, af = bpp * points_per_step, fpp * points_per_step
        st = traffic.get("_step", {})
        meas = st.get("bytes_per_step") if st.get("workload") == f"{variant}:{points_per_step}" else None
        out["roofline_step"] = {
            "algorithmic_GB": round(ab / 1e9, 3), "algorithmic_GFLOP": round(af / 1e9, 1), "ms": round(ms_per_step, 3),
            "hbm_GBps": round(ab / (ms_per_step * 1e-3) / 1e9, 1), "hbm_frac": round(ab / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "mfma_TFLOPs": round(af / (ms_per_step * 1e-3) / 1e12, 2),
            "mfma_frac": round(af / (ms_per_step * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
            "measured_hbm_GB": round(meas / 1e9, 2) if meas else None,
            "traffic_over_algorithmic": round(meas / ab, 2) if meas else None,
            "algorithmic_over_measured": round(ab / meas, 3) if meas else None,
            "operators_algorithmic_GB": round(sum(v["bytes"] for v in kernels.values()) / steps / 1e9, 3),
            "note": "algorithmic = SURVEY 8(d) (ideal fusion, forward x 3); measured = sum over every kernel of one step of FETCH_SIZE x 2 "
                    "+ WRITE_SIZE (profiles/hbm_traffic.json, rocprofv3 --pmc passes); operators_algorithmic = sum of the per-operator "
                    "figures (each operator's own inputs + outputs: what fusion of neighbouring operators could still remove)"}
    if fps and "furthest_point_sampling" in kernels:
        its = sum(fps["points"] // 4 ** k for k in range(1, fps["levels"]))
        out["latency_chain"] = {
            "kernel": "furthest_point_sampling", "iterations_per_cloud_all_levels": its, "joint_launch_ms": fps["joint_ms"],
            "us_per_iteration": round(fps["joint_ms"] * 1e3 / its, 3) if fps["joint_ms"] else None,
            "clouds_per_launch": fps["clouds"], "launch_every_steps": fps["lanes"],
            "hbm_bytes_per_launch": traffic.get("furthest_point_sampling", {}).get("bytes_per_launch"),
            "note": (f"serial arg-max chain, off the critical path: all sampling levels of {fps['lanes']} future batches run as one "
                     f"launch every {fps['lanes']} steps on the sampling queue, one workgroup per cloud")}
    return out
