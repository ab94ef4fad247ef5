#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of tools/profile_round.sh.

usage: tools/pmc_summarize.py OUTDIR
Writes OUTDIR/<mode>_4k_pmc_summary.json (counter means per dispatch of the strip kernel and the
derived figures) and OUTDIR/<mode>_4k_kernel_stats.csv (per-kernel rows of the --stats pass).
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports
HALF the bytes of a streaming read whatever its width (byte, dword and dwordx4 loads all calibrated with
tools/pmc_calib.hip: profiles/r02/pmc_calibration.txt), WRITE_SIZE is exact within 2 %, so
reads = FETCH_SIZE * 1024 * 2, writes = WRITE_SIZE * 1024 for every kernel.
"""
import csv
import re, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
# kernels of one step: the strip kernel does the work; with seams (float32 mode) one small launch (srcnn_seams_merged_kernel;
# two launches, srcnn_seam_kernel + srcnn_cseam_kernel, for plans whose seam windows are not kept apart) finishes the rows /
# columns at item and strip boundaries -- their HBM bytes belong to the step's traffic
KERNEL = {"mfma": ["srcnn_strip_kernel", "srcnn_seams_merged_kernel", "srcnn_seam_kernel", "srcnn_cseam_kernel"], "split16": ["srcnn_split16_kernel"]}


def writes_flags(name):
    """The SRCNN_MODE_REFBYTES instantiations (template argument FIX = true) also run in a default bench -- its `refbytes`
    leg -- and store a flag plane besides: they are not the kernels of the float32 headline step."""
    m = re.search(r"<([^<>]*)>\(", name)
    args = [a.strip() for a in m.group(1).split(",")] if m else []
    # srcnn_strip_kernel<MODE, PRE, DIAG, FIX, HALO3>; the seam kernels' FIX is their last argument
    return bool(args) and (args[3] == "true" if len(args) >= 5 else args[-1] == "true")


traffic_rec = {}
for mode, knames in KERNEL.items():
    kname = knames[0]
    # Round 5: with seam deferral (the bench default) the step IS srcnn_strip_fold_kernel -- the strip kernel's row body on the
    # work items plus the previous step's seam blocks behind them; the plain strip kernel and the seam launch then run once per
    # process (first and last step).  The fold kernel's counters are the step's.
    files = glob.glob(os.path.join(out, f"pmc_{mode}_*", "**", "*counter_collection.csv"), recursive=True)
    folded = False
    if mode == "mfma":
        for f in files:
            with open(f, newline="") as fh:
                if any("srcnn_strip_fold_kernel" in row["Kernel_Name"] for row in csv.DictReader(fh)):
                    folded = True
                    break
    if folded:
        kname, knames = "srcnn_strip_fold_kernel", ["srcnn_strip_fold_kernel"]
    sums, cnt = defaultdict(float), defaultdict(int)
    extra = defaultdict(lambda: defaultdict(float))
    extra_cnt = defaultdict(lambda: defaultdict(int))
    grid = None
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if writes_flags(row["Kernel_Name"]):
                    continue
                if kname in row["Kernel_Name"]:
                    sums[row["Counter_Name"]] += float(row["Counter_Value"])
                    cnt[row["Counter_Name"]] += 1
                    grid = int(row["Grid_Size"]) // int(row["Workgroup_Size"])
                for k2 in knames[1:]:
                    if k2 in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                        extra[k2][row["Counter_Name"]] += float(row["Counter_Value"])
                        extra_cnt[k2][row["Counter_Name"]] += 1
    if not sums:
        continue
    mean = {k: sums[k] / cnt[k] for k in sums}
    d = {}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        d["hbm_read_bytes"] = mean["FETCH_SIZE"] * 1024 * 2
        d["hbm_write_bytes"] = mean["WRITE_SIZE"] * 1024
        d["hbm_bytes"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
        d["hbm_bytes_step"] = d["hbm_bytes"]
        for k2 in knames[1:]:
            if extra[k2]:
                # the seam kernels' dword loads are reported at half their size like every read (pmc_calibration.txt)
                b = sum(extra[k2][c] / extra_cnt[k2][c] * (2 if c == "FETCH_SIZE" else 1) for c in extra[k2]) * 1024
                d[f"hbm_bytes_{k2}"] = b
                d["hbm_bytes_step"] += b
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
        d["mfma_busy_frac_of_simd_cycles"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (mean["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_INSTS_MFMA" in mean:
        wave_rows = mean["SQ_INSTS_MFMA"] / (130 if mode == "mfma" else 42)
        d["wave_rows"] = wave_rows
        d["non_mfma_valu_per_wave_row"] = (mean.get("SQ_INSTS_VALU", 0) - mean["SQ_INSTS_MFMA"]) / wave_rows
        d["lds_insts_per_wave_row"] = mean.get("SQ_INSTS_LDS", 0) / wave_rows
        d["salu_insts_per_wave_row"] = mean.get("SQ_INSTS_SALU", 0) / wave_rows
    stats_rows = []
    avg_ns = None
    for f in glob.glob(os.path.join(out, f"trace_{mode}", "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows = list(csv.reader(fh))
        stats_rows = rows
        for r in rows[1:]:
            if kname in r[0] and not writes_flags(r[0]):
                avg_ns = float(r[3])
    if stats_rows:
        with open(os.path.join(out, f"{mode}_4k_kernel_stats.csv"), "w", newline="") as fh:
            csv.writer(fh).writerows(stats_rows)
    traffic_rec[("fused" if mode == "mfma" else mode) + "_3840x2160x1"] = d.get("hbm_bytes_step")
    if "mfma_busy_frac_of_simd_cycles" in d:
        traffic_rec[("fused" if mode == "mfma" else mode) + "_3840x2160x1_mfma_busy_frac"] = round(d["mfma_busy_frac_of_simd_cycles"], 4)
    summary = {"kernel": f"{kname} 3840x2160x1, {grid} workgroups" + (" (512 work items + the previous step's seam blocks)" if folded else ""),
               "rocprof_kernel_trace_avg_ns": avg_ns,
               "counters_mean_per_dispatch": mean, "derived": d}
    with open(os.path.join(out, f"{mode}_4k_pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(mode, json.dumps(d), "avg_ns", avg_ns)

import datetime, subprocess
try:
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = ""
traffic_rec["_taken_at"] = {"date": datetime.date.today().isoformat(), "commit": head or "unknown (the GPU box holds no .git)"}
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from srcnn_cpp_amd.build import kernel_sources_fingerprint
traffic_rec["_kernel_sources"] = kernel_sources_fingerprint()      # bench.py quotes these figures only for this build

# the two reference functions alone (unfused path, one 3840x2160 frame): bytes per launch against the algorithmic 1 + 128 B/pixel
unf = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_unfused_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            for tag, name in (("conv99x11_L12", "srcnn_strip_kernel<1"), ("conv55_L3", "srcnn_strip_kernel<2"), ("conv55_cseam", "srcnn_cseam_kernel")):
                if name in row["Kernel_Name"]:
                    unf[tag][row["Counter_Name"]].append(float(row["Counter_Value"]))
if unf:
    px = 3840 * 2160
    rec = {}
    for tag, ctrs in unf.items():
        m = {k: sum(v) / len(v) for k, v in ctrs.items()}
        d = {"counters_mean_per_dispatch": m}
        if "FETCH_SIZE" in m: d["hbm_read_bytes"] = m["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in m: d["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
        if "SQ_INSTS_MFMA" in m and m["SQ_INSTS_MFMA"] > 0:
            wr = m["SQ_INSTS_MFMA"] / (114 if "L12" in tag else 16)
            d["wave_rows"] = wr
            d["non_mfma_valu_per_wave_row"] = (m.get("SQ_INSTS_VALU", 0) - m["SQ_INSTS_MFMA"]) / wr
            d["lds_insts_per_wave_row"] = m.get("SQ_INSTS_LDS", 0) / wr
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            d["mfma_busy_frac_of_simd_cycles"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
        rec[tag] = d
    if "conv99x11_L12" in rec and "hbm_write_bytes" in rec["conv99x11_L12"]:
        r = rec["conv99x11_L12"]
        r["algorithmic_bytes"] = 129 * px
        r["traffic_over_algorithmic"] = (r.get("hbm_read_bytes", 0) + r["hbm_write_bytes"]) / (129 * px)
    if "conv55_L3" in rec and "hbm_read_bytes" in rec["conv55_L3"]:
        r = rec["conv55_L3"]
        extra = rec.get("conv55_cseam", {})
        tot = r["hbm_read_bytes"] + r.get("hbm_write_bytes", 0) + extra.get("hbm_read_bytes", 0) + extra.get("hbm_write_bytes", 0)
        r["algorithmic_bytes"] = 129 * px
        r["traffic_over_algorithmic"] = tot / (129 * px)
    with open(os.path.join(out, "unfused_4k_pmc_summary.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print("unfused", json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_mean_per_dispatch"} for k, v in rec.items()}))

# ---- the kernels beside the headline strip kernel: the REFBYTES fix-up, Convolution55 alone, the pipeline
# byte kernels -- counters per dispatch and what follows from them, one file per run:
#   refbytes_4k_pmc_summary.json   (pmc_refbytes_*: bench.py --mode refbytes)     fix_apply_kernel, fix_collect_kernel, fix_rerun_kernel,
#                                                                                 the flag-writing strip kernel and its seam launch
#   pipeline_pmc_summary.json      (pmc_pipeline_*: bench.py --path pipeline)     bgr_to_y_resized_kernel, resize_merge_kernel
#   l3_4k_pmc_summary.json         (pmc_unfused_*)                                Convolution55 alone: TB/s FROM THE COUNTERS
def kernel_table(run, patterns, trace_tag):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, f"pmc_{run}_*", "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                for tag, pat in patterns:
                    if pat in row["Kernel_Name"]:
                        acc[tag][row["Counter_Name"]].append(float(row["Counter_Value"]))
                        break
    avg_ns = {}
    for f in glob.glob(os.path.join(out, f"trace_{trace_tag}", "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in list(csv.reader(fh))[1:]:
                for tag, pat in patterns:
                    if pat in r[0]:
                        avg_ns.setdefault(tag, float(r[3]))
                        break
    rec = {}
    for tag, ctrs in acc.items():
        m = {k: sum(v) / len(v) for k, v in ctrs.items()}
        d = {"counters_mean_per_dispatch": m, "rocprof_kernel_trace_avg_ns": avg_ns.get(tag)}
        if "FETCH_SIZE" in m: d["hbm_read_bytes"] = m["FETCH_SIZE"] * 2048          # x2: pmc_calibration.txt
        if "WRITE_SIZE" in m: d["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
        if avg_ns.get(tag) and ("hbm_read_bytes" in d or "hbm_write_bytes" in d):
            d["hbm_TB_per_s_from_counters"] = (d.get("hbm_read_bytes", 0) + d.get("hbm_write_bytes", 0)) / avg_ns[tag] / 1e3
        if "GRBM_GUI_ACTIVE" in m and m.get("SQ_INSTS_VALU", 0) > 0:
            simd_cycles = m["GRBM_GUI_ACTIVE"] / 8 * 1024                            # 1,024 SIMDs x the kernel's cycles
            d["simd_cycles_per_valu_instruction"] = simd_cycles / m["SQ_INSTS_VALU"]  # 4 = one wave's issue rate; packed f32: 4.6-4.8 in the issue probe
            if avg_ns.get(tag):
                d["effective_clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8 / avg_ns[tag]
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
            for k2, name in (("SQ_ACTIVE_INST_VALU", "valu_issue_frac_of_wave_cycles"), ("SQ_ACTIVE_INST_LDS", "lds_issue_frac_of_wave_cycles"),
                             ("SQ_ACTIVE_INST_ANY", "any_issue_frac_of_wave_cycles"), ("SQ_WAIT_INST_ANY", "issue_stall_frac_of_wave_cycles"),
                             ("SQ_WAIT_ANY", "parked_frac_of_wave_cycles")):
                if k2 in m: d[name] = m[k2] / m["SQ_WAVE_CYCLES"]
        if m.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in m:
            d["lds_bank_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
        rec[tag] = d
    return rec


rb = kernel_table("refbytes", [("fix_apply", "fix_apply_kernel"), ("fix_collect", "fix_collect_kernel"), ("fix_rerun", "fix_rerun_kernel"),
                               ("strip_with_flags", "srcnn_strip_kernel<0"), ("seams_with_flags", "srcnn_seams_merged_kernel")], "refbytes")
if rb:
    if "fix_apply" in rb and rb["fix_apply"]["counters_mean_per_dispatch"].get("SQ_INSTS_VALU"):
        rb["fix_apply"]["note"] = ("the fix-up is vector-ALU work in the reference's arithmetic (no FMA): simd_cycles_per_valu_instruction against "
                                   "the 4-cycle issue rate of one wave's stream says how close the kernel runs to the instruction-issue bound")
    with open(os.path.join(out, "refbytes_4k_pmc_summary.json"), "w") as fh:
        json.dump(rb, fh, indent=1)
    print("refbytes", json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_mean_per_dispatch"} for k, v in rb.items()}))
pl = kernel_table("pipeline", [("bgr_to_y_resized", "bgr_to_y_resized_kernel"), ("resize_merge", "resize_merge_kernel")], "pipeline")
if pl:
    with open(os.path.join(out, "pipeline_pmc_summary.json"), "w") as fh:
        json.dump(pl, fh, indent=1)
    print("pipeline", json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_mean_per_dispatch"} for k, v in pl.items()}))
l3 = kernel_table("unfused", [("conv55_L3", "srcnn_strip_kernel<2"), ("conv99x11_L12", "srcnn_strip_kernel<1")], "unfused1")
if l3:
    if "conv55_L3" in l3 and "hbm_TB_per_s_from_counters" in l3["conv55_L3"]:
        l3["conv55_L3"]["algorithmic_TB_per_s"] = 129 * 3840 * 2160 / l3["conv55_L3"]["rocprof_kernel_trace_avg_ns"] / 1e3
    with open(os.path.join(out, "l3_4k_pmc_summary.json"), "w") as fh:
        json.dump(l3, fh, indent=1)
    print("l3", json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_mean_per_dispatch"} for k, v in l3.items()}))

traffic_rec["_note"] = ("per step (one 3840x2160 plane): HBM bytes of the strip kernel plus, in the float32 mode, the seam blocks (with seam deferral they run inside srcnn_strip_fold_kernel, whose counters are the step's); "
                        "FETCH_SIZE x2 (every read width is reported at half its size: profiles/r02/pmc_calibration.txt) + WRITE_SIZE, KB -> bytes; "
                        "separate --pmc passes (tools/profile_round.sh); *_mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) of the strip kernel")
with open(os.path.join(out, "pmc_traffic.json"), "w") as fh:
    json.dump(traffic_rec, fh, indent=1)

# per-kernel averages of the other runs (rocprofv3 --stats): OUTDIR/other_kernels_stats.csv
rows = [["run", "kernel", "calls", "avg_us"]]
for tag in ("unfused", "pipeline", "exact", "pipeline_split16", "refbytes", "refbytes16"):
    for f in glob.glob(os.path.join(out, f"trace_{tag}", "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in list(csv.reader(fh))[1:]:
                if "srcnn::" in r[0]:
                    name = r[0].replace("void ", "").split("(srcnn::StripParams")[0].split("(srcnn::FixParams")[0].split("(unsigned char")[0].split("(float")[0]
                    rows.append([tag, name, r[1], f"{float(r[3]) / 1000:.1f}"])
if len(rows) > 1:
    with open(os.path.join(out, "other_kernels_stats.csv"), "w", newline="") as fh:
        csv.writer(fh).writerows(rows)
