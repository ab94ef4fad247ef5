#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of tools/profile_round.sh.

usage: tools/pmc_summarize.py OUTDIR
Writes OUTDIR/<mode>_4k_pmc_summary.json (counter means per dispatch of the strip kernel and the
derived figures) and OUTDIR/<mode>_4k_kernel_stats.csv (per-kernel rows of the --stats pass).
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 byte-wide loads
are reported at half their size (calibrated with tools/pmc_calib.hip: profiles/r01/pmc_calibration.txt),
so reads = FETCH_SIZE * 1024 * 2, writes = WRITE_SIZE * 1024.
"""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
# kernels of one step: the strip kernel does the work; with seams (float32 mode) two small kernels finish the
# rows / columns at item and strip boundaries -- their HBM bytes belong to the step's traffic
KERNEL = {"mfma": ["srcnn_strip_kernel", "srcnn_seam_kernel", "srcnn_cseam_kernel"], "split16": ["srcnn_split16_kernel"]}
traffic_rec = {}
for mode, knames in KERNEL.items():
    kname = knames[0]
    sums, cnt = defaultdict(float), defaultdict(int)
    extra = defaultdict(lambda: defaultdict(float))
    extra_cnt = defaultdict(lambda: defaultdict(int))
    grid = None
    for f in glob.glob(os.path.join(out, f"pmc_{mode}_*", "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
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
                # dword loads / byte stores of the seam kernels: FETCH_SIZE and WRITE_SIZE both exact (pmc_calibration.txt)
                b = sum(extra[k2][c] / extra_cnt[k2][c] for c in extra[k2]) * 1024
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
            if kname in r[0]:
                avg_ns = float(r[3])
    if stats_rows:
        with open(os.path.join(out, f"{mode}_4k_kernel_stats.csv"), "w", newline="") as fh:
            csv.writer(fh).writerows(stats_rows)
    traffic_rec[("fused" if mode == "mfma" else mode) + "_3840x2160x1"] = d.get("hbm_bytes_step")
    if "mfma_busy_frac_of_simd_cycles" in d:
        traffic_rec[("fused" if mode == "mfma" else mode) + "_3840x2160x1_mfma_busy_frac"] = round(d["mfma_busy_frac_of_simd_cycles"], 4)
    summary = {"kernel": f"{kname} 3840x2160x1, {grid} workgroups", "rocprof_kernel_trace_avg_ns": avg_ns,
               "counters_mean_per_dispatch": mean, "derived": d}
    with open(os.path.join(out, f"{mode}_4k_pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(mode, json.dumps(d), "avg_ns", avg_ns)

traffic_rec["_note"] = ("per step (one 3840x2160 plane): HBM bytes of the strip kernel plus, in the float32 mode, the two seam kernels; "
                        "FETCH_SIZE x2 for the strip kernels' byte loads (calibrated: profiles/r01/pmc_calibration.txt) + WRITE_SIZE, KB -> bytes; "
                        "separate --pmc passes (tools/profile_round.sh); *_mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) of the strip kernel")
with open(os.path.join(out, "pmc_traffic.json"), "w") as fh:
    json.dump(traffic_rec, fh, indent=1)

# per-kernel averages of the other runs (rocprofv3 --stats): OUTDIR/other_kernels_stats.csv
rows = [["run", "kernel", "calls", "avg_us"]]
for tag in ("unfused", "pipeline", "exact", "pipeline_split16"):
    for f in glob.glob(os.path.join(out, f"trace_{tag}", "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in list(csv.reader(fh))[1:]:
                if "srcnn::" in r[0]:
                    name = r[0].replace("void ", "").split("(srcnn::StripParams")[0].split("(unsigned char")[0].split("(float")[0]
                    rows.append([tag, name, r[1], f"{float(r[3]) / 1000:.1f}"])
if len(rows) > 1:
    with open(os.path.join(out, "other_kernels_stats.csv"), "w", newline="") as fh:
        csv.writer(fh).writerows(rows)
