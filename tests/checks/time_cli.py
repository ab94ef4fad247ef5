#!/usr/bin/env python3
"""The reference's ONLY use case, timed as a user meets it: one image, one process (src/srcnn.cpp:707-731 runs one picture
per invocation; its "Performace" line covers :505-659 only).  Wall clock of `srcnn_amd --timing ...` from exec to exit for
  * the reference's own example, butterfly.png x1.5 (BASELINE configs[0]), and
  * a 1920x1080 picture x2.0 (the input of BASELINE configs[1]),
split into the phases the tool prints (decode, HIP runtime start, srcnn_create, weights, warm-up launch, the timed region,
encode, destroy) + what lies outside main() (exec, dynamic linking of libamdhip64 / libsrcnn_amd, exit), beside
  * a bare HIP process (tools/hip_init_probe.hip: runtime start, first allocation, code-object load, first launch) and
  * oracle.process_bgr (the reference's arithmetic, -O3, OpenMP on every core) on the same pictures.
Run on the GPU box:  python tests/checks/time_cli.py [runs] > profiles/rNN/cli_process_cold.txt"""
import re
import statistics
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from PIL import Image

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 7
tmp = Path(tempfile.mkdtemp(prefix="time_cli_"))
cli, bare = tmp / "srcnn_amd", tmp / "hip_init_probe"
subprocess.run(["g++", "-std=c++17", "-O2", f"-I{ROOT / 'include'}", f"-I{ROOT / 'tools'}", str(ROOT / "tools" / "srcnn_cli.cpp"),
                f"-L{ROOT / 'srcnn_cpp_amd'}", "-lsrcnn_amd", "-lz", "-ldl", f"-Wl,-rpath,{ROOT / 'srcnn_cpp_amd'}",
                "-Wl,-rpath,/opt/rocm/lib", "-o", str(cli)], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", str(ROOT / "tools" / "hip_init_probe.hip"), "-o", str(bare)],
               check=True, stderr=subprocess.DEVNULL)

fx = np.load(ROOT / "tests" / "golden" / "butterfly_bgr.npz")
Image.fromarray(np.ascontiguousarray(fx["src_bgr"][:, :, ::-1])).save(tmp / "butterfly.png")
from srcnn_cpp_amd.synth import synth_luma
y = synth_luma(1920, 1080).astype(np.int32)
hd = np.clip(np.stack([y + 8, y, y - 12], axis=-1), 0, 255).astype(np.uint8)              # R, G, B
Image.fromarray(hd).save(tmp / "hd.png")

PHASE = re.compile(r"^- timing : (.+?)\s+([0-9.]+) ms")


def run(cmd):
    t = time.perf_counter()
    r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True)
    wall = (time.perf_counter() - t) * 1e3
    if r.returncode != 0:
        raise SystemExit(f"{cmd}: exit {r.returncode}\n{r.stdout[-800:]}\n{r.stderr[-800:]}")
    phases = [(m.group(1).strip(), float(m.group(2))) for m in map(PHASE.match, r.stdout.splitlines()) if m]
    return wall, phases


def table(title, cmd, runs=RUNS):
    walls, per = [], {}
    order = []
    for i in range(runs + 1):
        wall, phases = run(cmd)
        if i == 0:
            first = (wall, phases)                  # the very first process on the box: page cache and driver state cold
            continue
        walls.append(wall)
        for k, v in phases:
            per.setdefault(k, []).append(v)
            if k not in order:
                order.append(k)
    print(f"\n## {title}\n   {' '.join(Path(str(c)).name if str(c).startswith('/') else str(c) for c in cmd)}")
    print(f"   median of {runs} processes (the first process of this command on the box, page cache cold: {first[0]:.0f} ms wall)")
    inside = 0.0
    for k in order:
        med = statistics.median(per[k])
        inside += med
        print(f"   {k:<46} {med:9.2f} ms   (min {min(per[k]):8.2f}, max {max(per[k]):8.2f})")
    w = statistics.median(walls)
    print(f"   {'outside main(): exec, dynamic linking, exit':<46} {w - inside:9.2f} ms")
    print(f"   {'WALL, exec -> exit':<46} {w:9.2f} ms   (min {min(walls):8.2f}, max {max(walls):8.2f})")
    return w, {k: statistics.median(v) for k, v in per.items()}


print("# Process-cold latency of the command-line tool on one MI355X box (tests/checks/time_cli.py); every line is a fresh process.")
res = {}
res["bare"] = table("bare HIP process: what any program pays before its first kernel", [bare])
for name, img, scale in (("butterfly", tmp / "butterfly.png", 1.5), ("hd", tmp / "hd.png", 2.0)):
    for mode in ("mfma", "refbytes"):
        extra = ["--refbytes"] if mode == "refbytes" else []
        res[name, mode] = table(f"{img.name} x{scale} ({Image.open(img).size[0]}x{Image.open(img).size[1]} -> x{scale}), "
                                f"{'SRCNN_MODE_REFBYTES' if mode == 'refbytes' else 'SRCNN_MODE_MFMA'}",
                                [cli, f"--scale={scale}", "--timing", *extra, img, tmp / f"{name}_{mode}_out.png"])

# the two outputs against the oracle (REFBYTES: every byte), and the oracle's own time on the host cores
import oracle
import srcnn_cpp_amd as S
blob = S.load_weights()
print("\n## the reference's arithmetic on the host (oracle.process_bgr: -O3 -ffp-contract=off, OpenMP, every core), in-process, best of 3")
for name, src, scale in (("butterfly", fx["src_bgr"], 1.5), ("hd", np.ascontiguousarray(hd[:, :, ::-1]), 2.0)):
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        ref = oracle.process_bgr(src, scale, blob)
        best = min(best, (time.perf_counter() - t) * 1e3)
    got = np.asarray(Image.open(tmp / f"{name}_refbytes_out.png"))[:, :, ::-1]
    got_m = np.asarray(Image.open(tmp / f"{name}_mfma_out.png"))[:, :, ::-1]
    print(f"   {name:<10} {src.shape[1]}x{src.shape[0]} x{scale}: {best:9.1f} ms   | the tool's --refbytes file equals it: {bool(np.array_equal(got, ref))}; "
          f"MFMA-mode file: {int((got_m != ref).sum())} of {ref.size} bytes differ")
b = res["bare"][1]
bare_floor = sum(v for k, v in b.items() if "second" not in k)
for key in (("butterfly", "mfma"), ("hd", "mfma")):
    p = res[key][1]
    create = p.get("srcnn_create (stream, interlock probe)", float("nan"))
    print(f"\n# {key[0]}: srcnn_create {create:.1f} ms + HIP runtime start {p.get('HIP runtime start (hipInit, device count)', float('nan')):.1f} ms; "
          f"bare process to its first finished launch: {bare_floor:.1f} ms")
