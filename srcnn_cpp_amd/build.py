"""Build the HIP extension in-tree: srcnn_cpp_amd/libsrcnn_amd.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
# SRCNN_BUILD_VARIANT=name [+ SRCNN_BUILD_DEFINES="-DX -DY"]: an alternative library libsrcnn_amd_<name>.so with its own
# object directory, for same-box A/B timing (load it with SRCNN_LIB=...); the default build is the product
VARIANT = os.environ.get("SRCNN_BUILD_VARIANT", "")
LIB = PKG / (f"libsrcnn_amd_{VARIANT}.so" if VARIANT else "libsrcnn_amd.so")
OBJ = PKG.parent / ("build" + (f"_{VARIANT}" if VARIANT else ""))

ARCH = "gfx950"
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
COMMON += os.environ.get("SRCNN_BUILD_DEFINES", "").split()
# (source, extra flags).  srcnn_exact.hip reproduces the reference's
# multiply-then-add arithmetic: contraction to FMA must stay off there.
UNITS = [
    # the SLP vectoriser packs the few scalar adds of the layer-3 sums into v_pk_add_f32 behind v_mov shuffles: more
    # vector instructions, not fewer, and every one of them costs MFMA issue time
    ("srcnn_mfma.hip", ["-fno-slp-vectorize"]),
    # MFMA results are consumed by vector instructions: keep them in architectural VGPRs (the 1-wave/SIMD
    # variant pins its weight fragments to AGPRs instead)
    ("srcnn_split16.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"]),
    # ... and no SLP packing either: v_pk_mul_f32 / v_pk_add_f32 run no faster than two plain instructions here (measured:
    # SRCNN_MODE_EXACT 2.92 -> 2.81 ms without them; a hand-packed layer-1 product made it 3.09)
    ("srcnn_exact.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),
    # the resize's vertical pass is OpenCV's float32 multiply-then-add (every product and sum rounded on its own):
    # __fmul_rn / __fadd_rn are plain * and + in HIP's headers, so contraction must be off here too
    ("srcnn_pipeline.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),
    ("srcnn_api.cpp", ["-x", "hip"]),
]


def kernel_sources_fingerprint() -> str:
    """sha256 over the sources that decide what the conv-path kernels do and how they are launched.  The rocprofv3 --pmc
    figures bench.py quotes (profiles/pmc_traffic.json) carry the fingerprint they were taken with; a line never quotes
    counters of a different kernel build (the GPU box holds no .git, so a commit id cannot serve)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in ("srcnn_mfma.hip", "srcnn_split16.hip", "srcnn_kernels.h", "srcnn_api.cpp"):
        text = (CSRC / name).read_text()
        # the CODE, not its commentary: comments removed (no string literal of these files holds "//" or "/*"), white space collapsed
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(" ".join(text.split()).encode())
    return "c2-" + h.hexdigest()[:16]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _stale(target: Path, deps) -> bool:
    return (not target.exists()) or any(d.stat().st_mtime > target.stat().st_mtime for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    OBJ.mkdir(exist_ok=True)
    headers = [CSRC / "srcnn_kernels.h", PKG.parent / "include" / "srcnn_amd.h", Path(__file__)]
    objs = []
    for src, extra in UNITS:
        s = CSRC / src
        o = OBJ / (Path(src).stem + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc()] + COMMON + extra + ["-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
    if force or _stale(LIB, objs):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB)] + [str(o) for o in objs]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
