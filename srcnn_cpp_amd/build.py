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
# object directory, for same-box A/B timing (load it with srcnn_cpp_amd.use_library() / bench.py --lib); the default build
# makes the PRODUCT library and, from the same kernel objects, the TUNING library (host units recompiled with
# -DSRCNN_TUNING_BUILD: experiment knobs and test hooks, which the product does not contain)
VARIANT = os.environ.get("SRCNN_BUILD_VARIANT", "")
LIB = PKG / (f"libsrcnn_amd_{VARIANT}.so" if VARIANT else "libsrcnn_amd.so")
OBJ = PKG.parent / ("build" + (f"_{VARIANT}" if VARIANT else ""))

ARCH = "gfx950"
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
COMMON += os.environ.get("SRCNN_BUILD_DEFINES", "").split()
# (source, extra flags).  srcnn_exact.hip reproduces the reference's
# multiply-then-add arithmetic: contraction to FMA must stay off there.
EXPORTS = PKG / "csrc" / "exports.map"       # the product exports the C symbols of include/srcnn_amd.h and nothing else
UNITS = [
    # the SLP vectoriser packs the few scalar adds of the layer-3 sums into v_pk_add_f32 behind v_mov shuffles: more
    # vector instructions, not fewer, and every one of them costs MFMA issue time
    ("srcnn_mfma.hip", ["-fno-slp-vectorize"]),
    # ... and the same strip kernels with every MFMA <-> vector-ALU hazard visible to the compiler (namespace srcnn::safe,
    # launch_strip_safe): what a context launches when the interlock probe fails on its device
    ("srcnn_mfma.hip", ["-fno-slp-vectorize", "-DSRCNN_SAFE_HAZARDS=1"], "srcnn_mfma_safe"),
    # the probe must run the FAST row body's sequences as the strip kernels have them: accumulators in architectural VGPRs
    # (left alone the compiler moves the second chain to AGPRs and pads the read-back: another dependency than the one probed;
    # tests/test_abi.py checks the listing)
    ("srcnn_probe.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"]),
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
    ("srcnn_model.cpp", ["-x", "hip"]),
    ("srcnn_plan.cpp", ["-x", "hip"]),
    ("srcnn_launch.cpp", ["-x", "hip"]),
    ("srcnn_host.cpp", ["-x", "hip", "-ffp-contract=off"]),      # cubic_table(): OpenCV's float arithmetic, nothing contracted
    ("srcnn_multi.cpp", ["-x", "hip"]),
]
HOST_UNITS = {"srcnn_api.cpp", "srcnn_model.cpp", "srcnn_plan.cpp", "srcnn_launch.cpp", "srcnn_host.cpp", "srcnn_multi.cpp"}


def kernel_sources_fingerprint() -> str:
    """sha256 over the sources that decide what the conv-path kernels do and how they are launched.  The rocprofv3 --pmc
    figures bench.py quotes (profiles/pmc_traffic.json) carry the fingerprint they were taken with; a line never quotes
    counters of a different kernel build (the GPU box holds no .git, so a commit id cannot serve)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in ("srcnn_mfma.hip", "srcnn_split16.hip", "srcnn_exact.hip", "srcnn_probe.hip", "srcnn_kernels.h", "srcnn_ctx.h",
                 "srcnn_model.cpp", "srcnn_plan.cpp", "srcnn_launch.cpp"):
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


def _compile(src: str, extra, obj: Path, headers, force: bool, verbose: bool) -> None:
    s = CSRC / src
    if force or _stale(obj, [s] + headers):
        cmd = [hipcc()] + COMMON + list(extra) + ["-c", str(s), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)


def _link(lib: Path, objs, exports, force: bool, verbose: bool) -> None:
    if force or _stale(lib, list(objs) + [exports]):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", f"-Wl,--version-script={exports}", "-o", str(lib)]
        cmd += [str(o) for o in objs]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)


TUNING_LIB = PKG / "libsrcnn_amd_tuning.so"


def build(force: bool = False, verbose: bool = False) -> Path:
    OBJ.mkdir(exist_ok=True)
    headers = [CSRC / "srcnn_kernels.h", CSRC / "srcnn_ctx.h", PKG.parent / "include" / "srcnn_amd.h", Path(__file__)]
    objs, tuning_objs = [], []
    for unit in UNITS:
        src, extra = unit[0], unit[1]
        o = OBJ / ((unit[2] if len(unit) > 2 else Path(src).stem) + ".o")
        _compile(src, extra, o, headers, force, verbose)
        objs.append(o)
        if src in HOST_UNITS and not VARIANT:
            t = OBJ / ("tuning_" + Path(src).stem + ".o")
            _compile(src, list(extra) + ["-DSRCNN_TUNING_BUILD"], t, headers, force, verbose)
            tuning_objs.append(t)
        else:
            tuning_objs.append(o)            # the kernels are the product's own objects
    _link(LIB, objs, EXPORTS, force, verbose)
    if not VARIANT:
        _link(TUNING_LIB, tuning_objs, EXPORTS, force, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
