"""AddressSanitizer + UndefinedBehaviorSanitizer passes over the HOST code, CPU only -- never on
the GPU: (1) the library's host-only logic (work-item planner, fragment packers, cubic tables) by compiling
the host units of the C-ABI layer themselves with the host compiler and the sanitizers; (2) the hand-written PNG / PNM decoders of
tools/image_io.hpp against truncated, corrupted and crafted files (untrusted input)."""
import shutil
import struct
import subprocess
import zlib
from pathlib import Path

import numpy as np
import pytest
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
CLANG = Path("/opt/rocm/lib/llvm/bin/clang++")       # host C++ compiler with _Float16 and the sanitizer runtimes
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1", "-std=c++17"]
ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1", "PATH": "/usr/bin:/bin"}

pytestmark = pytest.mark.skipif(not CLANG.exists(), reason="ROCm clang++ not found")
HOST_UNITS = [str(ROOT / "srcnn_cpp_amd" / "csrc" / f"srcnn_{u}.cpp") for u in ("api", "model", "plan", "launch", "host", "multi")]


def test_host_logic_under_asan_ubsan(tmp_path):
    exe = tmp_path / "san_host"
    subprocess.run([str(CLANG), *SAN, "-D__HIP_PLATFORM_AMD__", "-DSRCNN_TUNING_BUILD", "-I/opt/rocm/include",
                    f"-I{ROOT / 'srcnn_cpp_amd' / 'csrc'}", *HOST_UNITS, str(ROOT / "tests" / "checks" / "san_host.cpp"),
                    "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), str(ROOT / "srcnn_cpp_amd" / "data" / "srcnn915_weights.f32")], capture_output=True,
                       text=True, env=ENV, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.startswith("ok:") and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_worker_pool_under_thread_sanitizer(tmp_path):
    """The persistent worker threads of the several-GPUs entry points (WorkerPool in srcnn_ctx.h: one mutex + condition
    variable per worker, tasks handed over and results collected per call) under ThreadSanitizer: a data race or a lost wake-up
    in that hand-over would be a wrong row stripe once in a million steps.  The same harness, SRCNN_SAN_POOL_ONLY: only the pool."""
    exe = tmp_path / "tsan_host"
    subprocess.run([str(CLANG), "-fsanitize=thread", "-g", "-O1", "-std=c++17", "-DSRCNN_SAN_POOL_ONLY", "-D__HIP_PLATFORM_AMD__",
                    "-DSRCNN_TUNING_BUILD", "-I/opt/rocm/include", f"-I{ROOT / 'srcnn_cpp_amd' / 'csrc'}", *HOST_UNITS,
                    str(ROOT / "tests" / "checks" / "san_host.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
                    "-pthread", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env={"PATH": "/usr/bin:/bin", "TSAN_OPTIONS": "halt_on_error=1"},
                       timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok: worker pool"), (r.stdout + r.stderr)[-3000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


def test_oracle_under_asan_ubsan_and_thread_oversubscription(tmp_path):
    """The CHECKER under the sanitizers (advisor, round 5): oracle/srcnn_oracle.c's conv path compiled with AddressSanitizer +
    UBSan, on the edge sizes (1 x 1 ... 40 x 2) and random planes, every plane twice with 256 OpenMP threads on this container's
    8 CPUs and once with one thread -- exact-size buffers, no error, no disagreement between the three runs."""
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not found")
    exe = tmp_path / "san_oracle"
    subprocess.run([gcc, "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1",
                    "-ffp-contract=off", "-fopenmp", "-std=c11", str(ROOT / "oracle" / "srcnn_oracle.c"),
                    str(ROOT / "tests" / "checks" / "san_oracle.c"), "-lm", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), str(ROOT / "srcnn_cpp_amd" / "data" / "srcnn915_weights.f32"), "40", "256"], capture_output=True,
                       text=True, env=dict(ENV, OMP_WAIT_POLICY="passive"), timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), (r.stdout + r.stderr)[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def png_chunk(kind: bytes, body: bytes) -> bytes:
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))


def crafted_png(w, h, ctype=6, depth=8, raw=b"\0" * 64, extra_first=b"", ihdr_len=13, twice=False):
    ihdr = struct.pack(">IIBBBBB", w & 0xFFFFFFFF, h & 0xFFFFFFFF, depth, ctype, 0, 0, 0)[:ihdr_len]
    out = b"\x89PNG\r\n\x1a\n" + extra_first + png_chunk(b"IHDR", ihdr)
    if twice:
        out += png_chunk(b"IHDR", struct.pack(">IIBBBBB", 1, 1, 8, 2, 0, 0, 0))
    return out + png_chunk(b"IDAT", zlib.compress(raw)) + png_chunk(b"IEND", b"")


def test_image_decoders_under_asan_ubsan(tmp_path):
    exe = tmp_path / "san_image_io"
    subprocess.run([str(CLANG), *SAN, f"-I{ROOT / 'tools'}", str(ROOT / "tests" / "checks" / "san_image_io.cpp"), "-lz", "-o",
                    str(exe)], check=True)
    rng = np.random.default_rng(11)
    files, must_accept = [], []
    base = (rng.integers(0, 256, (23, 31, 3)) // 32 * 32).astype(np.uint8)
    base[4:15, 3:20] = [10, 200, 77]
    for mode in ("RGB", "L", "RGBA", "P", "LA"):
        img = Image.fromarray(base).convert(mode)
        p = tmp_path / f"good_{mode}.png"
        img.save(p)
        files.append(p)
        must_accept.append(p)
        data = p.read_bytes()
        for cut in sorted({8, 20, 33, 34, 40, len(data) // 2, len(data) - 13, len(data) - 5, len(data) - 1}):
            q = tmp_path / f"trunc_{mode}_{cut}.png"
            q.write_bytes(data[:cut])
            files.append(q)
        for k in range(40):                                          # byte flips anywhere (headers, lengths, zlib stream)
            d = bytearray(data)
            for _ in range(1 + k % 3):
                d[int(rng.integers(8, len(d)))] ^= int(rng.integers(1, 256))
            q = tmp_path / f"flip_{mode}_{k}.png"
            q.write_bytes(bytes(d))
            files.append(q)
    ppm = tmp_path / "good.ppm"
    Image.fromarray(base).save(ppm)
    pgm = tmp_path / "good.pgm"
    Image.fromarray(base[:, :, 0]).save(pgm)
    files += [ppm, pgm]
    must_accept += [ppm, pgm]
    crafted = {
        "huge_dims.png": crafted_png(0xFFFFFFFF, 0xFFFFFFFF),                      # (stride+1)*h wraps size_t
        "wrap_rgba.png": crafted_png(0x40000001, 4),                               # w*4 wraps 32 bits
        "tall.png": crafted_png(1, 0x7FFFFFFF, ctype=0),
        "wide_cap.png": crafted_png(65535, 65535, ctype=0),                        # inside the side cap, over the pixel cap
        "zero_w.png": crafted_png(0, 5),
        "short_ihdr.png": crafted_png(4, 4, ihdr_len=9),
        "two_ihdr.png": crafted_png(4, 4, twice=True),
        "ihdr_not_first.png": crafted_png(4, 4, extra_first=png_chunk(b"tEXt", b"x")),
        "short_data.png": crafted_png(16, 16, raw=b"\0" * 10),
        "bad_filter.png": crafted_png(2, 2, ctype=0, raw=b"\x09\1\2\x09\3\4"),
        "palette_oob.png": crafted_png(2, 1, ctype=3, raw=b"\0\xff\xfe"),
        "no_idat.png": b"\x89PNG\r\n\x1a\n" + png_chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 8, 2, 0, 0, 0)) + png_chunk(b"IEND", b""),
        "len_overflow.png": b"\x89PNG\r\n\x1a\n" + struct.pack(">I", 0xFFFFFFF0) + b"IHDR" + b"\0" * 30,
        "digits.ppm": b"P6\n" + b"9" * 40 + b" 2\n255\n" + b"\0" * 12,               # v*10 overflows a long
        "wrap.ppm": b"P6\n65535 65535\n255\n" + b"\0" * 64,
        "wrap2.ppm": b"P6\n4294967297 1\n255\n" + b"\0" * 64,
        "neg.ppm": b"P6\n-3 2\n255\n" + b"\0" * 64,
        "short.ppm": b"P6\n4 4\n255\n" + b"\0" * 10,
        "maxval.ppm": b"P6\n2 2\n65535\n" + b"\0" * 24,
        "eof_hdr.ppm": b"P6\n2 2\n255",
        "comment.ppm": b"P6\n# c\n2 1\n255\n" + b"\1\2\3\4\5\6",
        "empty.png": b"",
        "sig_only.png": b"\x89PNG\r\n\x1a\n",
    }
    for name, data in crafted.items():
        q = tmp_path / name
        q.write_bytes(data)
        files.append(q)
    must_accept.append(tmp_path / "comment.ppm")
    r = subprocess.run([str(exe), *map(str, files)], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    lines = r.stdout.splitlines()
    for p in must_accept:
        assert any(l.startswith(f"ok {p} ") for l in lines), f"{p.name} was rejected"
    for name in crafted:
        if name != "comment.ppm":
            assert f"reject {tmp_path / name}" in lines, f"{name} was accepted"
