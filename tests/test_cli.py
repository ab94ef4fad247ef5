"""Command-line tool (tools/srcnn_cli.cpp): the
reference tool's argument rules and exit codes (src/srcnn.cpp:331-447, :707-731),
own PNG/PPM codecs.  CPU tests cover parsing, codecs and failure exits; the GPU
test runs the whole tool."""
import subprocess
from pathlib import Path

import numpy as np
import pytest
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def cli(tmp_path_factory):
    from srcnn_cpp_amd import build as B
    B.build()
    exe = tmp_path_factory.mktemp("cli") / "srcnn_amd"
    subprocess.run(["g++", "-std=c++17", "-O2", f"-I{ROOT / 'include'}", f"-I{ROOT / 'tools'}",
                    str(ROOT / "tools" / "srcnn_cli.cpp"), f"-L{ROOT / 'srcnn_cpp_amd'}", "-lsrcnn_amd", "-lz", "-ldl",
                    f"-Wl,-rpath,{ROOT / 'srcnn_cpp_amd'}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    return exe


def run(exe, *args, cwd=None):
    return subprocess.run([str(exe), *map(str, args)], capture_output=True, text=True, cwd=cwd)


def test_no_arguments_prints_title_and_help(cli):
    r = run(cli)                                            # src/srcnn.cpp:709-715: returns 0
    assert r.returncode == 0
    assert "Super-Resolution with deep Convolutional Neural Networks" in r.stdout
    assert "--scale=( ratio: 0.1 to .. )" in r.stdout and "--noverbose" in r.stdout
    assert run(cli, "--help", "x.png").returncode == 0      # --help wins over a source (:394)


def test_load_failure_exit_code(cli, tmp_path):
    r = run(cli, tmp_path / "missing.png")
    assert r.returncode == 255 and "load failure" in r.stdout          # t_exit_code = -1, :479
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not an image")
    assert run(cli, "--noverbose", bad).returncode == 255
    assert run(cli, "--noverbose", bad).stdout == ""                    # --noverbose silences progress (:372-375)


def test_scale_too_small(cli, tmp_path):
    p = tmp_path / "a.ppm"
    Image.fromarray(np.zeros((3, 3, 3), np.uint8)).save(p)
    r = run(cli, "--scale=0.2", p)                                      # (int)(3*0.2) == 0, :485-495
    assert r.returncode == 255 and "ratio too small" in r.stdout
    r = run(cli, "--scale=-3", "--copy", p, tmp_path / "b.ppm")         # non-positive ratio ignored (:365)
    assert r.returncode == 0


@pytest.mark.parametrize("mode", ["RGB", "L", "RGBA", "P", "LA"])
def test_png_decoder_matches_pil(cli, tmp_path, mode):
    rng = np.random.default_rng(5)
    base = (rng.integers(0, 256, (37, 53, 3)) // 16 * 16).astype(np.uint8)
    base[5:20, 7:30] = [200, 30, 99]                                    # flat areas: exercises all PNG filters
    img = Image.fromarray(base).convert(mode) if mode != "P" else Image.fromarray(base).quantize(64)
    src = tmp_path / f"in_{mode}.png"
    img.save(src)
    want = np.asarray(Image.open(src).convert("RGB"))                   # cv::imread(IMREAD_COLOR): alpha dropped
    for ext in (".ppm", ".png"):
        dst = tmp_path / f"out_{mode}{ext}"
        assert run(cli, "--noverbose", "--copy", src, dst).returncode == 0
        assert np.array_equal(np.asarray(Image.open(dst).convert("RGB")), want)


def test_pnm_reader_and_default_output_name(cli, tmp_path):
    arr = np.random.default_rng(1).integers(0, 256, (9, 14, 3), dtype=np.uint8)
    Image.fromarray(arr).save(tmp_path / "pic.ppm")
    Image.fromarray(arr[:, :, 0]).save(tmp_path / "grey.pgm")
    assert run(cli, "--noverbose", "--copy", "pic.ppm", cwd=tmp_path).returncode == 0
    out = tmp_path / "pic_resized.ppm"                                  # "<name>_resized<ext>", :396-416
    assert out.exists() and np.array_equal(np.asarray(Image.open(out)), arr)
    assert run(cli, "--noverbose", "--copy", "grey.pgm", "g.png", cwd=tmp_path).returncode == 0
    assert np.array_equal(np.asarray(Image.open(tmp_path / "g.png"))[:, :, 1], arr[:, :, 0])


@pytest.mark.gpu
def test_cli_end_to_end_on_gpu(cli, tmp_path, gpu_ctx):
    from srcnn_cpp_amd.synth import synth_luma
    bgr = np.stack([synth_luma(90, 60, frame=k, seed=7 + k) for k in range(3)], axis=2)
    Image.fromarray(bgr[:, :, ::-1]).save(tmp_path / "in.png")
    r = run(cli, "--scale=1.5", "in.png", cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "- Scale multiply ratio : 1.50" in r.stdout and "- Performace : " in r.stdout and "ms took." in r.stdout
    got = np.asarray(Image.open(tmp_path / "in_resized.png"))[:, :, ::-1]
    assert np.array_equal(got, gpu_ctx.process_bgr(bgr, 1.5))
