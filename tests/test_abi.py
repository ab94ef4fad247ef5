"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/srcnn_amd.h declares; argument validation and the
"no GPU -> loud failure, no CPU fallback" contract.  No compute calls."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

import srcnn_cpp_amd as S
from srcnn_cpp_amd import build as B

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    B.build()                      # hipcc cross-compiles gfx950 without a GPU
    return S.load_library()


def header_symbols():
    text = (ROOT / "include" / "srcnn_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srcnn_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    syms = header_symbols()
    assert len(syms) >= 20
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in srcnn_amd.h but not exported"
    assert sorted(S.ABI_SYMBOLS) == syms
    assert lib.srcnn_abi_version() == 1


def test_product_library_has_no_debug_surface(lib):
    """The product library reads no environment variable and exports exactly the ABI: the SRCNN_DEBUG_* experiment knobs,
    SRCNN_HOST_COPY_THREADS and the srcnn_debug_* hooks exist only in libsrcnn_amd_tuning.so (-DSRCNN_TUNING_BUILD), and the
    Python binding is redirected to another build only by an explicit use_library() call (no SRCNN_LIB variable)."""
    import subprocess
    prod, tune = S.library_path(), S.tuning_library_path()
    assert prod.name == "libsrcnn_amd.so" and tune.exists()
    text = prod.read_bytes()
    for needle in (b"SRCNN_DEBUG", b"SRCNN_HOST_COPY", b"SRCNN_LIB", b"getenv"):
        assert needle not in text, needle
    assert b"SRCNN_DEBUG_TUNE" in tune.read_bytes()
    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", str(path)], check=True, capture_output=True, text=True).stdout
        return sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert exported(prod) == header_symbols()
    extra = sorted(set(exported(tune)) - set(header_symbols()))
    assert extra and all(n.startswith("srcnn_debug_") for n in extra), extra
    assert "SRCNN_LIB" not in (ROOT / "srcnn_cpp_amd" / "__init__.py").read_text().replace("no SRCNN_LIB", "")


def test_no_torch_or_hip_types_in_header():
    text = (ROOT / "include" / "srcnn_amd.h").read_text()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for banned in ("hipStream_t", "at::", "torch", "cv::", "std::"):
        assert banned not in code


def test_reference_mirror_header_compiles():
    """include/srcnn_amd.hpp restates the reference prototypes (src/srcnn.cpp:60-73)
    over the C ABI; it must compile as plain C++ without OpenCV or HIP."""
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        src = Path(d) / "t.cpp"
        src.write_text('#include "srcnn_amd.hpp"\n'
                       'int main(){ srcnn::Plane<unsigned char> y(4,3); srcnn::Plane<float> f(4,3);\n'
                       ' return (y.rows==3 && f.cols==4) ? 0 : 1; }\n')
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", f"-I{ROOT / 'include'}", str(src)], check=True)


def test_reference_mirror_templates_accept_a_cv_mat_shaped_type():
    """OpenCV is absent from this image, so the templates of srcnn_amd.hpp cannot be instantiated with the real cv::Mat
    here.  They only use what the reference uses of it: `rows`, `cols`, `data` (uchar*) and `step`, which in OpenCV is
    a cv::MatStep OBJECT convertible to size_t, not an integer.  Instantiate all four prototypes (and ForwardY) with a
    type of exactly that shape -- compile only, nothing is called."""
    import subprocess, tempfile
    code = r'''
#include "srcnn_amd.hpp"
struct MatStepLike { std::size_t p[2]; operator std::size_t() const { return p[0]; } };
struct MatLike { int flags, dims, rows, cols; unsigned char *data; MatStepLike step; };
static float k99[64][9][9], b99[64], k11[32][64], b11[32], k55[32][5][5];
void instantiate(MatLike &y, MatLike &f, std::vector<MatLike> &v32, std::vector<MatLike> &v64) {
    srcnn::Convolution99(y, f, k99[0], 0.f);                       // src/srcnn.cpp:60-61
    srcnn::Convolution11(v64, f, k11[0], 0.f);                     // :63-64
    srcnn::Convolution55(v32, y, k55, 0.f);                        // :66-67
    srcnn::Convolution99x11(y, v32, k99, b99, k11, b11);           // :69-73
    srcnn::ForwardY(y, y, k99, b99, k11, b11, k55, 0.f);
}
int main() { return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = Path(d) / "m.cpp"
        src.write_text(code)
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src)], check=True)


def test_reference_call_sites_compile_with_device_planes():
    """The text of the reference's two call sites (src/srcnn.cpp:609, :627) with ONLY the element type of the vector allocated at
    :602-607 changed to srcnn::DevicePlane<float>: overload resolution must pick the device-plane overloads next to the generic
    templates (a cv::Mat-shaped type for the u8 planes, as above) -- compile only."""
    import subprocess, tempfile
    code = r'''
#include "srcnn_amd.hpp"
using namespace srcnn;
struct MatStepLike { std::size_t p[2]; operator std::size_t() const { return p[0]; } };
struct MatLike { int flags, dims, rows, cols; unsigned char *data; MatStepLike step; };
#define CONV2_FILTERS 32
static float weights_conv1_data[64][9][9], biases_conv1[64], weights_conv2_data[32][64], biases_conv2[32], weights_conv3_data[32][5][5];
static float biases_conv3;
void driver(std::vector<MatLike> &pImg, MatLike &pImgConv3) {
    std::vector<srcnn::DevicePlane<float>> pImgConv2 = srcnn::DevicePlanes<float>(CONV2_FILTERS, pImg[0].cols, pImg[0].rows);
    Convolution99x11( pImg[0], pImgConv2, weights_conv1_data, biases_conv1, weights_conv2_data, biases_conv2 );
    Convolution55( pImgConv2, pImgConv3, weights_conv3_data, biases_conv3 );
    // ... and the host-plane form still resolves to the generic templates
    std::vector<MatLike> host32(32);
    Convolution99x11( pImg[0], host32, weights_conv1_data, biases_conv1, weights_conv2_data, biases_conv2 );
    Convolution55( host32, pImgConv3, weights_conv3_data, biases_conv3 );
}
int main() { return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = Path(d) / "m.cpp"
        src.write_text(code)
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src)], check=True)


def test_reference_call_sites_compile_against_opencv4s_mat_interface():
    """the literal drop-in.  tests/helpers/opencv4_mat_shape.hpp declares cv::Mat with OpenCV 4's public
    member TYPES (int rows, cols; uchar* data; MatStep step with operator size_t; MatSize size) -- the real headers
    (src/srcnn.h:6-9) cannot be satisfied in this image.  The TEXT of src/srcnn.cpp:602-627 (allocation of the 32 planes, the two
    calls) then compiles against include/srcnn_amd.hpp's adapters as it stands, `using namespace cv; using namespace std;` as
    the reference has them -- and again with only the vector's element type changed to srcnn::DevicePlane<float>.  Also the two
    un-fused functions (:60-67).  Compile only: nothing is linked."""
    import subprocess, tempfile
    code = r'''
#include "opencv4_mat_shape.hpp"
#include "srcnn_amd.hpp"
using namespace cv;
using namespace std;
using namespace srcnn;
#define CONV1_FILTERS 64
#define CONV2_FILTERS 32
typedef float ConvKernel64_99[CONV1_FILTERS][9][9];
typedef float ConvKernel32_55[CONV2_FILTERS][5][5];
typedef float ConvKernel21[CONV2_FILTERS][CONV1_FILTERS];
typedef float ConvKernel1[CONV1_FILTERS];
typedef float ConvKernel2[CONV2_FILTERS];
extern const ConvKernel64_99 weights_conv1_data;
extern const ConvKernel1 biases_conv1;
extern const ConvKernel21 weights_conv2_data;
extern const ConvKernel2 biases_conv2;
extern const ConvKernel32_55 weights_conv3_data;
extern const float biases_conv3;
void reference_text(vector<Mat> &pImg)
{
    vector<Mat> pImgConv2(CONV2_FILTERS);
    for ( unsigned cnt=0; cnt<CONV2_FILTERS; cnt++)
    {
        pImgConv2[cnt].create( pImg[0].size(), CV_32F );
    }

    Convolution99x11( pImg[0], pImgConv2, weights_conv1_data, biases_conv1, weights_conv2_data, biases_conv2 );

    Mat pImgConv3;
    pImgConv3.create(pImg[0].size(), CV_8U);
    Convolution55(pImgConv2, pImgConv3, weights_conv3_data, biases_conv3);
}
void device_planes(vector<Mat> &pImg)
{
    vector<DevicePlane<float>> pImgConv2 = DevicePlanes<float>(CONV2_FILTERS, pImg[0].cols, pImg[0].rows);
    Convolution99x11( pImg[0], pImgConv2, weights_conv1_data, biases_conv1, weights_conv2_data, biases_conv2 );
    Mat pImgConv3;
    pImgConv3.create(pImg[0].size(), CV_8U);
    Convolution55(pImgConv2, pImgConv3, weights_conv3_data, biases_conv3);
}
void unfused(Mat &y, vector<Mat> &conv1, Mat &dst)
{
    Convolution99( y, conv1[0], weights_conv1_data[0], biases_conv1[0] );
    Convolution11( conv1, dst, weights_conv2_data[0], biases_conv2[0] );
}
int main() { return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = Path(d) / "m.cpp"
        src.write_text(code)
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", f"-I{ROOT / 'include'}",
                        f"-I{ROOT / 'tests' / 'helpers'}", str(src)], check=True)


def test_lanes_adapter_compiles():
    """srcnn::ForwardYLanes over a SessionSet (two sessions on one device = two lanes of it): compile only."""
    import subprocess, tempfile
    code = r'''
#include "srcnn_amd.hpp"
void stream(srcnn::SessionSet &lanes, std::vector<srcnn::DevicePlane<unsigned char>> &in, std::vector<srcnn::DevicePlane<unsigned char>> &out)
{
    srcnn::ForwardYLanes(lanes, in, out);
    lanes.synchronize();
}
int main() { return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = Path(d) / "m.cpp"
        src.write_text(code)
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src)], check=True)


def test_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    h = ctypes.c_void_p()
    assert lib.srcnn_create(ctypes.byref(h), 0) == -4          # SRCNN_ERR_NODEVICE
    assert not h
    with pytest.raises(S.SrcnnError):
        S.Context(0)
    with pytest.raises(S.SrcnnError):
        S.Convolution99(np.zeros((4, 4), np.uint8), np.zeros((4, 4), np.float32), np.zeros(81, np.float32), 0.0)


def test_null_and_bad_arguments(lib):
    assert lib.srcnn_create(None, 0) == -1
    assert lib.srcnn_set_mode(None, 0) == -1
    assert lib.srcnn_get_mode(None) == -1
    assert lib.srcnn_last_error(None) == b"null context"
    lib.srcnn_destroy(None)                                    # harmless
    assert lib.srcnn_kernel_variant(None) == -1 and lib.srcnn_halo_transport(None) == -1
    assert lib.srcnn_ipc_export(None, None, None) == -1 and lib.srcnn_ipc_open(None, None, None) == -1
    assert lib.srcnn_set_fixup_strict(None, 1) == -1


def test_python_binding_validates_planes():
    with pytest.raises(TypeError):
        S._plane(np.zeros((4, 4), np.float32), np.uint8, "src")
    with pytest.raises(ValueError):
        S._plane(np.zeros((4, 8), np.uint8)[:, ::2], np.uint8, "src")
    a, stride = S._plane(np.zeros((4, 8), np.uint8)[:, :5], np.uint8, "src")
    assert stride == 8


@pytest.mark.parametrize("unit", ["srcnn_exact.hip", "srcnn_pipeline.hip"])
def test_exact_kernels_have_no_fused_multiply_add(tmp_path, unit):
    """srcnn_exact.hip must keep the reference's multiply-then-add arithmetic, srcnn_pipeline.hip OpenCV's separately
    rounded float32 products and sums in the resize's vertical pass (HIP's __fmul_rn / __fadd_rn are plain * and +, so
    only the build flag keeps them apart).  Compiled with the flags the build uses (srcnn_cpp_amd/build.py)."""
    import subprocess
    flags = {u[0]: u[1] for u in B.UNITS if len(u) == 2}[unit]
    assert "-ffp-contract=off" in flags
    out = tmp_path / "unit.s"
    subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-S",
                    "--cuda-device-only", "-o", str(out), str(B.CSRC / unit)],
                   check=True, stderr=subprocess.DEVNULL)
    asm = out.read_text()
    assert "v_mul_f32" in asm and "v_add_f32" in asm
    if unit == "srcnn_exact.hip":
        assert "v_add_f64" in asm
    for banned in ("v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_mad_f32", "v_pk_fma_f32", "v_fma_f64",
                   "v_fma_mix"):
        assert banned not in asm, banned


def test_interlock_probe_runs_the_strip_kernels_own_sequences(tmp_path):
    """srcnn_create's interlock probe (csrc/srcnn_probe.hip) is only evidence if its no-wait kernel issues the FAST row body's
    dependent pairs back to back (advisor, round 4): the MFMA, then the eight packed clamp multiplies rewriting ITS registers in
    place, then an MFMA reading the first rewritten register -- no copy through AGPRs, no v_mov and no more than the one-cycle
    s_nop the assembler's own rules ask for in between.  Checked on the listing the build's flags produce."""
    import re
    import subprocess
    flags = [u[1] for u in B.UNITS if u[0] == "srcnn_probe.hip"][0]
    out = tmp_path / "probe.s"
    subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, f"-I{B.CSRC}", "-S", "--cuda-device-only",
                    "-o", str(out), str(B.CSRC / "srcnn_probe.hip")], check=True, stderr=subprocess.DEVNULL)
    text = out.read_text()
    m = re.search(r"^(\S*interlock_probe_kernelILb0\S*):", text, re.M)
    body = text[m.end():text.index(".Lfunc_end", m.end())]
    ins = [l.split(";")[0].strip() for l in body.splitlines()]
    # (scalar-ALU bookkeeping of the probe's loop -- its counter -- issues beside the vector stream and is no wait state of note)
    ins = [i for i in ins if i and not i.startswith(".") and not i.endswith(":") and not re.match(r"s_(add|cmp|mov|sub)", i)]
    assert not any(i.startswith("v_accvgpr") for i in ins), "the probe's accumulators must stay in architectural VGPRs"
    mfma = [k for k, i in enumerate(ins) if i.startswith("v_mfma_f32_32x32x2")]
    assert len(mfma) == 1 + 16 + 16          # one MFMA feeding the first clamp, then the two 16-step chains
    chains = 0
    for k in mfma:
        dst = re.match(r"v_mfma_f32_32x32x2_f32 v\[(\d+):(\d+)\]", ins[k])
        nxt = ins[k + 1:k + 9]
        if dst and all(n.startswith("v_pk_mul_f32") and "clamp" in n for n in nxt):
            lo = int(dst.group(1))
            # (1) the eight multiplies rewrite exactly the MFMA's result registers, in place, directly behind it
            assert [re.match(r"v_pk_mul_f32 v\[(\d+):", n).group(1) for n in nxt] == [str(lo + 2 * q) for q in range(8)], (ins[k], nxt)
            assert all(re.match(r"v_pk_mul_f32 (v\[\d+:\d+\]), \1,", n) for n in nxt)
            # (2)/(3) the next MFMA reads the first rewritten register as its B operand, at most an `s_nop 0` in between
            follow = [i for i in ins[k + 9:k + 11] if i != "s_nop 0"]
            assert follow[0].startswith("v_mfma_f32_32x32x2") and re.search(rf", v\d+, v{lo}, ", follow[0] + " "), follow
            chains += 1
    assert chains == 2, "both rewritten accumulators (layer 1 -> 2, layer 2 -> 3) must be probed"


class _NoCall:
    """Stands in for the C library: the binding must reject a bad call BEFORE reaching the ABI."""
    def __getattr__(self, name):
        raise AssertionError(f"{name} reached the C ABI with mismatched planes")


def _shell_context():
    ctx = object.__new__(S.Context)           # no device needed: only the Python-side validation runs
    ctx._lib, ctx._h = _NoCall(), None
    return ctx


def test_python_binding_rejects_mismatched_shapes():
    """A dst / preclamp / plane smaller than the plane the dims are taken from would be overrun by the
    device-to-host copies (the C side only sees pointers and one width x height)."""
    ctx = _shell_context()
    u8, f32 = (lambda h, w: np.zeros((h, w), np.uint8)), (lambda h, w: np.zeros((h, w), np.float32))
    with pytest.raises(ValueError):
        ctx.forward_y(u8(8, 8), dst=u8(4, 8))
    with pytest.raises(ValueError):
        ctx.forward_y(u8(8, 8), dst=u8(8, 8), preclamp=f32(8, 7))
    with pytest.raises(ValueError):
        ctx.conv99(u8(4, 8), f32(8, 8), np.zeros(81, np.float32), 0.0)
    with pytest.raises(ValueError):
        ctx.conv11([f32(4, 8)] * 64, f32(8, 8), np.zeros(64, np.float32), 0.0)
    with pytest.raises(ValueError):
        ctx.conv11([f32(8, 8)] * 63 + [f32(4, 8)], f32(8, 8), np.zeros(64, np.float32), 0.0)
    with pytest.raises(ValueError):
        ctx.conv55([f32(8, 4)] * 32, u8(8, 8), np.zeros(800, np.float32), 0.0)
    with pytest.raises(ValueError):
        ctx.conv99x11(u8(8, 8), [f32(8, 8)] * 31 + [f32(7, 8)], np.zeros(5184, np.float32), np.zeros(64, np.float32),
                      np.zeros(2048, np.float32), np.zeros(32, np.float32))
    with pytest.raises(ValueError):
        ctx.conv99x11(u8(8, 8), [f32(6, 8)] * 32, np.zeros(5184, np.float32), np.zeros(64, np.float32),
                      np.zeros(2048, np.float32), np.zeros(32, np.float32))
    frames = np.zeros((3, 8, 8), np.uint8)
    with pytest.raises(ValueError):
        ctx.forward_y_frames(frames, out=np.zeros((2, 8, 8), np.uint8))
    with pytest.raises(TypeError):
        ctx.forward_y_frames(frames, out=np.zeros((3, 8, 8), np.float32))
    with pytest.raises(ValueError):
        ctx.forward_y_frames(frames, out=np.zeros((3, 8, 16), np.uint8)[:, :, ::2])
    ro = np.zeros((3, 8, 8), np.uint8)
    ro.flags.writeable = False
    with pytest.raises(ValueError):
        ctx.forward_y_frames(frames, out=ro)
    with pytest.raises(ValueError):
        ctx.forward_y_frames(np.zeros((0, 8, 8), np.uint8))


@pytest.mark.parametrize("unit", ["srcnn_mfma.hip", "srcnn_split16.hip", "srcnn_exact.hip"])
def test_no_kernel_keeps_registers_in_scratch_memory(tmp_path, unit):
    """A kernel that spills -- or keeps an array behind a selected reference in private memory -- pays for it at dispatch and in
    every row: the REFBYTES16 instantiation of the split-f16 strip kernel ran at 503 us per 3840x2160 plane instead of 272 for
    four rounds because `last ? tA : tB` on two accumulator tiles sent both to scratch (round 5).  Every kernel of the three hot
    translation units must report a private segment of 0 bytes, with the flags the build uses."""
    import re
    import subprocess
    flags = [u[1] for u in B.UNITS if u[0] == unit and len(u) == 2][0]
    out = tmp_path / "unit.s"
    subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, f"-I{B.CSRC}", "-S", "--cuda-device-only",
                    "-o", str(out), str(B.CSRC / unit)], check=True, stderr=subprocess.DEVNULL)
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", out.read_text(), re.S)
    assert len(kernels) >= 4
    for name, body in kernels:
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)) == 0, name
