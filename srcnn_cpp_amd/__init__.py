"""srcnn_cpp_amd -- MI355X-native SRCNN Y-channel conv path.

This package is a thin ctypes binding of the C ABI in ``include/srcnn_amd.h``
(``libsrcnn_amd.so``, hand-written HIP for gfx950) plus a Python restatement of
the reference's call surface for tests and the bench:

    Convolution99(src, dst, kernel, bias)                  src/srcnn.cpp:92
    Convolution11(src, dst, kernel, bias)                  src/srcnn.cpp:151
    Convolution55(src, dst, kernel, bias)                  src/srcnn.cpp:189
    Convolution99x11(src, dst, k99, b99, k11, b11)         src/srcnn.cpp:254

with the reference's argument meaning: ``dst`` is pre-allocated by the caller
and written in place, planes are 2-D numpy arrays (any row stride), feature
maps are sequences of 32 / 64 planes.  The C++ host-side mirror a reference
maintainer would use is ``include/srcnn_amd.hpp``.

There is NO CPU fallback: importing works anywhere, but every compute call
needs the HIP library and a gfx950 device and raises ``SrcnnError`` otherwise.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

__all__ = [
    "Context", "SrcnnError", "forward_y_striped_frames", "load_library", "library_path", "tuning_library_path", "use_library", "load_weights", "split_weights",
    "Convolution99", "Convolution11", "Convolution55", "Convolution99x11", "default_context",
    "MODE_MFMA", "MODE_EXACT", "MODE_SPLIT16", "MODE_REFBYTES", "MODE_REFBYTES16", "FLOP_PER_PIXEL",
    "ERR_INVALID", "ERR_HIP", "ERR_NOMEM", "ERR_NODEVICE", "ERR_STATE",
    "stripe_rows", "forward_y_frames_multi", "forward_y_lanes_dev", "forward_y_striped", "forward_y_striped_dev",
]

_PKG = Path(__file__).resolve().parent
_LIB_PATH = _PKG / "libsrcnn_amd.so"              # the product library; use_library() for another build, before the first load
_WEIGHTS_PATH = _PKG / "data" / "srcnn915_weights.f32"

MODE_MFMA = 0
MODE_EXACT = 1
MODE_SPLIT16 = 2
MODE_REFBYTES = 3          # float32 MFMA + exact fix-up of the pixels next to a truncation boundary: the reference's bytes
MODE_REFBYTES16 = 4        # opt-in: the same behind the split-f16 kernel
N_WEIGHTS = 8129
# 2 x (64*81 + 32*64 + 32*25) MAC per output pixel
FLOP_PER_PIXEL = 16064

ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_NODEVICE, ERR_STATE = -1, -2, -3, -4, -5      # include/srcnn_amd.h
_ERR = {-1: "invalid argument", -2: "HIP runtime error", -3: "out of memory",
        -4: "no gfx950 device", -5: "bad state"}

_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_f32pp = C.POINTER(_f32p)
_lib = None


class SrcnnError(RuntimeError):
    def __init__(self, code: int, msg: str = ""):
        self.code = code
        super().__init__(f"srcnn error {code} ({_ERR.get(code, '?')}): {msg}")


def library_path() -> Path:
    return _LIB_PATH


def tuning_library_path() -> Path:
    """The TUNING build of the library (srcnn_cpp_amd/build.py): the product's code plus the SRCNN_DEBUG_* experiment knobs
    and the srcnn_debug_* test hooks, which the product library does not contain."""
    return _PKG / "libsrcnn_amd_tuning.so"


def use_library(path) -> None:
    """Bind this process to another build of the library (the tuning build, an A/B variant of tools/ab.sh).  Explicit and
    in-process: no environment variable redirects the binding.  Must be called before the first load."""
    global _LIB_PATH
    if _lib is not None:
        raise RuntimeError("the library is already loaded")
    _LIB_PATH = Path(path)


def load_library() -> C.CDLL:
    """dlopen libsrcnn_amd.so; fails loudly when the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise SrcnnError(-4, f"{_LIB_PATH} not built: run `python -m srcnn_cpp_amd.build` "
                             "(there is no CPU fallback)")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; a process must hold ONE
    # HIP runtime or torch tensors/streams and this library would not share a device
    # context.  Importing torch first makes the dynamic linker resolve our NEEDED
    # libamdhip64.so.7 to the copy torch already loaded (same SONAME).
    import importlib.util
    import sys
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
    lib = C.CDLL(str(_LIB_PATH))
    sz, i, vp = C.c_size_t, C.c_int, C.c_void_p
    sigs = {
        "srcnn_abi_version": ([], i),
        "srcnn_create": ([C.POINTER(vp), i], i),
        "srcnn_destroy": ([vp], None),
        "srcnn_last_error": ([vp], C.c_char_p),
        "srcnn_set_mode": ([vp, i], i),
        "srcnn_get_mode": ([vp], i),
        "srcnn_set_stream": ([vp, vp], i),
        "srcnn_synchronize": ([vp], i),
        "srcnn_kernel_variant": ([vp], i),
        "srcnn_conv99": ([vp, _u8p, sz, _f32p, sz, i, i, _f32p, C.c_float], i),
        "srcnn_conv11": ([vp, _f32pp, sz, _f32p, sz, i, i, _f32p, C.c_float], i),
        "srcnn_conv55": ([vp, _f32pp, sz, _u8p, sz, i, i, _f32p, C.c_float], i),
        "srcnn_conv99x11": ([vp, _u8p, sz, _f32pp, sz, i, i, _f32p, _f32p, _f32p, _f32p], i),
        "srcnn_set_weights": ([vp, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float], i),
        "srcnn_forward_y": ([vp, _u8p, sz, _u8p, sz, i, i, _f32p, sz], i),
        "srcnn_forward_y_frames": ([vp, C.POINTER(_u8p), sz, C.POINTER(_u8p), sz, i, i, i], i),
        "srcnn_forward_y_dev": ([vp, vp, sz, sz, vp, sz, sz, i, i, i, vp], i),
        "srcnn_forward_y_rows_dev": ([vp, vp, sz, i, vp, sz, i, i, i, i, i], i),
        "srcnn_forward_y_rows_halo_dev": ([vp, vp, sz, i, i, vp, vp, sz, vp, sz, i, i, i, i, i], i),
        "srcnn_halo_transport": ([vp], i),
        "srcnn_forward_y_unfused_dev": ([vp, vp, sz, sz, vp, sz, sz, i, i, i, vp], i),
        "srcnn_conv99x11_dev": ([vp, vp, sz, vp, sz, sz, i, i], i),
        "srcnn_conv55_dev": ([vp, vp, sz, sz, vp, sz, i, i, vp], i),
        "srcnn_conv99x11_to_dev": ([vp, _u8p, sz, vp, sz, sz, i, i, _f32p, _f32p, _f32p, _f32p], i),
        "srcnn_conv55_from_dev": ([vp, vp, sz, sz, _u8p, sz, i, i, _f32p, C.c_float], i),
        "srcnn_dev_alloc": ([vp, sz, C.POINTER(vp)], i),
        "srcnn_dev_free": ([vp, vp], i),
        "srcnn_dev_download": ([vp, vp, vp, sz], i),
        "srcnn_dev_upload": ([vp, vp, vp, sz], i),
        "srcnn_ipc_export": ([vp, vp, C.POINTER(C.c_ubyte * 64)], i),
        "srcnn_ipc_open": ([vp, C.POINTER(C.c_ubyte * 64), C.POINTER(vp)], i),
        "srcnn_ipc_close": ([vp, vp], i),
        "srcnn_query_plan": ([vp, i, i, i, C.POINTER(i * 6)], i),
        "srcnn_fixup_stats": ([vp, C.POINTER(C.c_ulonglong * 4), C.POINTER(C.c_float), C.POINTER(C.c_float)], i),
        "srcnn_set_fixup_strict": ([vp, i], i),
        "srcnn_set_seam_deferral": ([vp, i], i),
        "srcnn_flush": ([vp], i),
        "srcnn_set_fixup_margin": ([vp, C.c_float], i),
        "srcnn_set_kernel_variant": ([vp, i], i),
        "srcnn_set_fixup_local": ([vp, C.c_float], i),
        "srcnn_fixup_local_stats": ([vp, C.POINTER(C.c_float), C.POINTER(C.c_float)], i),
        "srcnn_scaled_size": ([i, i, C.c_float, C.POINTER(i), C.POINTER(i)], i),
        "srcnn_bgr2ycrcb": ([vp, _u8p, sz, i, i, _u8p, _u8p, _u8p, sz], i),
        "srcnn_ycrcb2bgr": ([vp, _u8p, _u8p, _u8p, sz, i, i, _u8p, sz], i),
        "srcnn_resize_cubic": ([vp, _u8p, sz, i, i, _u8p, sz, i, i], i),
        "srcnn_process_bgr": ([vp, _u8p, sz, i, i, C.c_float, _u8p, sz], i),
        "srcnn_process_bgr_dev": ([vp, vp, sz, i, i, C.c_float, vp, sz], i),
        "srcnn_stripe_rows": ([i, i, i, C.POINTER(i), C.POINTER(i)], i),
        "srcnn_forward_y_frames_multi": ([C.POINTER(vp), i, C.POINTER(_u8p), sz, C.POINTER(_u8p), sz, i, i, i], i),
        "srcnn_forward_y_lanes_dev": ([C.POINTER(vp), i, C.POINTER(vp), sz, C.POINTER(vp), sz, i, i, i], i),
        "srcnn_forward_y_striped": ([C.POINTER(vp), i, _u8p, sz, _u8p, sz, i, i], i),
        "srcnn_forward_y_striped_frames": ([C.POINTER(vp), i, C.POINTER(_u8p), sz, C.POINTER(_u8p), sz, i, i, i], i),
        "srcnn_forward_y_striped_dev": ([C.POINTER(vp), i, C.POINTER(vp), sz, C.POINTER(vp), sz, i, i], i),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


ABI_SYMBOLS = (
    "srcnn_abi_version", "srcnn_create", "srcnn_destroy", "srcnn_last_error", "srcnn_set_mode",
    "srcnn_get_mode", "srcnn_set_stream", "srcnn_synchronize", "srcnn_kernel_variant", "srcnn_set_kernel_variant", "srcnn_conv99", "srcnn_conv11",
    "srcnn_conv55", "srcnn_conv99x11", "srcnn_set_weights", "srcnn_forward_y", "srcnn_forward_y_frames",
    "srcnn_forward_y_dev",
    "srcnn_forward_y_rows_dev", "srcnn_forward_y_rows_halo_dev", "srcnn_halo_transport", "srcnn_forward_y_unfused_dev", "srcnn_conv99x11_dev",
    "srcnn_conv55_dev", "srcnn_conv99x11_to_dev", "srcnn_conv55_from_dev", "srcnn_dev_alloc", "srcnn_dev_free",
    "srcnn_dev_download", "srcnn_dev_upload", "srcnn_ipc_export", "srcnn_ipc_open", "srcnn_ipc_close", "srcnn_query_plan", "srcnn_fixup_stats", "srcnn_set_fixup_strict", "srcnn_set_fixup_margin", "srcnn_set_fixup_local", "srcnn_fixup_local_stats", "srcnn_set_seam_deferral", "srcnn_flush", "srcnn_scaled_size", "srcnn_bgr2ycrcb", "srcnn_ycrcb2bgr",
    "srcnn_resize_cubic", "srcnn_process_bgr", "srcnn_process_bgr_dev",
    "srcnn_stripe_rows", "srcnn_forward_y_frames_multi", "srcnn_forward_y_lanes_dev", "srcnn_forward_y_striped", "srcnn_forward_y_striped_frames", "srcnn_forward_y_striped_dev",
)


def load_weights(path: Optional[Path] = None) -> np.ndarray:
    """The SRCNN 9-1-5 parameters as an 8,129-float blob in convdata.h order
    (b1|W1|b2|W2|b3|W3; provenance: oracle/dump_weights.c)."""
    blob = np.fromfile(str(path or _WEIGHTS_PATH), dtype="<f4")
    if blob.size != N_WEIGHTS:
        raise ValueError(f"weight blob has {blob.size} floats, expected {N_WEIGHTS}")
    return blob


def split_weights(blob: np.ndarray):
    """blob -> (w1[64,9,9], b1[64], w2[32,64], b2[32], w3[32,5,5], b3)."""
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    return (blob[64:5248].reshape(64, 9, 9), blob[0:64], blob[5280:7328].reshape(32, 64),
            blob[5248:5280], blob[7329:8129].reshape(32, 5, 5), float(blob[7328]))


def _plane(a, dtype, name, writable=False):
    if not isinstance(a, np.ndarray) or a.ndim != 2 or a.dtype != dtype:
        raise TypeError(f"{name}: expected a 2-D numpy array of {np.dtype(dtype).name}")
    if a.strides[1] != a.itemsize or a.strides[0] % a.itemsize or a.strides[0] < a.shape[1] * a.itemsize:
        raise ValueError(f"{name}: rows must be contiguous (row stride may be padded)")
    if writable and not a.flags.writeable:
        raise ValueError(f"{name}: output plane is read-only")
    return a, a.strides[0] // a.itemsize


def _same_shape(name, got, want):
    """Every plane of a call has the dims the C side takes from ONE of them (the reference reads them from
    dst or src, src/srcnn.cpp:94-95, :262-263, and never checks the others): a smaller buffer would be
    overrun by the device-to-host copies, so reject it here."""
    if tuple(got) != tuple(want):
        raise ValueError(f"{name}: shape {tuple(got)} does not match the call's plane shape {tuple(want)}")


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _ptr_array(planes, n, name, writable=False):
    if len(planes) != n:
        raise ValueError(f"{name}: expected {n} planes, got {len(planes)}")
    stride = None
    shape = None
    for k, p in enumerate(planes):
        _, s = _plane(p, np.float32, f"{name}[{k}]", writable)
        if stride is None:
            stride, shape = s, p.shape
        elif s != stride or p.shape != shape:
            raise ValueError(f"{name}: all planes must share shape and row stride")
    arr = (_f32p * n)(*[_fp(p) for p in planes])
    return arr, stride, shape


def _wt(a, n, name):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.size != n:
        raise ValueError(f"{name}: expected {n} floats, got {a.size}")
    return a


class Context:
    """One GPU + one stream (``srcnn_ctx``).  Not thread-safe; one per thread."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.srcnn_create(C.byref(h), int(device))
        if rc != 0:
            raise SrcnnError(rc, f"srcnn_create(device={device}) -- a gfx950 GPU is required, "
                                 "there is no CPU fallback")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.srcnn_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise SrcnnError(rc, self._lib.srcnn_last_error(self._h).decode())

    # -- configuration ------------------------------------------------------
    def set_mode(self, mode: int):
        self._check(self._lib.srcnn_set_mode(self._h, int(mode)))

    def set_stream(self, hip_stream: int):
        self._check(self._lib.srcnn_set_stream(self._h, C.c_void_p(int(hip_stream) or None)))
        self.stream_ptr = int(hip_stream)       # 0: the context's own stream (not visible to the caller's framework)

    def synchronize(self):
        self._check(self._lib.srcnn_synchronize(self._h))

    def kernel_variant(self) -> int:
        """0 = fast strip kernels (hardware interlock verified at create), 1 = hazard-safe kernels (srcnn_kernel_variant)."""
        return int(self._lib.srcnn_kernel_variant(self._h))

    def set_kernel_variant(self, variant: int):
        """1 = pin the hazard-safe strip kernels (same bytes, ~3 % slower), 0 = what the interlock probe allows (srcnn_set_kernel_variant)."""
        self._check(self._lib.srcnn_set_kernel_variant(self._h, int(variant)))

    def set_weights(self, w1, b1, w2, b2, w3, b3):
        w1, b1 = _wt(w1, 5184, "kernel99"), _wt(b1, 64, "bias99")
        w2, b2 = _wt(w2, 2048, "kernel11"), _wt(b2, 32, "bias11")
        w3 = _wt(w3, 800, "kernel55")
        self._check(self._lib.srcnn_set_weights(self._h, _fp(w1), _fp(b1), _fp(w2), _fp(b2), _fp(w3), float(b3)))

    def set_weights_blob(self, blob):
        self.set_weights(*split_weights(blob))

    def query_plan(self, width, height, n_frames=1):
        out = (C.c_int * 6)()
        self._check(self._lib.srcnn_query_plan(self._h, width, height, n_frames, C.byref(out)))
        return dict(workgroups=out[0], seg_rows=out[1], strips=out[2], segments=out[3],
                    lds_bytes=out[4], threads=out[5])

    # -- the reference call surface (host buffers, dst written in place) ----
    def conv99(self, src, dst, kernel, bias):
        src, ss = _plane(src, np.uint8, "src")
        dst, ds = _plane(dst, np.float32, "dst", True)
        k = _wt(kernel, 81, "kernel")
        h, w = dst.shape                       # dims come from dst: src/srcnn.cpp:94-95
        _same_shape("src", src.shape, dst.shape)
        self._check(self._lib.srcnn_conv99(self._h, src.ctypes.data_as(_u8p), ss, _fp(dst), ds, w, h,
                                           _fp(k), float(bias)))

    def conv11(self, src, dst, kernel, bias):
        arr, ss, shape = _ptr_array(src, 64, "src")
        dst, ds = _plane(dst, np.float32, "dst", True)
        k = _wt(kernel, 64, "kernel")
        h, w = dst.shape
        _same_shape("src planes", shape, dst.shape)
        self._check(self._lib.srcnn_conv11(self._h, arr, ss, _fp(dst), ds, w, h, _fp(k), float(bias)))

    def conv55(self, src, dst, kernel, bias):
        arr, ss, shape = _ptr_array(src, 32, "src")
        dst, ds = _plane(dst, np.uint8, "dst", True)
        k = _wt(kernel, 800, "kernel")
        h, w = dst.shape
        _same_shape("src planes", shape, dst.shape)
        self._check(self._lib.srcnn_conv55(self._h, arr, ss, dst.ctypes.data_as(_u8p), ds, w, h,
                                           _fp(k), float(bias)))

    def conv99x11(self, src, dst, k99, b99, k11, b11):
        src, ss = _plane(src, np.uint8, "src")
        arr, ds, shape = _ptr_array(dst, 32, "dst", True)
        k99, b99 = _wt(k99, 5184, "kernel99"), _wt(b99, 64, "bias99")
        k11, b11 = _wt(k11, 2048, "kernel11"), _wt(b11, 32, "bias11")
        h, w = src.shape                       # dims come from src: src/srcnn.cpp:262-263
        _same_shape("dst planes", shape, src.shape)
        self._check(self._lib.srcnn_conv99x11(self._h, src.ctypes.data_as(_u8p), ss, arr, ds, w, h,
                                              _fp(k99), _fp(b99), _fp(k11), _fp(b11)))

    def fixup_stats(self):
        """SRCNN_MODE_REFBYTES counters since the context was created (srcnn_fixup_stats; synchronises)."""
        out, delta, dev = (C.c_ulonglong * 4)(), C.c_float(), C.c_float()
        self._check(self._lib.srcnn_fixup_stats(self._h, C.byref(out), C.byref(delta), C.byref(dev)))
        return {"scattered_pixels": int(out[0]), "dense_tiles": int(out[1]), "bytes_changed": int(out[2]),
                "exact_reruns": int(out[3]), "delta": float(delta.value), "max_dev": float(dev.value)}

    def set_fixup_strict(self, on: bool = True):
        """SRCNN_MODE_REFBYTES: redo a launch in the reference's arithmetic on every pixel when its monitored deviation exceeds
        delta / 2 -- on the device, no host read; ON by default."""
        self._check(self._lib.srcnn_set_fixup_strict(self._h, int(bool(on))))

    def set_seam_deferral(self, on: bool = True):
        """Fused float32 launches queued back to back: the seam blocks of a launch ride behind the NEXT launch's work items instead
        of a launch of their own (srcnn_set_seam_deferral).  The last launch's output is complete only after ``flush()`` or any
        other call on the context."""
        self._check(self._lib.srcnn_set_seam_deferral(self._h, int(bool(on))))

    def flush(self):
        """Queue pending deferred seam work on its stream (srcnn_flush; does not wait)."""
        self._check(self._lib.srcnn_flush(self._h))

    def set_fixup_margin(self, factor: float):
        """SRCNN_MODE_REFBYTES: delta = factor x (noise scale of the model) + absolute term; default 4."""
        self._check(self._lib.srcnn_set_fixup_margin(self._h, float(factor)))

    def set_fixup_local(self, k_local: float):
        """SRCNN_MODE_REFBYTES: the per-pixel flag threshold min(delta, margin * k_local * 2^-24 * S1(x) + abs); 0 = the one
        global threshold of rounds 3-5 (srcnn_set_fixup_local)."""
        self._check(self._lib.srcnn_set_fixup_local(self._h, float(k_local)))

    def fixup_local_stats(self):
        """(k in effect, largest |v_mfma - v_reference| / the pixel's own threshold met so far) -- srcnn_fixup_local_stats."""
        k, r = C.c_float(), C.c_float()
        self._check(self._lib.srcnn_fixup_local_stats(self._h, C.byref(k), C.byref(r)))
        return float(k.value), float(r.value)

    # the two reference calls with the 32-plane map kept in device memory between them (include/srcnn_amd.h)
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._lib.srcnn_dev_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, d_ptr: int):
        self._check(self._lib.srcnn_dev_free(self._h, d_ptr))

    def dev_download(self, dst: np.ndarray, d_src: int):
        if not (isinstance(dst, np.ndarray) and dst.flags.c_contiguous and dst.flags.writeable):
            raise ValueError("dst: expected a writeable C-contiguous numpy array")
        self._check(self._lib.srcnn_dev_download(self._h, dst.ctypes.data_as(C.c_void_p), d_src, dst.nbytes))
        return dst

    def dev_upload(self, d_dst: int, src: np.ndarray):
        src = np.ascontiguousarray(src)
        self._check(self._lib.srcnn_dev_upload(self._h, d_dst, src.ctypes.data_as(C.c_void_p), src.nbytes))

    # device memory shared with the other ranks of the node (srcnn_ipc_*): a 64-byte handle out, a device address in
    def ipc_export(self, d_ptr: int) -> bytes:
        h = (C.c_ubyte * 64)()
        self._check(self._lib.srcnn_ipc_export(self._h, d_ptr, C.byref(h)))
        return bytes(h)

    def ipc_open(self, handle: bytes) -> int:
        if len(handle) != 64:
            raise ValueError("an IPC handle is 64 bytes")
        h = (C.c_ubyte * 64).from_buffer_copy(handle)
        p = C.c_void_p()
        self._check(self._lib.srcnn_ipc_open(self._h, C.byref(h), C.byref(p)))
        return p.value

    def ipc_close(self, d_ptr: int):
        self._check(self._lib.srcnn_ipc_close(self._h, d_ptr))

    def conv99x11_to_dev(self, src, d_planes, plane_stride, plane_pitch, k99, b99, k11, b11):
        src, ss = _plane(src, np.uint8, "src")
        k99, b99 = _wt(k99, 5184, "kernel99"), _wt(b99, 64, "bias99")
        k11, b11 = _wt(k11, 2048, "kernel11"), _wt(b11, 32, "bias11")
        h, w = src.shape
        self._check(self._lib.srcnn_conv99x11_to_dev(self._h, src.ctypes.data_as(_u8p), ss, d_planes, plane_stride, plane_pitch,
                                                     w, h, _fp(k99), _fp(b99), _fp(k11), _fp(b11)))

    def conv55_from_dev(self, d_planes, plane_stride, plane_pitch, dst, kernel, bias):
        dst, ds = _plane(dst, np.uint8, "dst", True)
        k = _wt(kernel, 800, "kernel")
        h, w = dst.shape
        self._check(self._lib.srcnn_conv55_from_dev(self._h, d_planes, plane_stride, plane_pitch, dst.ctypes.data_as(_u8p), ds,
                                                    w, h, _fp(k), float(bias)))

    def forward_y(self, src, dst=None, preclamp=None):
        """Fused Convolution99x11 + Convolution55 (needs set_weights)."""
        src, ss = _plane(src, np.uint8, "src")
        h, w = src.shape
        if dst is None:
            dst = np.empty((h, w), np.uint8)
        dst, ds = _plane(dst, np.uint8, "dst", True)
        _same_shape("dst", dst.shape, src.shape)
        pp, ps = None, 0
        if preclamp is not None:
            preclamp, ps = _plane(preclamp, np.float32, "preclamp", True)
            _same_shape("preclamp", preclamp.shape, src.shape)
            pp = _fp(preclamp)
        self._check(self._lib.srcnn_forward_y(self._h, src.ctypes.data_as(_u8p), ss,
                                              dst.ctypes.data_as(_u8p), ds, w, h, pp, ps))
        return dst

    def forward_y_frames(self, frames, out=None):
        """A stream of equally sized host frames ([n,h,w] uint8), PCIe transfers overlapped with compute."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim != 3 or 0 in frames.shape:
            raise ValueError("frames: expected a non-empty [n, h, w] uint8 array")
        n, h, w = frames.shape
        if out is None:
            out = np.empty_like(frames)
        if not isinstance(out, np.ndarray) or out.dtype != np.uint8:
            raise TypeError("out: expected a uint8 numpy array")
        _same_shape("out", out.shape, frames.shape)
        if not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("out: must be C-contiguous and writeable")
        srcs = (_u8p * n)(*[frames[k].ctypes.data_as(_u8p) for k in range(n)])
        dsts = (_u8p * n)(*[out[k].ctypes.data_as(_u8p) for k in range(n)])
        self._check(self._lib.srcnn_forward_y_frames(self._h, srcs, w, dsts, w, w, h, n))
        return out

    # -- device-resident entry points (integer device addresses) -------------
    def forward_y_dev(self, d_src, src_stride, src_frame_pitch, d_dst, dst_stride, dst_frame_pitch,
                      width, height, n_frames=1, d_preclamp=0):
        self._check(self._lib.srcnn_forward_y_dev(self._h, d_src, src_stride, src_frame_pitch, d_dst,
                                                  dst_stride, dst_frame_pitch, width, height, n_frames,
                                                  d_preclamp or None))

    def forward_y_rows_dev(self, d_src, src_stride, src_row0, d_dst, dst_stride, dst_row0,
                           width, height, row_begin, row_end):
        self._check(self._lib.srcnn_forward_y_rows_dev(self._h, d_src, src_stride, src_row0, d_dst,
                                                       dst_stride, dst_row0, width, height,
                                                       row_begin, row_end))

    def forward_y_rows_halo_dev(self, d_src, src_stride, src_row0, src_rows, d_halo_top, d_halo_bot, halo_stride,
                                d_dst, dst_stride, dst_row0, width, height, row_begin, row_end):
        """A row stripe with its 6 halo rows either side in buffers of their own (0 / None = no rows on that side)."""
        self._check(self._lib.srcnn_forward_y_rows_halo_dev(self._h, d_src, src_stride, src_row0, src_rows,
                                                            d_halo_top or None, d_halo_bot or None, halo_stride, d_dst,
                                                            dst_stride, dst_row0, width, height, row_begin, row_end))

    def halo_transport(self) -> int:
        """0 none yet, 1 same device, 2 peer access (xGMI), 3 staged through the host (srcnn_halo_transport)."""
        return int(self._lib.srcnn_halo_transport(self._h))

    def forward_y_unfused_dev(self, d_src, src_stride, src_frame_pitch, d_dst, dst_stride,
                              dst_frame_pitch, width, height, n_frames, d_work):
        self._check(self._lib.srcnn_forward_y_unfused_dev(self._h, d_src, src_stride, src_frame_pitch,
                                                          d_dst, dst_stride, dst_frame_pitch, width,
                                                          height, n_frames, d_work))

    def conv99x11_dev(self, d_src, src_stride, d_planes, plane_stride, plane_pitch, width, height):
        self._check(self._lib.srcnn_conv99x11_dev(self._h, d_src, src_stride, d_planes, plane_stride,
                                                  plane_pitch, width, height))

    def conv55_dev(self, d_planes, plane_stride, plane_pitch, d_dst, dst_stride, width, height,
                   d_preclamp=0):
        self._check(self._lib.srcnn_conv55_dev(self._h, d_planes, plane_stride, plane_pitch, d_dst,
                                               dst_stride, width, height, d_preclamp or None))


def scaled_size(width: int, height: int, scale: float):
    """(int)(w*scale), (int)(h*scale) -- src/srcnn.cpp:573-575."""
    ow, oh = C.c_int(), C.c_int()
    rc = load_library().srcnn_scaled_size(width, height, float(scale), C.byref(ow), C.byref(oh))
    if rc != 0:
        raise SrcnnError(rc, "scale too small")
    return ow.value, oh.value


def _image(a, name, writable=False):
    if not isinstance(a, np.ndarray) or a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
        raise TypeError(f"{name}: expected an HxWx3 uint8 array (B,G,R)")
    if a.strides[2] != 1 or a.strides[1] != 3 or a.strides[0] < 3 * a.shape[1]:
        raise ValueError(f"{name}: pixels must be packed B,G,R (row stride may be padded)")
    if writable and not a.flags.writeable:
        raise ValueError(f"{name}: output image is read-only")
    return a, a.strides[0]


def _ctx_method(fn):
    setattr(Context, fn.__name__, fn)
    return fn


@_ctx_method
def bgr2ycrcb(self, bgr):
    """cvtColor(CV_BGR2YCrCb) + split (src/srcnn.cpp:509,540) -> (y, cr, cb) planes."""
    bgr, st = _image(bgr, "bgr")
    h, w, _ = bgr.shape
    out = [np.empty((h, w), np.uint8) for _ in range(3)]
    self._check(self._lib.srcnn_bgr2ycrcb(self._h, bgr.ctypes.data_as(_u8p), st, w, h,
                                          *[o.ctypes.data_as(_u8p) for o in out], w))
    return out


@_ctx_method
def ycrcb2bgr(self, y, cr, cb):
    """merge + cvtColor(CV_YCrCb2BGR) (src/srcnn.cpp:639,657) -> HxWx3 B,G,R."""
    planes = [np.ascontiguousarray(p, dtype=np.uint8) for p in (y, cr, cb)]
    h, w = planes[0].shape
    out = np.empty((h, w, 3), np.uint8)
    self._check(self._lib.srcnn_ycrcb2bgr(self._h, *[p.ctypes.data_as(_u8p) for p in planes], w, w, h,
                                          out.ctypes.data_as(_u8p), 3 * w))
    return out


@_ctx_method
def resize_cubic(self, src, dst_w, dst_h):
    """resize(.., CV_INTER_CUBIC) of one 8-bit plane (src/srcnn.cpp:577-582)."""
    src, ss = _plane(src, np.uint8, "src")
    h, w = src.shape
    out = np.empty((dst_h, dst_w), np.uint8)
    self._check(self._lib.srcnn_resize_cubic(self._h, src.ctypes.data_as(_u8p), ss, w, h,
                                             out.ctypes.data_as(_u8p), dst_w, dst_w, dst_h))
    return out


@_ctx_method
def process_bgr(self, bgr, scale):
    """The reference's timed pipeline region (src/srcnn.cpp:505-659) in one call."""
    bgr, st = _image(bgr, "bgr")
    h, w, _ = bgr.shape
    ow, oh = scaled_size(w, h, scale)
    out = np.empty((oh, ow, 3), np.uint8)
    self._check(self._lib.srcnn_process_bgr(self._h, bgr.ctypes.data_as(_u8p), st, w, h, float(scale),
                                            out.ctypes.data_as(_u8p), 3 * ow))
    return out


@_ctx_method
def process_bgr_dev(self, d_bgr, stride, width, height, scale, d_out, out_stride):
    self._check(self._lib.srcnn_process_bgr_dev(self._h, d_bgr, stride, width, height, float(scale),
                                                d_out, out_stride))


def stripe_rows(height: int, n_parts: int, index: int):
    """[begin, end) of part `index` of `n_parts` (srcnn_stripe_rows; equals sharding.split_range)."""
    a, b = C.c_int(), C.c_int()
    rc = load_library().srcnn_stripe_rows(height, n_parts, index, C.byref(a), C.byref(b))
    if rc != 0:
        raise SrcnnError(rc, "bad split")
    return a.value, b.value


def _ctx_array(ctxs):
    if not ctxs:
        raise ValueError("need at least one context")
    return (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])


def forward_y_frames_multi(ctxs: Sequence[Context], frames, out=None):
    """A stream of host frames over several contexts / GPUs (srcnn_forward_y_frames_multi)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    if frames.ndim != 3 or 0 in frames.shape:
        raise ValueError("frames: expected a non-empty [n, h, w] uint8 array")
    n, h, w = frames.shape
    if out is None:
        out = np.empty_like(frames)
    if not isinstance(out, np.ndarray) or out.dtype != np.uint8:
        raise TypeError("out: expected a uint8 numpy array")
    _same_shape("out", out.shape, frames.shape)
    if not out.flags.c_contiguous or not out.flags.writeable:
        raise ValueError("out: must be C-contiguous and writeable")
    srcs = (_u8p * n)(*[frames[k].ctypes.data_as(_u8p) for k in range(n)])
    dsts = (_u8p * n)(*[out[k].ctypes.data_as(_u8p) for k in range(n)])
    ctxs[0]._check_multi(ctxs, load_library().srcnn_forward_y_frames_multi(_ctx_array(ctxs), len(ctxs), srcs, w, dsts, w,
                                                                          w, h, n))
    return out


def forward_y_lanes_dev(ctxs: Sequence[Context], d_src_ptrs, src_stride, d_dst_ptrs, dst_stride, width, height):
    """Device-resident planes of a stream over the contexts used as lanes: plane f on ctxs[f % len(ctxs)], asynchronous
    (srcnn_forward_y_lanes_dev).  Two contexts on one GPU = two lanes of it."""
    n = len(d_src_ptrs)
    if n == 0 or len(d_dst_ptrs) != n:
        raise ValueError("need as many output planes as input planes, at least one")
    srcs = (C.c_void_p * n)(*[int(p) for p in d_src_ptrs])
    dsts = (C.c_void_p * n)(*[int(p) for p in d_dst_ptrs])
    ctxs[0]._check_multi(ctxs, load_library().srcnn_forward_y_lanes_dev(_ctx_array(ctxs), len(ctxs), srcs, src_stride, dsts, dst_stride,
                                                                       width, height, n))


def forward_y_striped(ctxs: Sequence[Context], src, dst=None):
    """ONE host plane row-striped over several contexts / GPUs (srcnn_forward_y_striped)."""
    src, ss = _plane(src, np.uint8, "src")
    h, w = src.shape
    if dst is None:
        dst = np.empty((h, w), np.uint8)
    dst, ds = _plane(dst, np.uint8, "dst", True)
    _same_shape("dst", dst.shape, src.shape)
    ctxs[0]._check_multi(ctxs, load_library().srcnn_forward_y_striped(_ctx_array(ctxs), len(ctxs), src.ctypes.data_as(_u8p),
                                                                     ss, dst.ctypes.data_as(_u8p), ds, w, h))
    return dst


def forward_y_striped_frames(ctxs: Sequence[Context], planes, out=None):
    """A STREAM of host planes [n, h, w], each row-striped over the contexts, pipelined (srcnn_forward_y_striped_frames)."""
    if not isinstance(planes, np.ndarray) or planes.dtype != np.uint8 or planes.ndim != 3 or not planes.flags.c_contiguous:
        raise ValueError("planes: expected a C-contiguous [n, h, w] uint8 array")
    n, h, w = planes.shape
    if out is None:
        out = np.empty_like(planes)
    _same_shape("out", out.shape, planes.shape)
    if not isinstance(out, np.ndarray) or out.dtype != np.uint8 or not out.flags.c_contiguous or not out.flags.writeable:
        raise ValueError("out: must be a C-contiguous writeable uint8 array")
    srcs = (_u8p * n)(*[planes[k].ctypes.data_as(_u8p) for k in range(n)])
    dsts = (_u8p * n)(*[out[k].ctypes.data_as(_u8p) for k in range(n)])
    ctxs[0]._check_multi(ctxs, load_library().srcnn_forward_y_striped_frames(_ctx_array(ctxs), len(ctxs), srcs, w, dsts, w, w, h, n))
    return out


def forward_y_striped_dev(ctxs: Sequence[Context], d_stripes, stripe_stride, d_out, out_stride, width, height):
    """Device-resident striped step: d_stripes[k] / d_out[k] are integer device addresses on ctxs[k]'s GPU."""
    n = len(ctxs)
    ins = (C.c_void_p * n)(*[int(p) for p in d_stripes])
    outs = (C.c_void_p * n)(*[int(p) for p in d_out])
    ctxs[0]._check_multi(ctxs, load_library().srcnn_forward_y_striped_dev(_ctx_array(ctxs), n, ins, stripe_stride, outs,
                                                                         out_stride, width, height))


def _check_multi(self, ctxs, rc):
    if rc != 0:
        msgs = [self._lib.srcnn_last_error(c._h).decode() for c in ctxs]
        raise SrcnnError(rc, " | ".join(m for m in msgs if m != "no error") or "multi-context call failed")


Context._check_multi = _check_multi

_default: Optional[Context] = None


def default_context() -> Context:
    global _default
    if _default is None:
        _default = Context(0)
    return _default


# The reference's free functions (src/srcnn.cpp:60-73), same names and argument order.
def Convolution99(src, dst, kernel, bias):
    default_context().conv99(src, dst, kernel, bias)


def Convolution11(src: Sequence[np.ndarray], dst, kernel, bias):
    default_context().conv11(src, dst, kernel, bias)


def Convolution55(src: Sequence[np.ndarray], dst, kernel, bias):
    default_context().conv55(src, dst, kernel, bias)


def Convolution99x11(src, dst: Sequence[np.ndarray], kernel99, bias99, kernel11, bias11):
    default_context().conv99x11(src, dst, kernel99, bias99, kernel11, bias11)
