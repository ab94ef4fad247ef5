"""Plain PyTorch restatement of the path as three conv2d calls -- an implementation that shares
nothing with oracle/ or the HIP kernels (different loops, different summation order, library
convolution) -- used two ways:

* CPU: the oracle (reference arithmetic, src/srcnn.cpp:254-325 + :189-243) against float64
  conv2d.  Confirms the restated semantics: cross-correlation orientation of the [out][kh][kw]
  tables (src/convdata.h), replicate padding of EACH layer's own input (:266-280, :196-210),
  ReLU between layers, truncation toward zero + clamp at the end (:238-240).
* GPU: the HIP path (float32 MFMA mode and the opt-in split-f16 mode) against float32 conv2d,
  with the tolerance stated in tests/test_gpu_parity.py (pre-clamp |d| <= 5e-3).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

TOL_PRE_ABS = 5e-3


def torch_forward(y, blob, dtype, device="cpu"):
    w1, b1, w2, b2, w3, b3 = S.split_weights(blob)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)
    x = t(y.astype(np.float64))[None, None]
    x = F.relu(F.conv2d(F.pad(x, (4, 4, 4, 4), mode="replicate"), t(w1)[:, None], t(b1)))
    x = F.relu(F.conv2d(x, t(w2)[:, :, None, None], t(b2)))
    x = F.conv2d(F.pad(x, (2, 2, 2, 2), mode="replicate"), t(w3)[None], t(np.array([b3])))
    return x[0, 0].to(torch.float64).cpu().numpy()


def u8_of(pre):
    return np.clip(np.trunc(pre), 0, 255).astype(np.uint8)


def assert_u8_consistent(out, ref_pre):
    """`out` may differ from trunc(ref_pre) by 1 LSB, and only where ref_pre is next to an integer."""
    d = np.abs(out.astype(int) - u8_of(ref_pre).astype(int))
    assert d.max() <= 1
    if d.any():
        assert np.abs(ref_pre - np.rint(ref_pre))[d != 0].max() <= TOL_PRE_ABS


@pytest.mark.parametrize("w,h,frame", [(97, 61, 0), (200, 33, 3), (9, 5, 1), (1, 1, 0), (130, 140, 7)])
def test_oracle_matches_float64_conv2d(weights_blob, w, h, frame):
    y = synth_luma(w, h, frame=frame)
    r_out, r_pre = oracle.forward_y(y, weights_blob)
    t_pre = torch_forward(y, weights_blob, torch.float64)
    # the oracle accumulates in float32 (as the reference does): ~1e-4 of rounding noise on 0..255
    assert np.abs(r_pre - t_pre).max() <= 1e-3
    assert_u8_consistent(r_out, t_pre)


def test_oracle_32_channel_map_matches_conv2d(weights_blob):
    w1, b1, w2, b2, _, _ = S.split_weights(weights_blob)
    y = synth_luma(120, 50, frame=2)
    planes = oracle.conv99x11(y, w1, b1, w2, b2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.float64)
    x = t(y.astype(np.float64))[None, None]
    x = F.relu(F.conv2d(F.pad(x, (4, 4, 4, 4), mode="replicate"), t(w1)[:, None], t(b1)))
    x = F.relu(F.conv2d(x, t(w2)[:, :, None, None], t(b2)))[0].numpy()
    got = np.stack([np.asarray(p) for p in planes])
    assert np.abs(got - x).max() <= 1e-3 * max(1.0, np.abs(x).max())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["mfma", "split16"])
@pytest.mark.parametrize("w,h", [(300, 70), (1920, 1080)])
def test_hip_path_matches_float32_conv2d(gpu_ctx, weights_blob, mode, w, h):
    y = synth_luma(w, h, frame=5)
    gpu_ctx.set_mode(S.MODE_SPLIT16 if mode == "split16" else S.MODE_MFMA)
    try:
        pre = np.empty((h, w), np.float32)
        out = gpu_ctx.forward_y(y, preclamp=pre)
    finally:
        gpu_ctx.set_mode(S.MODE_MFMA)
    t_pre = torch_forward(y, weights_blob, torch.float32, device="cuda")
    assert np.abs(pre - t_pre).max() <= TOL_PRE_ABS
    assert_u8_consistent(out, torch_forward(y, weights_blob, torch.float64))
