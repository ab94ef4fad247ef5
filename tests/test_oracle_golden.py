"""CPU tests of the oracle (oracle/srcnn_oracle.c) -- the restatement of
src/srcnn.cpp:77-325 that every GPU parity test is judged against.

Pins available for this path: the reference has NO test
vectors; its conv path cannot be compiled here (OpenCV absent, stand-in headers
not allowed).  The only reference-produced artefact is
Pictures/butterfly-srcnn.png -> tests/golden/butterfly_y_*.u8 (made by
tests/golden/make_butterfly_fixture.py).  So: one loose known-answer pin
(PSNR), plus structural checks of the restatement against itself.
"""
import hashlib
from pathlib import Path

import numpy as np
import pytest

import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

GOLD = Path(__file__).resolve().parent / "golden"


def psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return 10 * np.log10(255.0 ** 2 / np.mean(d * d))


def test_weight_blob_provenance(weights_blob):
    # sha256 of the dump of src/convdata.h made by oracle/dump_weights.c
    h = hashlib.sha256(weights_blob.astype("<f4").tobytes()).hexdigest()
    assert h == "822a078c4a6c04a499a98e1e6dc63ba541b39b83a0f27eb681e0cce42fcfc699"
    w1, b1, w2, b2, w3, b3 = S.split_weights(weights_blob)
    assert w1.shape == (64, 9, 9) and w2.shape == (32, 64) and w3.shape == (32, 5, 5)
    assert abs(b3 - 12.8460) < 1e-6                       # src/convdata.h:979
    assert abs(b1[8] - 177.2564) < 1e-4                   # src/convdata.h:22
    assert abs(w1[0, 0, 0] - (-0.0887)) < 1e-6            # src/convdata.h:35


def test_butterfly_known_answer(weights_blob):
    """Oracle output on the README example vs the reference's own output image
    (README.md:39-45).  Residual = OpenCV's fixed-point resize / colour round
    trip, which are outside /root/reference."""
    y_in = np.fromfile(GOLD / "butterfly_y_in_576.u8", np.uint8).reshape(576, 576)
    y_ref = np.fromfile(GOLD / "butterfly_y_ref_576.u8", np.uint8).reshape(576, 576)
    out, _ = oracle.forward_y(y_in, weights_blob)
    assert psnr(y_in, y_ref) < 34.0           # bicubic alone is far away ...
    assert psnr(out, y_ref) >= 50.0           # ... the SRCNN path lands on the reference's picture
    assert np.abs(out.astype(int) - y_ref.astype(int)).max() <= 8


def test_inttrim_semantics(weights_blob):
    """Final store truncates toward zero and clamps (src/srcnn.cpp:77-81, :238-240)."""
    planes = np.zeros((32, 4, 6), np.float32)
    k = np.zeros((32, 5, 5), np.float32)
    for bias, want in [(-3.7, 0), (0.0, 0), (0.99, 0), (1.0, 1), (254.999, 254), (255.0, 255),
                       (255.5, 255), (300.0, 255), (17.9999, 17)]:
        out, pre = oracle.conv55(planes, k, bias)
        assert (out == want).all(), (bias, out[0, 0])
        assert (pre == np.float32(bias)).all()


@pytest.mark.parametrize("w,h", [(1, 1), (3, 3), (9, 5), (5, 9), (17, 4), (97, 61)])
def test_fused_equals_unfused(weights_blob, w, h):
    """G5: Convolution99x11 == 64 x Convolution99 then 32 x Convolution11, bitwise
    (src/srcnn.cpp:124-137 == :293-304, :166-175 == :312-321)."""
    w1, b1, w2, b2, w3, b3 = S.split_weights(weights_blob)
    y = synth_luma(w, h, frame=3)
    fused = oracle.conv99x11(y, w1, b1, w2, b2)
    l1 = np.stack([oracle.conv99(y, w1[k], b1[k]) for k in range(64)])
    l2 = np.stack([oracle.conv11(l1, w2[k], b2[k]) for k in range(32)])
    assert np.array_equal(fused, l2)
    assert (fused >= 0).all()


def test_replicate_border_is_per_layer(weights_blob):
    """Layer 3 clamps FEATURE-map coordinates (src/srcnn.cpp:196-210): on a
    constant image every feature pixel is identical, so every output pixel is."""
    for v in (0, 128, 255):
        y = np.full((11, 14), v, np.uint8)
        out, pre = oracle.forward_y(y, weights_blob)
        assert (out == out[0, 0]).all() and (pre == pre[0, 0]).all()


def test_crop_locality(weights_blob):
    """13x13 receptive field: an interior crop with a 6-pixel margin reproduces
    the full-image result exactly."""
    y = synth_luma(80, 50, frame=1)
    full, _ = oracle.forward_y(y, weights_blob)
    crop, _ = oracle.forward_y(y[10:40, 20:70], weights_blob)
    assert np.array_equal(crop[6:-6, 6:-6], full[16:34, 26:64])


def test_gpuorder_model_within_tolerance(weights_blob):
    """The FMA-order model of the HIP kernels stays inside the tolerance that
    the GPU parity tests state against the oracle."""
    y = synth_luma(97, 61)
    out, pre = oracle.forward_y(y, weights_blob)
    out2, pre2 = oracle.gpuorder_forward_y(y, weights_blob)
    assert np.abs(pre - pre2).max() <= 2e-3
    d = np.abs(out.astype(int) - out2.astype(int))
    assert d.max() <= 1 and (d != 0).mean() <= 1e-3


def test_oracle_regression_vectors(weights_blob):
    """Self-pins (NOT reference pins): checksums of the oracle on seeded inputs,
    so that an accidental edit of the restatement is noticed."""
    sums = {}
    for (w, h, f) in [(97, 61, 0), (9, 5, 1), (1, 1, 2), (130, 33, 3)]:
        out, pre = oracle.forward_y(synth_luma(w, h, f), weights_blob)
        sums[(w, h, f)] = (int(out.astype(np.int64).sum()), hashlib.sha256(out.tobytes()).hexdigest()[:16])
    want = eval((GOLD / "oracle_selfpins.txt").read_text())
    assert sums == want


def test_synthetic_generator_is_pinned():
    """The integer-only generator must give identical bytes on every host (bench input, 4K pins)."""
    import json
    pins = json.loads((GOLD / "synthetic_4k_checksums.json").read_text())
    assert hashlib.sha256(synth_luma(3840, 2160).tobytes()).hexdigest() == pins["input_sha256"]
    a = synth_luma(64, 48, frame=3)
    assert a.dtype == np.uint8 and a.max() <= 241 and np.array_equal(a, synth_luma(64, 48, frame=3))
    assert not np.array_equal(a, synth_luma(64, 48, frame=4))


def _edge_pins():
    z = np.load(GOLD / "oracle_edge_pins.npz")
    return [(k[:-3], z[k], z[k[:-3] + ".out"], z[k[:-3] + ".pre"]) for k in z.files if k.endswith(".in")]


@pytest.mark.parametrize("name", [p[0] for p in _edge_pins()])
def test_edge_pins_hold_for_the_c_oracle_and_the_numpy_restatement(weights_blob, name):
    """The G2 edge set as committed input / output vectors (tests/golden/make_oracle_edge_pins.py): planes smaller
    than the 9 x 9 and 5 x 5 windows, constant 0 / 255.  The reference's picture pins the oracle's interior arithmetic; these pin
    its borders and saturating ends -- against the C oracle as it is NOW, and against the independent numpy float32 restatement
    (tests/test_oracle_numpy.py), so an edit of oracle/srcnn_oracle.c that still reproduces the picture but breaks W < 9 is
    caught without a GPU."""
    from test_oracle_numpy import np_conv11, np_conv55, np_conv99
    y, want_u8, want_pre = next((a, b, c) for n, a, b, c in _edge_pins() if n == name)
    got_u8, got_pre = oracle.forward_y(y, weights_blob)
    assert np.array_equal(got_u8, want_u8) and np.array_equal(got_pre, want_pre)
    w1, b1, w2, b2, w3, b3 = S.split_weights(weights_blob)
    l1 = [np_conv99(y, w1[k], b1[k]) for k in range(64)]
    l2 = [np_conv11(l1, w2[k], b2[k]) for k in range(32)]
    n_u8, n_pre = np_conv55(l2, w3, b3)
    assert np.array_equal(n_u8, want_u8) and np.array_equal(n_pre, want_pre)
    if name.startswith("const255"):
        assert want_u8.min() >= 250          # the saturating end really is exercised
