#!/usr/bin/env python3
"""Generate the butterfly known-answer fixture (run in the BUILD container only).

The reference repo holds exactly one artefact produced by its own conv path:
Pictures/butterfly-srcnn.png, the README's example output for
`srcnn --scale=1.5 butterfly.png` (README.md:39-45).  This script turns it into
raw vectors so the oracle can be pinned against it without PNG decoding or the
reference tree on the GPU box:

  butterfly_y_in_576.u8    luma of butterfly.png (384x384), OpenCV-style
                           BGR->YCrCb fixed point, then bicubic x1.5 upsample
                           (a=-0.75, half-pixel centres, replicate border, as
                           cv::resize INTER_CUBIC; src/srcnn.cpp:509,577-582).
                           This is the INPUT of the conv path.
  butterfly_y_ref_576.u8   luma recomputed from butterfly-srcnn.png = the
                           reference's OUTPUT of the conv path, up to the
                           YCrCb->BGR->Y round trip and OpenCV's fixed-point
                           resize (neither is in /root/reference: un-pinned
                           third-party arithmetic), hence a PSNR pin, not bits.

Both are 576*576 bytes, row-major.

  butterfly_bgr.npz        src_bgr [384,384,3] = butterfly.png, ref_bgr [576,576,3] =
                           butterfly-srcnn.png, both as cv::imread would hand them
                           over (B,G,R byte order).  Input and expected output of
                           the WHOLE pipeline region src/srcnn.cpp:505-659 at
                           --scale=1.5: the strongest pin the reference offers.

Only data is committed; no reference source.
"""
import sys
from pathlib import Path
import numpy as np
from PIL import Image

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent


def luma_cv(rgb):
    """OpenCV 8-bit BGR2YCrCb luma: (R*4899 + G*9617 + B*1868 + 2^13) >> 14."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)


def cubic_w(t, a=-0.75):
    t = abs(t)
    if t <= 1:
        return (a + 2) * t**3 - (a + 3) * t**2 + 1
    if t < 2:
        return a * t**3 - 5 * a * t**2 + 8 * a * t - 4 * a
    return 0.0


def resize_axis(img, n_out, scale, axis):
    img = np.moveaxis(img.astype(np.float64), axis, 0)
    n_in = img.shape[0]
    out = np.zeros((n_out,) + img.shape[1:])
    for o in range(n_out):
        fx = (o + 0.5) / scale - 0.5
        s = int(np.floor(fx))
        d = fx - s
        for k in range(-1, 3):
            idx = min(max(s + k, 0), n_in - 1)
            out[o] += cubic_w(k - d) * img[idx]
    return np.moveaxis(out, 0, axis)


def bicubic_cv(y, scale):
    h, w = y.shape
    oh, ow = int(h * scale), int(w * scale)      # src/srcnn.cpp:573-575
    t = resize_axis(y, ow, ow / w, 1)
    t = resize_axis(t, oh, oh / h, 0)
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


def main():
    src = np.asarray(Image.open(REF / "Pictures/butterfly.png").convert("RGB"))
    ref = np.asarray(Image.open(REF / "Pictures/butterfly-srcnn.png").convert("RGB"))
    y_in = bicubic_cv(luma_cv(src), 1.5)
    y_ref = luma_cv(ref)
    assert y_in.shape == y_ref.shape == (576, 576)
    np.savez_compressed(OUT / "butterfly_bgr.npz", src_bgr=src[:, :, ::-1].copy(), ref_bgr=ref[:, :, ::-1].copy())
    y_in.tofile(OUT / "butterfly_y_in_576.u8")
    y_ref.tofile(OUT / "butterfly_y_ref_576.u8")
    print("wrote", y_in.shape, y_ref.shape)


if __name__ == "__main__":
    main()
