/*
 * opencv_steps.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the three OpenCV calls that sit either side of the conv
 * path in the reference's pipeline driver:
 *
 *   cvtColor(.., CV_BGR2YCrCb)      src/srcnn.cpp:509
 *   resize(.., CV_INTER_CUBIC)      src/srcnn.cpp:577-582   (x3 planes, size (int)(w*s) x (int)(h*s), :573-575)
 *   cvtColor(.., CV_YCrCb2BGR)      src/srcnn.cpp:657
 *
 * PARITY PIN STATUS: UNPINNED third-party arithmetic.  The algorithm lives in
 * OpenCV 4 (un-pinned: `pkg-config opencv4`, reference Makefile:8-9), which is
 * neither under /root/reference nor installed here, and no reference test
 * holds vectors for it.  What follows restates OpenCV 4.x's published 8-bit
 * algorithms:
 *   - colour: fixed point, yuv_shift = 14, coefficients
 *       Y  = (4899 R + 9617 G + 1868 B + 2^13) >> 14
 *       Cr = ((R - Y) * 11682 + (128 << 14) + 2^13) >> 14,  Cb = ((B - Y) * 9241 + ...) >> 14
 *       B  = Y + ((Cb-128)*29049 + 2^13 >> 14), G = Y + (((Cb-128)*-5636 + (Cr-128)*-11698 + 2^13) >> 14),
 *       R  = Y + ((Cr-128)*22987 + 2^13 >> 14), all saturated to 0..255
 *     (modules/imgproc/src/color_yuv: RGB2YCrCb_i<uchar>, YCrCb2RGB_i<uchar>);
 *   - resize: half-pixel centres, Keys cubic A = -0.75 evaluated in float,
 *     coefficients rounded to 11-bit fixed point (INTER_RESIZE_COEF_BITS),
 *     replicate border, horizontal pass in int, vertical pass in int,
 *     (sum + 2^21) >> 22, saturate -- the scalar HResizeCubic / VResizeCubic /
 *     FixedPtCast path of modules/imgproc/src/resize.cpp.  (OpenCV's SIMD builds
 *     run the vertical pass in float; the two can differ by 1 LSB on rare pixels.)
 * The only anchor is the reference's example picture: the whole pipeline lands on
 * Pictures/butterfly-srcnn.png at >= 45 dB (tests/test_pipeline_oracle.py).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define YUV_SHIFT 14
#define DESCALE(x) (((x) + (1 << (YUV_SHIFT - 1))) >> YUV_SHIFT)

static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* interleaved BGR -> three planes */
int opencv_bgr2ycrcb(const uint8_t *bgr, size_t stride, int w, int h,
                     uint8_t *y, uint8_t *cr, uint8_t *cb, size_t pstride)
{
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            const uint8_t *p = bgr + (size_t)r * stride + 3 * (size_t)c;
            const int B = p[0], G = p[1], R = p[2];
            const int Y = DESCALE(B * 1868 + G * 9617 + R * 4899);
            const int Cr = DESCALE((R - Y) * 11682 + (128 << YUV_SHIFT));
            const int Cb = DESCALE((B - Y) * 9241 + (128 << YUV_SHIFT));
            y[(size_t)r * pstride + c] = sat_u8(Y);
            cr[(size_t)r * pstride + c] = sat_u8(Cr);
            cb[(size_t)r * pstride + c] = sat_u8(Cb);
        }
    return 0;
}

int opencv_ycrcb2bgr(const uint8_t *y, const uint8_t *cr, const uint8_t *cb, size_t pstride,
                     int w, int h, uint8_t *bgr, size_t stride)
{
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            const int Y = y[(size_t)r * pstride + c];
            const int Cr = cr[(size_t)r * pstride + c] - 128, Cb = cb[(size_t)r * pstride + c] - 128;
            uint8_t *p = bgr + (size_t)r * stride + 3 * (size_t)c;
            p[0] = sat_u8(Y + DESCALE(Cb * 29049));
            p[1] = sat_u8(Y + DESCALE(Cb * -5636 + Cr * -11698));
            p[2] = sat_u8(Y + DESCALE(Cr * 22987));
        }
    return 0;
}

/* Coefficient table of one axis: ofs[d] = floor(source coordinate), coef[d][4]
 * = Keys cubic (A = -0.75) at the fractional part, 11-bit fixed point. */
int opencv_cubic_table(int n_src, int n_dst, int *ofs, int16_t *coef)
{
    if (n_src <= 0 || n_dst <= 0) return -1;
    const double scale = 1.0 / ((double)n_dst / n_src);
    const float A = -0.75f;
    for (int d = 0; d < n_dst; d++) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        const int sx = (int)floorf(fx);
        fx -= sx;
        float cf[4];
        cf[0] = ((A * (fx + 1) - 5 * A) * (fx + 1) + 8 * A) * (fx + 1) - 4 * A;
        cf[1] = ((A + 2) * fx - (A + 3)) * fx * fx + 1;
        cf[2] = ((A + 2) * (1 - fx) - (A + 3)) * (1 - fx) * (1 - fx) + 1;
        cf[3] = 1.f - cf[0] - cf[1] - cf[2];
        ofs[d] = sx;
        for (int k = 0; k < 4; k++) {
            long q = lrintf(cf[k] * 2048.f);           /* saturate_cast<short>(x) == cvRound + clamp */
            coef[4 * d + k] = (int16_t)(q < -32768 ? -32768 : (q > 32767 ? 32767 : q));
        }
    }
    return 0;
}

int opencv_resize_cubic(const uint8_t *src, size_t sstride, int sw, int sh,
                        uint8_t *dst, size_t dstride, int dw, int dh)
{
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw), *yofs = (int *)malloc(sizeof(int) * (size_t)dh);
    int16_t *alpha = (int16_t *)malloc(8 * (size_t)dw), *beta = (int16_t *)malloc(8 * (size_t)dh);
    int rc = -1;
    if (xofs && yofs && alpha && beta && opencv_cubic_table(sw, dw, xofs, alpha) == 0 &&
        opencv_cubic_table(sh, dh, yofs, beta) == 0) {
#pragma omp parallel for
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++) {
                int acc = 0;
                for (int ky = 0; ky < 4; ky++) {
                    int sy = yofs[dy] - 1 + ky;
                    sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
                    int row = 0;
                    for (int kx = 0; kx < 4; kx++) {
                        int sx = xofs[dx] - 1 + kx;
                        sx = sx < 0 ? 0 : (sx >= sw ? sw - 1 : sx);
                        row += src[(size_t)sy * sstride + sx] * alpha[4 * dx + kx];
                    }
                    acc += row * beta[4 * dy + ky];
                }
                dst[(size_t)dy * dstride + dx] = sat_u8((acc + (1 << 21)) >> 22);
            }
        rc = 0;
    }
    free(xofs); free(yofs); free(alpha); free(beta);
    return rc;
}

/* (int)(n * scale) as `newsz.width *= image_multiply` does, src/srcnn.cpp:573-575 */
int opencv_scaled_dim(int n, float scale) { return (int)((float)n * scale); }
