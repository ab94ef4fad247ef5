/*
 * srcnn_gpuorder.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A CPU model of the ARITHMETIC ORDER the HIP kernels in
 * srcnn_cpp_amd/csrc/ use, so that GPU results can be regression-checked
 * BITWISE (any indexing / layout bug shows up as a hard mismatch instead of
 * hiding inside a floating-point tolerance).  It is NOT the reference's
 * arithmetic: the parity claim against the reference is always made against
 * srcnn_oracle.c (strict multiply-then-add) within the stated tolerance.
 *
 * Differences from the reference order (src/srcnn.cpp:288-321, :218-240):
 *   layers 1+2  same summation order (layer 1: taps i-major/j-minor, then bias;
 *               layer 2: the chain STARTS from the bias -- it is the MFMA's
 *               initial accumulator -- then input channels ascending), and
 *               every multiply-add is ONE fused multiply-add:
 *               v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain.
 *               The layer-1 bias enters as a 82nd "tap" (x = 1.0), i.e.
 *               fmaf(b, 1, t) == t + b rounded once.  (The kernels scale both
 *               layers by exact powers of two; that changes no bit.)
 *   layer 3     the 5x5x32 contraction is split as  T[tap] = sum_c W3[c][tap]*F[c]
 *               (fmaf chain, channels ascending, float) per FEATURE pixel,
 *               followed by a float shifted sum of the 25 tap planes, tap
 *               rows first: F_n = (((T[n]+T[5+n])+T[10+n])+T[15+n])+T[20+n]
 *               over the feature rows y-2..y+2 at column x+n-2, then
 *               out = ((((F_0+F_1)+F_2)+F_3)+F_4) + b3.
 *               (The reference sums 25 taps in double per channel and then
 *               the 32 channels in float.)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int srcnn_gpuorder_conv99x11(const uint8_t *src, size_t sstride, float *const *dst, size_t dstride,
                             int width, int height,
                             const float *kernel99, const float *bias99,
                             const float *kernel11, const float *bias11)
{
#pragma omp parallel for
    for (int row = 0; row < height; row++) {
        float px[81], t[64];
        for (int col = 0; col < width; col++) {
            for (int i = 0; i < 9; i++)
                for (int j = 0; j < 9; j++)
                    px[i * 9 + j] = (float)src[(size_t)clampi(row + i - 4, 0, height - 1) * sstride +
                                               clampi(col + j - 4, 0, width - 1)];
            for (int k = 0; k < 64; k++) {
                float a = 0.f;
                for (int q = 0; q < 81; q++) a = fmaf(kernel99[k * 81 + q], px[q], a);
                a = fmaf(bias99[k], 1.0f, a);
                t[k] = a < 0 ? 0 : a;
            }
            for (int k = 0; k < 32; k++) {
                float r = bias11[k];                  /* the MFMA chain starts from the bias */
                for (int i = 0; i < 64; i++) r = fmaf(kernel11[k * 64 + i], t[i], r);
                dst[k][(size_t)row * dstride + col] = r < 0 ? 0 : r;
            }
        }
    }
    return 0;
}

int srcnn_gpuorder_conv55(const float *const *src, size_t sstride, uint8_t *dst, size_t dstride,
                          int width, int height, const float *kernel, float bias, float *preclamp)
{
    /* tap planes T[tap][y][x] */
    size_t plane = (size_t)width * height;
    float *T = (float *)malloc(sizeof(float) * plane * 25);
    if (!T) return -1;
#pragma omp parallel for
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            for (int tap = 0; tap < 25; tap++) {
                float a = 0.f;
                for (int c = 0; c < 32; c++)
                    a = fmaf(kernel[c * 25 + tap], src[c][(size_t)y * sstride + x], a);
                T[tap * plane + (size_t)y * width + x] = a;
            }
#pragma omp parallel for
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            /* F_n = vertical sum over the 5 tap rows (m ascending) at column clamp(x+n-2),
             * then the 5-term horizontal sum (n ascending), then the bias */
            float acc = 0.f;
            for (int n = 0; n < 5; n++) {
                int fx = clampi(x + n - 2, 0, width - 1);
                float fn = 0.f;
                for (int m = 0; m < 5; m++) {
                    int fy = clampi(y + m - 2, 0, height - 1);
                    float v = T[(m * 5 + n) * plane + (size_t)fy * width + fx];
                    fn = (m == 0) ? v : fn + v;
                }
                acc = (n == 0) ? fn : acc + fn;
            }
            float v = acc + bias;
            if (preclamp) preclamp[(size_t)y * dstride + x] = v;
            int q = (int)v;
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            dst[(size_t)y * dstride + x] = (uint8_t)q;
        }
    free(T);
    return 0;
}

int srcnn_gpuorder_forward_y(const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                             int width, int height, const float *weights, float *preclamp)
{
    const float *b1 = weights, *w1 = b1 + 64, *b2 = w1 + 64 * 81, *w2 = b2 + 32,
                *b3 = w2 + 32 * 64, *w3 = b3 + 1;
    size_t plane = (size_t)width * height;
    float *buf = (float *)malloc(sizeof(float) * plane * 32);
    if (!buf) return -1;
    float *planes[32];
    for (int k = 0; k < 32; k++) planes[k] = buf + plane * k;
    int rc = srcnn_gpuorder_conv99x11(src, sstride, planes, (size_t)width, width, height, w1, b1, w2, b2);
    if (rc == 0)
        rc = srcnn_gpuorder_conv55((const float *const *)planes, (size_t)width, dst, dstride,
                                   width, height, w3, *b3, preclamp);
    free(buf);
    return rc;
}
