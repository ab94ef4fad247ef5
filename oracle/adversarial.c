/*
 * adversarial.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Attacks the flag threshold of SRCNN_MODE_REFBYTES (srcnn_cpp_amd/csrc/srcnn_api.cpp, fixup_delta()).  That mode is
 * bit-identical to the reference only while |v_gpu - v_ref| <= delta for every pixel, where v_ref is the value the
 * reference truncates at src/srcnn.cpp:238-240 and v_gpu the float32 MFMA kernel's value for the same pixel.  delta was
 * chosen from statistics over natural and synthetic content; this file SEARCHES for luma windows that maximise the
 * difference instead of sampling it.
 *
 * One output pixel depends on a 13 x 13 luma window (4 + 2 pixels either side).  srcnn_adv_point() evaluates that one
 * pixel in both arithmetics:
 *   v_ref  -- the reference's: rounded multiply then rounded add in the loop order of Convolution99x11
 *             (src/srcnn.cpp:288-321), then Convolution55's float product / double 25-term sum per channel / float
 *             running sum over channels / + bias (:218-235).  Same statements as oracle/srcnn_oracle.c, for one pixel.
 *   v_gpu  -- the HIP kernels' order, as in oracle/srcnn_gpuorder.c: fused multiply-add chains, layer-2 chain started
 *             from its bias, layer 3 as tap partials per feature pixel, tap rows first, then tap columns, then b3.
 * Both are checked bit for bit against the whole-plane functions of those two files (tests/test_adversarial.py), so
 * a deviation found here is a deviation the GPU would show.
 *
 * srcnn_adv_search() is plain randomised coordinate ascent on |v_gpu - v_ref| over the 169 bytes of a window, from
 * caller-supplied starts (random, natural, extreme), one restart per start, restarts in parallel (OpenMP).
 *
 * Must be compiled with -ffp-contract=off (oracle/Makefile): the reference arithmetic must not be contracted; the
 * GPU-order arithmetic asks for its fusions explicitly (fmaf).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define WIN 13

/* weights: b1[64] | W1[64][81] | b2[32] | W2[32][64] | b3 | W3[32][25]  (convdata.h declaration order) */
typedef struct {
    float w1t[81][64];   /* tap-major copies: the channel loop is the inner, vectorisable one -- every CHAIN keeps its order */
    float w2t[64][32];
    float w3t[25][32];
    float b1[64], b2[32], b3;
} AdvModel;

static void adv_model(const float *weights, AdvModel *m)
{
    const float *b1 = weights, *w1 = b1 + 64, *b2 = w1 + 64 * 81, *w2 = b2 + 32, *b3 = w2 + 32 * 64, *w3 = b3 + 1;
    for (int k = 0; k < 64; k++)
        for (int t = 0; t < 81; t++) m->w1t[t][k] = w1[k * 81 + t];
    for (int k = 0; k < 32; k++)
        for (int i = 0; i < 64; i++) m->w2t[i][k] = w2[k * 64 + i];
    for (int c = 0; c < 32; c++)
        for (int t = 0; t < 25; t++) m->w3t[t][c] = w3[c * 25 + t];
    memcpy(m->b1, b1, sizeof m->b1);
    memcpy(m->b2, b2, sizeof m->b2);
    m->b3 = *b3;
}

/* The centre pixel (6, 6) of a 13 x 13 window in both arithmetics.  No border is involved: the 25 feature positions
 * (4..8, 4..8) read luma rows / columns 0..12. */
static void adv_eval_scale2(const AdvModel *m, const uint8_t *win, float *v_ref, float *v_gpu, float *scale, float *local)
{
    float Fr[25][32], Fg[25][32];
    for (int p = 0; p < 25; p++) {
        const int fy = 4 + p / 5, fx = 4 + p % 5;
        float ar[64], ag[64];
        for (int k = 0; k < 64; k++) ar[k] = ag[k] = 0.f;
        for (int i = 0; i < 9; i++)
            for (int j = 0; j < 9; j++) {
                const float y = (float)win[(fy + i - 4) * WIN + (fx + j - 4)];
                const float *w = m->w1t[i * 9 + j];
                for (int k = 0; k < 64; k++) {
                    const float pr = w[k] * y;              /* src/srcnn.cpp:297 -- rounded product, rounded add */
                    ar[k] = ar[k] + pr;
                    ag[k] = fmaf(w[k], y, ag[k]);           /* v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain */
                }
            }
        for (int k = 0; k < 64; k++) {
            ar[k] = ar[k] + m->b1[k];                       /* :301 */
            ar[k] = ar[k] < 0 ? 0 : ar[k];                  /* :304 */
            ag[k] = fmaf(m->b1[k], 1.0f, ag[k]);            /* the bias is the 82nd tap */
            ag[k] = ag[k] < 0 ? 0 : ag[k];
        }
        float rr[32], rg[32];
        for (int k = 0; k < 32; k++) { rr[k] = 0.f; rg[k] = m->b2[k]; }      /* the MFMA chain starts from the bias */
        for (int i = 0; i < 64; i++) {
            const float *w = m->w2t[i];
            for (int k = 0; k < 32; k++) {
                const float pr = ar[i] * w[k];              /* :314 */
                rr[k] = rr[k] + pr;
                rg[k] = fmaf(w[k], ag[i], rg[k]);
            }
        }
        for (int k = 0; k < 32; k++) {
            rr[k] = rr[k] + m->b2[k];                       /* :316 */
            Fr[p][k] = rr[k] < 0 ? 0 : rr[k];               /* :319 */
            Fg[p][k] = rg[k] < 0 ? 0 : rg[k];
        }
    }
    /* reference layer 3, :218-235 */
    double tp[32];
    for (int c = 0; c < 32; c++) tp[c] = 0.0;
    for (int t = 0; t < 25; t++)
        for (int c = 0; c < 32; c++) {
            const float pr = m->w3t[t][c] * Fr[t][c];       /* :227-228 */
            tp[c] += pr;
        }
    float temp = 0;
    for (int c = 0; c < 32; c++) temp += tp[c];             /* :232 */
    temp += m->b3;                                          /* :235 */
    *v_ref = temp;
    /* GPU order: tap partials per feature pixel (channels ascending), tap rows, tap columns, bias */
    float acc = 0.f;
    for (int n = 0; n < 5; n++) {
        float fn = 0.f;
        for (int mm = 0; mm < 5; mm++) {
            const int t = mm * 5 + n;
            float a = 0.f;
            for (int c = 0; c < 32; c++) a = fmaf(m->w3t[t][c], Fg[t][c], a);
            fn = (mm == 0) ? a : fn + a;
        }
        acc = (n == 0) ? fn : acc + fn;
    }
    *v_gpu = acc + m->b3;
    /* the magnitude rounding errors of layer 3 scale with: the sum of its 800 |products| */
    if (scale) {
        float sc = 0.f;
        for (int t = 0; t < 25; t++)
            for (int c = 0; c < 32; c++) sc += fabsf(m->w3t[t][c] * Fr[t][c]);
        *scale = sc;
    }
    /* the LOCAL scale SRCNN_MODE_REFBYTES' per-pixel threshold is proportional to (round 6): S1 = the sum over the pixel's
     * 5 x 5 feature window of U = sum_c a_c * F_c with a_c = max_tap |W3[c][tap]|, on the kernels' own layer-2 map.  (The
     * kernels sum it in another order -- one more layer-3 accumulator row, then the vertical chains and the horizontal
     * 5-term sum; S1 only scales a threshold, its last bits decide nothing.) */
    if (local) {
        float amax[32], s1 = 0.f;
        for (int c = 0; c < 32; c++) {
            amax[c] = 0.f;
            for (int t = 0; t < 25; t++) amax[c] = fmaxf(amax[c], fabsf(m->w3t[t][c]));
        }
        for (int t = 0; t < 25; t++)
            for (int c = 0; c < 32; c++) s1 += amax[c] * Fg[t][c];
        *local = s1;
    }
}
static void adv_eval_scale(const AdvModel *m, const uint8_t *win, float *v_ref, float *v_gpu, float *scale)
{
    adv_eval_scale2(m, win, v_ref, v_gpu, scale, NULL);
}
static void adv_eval(const AdvModel *m, const uint8_t *win, float *v_ref, float *v_gpu) { adv_eval_scale(m, win, v_ref, v_gpu, NULL); }

int srcnn_adv_point(const uint8_t *win /*[13][13]*/, const float *weights, float *v_ref, float *v_gpu)
{
    AdvModel m;
    if (!win || !weights || !v_ref || !v_gpu) return -1;
    adv_model(weights, &m);
    adv_eval(&m, win, v_ref, v_gpu);
    return 0;
}

/* ... and the local scale S1 of that pixel (see adv_eval_scale2) */
int srcnn_adv_point_local(const uint8_t *win /*[13][13]*/, const float *weights, float *v_ref, float *v_gpu, float *s1)
{
    AdvModel m;
    if (!win || !weights || !v_ref || !v_gpu || !s1) return -1;
    adv_model(weights, &m);
    adv_eval_scale2(&m, win, v_ref, v_gpu, NULL, s1);
    return 0;
}

static inline uint64_t adv_rng(uint64_t *s)      /* splitmix64 */
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

/* Values outside (0.5, 255.5) are never flagged and need no margin: the store truncates toward zero and clamps
 * (srcnn_kernels.h, fix_code()).  The objective only counts windows whose value could be flagged. */
static inline float adv_score(float v_ref, float v_gpu)
{
    if (!(v_gpu > 0.5f && v_gpu < 255.5f) && !(v_ref > 0.5f && v_ref < 255.5f)) return 0.f;
    return fabsf(v_gpu - v_ref);
}

/* n restarts of coordinate ascent, `iters` candidate moves each.  The first `scale_iters` moves of a restart climb on the
 * MAGNITUDE of the layer-3 products instead (rounding noise scales with it: a window that drives the maps hard is where
 * a large deviation can live), the remaining ones on the deviation itself.  starts / out_wins: [n][169]; out_dev[n] = the final
 * |v_gpu - v_ref|, out_v[n][2] = (v_ref, v_gpu).  A move changes one byte: to a random value, by +-1 .. +-8, or to an
 * extreme.  Returns the number of point evaluations. */
long srcnn_adv_search(const float *weights, const uint8_t *starts, int n, int iters, int scale_iters, uint64_t seed,
                      uint8_t *out_wins, float *out_dev, float *out_v)
{
    AdvModel m;
    if (!weights || !starts || !out_wins || !out_dev || n <= 0 || iters < 0) return -1;
    adv_model(weights, &m);
    long evals = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : evals)
    for (int r = 0; r < n; r++) {
        uint8_t w[WIN * WIN];
        memcpy(w, starts + (size_t)r * WIN * WIN, sizeof w);
        uint64_t s = seed * 0x2545f4914f6cdd1dull + (uint64_t)r * 0x9e3779b97f4a7c15ull + 1;
        float vr, vg, sc0;
        adv_eval_scale(&m, w, &vr, &vg, &sc0);
        float best = adv_score(vr, vg), bvr = vr, bvg = vg, best_scale = sc0;
        ++evals;
        for (int it = 0; it < iters; it++) {
            const int climb_scale = it < scale_iters;
            const uint64_t z = adv_rng(&s);
            const int at = (int)(z % (WIN * WIN)), kind = (int)((z >> 16) & 7);
            const uint8_t old = w[at];
            int nv;
            if (kind < 3) nv = (int)((z >> 24) & 255);
            else if (kind < 6) nv = (int)old + (int)((z >> 24) % 17) - 8;
            else nv = (z >> 24) & 1 ? 255 : 0;
            nv = nv < 0 ? 0 : (nv > 255 ? 255 : nv);
            if (nv == old) continue;
            w[at] = (uint8_t)nv;
            float mag;
            adv_eval_scale(&m, w, &vr, &vg, &mag);
            ++evals;
            const float sc = adv_score(vr, vg);
            if (climb_scale) {
                /* keep the value where it can be flagged at all: (0.5, 255.5) */
                if (mag > best_scale && vg > 0.5f && vg < 255.5f) { best_scale = mag; best = sc; bvr = vr; bvg = vg; }
                else w[at] = old;
            } else if (sc > best) { best = sc; bvr = vr; bvg = vg; }
            else w[at] = old;
        }
        memcpy(out_wins + (size_t)r * WIN * WIN, w, sizeof w);
        out_dev[r] = best;
        if (out_v) { out_v[2 * r] = bvr; out_v[2 * r + 1] = bvg; }
    }
    return evals;
}

/* The same climb on what a PER-PIXEL threshold  k * 2^-24 * S1 + abs_term  has to cover (round 6):
 *   score = max(gain * |v_gpu - v_ref| - abs_term, 0) / (2^-24 * S1)
 * i.e. the factor k this window needs for the threshold to stay `gain` times above its deviation (gain = 1: the bare
 * requirement; gain = 1.73: the safety factor the global delta keeps over the worst deviation any search has found).  A window
 * may win by a large deviation or by a small local scale -- which is what sampling |v_gpu - v_ref| alone (srcnn_adv_search)
 * cannot show.  out_v[n][3] = (v_ref, v_gpu, S1); out_dev[n] = the score. */
long srcnn_adv_search_ratio(const float *weights, const uint8_t *starts, int n, int iters, float abs_term, float gain, uint64_t seed,
                            uint8_t *out_wins, float *out_dev, float *out_v)
{
    AdvModel m;
    if (!weights || !starts || !out_wins || !out_dev || n <= 0 || iters < 0) return -1;
    adv_model(weights, &m);
    long evals = 0;
    const float eps = 5.9604644775390625e-08f;      /* 2^-24 */
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : evals)
    for (int r = 0; r < n; r++) {
        uint8_t w[WIN * WIN];
        memcpy(w, starts + (size_t)r * WIN * WIN, sizeof w);
        uint64_t s = seed * 0x2545f4914f6cdd1dull + (uint64_t)r * 0x9e3779b97f4a7c15ull + 1;
        float vr, vg, s1;
        adv_eval_scale2(&m, w, &vr, &vg, NULL, &s1);
        float d = gain * adv_score(vr, vg) - abs_term;
        float best = (d > 0.f && s1 > 0.f) ? d / (eps * s1) : 0.f, bvr = vr, bvg = vg, bs1 = s1;
        ++evals;
        for (int it = 0; it < iters; it++) {
            const uint64_t z = adv_rng(&s);
            const int at = (int)(z % (WIN * WIN)), kind = (int)((z >> 16) & 7);
            const uint8_t old = w[at];
            int nv;
            if (kind < 3) nv = (int)((z >> 24) & 255);
            else if (kind < 6) nv = (int)old + (int)((z >> 24) % 17) - 8;
            else nv = (z >> 24) & 1 ? 255 : 0;
            nv = nv < 0 ? 0 : (nv > 255 ? 255 : nv);
            if (nv == old) continue;
            w[at] = (uint8_t)nv;
            adv_eval_scale2(&m, w, &vr, &vg, NULL, &s1);
            ++evals;
            d = gain * adv_score(vr, vg) - abs_term;
            const float sc = (d > 0.f && s1 > 0.f) ? d / (eps * s1) : 0.f;
            if (sc > best) { best = sc; bvr = vr; bvg = vg; bs1 = s1; }
            else w[at] = old;
        }
        memcpy(out_wins + (size_t)r * WIN * WIN, w, sizeof w);
        out_dev[r] = best;
        if (out_v) { out_v[3 * r] = bvr; out_v[3 * r + 1] = bvg; out_v[3 * r + 2] = bs1; }
    }
    return evals;
}
