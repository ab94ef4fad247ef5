/* san_oracle.c -- TEST INFRASTRUCTURE: the CPU oracle's conv path under AddressSanitizer + UBSan, and a run-to-run agreement
 * check under heavy thread oversubscription (advisor, round 5: one oracle call in ~5,000 on the GPU boxes' shared 256-thread
 * hosts returned a band of wrong rows; cause not found).  Linked against oracle/srcnn_oracle.c compiled with the sanitizers
 * (tests/test_sanitizers.py).  usage: san_oracle WEIGHTS.f32 [planes=60] [threads=256]
 * Every plane is computed twice with `threads` OpenMP threads (on the 8-CPU build container: 32 threads per core, every barrier
 * of the two parallel loops crossed by descheduled threads) and once with 1 thread; all three must agree byte for byte. */
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int srcnn_oracle_forward_y(const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride, int width, int height,
                           const float *weights, float *preclamp);

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const int planes = argc > 2 ? atoi(argv[2]) : 60, threads = argc > 3 ? atoi(argv[3]) : 256;
    float *w = malloc(8129 * sizeof(float));
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(w, sizeof(float), 8129, f) != 8129) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    static const int sizes[][2] = {{1, 1}, {3, 3}, {9, 5}, {5, 9}, {17, 4}, {2, 40}, {40, 2}, {33, 9}};
    long bad = 0;
    for (int n = 0; n < planes; n++) {
        int W, H;
        if (n < 8) { W = sizes[n][0]; H = sizes[n][1]; }
        else { W = 20 + (int)(rnd() % 200); H = 10 + (int)(rnd() % 120); }
        const size_t px = (size_t)W * H;
        uint8_t *y = malloc(px), *a = malloc(px), *b = malloc(px), *c = malloc(px);    /* exact sizes: ASan sees any overrun */
        float *pa = malloc(px * sizeof(float)), *pb = malloc(px * sizeof(float));
        for (size_t i = 0; i < px; i++) y[i] = (uint8_t)(n % 3 == 0 ? rnd() : 100 + rnd() % 60);
        omp_set_num_threads(threads);
        if (srcnn_oracle_forward_y(y, W, a, W, W, H, w, pa) || srcnn_oracle_forward_y(y, W, b, W, W, H, w, pb)) return 3;
        omp_set_num_threads(1);
        if (srcnn_oracle_forward_y(y, W, c, W, W, H, w, NULL)) return 3;
        if (memcmp(a, b, px) || memcmp(a, c, px) || memcmp(pa, pb, px * sizeof(float))) {
            fprintf(stderr, "plane %d (%d x %d): runs disagree\n", n, W, H);
            bad++;
        }
        free(y); free(a); free(b); free(c); free(pa); free(pb);
    }
    free(w);
    printf("%s: %d planes, 2 runs at %d threads + 1 run at 1 thread each, %ld disagreements\n", bad ? "FAILED" : "ok", planes, threads, bad);
    return bad ? 1 : 0;
}
