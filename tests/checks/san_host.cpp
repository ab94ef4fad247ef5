// san_host.cpp -- sanitizer harness for the host-only logic of srcnn_cpp_amd/csrc/srcnn_{api,model,plan,launch,host,multi}.cpp: the work-item
// planner (plan_items), the MFMA / split-f16 fragment packers and the cubic coefficient tables, reached through the
// tuning build's debug hooks.  tests/test_sanitizers.py compiles those files themselves (-DSRCNN_TUNING_BUILD) with
// -fsanitize=address,undefined (host compiler, no device code), links it with the kernel-launch stubs below and
// libamdhip64, and runs this on the CPU.  No HIP call is made: nothing here needs a device.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "srcnn_kernels.h"

// ---- link-time stand-ins for the kernel launchers (srcnn_*.hip); never called by this harness ----
namespace srcnn {
static hipError_t never() { std::abort(); }
size_t strip_lds_bytes(int) { return 35840; }
size_t split16_lds_bytes() { return 0; }
hipError_t launch_seams(const StripParams &, int, const int *, hipStream_t) { return never(); }
hipError_t launch_cseams(const StripParams &, int, hipStream_t) { return never(); }
hipError_t launch_seams_merged(const StripParams &, int, const int *, const unsigned char *, int, hipStream_t) { return never(); }
hipError_t launch_strip(int, const StripParams &, int, hipStream_t, size_t) { return never(); }
hipError_t launch_strip_fold(const StripParams &, const FoldParams &, hipStream_t, size_t) { return never(); }
hipError_t launch_strip_safe(int, const StripParams &, int, hipStream_t, size_t) { return never(); }
long interlock_probe_mismatches(int, hipStream_t) { return 0; }
hipError_t launch_fixup(const FixParams &, int, bool, bool, hipStream_t) { return never(); }
size_t fixup_list_entries(int, int, int, size_t *dense) { if (dense) *dense = 0; return 0; }
hipError_t launch_split16(const StripParams &, int, hipStream_t, size_t) { return never(); }
hipError_t launch_conv99_exact(const uint8_t *, long, float *, long, int, int, const float *, float, hipStream_t) { return never(); }
hipError_t launch_conv11_exact(const float *, long, long, float *, long, int, int, const float *, float, hipStream_t) { return never(); }
hipError_t launch_conv99x11_exact(const uint8_t *, long, long, float *, long, long, long, int, int, int, const float *, hipStream_t) { return never(); }
hipError_t launch_conv55_exact(const float *, long, long, long, uint8_t *, float *, long, long, int, int, int, const float *, float, hipStream_t) { return never(); }
hipError_t launch_copy_rows(uint8_t *, long, const uint8_t *, long, int, int, hipStream_t) { return never(); }
hipError_t launch_bgr2ycrcb(const uint8_t *, long, int, int, uint8_t *, long, long, hipStream_t) { return never(); }
hipError_t launch_ycrcb2bgr(const uint8_t *, long, const uint8_t *, long, long, int, int, uint8_t *, long, hipStream_t) { return never(); }
hipError_t launch_resize_cubic(const uint8_t *, long, long, int, int, uint8_t *, long, long, int, int, int, const int *, const short *, const int *, const short *, hipStream_t) { return never(); }
bool fused_pipeline_ok(int, int, int, int, const void *, long, const void *, long) { return false; }
hipError_t launch_bgr_to_y_resized(const uint8_t *, long, int, int, uint8_t *, long, int, int, const int *, const short *, const int *, const short *, hipStream_t) { return never(); }
hipError_t launch_resize_merge(const uint8_t *, long, int, int, const uint8_t *, long, uint8_t *, long, int, int, const int *, const short *, const int *, const short *, hipStream_t) { return never(); }
}  // namespace srcnn

extern "C" {
int srcnn_debug_plan_items(int n_cu, int n_strips, int row_begin, int row_end, int skew_pct, int wgs_per_cu, int want_seams,
                           int *items, int max_items, int *seams, int max_seams, int *n_seams);
int srcnn_debug_pack_fragments(const float *blob8129, float *frag, uint8_t *frag16, int *frag_floats, int *frag16_bytes);
int srcnn_debug_cubic_table(int n_src, int n_dst, int *ofs, short *coef);
int srcnn_debug_worker_pool(int n, int rounds);
int srcnn_stripe_rows(int height, int n_parts, int index, int *row_begin, int *row_end);
int srcnn_scaled_size(int width, int height, float scale, int *out_w, int *out_h);
}

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    // ---- the persistent worker threads of the several-GPUs entry points (condition variables, task hand-over, teardown) ----
    CHECK(srcnn_debug_worker_pool(1, 5) == 0);
    CHECK(srcnn_debug_worker_pool(8, 2000) == 0);
    CHECK(srcnn_debug_worker_pool(3, 301) == 0);
#ifdef SRCNN_SAN_POOL_ONLY      // the ThreadSanitizer pass: the pool is the only threaded host code
    std::printf("ok: worker pool\n");
    return 0;
#endif

    // ---- planner: exact-size buffers (one int beyond them is a sanitizer error), every geometry class ----
    long plans = 0, with_items = 0;
    for (int n_cu : {1, 4, 64, 256, 304})
        for (int n_strips : {1, 2, 3, 9, 15, 30, 31, 60, 61, 256, 300})
            for (int rows : {1, 7, 8, 9, 10, 23, 24, 45, 135, 270, 540, 1080, 2160, 4320, 8192})
                for (int row_begin : {0, 1000})
                    for (int skew : {0, 10, 35, 99})
                        for (int wpc : {1, 2})
                            for (int seams : {0, 1})
                              for (int edges : {0, 2 * 16 + 2 * 256, 2 * 16, 2 * 256}) {     // the planner's edge-row weights ride in the upper bits
                                if (edges && (wpc != 2 || !seams || skew != 10 || n_cu < 64)) continue;   // (only the balanced planner takes them; the product's skew)
                                if (edges && edges != 2 * 16 + 2 * 256 && n_cu != 256) continue;           // (one-sided: the product's CU count only)
                                int n_seams = -1;
                                const int cap = wpc * n_cu;
                                std::vector<int> items((size_t)srcnn::ITEM_INTS * cap), sm((size_t)2 * cap);
                                const int n = srcnn_debug_plan_items(n_cu, n_strips, row_begin, row_begin + rows, skew, wpc + edges, seams,
                                                                     items.data(), cap, sm.data(), cap, &n_seams);
                                ++plans;
                                CHECK(n >= 0 && n <= cap && n_seams >= 0 && n_seams <= cap);
                                if (n == 0) continue;
                                ++with_items;
                                // the items tile [row_begin, row_end) of every strip exactly once
                                std::vector<long> covered((size_t)n_strips, 0);
                                for (int i = 0; i < n; ++i) {
                                    const int *it = &items[(size_t)srcnn::ITEM_INTS * i];
                                    CHECK(it[0] >= 0 && it[0] < n_strips && it[1] >= row_begin && it[2] <= row_begin + rows && it[1] <= it[2]);
                                    CHECK(it[3] >= -1 && it[3] < n_seams && it[4] >= -1 && it[4] < n_seams);
                                    covered[(size_t)it[0]] += it[2] - it[1];
                                }
                                for (long cvr : covered) CHECK(cvr == rows);
                            }
    // ---- fragment packers: exact-size outputs, extreme but finite weights ----
    int nf = 0, nb = 0;
    srcnn_debug_pack_fragments(nullptr, nullptr, nullptr, &nf, &nb);
    CHECK(nf == srcnn::NFRAG * 64 && nb == (int)srcnn::S16_TABLE_BYTES);
    std::vector<float> blob(8129), frag((size_t)nf);
    std::vector<uint8_t> frag16((size_t)nb);
    unsigned s = 12345;
    for (float scale : {0.f, 1e-30f, 0.5f, 177.0f, 70000.0f, 3e38f})
        for (int rep = 0; rep < 3; ++rep) {
            for (auto &v : blob) { s = s * 1664525u + 1013904223u; v = scale * ((int)(s >> 8) % 2001 - 1000) / 1000.0f; }
            const int ok = srcnn_debug_pack_fragments(blob.data(), frag.data(), frag16.data(), nullptr, nullptr);
            CHECK(ok == 0 || ok == 1);
        }
    if (argc > 1) {     // the real model: must be accepted by the split-f16 range check
        FILE *f = std::fopen(argv[1], "rb");
        CHECK(f && std::fread(blob.data(), 4, 8129, f) == 8129);
        std::fclose(f);
        CHECK(srcnn_debug_pack_fragments(blob.data(), frag.data(), frag16.data(), nullptr, nullptr) == 1);
    }
    // ---- cubic tables: every (src, dst) pair class, exact-size outputs ----
    for (int n_src : {1, 2, 3, 4, 5, 384, 1080, 1920})
        for (int n_dst : {1, 2, 3, 7, 576, 2160, 3840, 8191}) {
            std::vector<int> ofs((size_t)n_dst);
            std::vector<short> coef((size_t)4 * n_dst);
            CHECK(srcnn_debug_cubic_table(n_src, n_dst, ofs.data(), coef.data()) == 0);
            for (int d = 0; d < n_dst; ++d) {
                const int sum = coef[4 * d] + coef[4 * d + 1] + coef[4 * d + 2] + coef[4 * d + 3];
                CHECK(sum >= 2046 && sum <= 2050 && ofs[d] >= -2 && ofs[d] <= n_src);
            }
        }
    // ---- small pure helpers ----
    for (int h : {0, 1, 5, 6, 2160, 4320})
        for (int parts : {1, 2, 3, 8, 16}) {
            int prev = 0, a, b;
            for (int k = 0; k < parts; ++k) {
                CHECK(srcnn_stripe_rows(h, parts, k, &a, &b) == 0 && a == prev && b >= a);
                prev = b;
            }
            CHECK(prev == h);
        }
    int ow, oh;
    CHECK(srcnn_scaled_size(1920, 1080, 2.0f, &ow, &oh) == 0 && ow == 3840 && oh == 2160);
    CHECK(srcnn_scaled_size(3, 3, 0.2f, &ow, &oh) != 0);
    std::printf("ok: %ld plans (%ld with explicit items), packers and tables clean\n", plans, with_items);
    return 0;
}
