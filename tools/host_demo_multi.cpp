// host_demo_multi.cpp -- a plain C++ host (no OpenCV, no HIP headers) that owns SEVERAL contexts / GPUs and
// serves the reference's conv-path call sites (src/srcnn.cpp:609,627) with them through include/srcnn_amd.hpp:
//   * ONE plane row-striped over the contexts (srcnn_forward_y_striped: own rows uploaded per device, 6 halo rows
//     per boundary device to device, interior rows first, edge bands after the copies)
//   * a stream of frames, contiguous ranges per context (srcnn_forward_y_frames_multi, no collective)
//   * device-resident planes of a stream alternately on the contexts used as lanes (srcnn_forward_y_lanes_dev)
// and checks all three bit for bit against the single-context ForwardY.
// Build: g++ -std=c++17 -pthread -Iinclude tools/host_demo_multi.cpp -Lsrcnn_cpp_amd -lsrcnn_amd \
//            -Wl,-rpath,$PWD/srcnn_cpp_amd -o build/host_demo_multi
// Run:   build/host_demo_multi weights.f32 W H N_FRAMES out.u8 dev0 [dev1 ...]   (a device may repeat)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "srcnn_amd.hpp"

// The integer-only synthetic luma of srcnn_cpp_amd/synth.py.
static unsigned char synth(int x, int y, int f, int W, int H, unsigned seed = 12345)
{
    auto tri = [](long t, long p) { long m = t % (2 * p); return labs(m - p); };
    unsigned h = seed ^ (unsigned)(((long)f * H + y) * W + x);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    long v = 2 * tri(x + 5 * f, 61) + tri(y + 3 * f, 89) + tri(x + y, 23) + (h >> 29);
    return (unsigned char)(v > 255 ? 255 : v);
}

int main(int argc, char **argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage: %s weights.f32 W H N_FRAMES out.u8 dev0 [dev1 ...]\n", argv[0]); return 2; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), NF = std::atoi(argv[4]);
    std::vector<int> devices;
    for (int i = 6; i < argc; ++i) devices.push_back(std::atoi(argv[i]));
    static float blob[8129];
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(blob, 4, 8129, f) != 8129) { std::fprintf(stderr, "bad weight file\n"); return 1; }
    std::fclose(f);
    const float *b1 = blob;
    auto w1 = reinterpret_cast<const float(*)[9][9]>(blob + 64);
    const float *b2 = blob + 5248;
    auto w2 = reinterpret_cast<const float(*)[64]>(blob + 5280);
    const float b3 = blob[7328];
    auto w3 = reinterpret_cast<const float(*)[5][5]>(blob + 7329);

    using Plane = srcnn::Plane<unsigned char>;
    std::vector<Plane> in((size_t)NF), ref((size_t)NF), out((size_t)NF);
    for (int k = 0; k < NF; ++k) {
        in[k].create(W, H); ref[k].create(W, H); out[k].create(W, H);
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c) in[k].at(r, c) = synth(c, r, k, W, H);
    }
    try {
        for (int k = 0; k < NF; ++k) srcnn::ForwardY(in[k], ref[k], w1, b1, w2, b2, w3, b3);      // one context, one GPU
        srcnn::SessionSet set(devices);
        set.set_weights(w1, b1, w2, b2, w3, b3);
        Plane striped(W, H);
        srcnn::ForwardYStriped(set, in[0], striped);                                               // warm-up + check
        if (std::memcmp(striped.data, ref[0].data, (size_t)W * H) != 0) { std::fprintf(stderr, "striped != single\n"); return 4; }
        const auto t0 = std::chrono::steady_clock::now();
        srcnn::ForwardYStriped(set, in[0], striped);
        const auto t1 = std::chrono::steady_clock::now();
        if (std::memcmp(striped.data, ref[0].data, (size_t)W * H) != 0) { std::fprintf(stderr, "striped (2nd) != single\n"); return 4; }
        srcnn::ForwardYFrames(set, in, out);
        const auto t2 = std::chrono::steady_clock::now();
        for (int k = 0; k < NF; ++k)
            if (std::memcmp(out[k].data, ref[k].data, (size_t)W * H) != 0) { std::fprintf(stderr, "frame %d != single\n", k); return 5; }
        // device-resident planes of a stream on the set's contexts used as LANES (two contexts on one GPU = two lanes of it:
        // srcnn_forward_y_lanes_dev): upload once, queue everything, wait once, compare
        {
            auto d_in = srcnn::DevicePlanes<unsigned char>(NF, W, H), d_out = srcnn::DevicePlanes<unsigned char>(NF, W, H);
            for (int k = 0; k < NF; ++k)
                if (srcnn_dev_upload(d_in[k].ctx, d_in[k].data, in[k].data, (size_t)W * H) != SRCNN_OK) { std::fprintf(stderr, "upload\n"); return 7; }
            srcnn::ForwardYLanes(set, d_in, d_out);
            set.synchronize();
            for (int k = 0; k < NF; ++k) {
                const Plane got = d_out[k].download();
                if (std::memcmp(got.data, ref[k].data, (size_t)W * H) != 0) { std::fprintf(stderr, "lane plane %d != single\n", k); return 7; }
            }
        }
        std::printf("ok %dx%d x%d on %d contexts: striped %.3f ms, frames %.3f ms (host buffers, PCIe-inclusive); lanes ok\n", W, H, NF,
                    set.size(), std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(t2 - t1).count());
    } catch (const srcnn::Error &e) {
        std::fprintf(stderr, "srcnn error %d: %s\n", e.code, e.what());
        return 3;
    }
    FILE *o = std::fopen(argv[5], "wb");
    if (!o || std::fwrite(out[NF - 1].data, 1, (size_t)W * H, o) != (size_t)W * H) return 6;
    std::fclose(o);
    return 0;
}
