// host_demo.cpp -- a plain C++ host (no OpenCV, no HIP headers) driving the HIP
// path through the reference's own call surface (include/srcnn_amd.hpp):
//   Convolution99x11 + Convolution55  exactly as src/srcnn.cpp:602-627 calls them,
//   then the fused ForwardY, then the same two calls on srcnn::DevicePlane<float> (the 32-plane map never leaves the
//   GPU), and checks that all three give the same plane.
// Build: g++ -std=c++17 -Iinclude tools/host_demo.cpp -Lsrcnn_cpp_amd -lsrcnn_amd \
//            -Wl,-rpath,$PWD/srcnn_cpp_amd -o build/host_demo
// Run:   build/host_demo srcnn_cpp_amd/data/srcnn915_weights.f32 W H out.u8
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
    if (argc < 5) { std::fprintf(stderr, "usage: %s weights.f32 W H out.u8\n", argv[0]); return 2; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
    static float blob[8129];
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(blob, 4, 8129, f) != 8129) { std::fprintf(stderr, "bad weight file\n"); return 1; }
    std::fclose(f);
    // convdata.h order: b1[64] | W1[64][9][9] | b2[32] | W2[32][64] | b3 | W3[32][5][5]
    const float *b1 = blob;
    auto w1 = reinterpret_cast<const float(*)[9][9]>(blob + 64);
    const float *b2 = blob + 5248;
    auto w2 = reinterpret_cast<const float(*)[64]>(blob + 5280);
    const float b3 = blob[7328];
    auto w3 = reinterpret_cast<const float(*)[5][5]>(blob + 7329);

    srcnn::Plane<unsigned char> y(W, H), out_a(W, H), out_b(W, H), out_c(W, H);
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) y.at(r, c) = synth(c, r, 0, W, H);
    try {
        // the reference's driver, src/srcnn.cpp:602-627
        std::vector<srcnn::Plane<float>> conv2(32);
        for (auto &p : conv2) p.create(W, H);
        srcnn::Convolution99x11(y, conv2, w1, b1, w2, b2);
        srcnn::Convolution55(conv2, out_a, w3, b3);
        // the fused replacement
        srcnn::ForwardY(y, out_b, w1, b1, w2, b2, w3, b3);
        // the same two calls with the 32 planes kept on the GPU: only the vector's element type differs (:602-607)
        std::vector<srcnn::DevicePlane<float>> conv2d = srcnn::DevicePlanes<float>(32, W, H);
        srcnn::Convolution99x11(y, conv2d, w1, b1, w2, b2);
        srcnn::Convolution55(conv2d, out_c, w3, b3);
        // ... and the device map is the map the host-plane call returned
        const srcnn::Plane<float> back = conv2d[17].download();
        if (std::memcmp(back.data, conv2[17].data, sizeof(float) * (size_t)W * H) != 0) { std::fprintf(stderr, "device map != host map\n"); return 6; }
    } catch (const srcnn::Error &e) {
        std::fprintf(stderr, "srcnn error %d: %s\n", e.code, e.what());
        return 3;
    }
    if (std::memcmp(out_a.data, out_b.data, (size_t)W * H) != 0) { std::fprintf(stderr, "fused != unfused\n"); return 4; }
    if (std::memcmp(out_a.data, out_c.data, (size_t)W * H) != 0) { std::fprintf(stderr, "device planes != host planes\n"); return 7; }
    unsigned long sum = 0;
    for (size_t i = 0; i < (size_t)W * H; ++i) sum += out_a.storage[i];
    FILE *o = std::fopen(argv[4], "wb");
    if (!o || std::fwrite(out_a.data, 1, (size_t)W * H, o) != (size_t)W * H) return 5;
    std::fclose(o);
    std::printf("ok %dx%d checksum %lu\n", W, H, sum);
    return 0;
}
