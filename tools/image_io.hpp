// image_io.hpp -- minimal image file I/O for the command-line tool (the
// reference uses cv::imread / cv::imwrite, src/srcnn.cpp:462,670; OpenCV is not
// a dependency here).  Reads 8-bit PNG (grey, RGB, palette, +alpha; non-interlaced)
// and binary PGM/PPM into packed B,G,R bytes like cv::imread(IMREAD_COLOR) does
// (alpha dropped, grey replicated); writes 8-bit RGB PNG or PPM chosen by file
// extension.  PNG needs zlib only.
#pragma once
#include <zlib.h>

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace imgio {

struct Image {
    int width = 0, height = 0;
    std::vector<unsigned char> bgr;   // packed B,G,R, row-major
    bool empty() const { return bgr.empty(); }
};

inline bool read_file(const std::string &path, std::vector<unsigned char> &out)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); return false; }
    out.resize((size_t)n);
    bool ok = n == 0 || std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    return ok;
}

// Files are untrusted input: dimensions above these caps are rejected before any size arithmetic
// (65,535 per side, 2^28 pixels: a 4-channel scanline buffer then stays below 2^31 bytes).
constexpr uint32_t kMaxSide = 65535;
constexpr uint64_t kMaxPixels = 1ull << 28;
inline bool sane_dims(uint64_t w, uint64_t h) { return w > 0 && h > 0 && w <= kMaxSide && h <= kMaxSide && w * h <= kMaxPixels; }

inline uint32_t be32(const unsigned char *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

inline bool decode_png(const std::vector<unsigned char> &d, Image &img)
{
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    bool have_ihdr = false, first = true;
    while (pos + 12 <= d.size()) {
        const uint32_t len = be32(&d[pos]);
        const char *type = reinterpret_cast<const char *>(&d[pos + 4]);
        if ((size_t)len > d.size() - pos - 12) return false;
        const unsigned char *body = &d[pos + 8];
        const bool is_ihdr = !std::memcmp(type, "IHDR", 4);
        if (first != is_ihdr) return false;             // IHDR is the first chunk and appears once
        first = false;
        if (is_ihdr) {
            if (len != 13) return false;
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (!sane_dims(w, h)) return false;
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || depth != 8 || interlace != 0 || idat.empty()) return false;
    int ch;
    switch (ctype) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: return false;
    }
    const size_t stride = (size_t)w * ch;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<unsigned char> pix(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const unsigned char *in = &raw[(stride + 1) * y];
        unsigned char *cur = &pix[stride * y];
        const unsigned char *up = y ? &pix[stride * (y - 1)] : nullptr;
        const int ft = in[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int v = in[1 + i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return false;
            }
            cur[i] = (unsigned char)v;
        }
    }
    img.width = (int)w;
    img.height = (int)h;
    img.bgr.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        unsigned char r, g, b;
        const unsigned char *p = &pix[i * ch];
        if (ctype == 0 || ctype == 4) r = g = b = p[0];
        else if (ctype == 3) {
            if ((size_t)p[0] * 3 + 2 >= plte.size()) return false;
            r = plte[p[0] * 3]; g = plte[p[0] * 3 + 1]; b = plte[p[0] * 3 + 2];
        } else { r = p[0]; g = p[1]; b = p[2]; }
        img.bgr[3 * i] = b; img.bgr[3 * i + 1] = g; img.bgr[3 * i + 2] = r;
    }
    return true;
}

inline bool decode_pnm(const std::vector<unsigned char> &d, Image &img)
{
    if (d.size() < 3 || d[0] != 'P' || (d[1] != '5' && d[1] != '6')) return false;
    const int ch = d[1] == '6' ? 3 : 1;
    size_t pos = 2;
    long vals[3];
    for (int k = 0; k < 3; ++k) {
        int digits = 0;
        for (;;) {
            while (pos < d.size() && std::isspace(d[pos])) ++pos;
            if (pos < d.size() && d[pos] == '#') { while (pos < d.size() && d[pos] != '\n') ++pos; continue; }
            break;
        }
        long v = 0; bool any = false;
        while (pos < d.size() && d[pos] >= '0' && d[pos] <= '9') {
            if (++digits > 6) return false;              // 65,535 has 5 digits: no overflow of `v` possible
            v = v * 10 + (d[pos++] - '0');
            any = true;
        }
        if (!any) return false;
        vals[k] = v;
    }
    if (pos >= d.size() || !std::isspace(d[pos])) return false;
    ++pos;   // single whitespace after maxval
    if (vals[0] <= 0 || vals[1] <= 0 || vals[2] != 255 || !sane_dims((uint64_t)vals[0], (uint64_t)vals[1])) return false;
    const size_t n = (size_t)vals[0] * vals[1];
    if (n * ch > d.size() - pos) return false;
    img.width = (int)vals[0]; img.height = (int)vals[1];
    img.bgr.resize(n * 3);
    for (size_t i = 0; i < n; ++i) {
        const unsigned char *p = &d[pos + i * ch];
        if (ch == 1) img.bgr[3 * i] = img.bgr[3 * i + 1] = img.bgr[3 * i + 2] = p[0];
        else { img.bgr[3 * i] = p[2]; img.bgr[3 * i + 1] = p[1]; img.bgr[3 * i + 2] = p[0]; }
    }
    return true;
}

inline Image imread(const std::string &path)
{
    Image img;
    std::vector<unsigned char> d;
    if (!read_file(path, d)) return img;
    if (!decode_png(d, img) && !decode_pnm(d, img)) img = Image();
    return img;
}

inline void put32(std::vector<unsigned char> &v, uint32_t x)
{
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}

inline void chunk(std::vector<unsigned char> &out, const char *type, const std::vector<unsigned char> &body)
{
    put32(out, (uint32_t)body.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    put32(out, (uint32_t)crc32(0L, &out[at], (uInt)(4 + body.size())));
}

inline bool imwrite(const std::string &path, const unsigned char *bgr, int w, int h)
{
    if (!bgr || w <= 0 || h <= 0) return false;
    std::vector<unsigned char> out;
    const size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    for (auto &c : ext) c = (char)std::tolower(c);
    if (ext == ".ppm" || ext == ".pnm") {
        char hdr[64];
        int n = std::snprintf(hdr, sizeof(hdr), "P6\n%d %d\n255\n", w, h);
        out.assign(hdr, hdr + n);
        for (size_t i = 0; i < (size_t)w * h; ++i) { out.push_back(bgr[3 * i + 2]); out.push_back(bgr[3 * i + 1]); out.push_back(bgr[3 * i]); }
    } else if (ext == ".png") {
        static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
        out.assign(sig, sig + 8);
        std::vector<unsigned char> ihdr;
        put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
        const unsigned char tail[5] = {8, 2, 0, 0, 0};
        ihdr.insert(ihdr.end(), tail, tail + 5);
        chunk(out, "IHDR", ihdr);
        // Rows with the SUB filter (each byte minus the same channel of the pixel to its left), deflate at its fastest level with
        // the run-length strategy: what cv::imwrite does for a PNG by default (IMWRITE_PNG_COMPRESSION 1, IMWRITE_PNG_STRATEGY
        // RLE, PNG_FILTER_SUB) -- the reference's writer, src/srcnn.cpp:670.  Unfiltered rows at level 6 took 0.8 s for a
        // 3840x2160 picture, most of the tool's wall clock (profiles/r05/cli_process_cold.txt); this form 0.1 s.
        std::vector<unsigned char> raw(((size_t)w * 3 + 1) * h);
        for (int y = 0; y < h; ++y) {
            unsigned char *row = &raw[((size_t)w * 3 + 1) * y];
            row[0] = 1;
            unsigned char pr = 0, pg = 0, pb = 0;
            for (int x = 0; x < w; ++x) {
                const unsigned char *p = &bgr[((size_t)y * w + x) * 3];
                row[1 + 3 * x] = (unsigned char)(p[2] - pr); row[2 + 3 * x] = (unsigned char)(p[1] - pg); row[3 + 3 * x] = (unsigned char)(p[0] - pb);
                pr = p[2]; pg = p[1]; pb = p[0];
            }
        }
        uLongf clen = compressBound((uLong)raw.size());
        std::vector<unsigned char> comp(clen);
        z_stream zs{};
        if (deflateInit2(&zs, 1, Z_DEFLATED, 15, 8, Z_RLE) != Z_OK) return false;
        zs.next_in = raw.data();
        zs.avail_in = (uInt)raw.size();
        zs.next_out = comp.data();
        zs.avail_out = (uInt)comp.size();
        const int zrc = deflate(&zs, Z_FINISH);
        clen = zs.total_out;
        deflateEnd(&zs);
        if (zrc != Z_STREAM_END) return false;
        comp.resize(clen);
        chunk(out, "IDAT", comp);
        chunk(out, "IEND", {});
    } else {
        return false;
    }
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

}  // namespace imgio
