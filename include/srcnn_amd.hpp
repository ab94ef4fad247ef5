// srcnn_amd.hpp -- C++ host-side mirror of the reference's call surface over
// the C ABI (srcnn_amd.h).  Header-only.
//
// The four free functions below have the names, argument order and argument
// meaning of the reference's prototypes at src/srcnn.cpp:60-73:
//
//   void Convolution99   (Mat& src, Mat& dst, const float kernel[9][9], float bias);
//   void Convolution11   (vector<Mat>& src, Mat& dst, const float kernel[64], float bias);
//   void Convolution55   (vector<Mat>& src, Mat& dst, const float kernel[32][5][5], float bias);
//   void Convolution99x11(Mat& src, vector<Mat>& dst, const float kernel99[64][9][9],
//                         const float bias99[64], const float kernel11[32][64], const float bias11[32]);
//
// They are templates over the matrix type, which only has to look like the
// part of cv::Mat the reference uses: `.rows`, `.cols`, `.data` (first byte)
// and `.step` convertible to size_t (row stride in BYTES).  cv::Mat satisfies
// that as is, so with OpenCV present a reference build only has to include
// this header instead of defining its own loops;
// srcnn::Plane<T> is a dependency-free stand-in for hosts without OpenCV.
//
// Like the reference functions they return void and write dst in place
// (src/srcnn.cpp:137,175,240,321); a failure of the GPU path throws
// srcnn::Error -- there is no CPU fallback.
#ifndef SRCNN_AMD_HPP
#define SRCNN_AMD_HPP

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "srcnn_amd.h"

namespace srcnn {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

// Minimal row-major plane with cv::Mat's field names.
template <class T>
struct Plane {
    int rows = 0, cols = 0;
    std::size_t step = 0;            // bytes per row
    unsigned char *data = nullptr;
    std::vector<T> storage;
    Plane() = default;
    Plane(int cols_, int rows_) { create(cols_, rows_); }
    void create(int cols_, int rows_)
    {
        rows = rows_;
        cols = cols_;
        storage.assign(static_cast<std::size_t>(rows) * cols, T());
        step = sizeof(T) * static_cast<std::size_t>(cols);
        data = reinterpret_cast<unsigned char *>(storage.data());
    }
    T &at(int r, int c) { return storage[static_cast<std::size_t>(r) * cols + c]; }
    const T &at(int r, int c) const { return storage[static_cast<std::size_t>(r) * cols + c]; }
    bool empty() const { return storage.empty(); }
};

// One lazily created context per host thread (the reference calls the path
// from a single worker thread, src/srcnn.cpp:720).
class Session {
public:
    explicit Session(int device = 0) : h_(std::make_shared<Handle>())
    {
        int rc = srcnn_create(&h_->ctx, device);
        if (rc != SRCNN_OK) throw Error(rc, "srcnn_create failed (a gfx950 GPU is required)");
    }
    Session(const Session &) = delete;
    Session &operator=(const Session &) = delete;
    srcnn_ctx *get() const { return h_->ctx; }
    // The context is destroyed when the Session AND every device allocation made through it are gone: a DevicePlane that
    // outlives its Session (a static vector, a vector handed to another thread that outlives the allocating one) keeps the
    // context alive through this handle instead of freeing into a destroyed one.
    std::shared_ptr<void> keep_alive() const { return h_; }
    void check(int rc) const
    {
        if (rc != SRCNN_OK) throw Error(rc, srcnn_last_error(h_->ctx));
    }
    static Session &thread_default()
    {
        thread_local Session s(0);
        return s;
    }
    // The reference's BYTES from the MFMA path (SRCNN_MODE_REFBYTES, srcnn_amd.h).  `net` (on by default, as in the C ABI): the
    // library acts on its own monitor -- a launch whose measured rounding noise comes within a factor two of the flag threshold
    // is redone in the reference's arithmetic on every pixel, on the device, without a host read (srcnn_set_fixup_strict).
    void reference_bytes(bool on = true, bool net = true)
    {
        check(srcnn_set_mode(get(), on ? SRCNN_MODE_REFBYTES : SRCNN_MODE_MFMA));
        check(srcnn_set_fixup_strict(get(), net ? 1 : 0));
    }
    // Seam deferral for callers that queue launches back to back on one stream (srcnn_set_seam_deferral): the last launch's
    // output is complete after flush() or any other call on the session.
    void seam_deferral(bool on = true) { check(srcnn_set_seam_deferral(get(), on ? 1 : 0)); }
    void flush() { check(srcnn_flush(get())); }
    // 0: the fast strip kernels (their hardware interlock was verified on this device at creation), 1: the hazard-safe ones
    int kernel_variant() const { return srcnn_kernel_variant(get()); }
    // A cautious deployment pins the hazard-safe kernels (same bytes, ~3 % slower): srcnn_set_kernel_variant
    void pin_safe_kernels(bool on = true) { check(srcnn_set_kernel_variant(get(), on ? 1 : 0)); }

private:
    struct Handle {
        srcnn_ctx *ctx = nullptr;
        Handle() = default;
        Handle(const Handle &) = delete;
        Handle &operator=(const Handle &) = delete;
        ~Handle() { srcnn_destroy(ctx); }
    };
    std::shared_ptr<Handle> h_;
};

namespace detail {
template <class T, class M>
inline T *ptr(M &m) { return reinterpret_cast<T *>(m.data); }
template <class T, class M>
inline std::size_t stride(const M &m) { return static_cast<std::size_t>(m.step) / sizeof(T); }
}  // namespace detail

// ---- the reference call surface -------------------------------------------

template <class MatU8, class MatF32>
inline void Convolution99(MatU8 &src, MatF32 &dst, const float kernel[9][9], float bias)
{
    Session &s = Session::thread_default();
    s.check(srcnn_conv99(s.get(), detail::ptr<const std::uint8_t>(src), detail::stride<std::uint8_t>(src),
                         detail::ptr<float>(dst), detail::stride<float>(dst), dst.cols, dst.rows,
                         &kernel[0][0], bias));                 // dims from dst: src/srcnn.cpp:94-95
}

template <class MatF32>
inline void Convolution11(std::vector<MatF32> &src, MatF32 &dst, const float kernel[SRCNN_CONV1_FILTERS],
                          float bias)
{
    if (src.size() != SRCNN_CONV1_FILTERS) throw Error(SRCNN_ERR_INVALID, "Convolution11: need 64 planes");
    const float *planes[SRCNN_CONV1_FILTERS];
    for (int k = 0; k < SRCNN_CONV1_FILTERS; ++k) planes[k] = detail::ptr<const float>(src[k]);
    Session &s = Session::thread_default();
    s.check(srcnn_conv11(s.get(), planes, detail::stride<float>(src[0]), detail::ptr<float>(dst),
                         detail::stride<float>(dst), dst.cols, dst.rows, kernel, bias));
}

template <class MatF32, class MatU8>
inline void Convolution55(std::vector<MatF32> &src, MatU8 &dst, const float kernel[32][5][5], float bias)
{
    if (src.size() != SRCNN_CONV2_FILTERS) throw Error(SRCNN_ERR_INVALID, "Convolution55: need 32 planes");
    const float *planes[SRCNN_CONV2_FILTERS];
    for (int k = 0; k < SRCNN_CONV2_FILTERS; ++k) planes[k] = detail::ptr<const float>(src[k]);
    Session &s = Session::thread_default();
    s.check(srcnn_conv55(s.get(), planes, detail::stride<float>(src[0]), detail::ptr<std::uint8_t>(dst),
                         detail::stride<std::uint8_t>(dst), dst.cols, dst.rows, &kernel[0][0][0], bias));
}

template <class MatU8, class MatF32>
inline void Convolution99x11(MatU8 &src, std::vector<MatF32> &dst,
                             const float kernel99[SRCNN_CONV1_FILTERS][9][9],
                             const float bias99[SRCNN_CONV1_FILTERS],
                             const float kernel11[SRCNN_CONV2_FILTERS][SRCNN_CONV1_FILTERS],
                             const float bias11[SRCNN_CONV2_FILTERS])
{
    if (dst.size() != SRCNN_CONV2_FILTERS) throw Error(SRCNN_ERR_INVALID, "Convolution99x11: need 32 planes");
    float *planes[SRCNN_CONV2_FILTERS];
    for (int k = 0; k < SRCNN_CONV2_FILTERS; ++k) planes[k] = detail::ptr<float>(dst[k]);
    Session &s = Session::thread_default();
    s.check(srcnn_conv99x11(s.get(), detail::ptr<const std::uint8_t>(src), detail::stride<std::uint8_t>(src),
                            planes, detail::stride<float>(dst[0]), src.cols, src.rows,   // dims from src: :262-263
                            &kernel99[0][0][0], bias99, &kernel11[0][0], bias11));
}

// ---- the same two call sites with the 32-plane map kept on the GPU --------------------------------------------
// The reference allocates 32 CV_32F Mats (src/srcnn.cpp:602-607), fills them at :609 and consumes them at :627: with host
// Mats that is 2 x 128 B/pixel over PCIe (42 ms per 3840x2160 plane, 40 x the kernels).  A maintainer who changes ONLY the
// element type of that vector keeps the text of :609 and :627 and never moves the map off the device:
//
//     std::vector<srcnn::DevicePlane<float>> pImgConv2 = srcnn::DevicePlanes<float>(CONV2_FILTERS, pImg[0].cols, pImg[0].rows);
//     Convolution99x11(pImg[0], pImgConv2, conv1_Weights, conv1_biases, conv2_Weights, conv2_biases);      // :609 unchanged
//     Convolution55(pImgConv2, pImgConv3, conv3_Weights, conv3_biases);                                     // :627 unchanged
//
// DevicePlane has cv::Mat's field names (rows, cols, step, data) but `data` is a DEVICE address; the planes of one
// DevicePlanes() call share one allocation on the calling thread's Session (plane k at k * rows * cols elements), which
// is freed when the last of them goes away; the Session's context stays alive until then.
template <class T>
struct DevicePlane {
    int rows = 0, cols = 0;
    std::size_t step = 0;            // bytes per row
    unsigned char *data = nullptr;   // DEVICE memory
    std::shared_ptr<void> owner;     // the allocation this plane lives in
    srcnn_ctx *ctx = nullptr;        // the context whose GPU holds it
    bool empty() const { return data == nullptr; }
    // a host copy, for inspection (synchronises the context's stream)
    Plane<T> download() const
    {
        Plane<T> h(cols, rows);
        if (srcnn_dev_download(ctx, h.data, data, sizeof(T) * static_cast<std::size_t>(rows) * cols) != SRCNN_OK)
            throw Error(SRCNN_ERR_HIP, srcnn_last_error(ctx));
        return h;
    }
};

template <class T>
inline std::vector<DevicePlane<T>> DevicePlanes(int n, int cols, int rows, Session &s = Session::thread_default())
{
    if (n <= 0 || cols <= 0 || rows <= 0) throw Error(SRCNN_ERR_INVALID, "DevicePlanes: bad size");
    const std::size_t pitch = static_cast<std::size_t>(rows) * cols;
    void *p = nullptr;
    s.check(srcnn_dev_alloc(s.get(), sizeof(T) * pitch * n, &p));
    srcnn_ctx *ctx = s.get();
    // (the deleter holds the Session's context alive: see Session::keep_alive)
    std::shared_ptr<void> owner(p, [ctx, keep = s.keep_alive()](void *q) { (void)srcnn_dev_free(ctx, q); });
    std::vector<DevicePlane<T>> v(static_cast<std::size_t>(n));
    for (int k = 0; k < n; ++k) {
        v[k].rows = rows;
        v[k].cols = cols;
        v[k].step = sizeof(T) * static_cast<std::size_t>(cols);
        v[k].data = static_cast<unsigned char *>(p) + sizeof(T) * pitch * k;
        v[k].owner = owner;
        v[k].ctx = ctx;
    }
    return v;
}

namespace detail {
// plane k of the vector at data[0] + k * pitch: the layout the device kernels address (one scalar base per plane pair)
template <class T>
inline std::size_t uniform_pitch(const std::vector<DevicePlane<T>> &v, const char *who)
{
    const std::size_t pitch = v.size() > 1 ? static_cast<std::size_t>(v[1].data - v[0].data) / sizeof(T)
                                           : static_cast<std::size_t>(v[0].rows) * (v[0].step / sizeof(T));
    for (std::size_t k = 0; k < v.size(); ++k)
        if (!v[k].data || v[k].rows != v[0].rows || v[k].cols != v[0].cols || v[k].step != v[0].step || v[k].ctx != v[0].ctx ||
            v[k].data != v[0].data + sizeof(T) * pitch * k)
            throw Error(SRCNN_ERR_INVALID, std::string(who) + ": the device planes must come from one DevicePlanes() call");
    return pitch;
}
}  // namespace detail

template <class MatU8>
inline void Convolution99x11(MatU8 &src, std::vector<DevicePlane<float>> &dst,
                             const float kernel99[SRCNN_CONV1_FILTERS][9][9], const float bias99[SRCNN_CONV1_FILTERS],
                             const float kernel11[SRCNN_CONV2_FILTERS][SRCNN_CONV1_FILTERS],
                             const float bias11[SRCNN_CONV2_FILTERS])
{
    if (dst.size() != SRCNN_CONV2_FILTERS) throw Error(SRCNN_ERR_INVALID, "Convolution99x11: need 32 planes");
    const std::size_t pitch = detail::uniform_pitch(dst, "Convolution99x11");
    srcnn_ctx *c = dst[0].ctx;
    const int rc = srcnn_conv99x11_to_dev(c, detail::ptr<const std::uint8_t>(src), detail::stride<std::uint8_t>(src),
                                          detail::ptr<float>(dst[0]), detail::stride<float>(dst[0]), pitch, src.cols, src.rows,
                                          &kernel99[0][0][0], bias99, &kernel11[0][0], bias11);      // dims from src: :262-263
    if (rc != SRCNN_OK) throw Error(rc, srcnn_last_error(c));
}

template <class MatU8>
inline void Convolution55(std::vector<DevicePlane<float>> &src, MatU8 &dst, const float kernel[32][5][5], float bias)
{
    if (src.size() != SRCNN_CONV2_FILTERS) throw Error(SRCNN_ERR_INVALID, "Convolution55: need 32 planes");
    const std::size_t pitch = detail::uniform_pitch(src, "Convolution55");
    srcnn_ctx *c = src[0].ctx;
    const int rc = srcnn_conv55_from_dev(c, detail::ptr<const float>(src[0]), detail::stride<float>(src[0]), pitch,
                                         detail::ptr<std::uint8_t>(dst), detail::stride<std::uint8_t>(dst), dst.cols, dst.rows,
                                         &kernel[0][0][0], bias);                                     // dims from dst: :191-192
    if (rc != SRCNN_OK) throw Error(rc, srcnn_last_error(c));
}

// The whole conv path of src/srcnn.cpp:602-627 in one fused kernel.
template <class MatU8>
inline void ForwardY(MatU8 &src, MatU8 &dst, const float kernel99[64][9][9], const float bias99[64],
                     const float kernel11[32][64], const float bias11[32], const float kernel55[32][5][5],
                     float bias55)
{
    Session &s = Session::thread_default();
    s.check(srcnn_set_weights(s.get(), &kernel99[0][0][0], bias99, &kernel11[0][0], bias11,
                              &kernel55[0][0][0], bias55));
    s.check(srcnn_forward_y(s.get(), detail::ptr<const std::uint8_t>(src), detail::stride<std::uint8_t>(src),
                            detail::ptr<std::uint8_t>(dst), detail::stride<std::uint8_t>(dst), src.cols,
                            src.rows, nullptr, 0));
}

// ---- more than one GPU from one host (srcnn_amd.h "several GPUs") -----------------------
// One context per entry of `devices` (a device may appear more than once); the model is uploaded to each.
class SessionSet {
public:
    explicit SessionSet(const std::vector<int> &devices)
    {
        for (int d : devices) {
            srcnn_ctx *c = nullptr;
            const int rc = srcnn_create(&c, d);
            if (rc != SRCNN_OK) {
                for (srcnn_ctx *p : ctxs_) srcnn_destroy(p);
                throw Error(rc, "srcnn_create failed (a gfx950 GPU is required)");
            }
            ctxs_.push_back(c);
        }
        if (ctxs_.empty()) throw Error(SRCNN_ERR_INVALID, "SessionSet: no devices");
    }
    ~SessionSet() { for (srcnn_ctx *c : ctxs_) srcnn_destroy(c); }
    SessionSet(const SessionSet &) = delete;
    SessionSet &operator=(const SessionSet &) = delete;
    int size() const { return (int)ctxs_.size(); }
    srcnn_ctx *const *data() const { return ctxs_.data(); }
    void synchronize() const
    {
        for (srcnn_ctx *c : ctxs_) check(srcnn_synchronize(c));
    }
    void set_weights(const float kernel99[64][9][9], const float bias99[64], const float kernel11[32][64],
                     const float bias11[32], const float kernel55[32][5][5], float bias55)
    {
        for (srcnn_ctx *c : ctxs_) {
            const int rc = srcnn_set_weights(c, &kernel99[0][0][0], bias99, &kernel11[0][0], bias11, &kernel55[0][0][0], bias55);
            if (rc != SRCNN_OK) throw Error(rc, srcnn_last_error(c));
        }
    }
    void check(int rc) const
    {
        if (rc == SRCNN_OK) return;
        std::string msg;
        for (srcnn_ctx *c : ctxs_) msg += std::string(msg.empty() ? "" : " | ") + srcnn_last_error(c);
        throw Error(rc, msg);
    }

private:
    std::vector<srcnn_ctx *> ctxs_;
};

// ONE plane row-striped over the set's GPUs (6 halo rows per boundary device to device): same bytes as ForwardY.
template <class MatU8>
inline void ForwardYStriped(SessionSet &set, MatU8 &src, MatU8 &dst)
{
    set.check(srcnn_forward_y_striped(set.data(), set.size(), detail::ptr<const std::uint8_t>(src),
                                      detail::stride<std::uint8_t>(src), detail::ptr<std::uint8_t>(dst),
                                      detail::stride<std::uint8_t>(dst), src.cols, src.rows));
}

// A stream of equally sized planes, contiguous frame ranges per GPU, no collective.
template <class MatU8>
inline void ForwardYFrames(SessionSet &set, std::vector<MatU8> &src, std::vector<MatU8> &dst)
{
    if (src.empty() || src.size() != dst.size()) throw Error(SRCNN_ERR_INVALID, "ForwardYFrames: need as many outputs as inputs");
    std::vector<const std::uint8_t *> in(src.size());
    std::vector<std::uint8_t *> out(src.size());
    for (std::size_t i = 0; i < src.size(); ++i) {
        if (src[i].rows != src[0].rows || src[i].cols != src[0].cols || dst[i].rows != src[0].rows || dst[i].cols != src[0].cols ||
            detail::stride<std::uint8_t>(src[i]) != detail::stride<std::uint8_t>(src[0]) ||
            detail::stride<std::uint8_t>(dst[i]) != detail::stride<std::uint8_t>(dst[0]))
            throw Error(SRCNN_ERR_INVALID, "ForwardYFrames: frames must share size and row stride");
        in[i] = detail::ptr<const std::uint8_t>(src[i]);
        out[i] = detail::ptr<std::uint8_t>(dst[i]);
    }
    set.check(srcnn_forward_y_frames_multi(set.data(), set.size(), in.data(), detail::stride<std::uint8_t>(src[0]), out.data(),
                                           detail::stride<std::uint8_t>(dst[0]), src[0].cols, src[0].rows, (int)src.size()));
}

// DEVICE-resident planes of a stream (DevicePlane<unsigned char>: `data` a device address on the GPU of the session that will
// run it) alternately on the sessions of the set, used as LANES: plane f on session f mod size().  `SessionSet lanes({0, 0})` is
// two lanes of GPU 0 -- the next plane's kernel fills the compute units the previous plane's slowest workgroups leave idle:
// what small planes lose most (576x576: 0.60 -> 0.75 of the f32 MFMA peak).  Asynchronous: synchronize() the set to wait.
template <class PlaneU8>
inline void ForwardYLanes(SessionSet &set, std::vector<PlaneU8> &d_src, std::vector<PlaneU8> &d_dst)
{
    if (d_src.empty() || d_src.size() != d_dst.size()) throw Error(SRCNN_ERR_INVALID, "ForwardYLanes: need as many outputs as inputs");
    std::vector<const std::uint8_t *> in(d_src.size());
    std::vector<std::uint8_t *> out(d_src.size());
    for (std::size_t i = 0; i < d_src.size(); ++i) {
        if (d_src[i].rows != d_src[0].rows || d_src[i].cols != d_src[0].cols || d_dst[i].rows != d_src[0].rows || d_dst[i].cols != d_src[0].cols ||
            detail::stride<std::uint8_t>(d_src[i]) != detail::stride<std::uint8_t>(d_src[0]) ||
            detail::stride<std::uint8_t>(d_dst[i]) != detail::stride<std::uint8_t>(d_dst[0]))
            throw Error(SRCNN_ERR_INVALID, "ForwardYLanes: planes must share size and row stride");
        in[i] = detail::ptr<const std::uint8_t>(d_src[i]);
        out[i] = detail::ptr<std::uint8_t>(d_dst[i]);
    }
    set.check(srcnn_forward_y_lanes_dev(set.data(), set.size(), in.data(), detail::stride<std::uint8_t>(d_src[0]), out.data(),
                                        detail::stride<std::uint8_t>(d_dst[0]), d_src[0].cols, d_src[0].rows, (int)d_src.size()));
}

// A stream of equally sized LARGE planes, each row-striped over the GPUs of the set, pipelined: uploads, kernels and downloads
// of neighbouring planes overlap, ordered across GPUs by events (srcnn_forward_y_striped_frames).
template <class MatU8>
inline void ForwardYStripedFrames(SessionSet &set, std::vector<MatU8> &src, std::vector<MatU8> &dst)
{
    if (src.empty() || src.size() != dst.size()) throw Error(SRCNN_ERR_INVALID, "ForwardYStripedFrames: need as many outputs as inputs");
    std::vector<const std::uint8_t *> in(src.size());
    std::vector<std::uint8_t *> out(src.size());
    for (std::size_t i = 0; i < src.size(); ++i) {
        if (src[i].rows != src[0].rows || src[i].cols != src[0].cols || dst[i].rows != src[0].rows || dst[i].cols != src[0].cols ||
            detail::stride<std::uint8_t>(src[i]) != detail::stride<std::uint8_t>(src[0]) ||
            detail::stride<std::uint8_t>(dst[i]) != detail::stride<std::uint8_t>(dst[0]))
            throw Error(SRCNN_ERR_INVALID, "ForwardYStripedFrames: planes must share size and row stride");
        in[i] = detail::ptr<const std::uint8_t>(src[i]);
        out[i] = detail::ptr<std::uint8_t>(dst[i]);
    }
    set.check(srcnn_forward_y_striped_frames(set.data(), set.size(), in.data(), detail::stride<std::uint8_t>(src[0]), out.data(),
                                             detail::stride<std::uint8_t>(dst[0]), src[0].cols, src[0].rows, (int)src.size()));
}

// The timed region of the reference's pipeline driver (src/srcnn.cpp:505-659) with the
// buffer-level shape of the sibling library's ProcessSRCNN (src/test.cpp:347-353):
// packed B,G,R bytes in, packed B,G,R bytes out at (int)(w*scale) x (int)(h*scale).
// Returns 0 and fills out/out_w/out_h; needs a model (pass the convdata.h tables).
inline int ProcessSRCNN(const unsigned char *bgr, unsigned w, unsigned h, float scale,
                        std::vector<unsigned char> &out, unsigned &out_w, unsigned &out_h,
                        const float kernel99[64][9][9], const float bias99[64], const float kernel11[32][64],
                        const float bias11[32], const float kernel55[32][5][5], float bias55)
{
    int ow = 0, oh = 0;
    if (!bgr || srcnn_scaled_size((int)w, (int)h, scale, &ow, &oh) != SRCNN_OK) return SRCNN_ERR_INVALID;
    Session &s = Session::thread_default();
    int rc = srcnn_set_weights(s.get(), &kernel99[0][0][0], bias99, &kernel11[0][0], bias11, &kernel55[0][0][0], bias55);
    if (rc != SRCNN_OK) return rc;
    out.assign((std::size_t)ow * oh * 3, 0);
    rc = srcnn_process_bgr(s.get(), bgr, 3 * (std::size_t)w, (int)w, (int)h, scale, out.data(), 3 * (std::size_t)ow);
    out_w = (unsigned)ow;
    out_h = (unsigned)oh;
    return rc;
}

}  // namespace srcnn
#endif  // SRCNN_AMD_HPP
