// opencv4_mat_shape.hpp -- TEST INFRASTRUCTURE (compile-only; nothing here is ever linked or run).
//
// OpenCV is absent from this image, so include/srcnn_amd.hpp's adapters have never met the real cv::Mat.  This header DECLARES,
// in namespace cv, the part of OpenCV 4's public cv::Mat interface the reference's conv path touches -- the public data members
// in the order and with the types <opencv2/core/mat.hpp> documents them (int flags, dims, rows, cols; uchar* data; ... MatSize
// size; MatStep step) and the handful of members src/srcnn.cpp:602-627 calls (create, size) -- so that the adapters are
// instantiated against those exact TYPES: `step` is a cv::MatStep OBJECT with operator size_t, `data` is uchar*, `rows` / `cols`
// are ints, `size` is a MatSize object with operator().  A restatement of an interface's shape, written for this test; no OpenCV
// source text.  tests/test_abi.py compiles the reference's two call sites against it with -fsyntax-only.
#pragma once
#include <cstddef>
#include <vector>

#define CV_8U 0
#define CV_32F 5

namespace cv {

typedef unsigned char uchar;

template <typename T>
struct Size_ {
    T width, height;
};
typedef Size_<int> Size;

struct MatSize {
    explicit MatSize(int *_p) : p(_p) {}
    Size operator()() const;
    const int &operator[](int i) const;
    int &operator[](int i);
    operator const int *() const;
    int *p;
};

struct MatStep {
    MatStep();
    explicit MatStep(std::size_t s);
    const std::size_t &operator[](int i) const;
    std::size_t &operator[](int i);
    operator std::size_t() const;
    MatStep &operator=(std::size_t s);
    std::size_t *p;
    std::size_t buf[2];

protected:
    MatStep &operator=(const MatStep &);
};

class MatAllocator;
struct UMatData;

class Mat {
public:
    Mat();
    Mat(const Mat &m);
    ~Mat();
    Mat &operator=(const Mat &m);
    void create(int rows, int cols, int type);
    void create(Size size, int type);
    bool empty() const;
    template <typename T>
    T &at(int row, int col);

    int flags;
    int dims;
    int rows, cols;
    uchar *data;
    const uchar *datastart;
    const uchar *dataend;
    const uchar *datalimit;
    MatAllocator *allocator;
    UMatData *u;
    MatSize size;
    MatStep step;
};

}  // namespace cv
