// ncnn/mat.h -- the part of ncnn::Mat (reference src/ncnn/mat.h, vendored there from Tencent/ncnn) that FeatherCNN's PUBLIC API
// touches: feather::Net::FeedInput(const char*, ncnn::Mat&) and Extract(std::string, ncnn::Mat&) (reference src/net.h:44,50).
// It exists so that application code written against the reference compiles unchanged against include/feather/net.h:
//
//     ncnn::Mat in(w, h, c);            // host memory, channel stride rounded up to 16 bytes (mat.h:288)
//     float* p = in.channel(0);         // fill it ...
//     net.FeedInput("data", in);  net.Forward();
//     ncnn::Mat out;  net.Extract("prob", out);   const float* prob = out.channel(0);
//
// Same include guard as the reference header: a program that already includes the real ncnn mat.h keeps using that one (the
// Net overloads only need data / w / h / c / cstep / elemsize / create() / channel()).
// Host-side fp32 container only: no allocators, no packing, no pixel conversion, no SIMD -- none of that is on the hot path,
// which runs on device blobs.
#ifndef NCNN_MAT_H
#define NCNN_MAT_H

#include <stddef.h>
#include <stdlib.h>
#include <string.h>

namespace ncnn
{

class Mat
{
  public:
    Mat() : data(0), refcount(0), elemsize(0), dims(0), w(0), h(0), c(0), cstep(0) {}
    Mat(int w_, size_t elemsize_ = 4u) : data(0), refcount(0), elemsize(0), dims(0), w(0), h(0), c(0), cstep(0) { create(w_, elemsize_); }
    Mat(int w_, int h_, size_t elemsize_ = 4u) : data(0), refcount(0), elemsize(0), dims(0), w(0), h(0), c(0), cstep(0) { create(w_, h_, elemsize_); }
    Mat(int w_, int h_, int c_, size_t elemsize_ = 4u) : data(0), refcount(0), elemsize(0), dims(0), w(0), h(0), c(0), cstep(0)
    {
        create(w_, h_, c_, elemsize_);
    }
    // external data (not owned), mat.h:54
    Mat(int w_, int h_, int c_, void* data_, size_t elemsize_ = 4u)
        : data(data_), refcount(0), elemsize(elemsize_), dims(3), w(w_), h(h_), c(c_), cstep(align_size((size_t)w_ * h_ * elemsize_, 16) / elemsize_)
    {
    }
    Mat(const Mat& m) : data(m.data), refcount(m.refcount), elemsize(m.elemsize), dims(m.dims), w(m.w), h(m.h), c(m.c), cstep(m.cstep)
    {
        if (refcount) ++*refcount;
    }
    ~Mat() { release(); }
    Mat& operator=(const Mat& m)
    {
        if (this == &m) return *this;
        if (m.refcount) ++*m.refcount;
        release();
        data = m.data;
        refcount = m.refcount;
        elemsize = m.elemsize;
        dims = m.dims;
        w = m.w;
        h = m.h;
        c = m.c;
        cstep = m.cstep;
        return *this;
    }

    void create(int w_, size_t elemsize_ = 4u) { alloc(1, w_, 1, 1, elemsize_, (size_t)w_); }
    void create(int w_, int h_, size_t elemsize_ = 4u) { alloc(2, w_, h_, 1, elemsize_, (size_t)w_ * h_); }
    void create(int w_, int h_, int c_, size_t elemsize_ = 4u)
    {
        alloc(3, w_, h_, c_, elemsize_, align_size((size_t)w_ * h_ * elemsize_, 16) / elemsize_);
    }
    void release()
    {
        if (refcount && --*refcount == 0)
        {
            free(refcount); // one allocation: [refcount | padding to 64 B | data]
        }
        data = 0;
        refcount = 0;
        elemsize = 0;
        dims = w = h = c = 0;
        cstep = 0;
    }
    bool empty() const { return data == 0 || total() == 0; }
    size_t total() const { return cstep * c; }
    void fill(float v)
    {
        float* p = (float*)data;
        for (size_t i = 0, n = total(); i < n; ++i) p[i] = v;
    }
    Mat clone() const
    {
        Mat m;
        if (dims == 1) m.create(w, elemsize);
        else if (dims == 2) m.create(w, h, elemsize);
        else if (dims == 3) m.create(w, h, c, elemsize);
        if (total()) memcpy(m.data, data, total() * elemsize);
        return m;
    }

    // a 2-D view of channel q (shares the data, like the reference's, mat.h:430-440)
    Mat channel(int q)
    {
        Mat m;
        m.data = (unsigned char*)data + cstep * q * elemsize;
        m.elemsize = elemsize;
        m.dims = 2;
        m.w = w;
        m.h = h;
        m.c = 1;
        m.cstep = (size_t)w * h;
        return m;
    }
    const Mat channel(int q) const { return const_cast<Mat*>(this)->channel(q); }
    float* row(int y) { return (float*)data + (size_t)w * y; }
    const float* row(int y) const { return (const float*)data + (size_t)w * y; }
    template <typename T>
    operator T*()
    {
        return (T*)data;
    }
    template <typename T>
    operator const T*() const
    {
        return (const T*)data;
    }
    float& operator[](int i) { return ((float*)data)[i]; }
    const float& operator[](int i) const { return ((const float*)data)[i]; }

    void* data;
    int* refcount;
    size_t elemsize;
    int dims;
    int w, h, c;
    size_t cstep;

  private:
    static size_t align_size(size_t sz, size_t n) { return (sz + n - 1) & ~(n - 1); }
    void alloc(int dims_, int w_, int h_, int c_, size_t elemsize_, size_t cstep_)
    {
        if (dims == dims_ && w == w_ && h == h_ && c == c_ && elemsize == elemsize_ && refcount) return;
        release();
        elemsize = elemsize_;
        dims = dims_;
        w = w_;
        h = h_;
        c = c_;
        cstep = cstep_;
        const size_t bytes = align_size(total() * elemsize, 4);
        if (!bytes) return;
        void* raw = 0;
        if (posix_memalign(&raw, 64, 64 + bytes) != 0) raw = 0;
        if (!raw)
        {
            dims = w = h = c = 0;
            cstep = 0;
            return;
        }
        refcount = (int*)raw;
        *refcount = 1;
        data = (unsigned char*)raw + 64;
    }
};

} // namespace ncnn

#endif // NCNN_MAT_H
