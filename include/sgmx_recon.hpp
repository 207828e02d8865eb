// sgmx_recon.hpp -- C++ adapter classes with the reference's own signatures on top of the C ABI
// (include/sgmx.h, libsgmx.so).  Header-only; a maintainer of ivankreso/stereo-vision replaces the
// bodies of reconstruction/base/stereo_sgm.{h,cc} and sgm.{h,cc} by an include of this file and
// links recon_base against libsgmx.so; the mains under reconstruction/main build unchanged.
//
//   recon::StereoSGMParams, recon::StereoSGM::compute   stereo_sgm.h:15-21, 53-57; stereo_sgm.cc:15-188
//   recon::SGMParams,       recon::SGM::Compute         sgm.h:18-23, 61-65;      sgm.cc:14-174
//   SGM::LoadRepresentationFromFile                     sgm.h:378-410
//   recon::SGMStereo::{Compute,Set...}                  new_sgm/sgm_stereo.h:24-34; sgm_stereo.cc:16-101
//
// The image type is a template parameter with cv::Mat's member names (rows, cols, data, step,
// type(), create(rows, cols, type)), so the same code serves cv::Mat where OpenCV is installed
// (define SGMX_WITH_OPENCV before including) and sgmx::Mat below where it is not -- the CLI
// programs of this repository use the latter.  CV_8UC1 == 0 and CV_16UC1 == 2 are OpenCV's values.
#ifndef SGMX_RECON_HPP
#define SGMX_RECON_HPP

#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "sgmx.h"

#ifdef SGMX_WITH_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace sgmx {

enum { MAT_8U = 0, MAT_16U = 2 }; // == CV_8UC1, CV_16UC1

// Minimal owning single-channel image with cv::Mat's member names (not an OpenCV stand-in for
// building the reference: it exists so that this repository's own programs have an image type).
class Mat {
  public:
    int rows = 0, cols = 0;
    unsigned char *data = nullptr;
    size_t step = 0; // bytes per row

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    // view of caller-owned pixels (cv::Mat's user-data constructor); a copy of a view owns its pixels
    Mat(int r, int c, int type, void *ext, size_t step_bytes = 0)
    {
        if (type != MAT_8U && type != MAT_16U)
            throw std::invalid_argument("sgmx::Mat: 8-bit or 16-bit single channel only");
        type_ = type;
        rows = r;
        cols = c;
        step = step_bytes ? step_bytes : (size_t)c * elem_size();
        data = static_cast<unsigned char *>(ext);
    }
    void create(int r, int c, int type)
    {
        if (type != MAT_8U && type != MAT_16U)
            throw std::invalid_argument("sgmx::Mat: 8-bit or 16-bit single channel only");
        type_ = type;
        rows = r;
        cols = c;
        step = (size_t)c * elem_size();
        buf_.assign((size_t)r * step, 0);
        data = buf_.data();
    }
    int type() const { return type_; }
    bool empty() const { return rows == 0 || cols == 0; }
    size_t elem_size() const { return type_ == MAT_16U ? 2 : 1; }
    template <class T> T &at(int y, int x) { return *reinterpret_cast<T *>(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    template <class T> const T &at(int y, int x) const
    {
        return *reinterpret_cast<const T *>(data + (size_t)y * step + (size_t)x * sizeof(T));
    }
    // deep copies only (a moved-from or copied Mat must not alias another one's buffer)
    Mat(const Mat &o) { *this = o; }
    Mat &operator=(const Mat &o)
    {
        if (this != &o) {
            type_ = o.type_;
            rows = o.rows;
            cols = o.cols;
            step = (size_t)cols * elem_size();
            buf_.resize((size_t)rows * step);
            for (int y = 0; y < rows; y++)
                std::memcpy(buf_.data() + (size_t)y * step, o.data + (size_t)y * o.step, step);
            data = buf_.data();
        }
        return *this;
    }

  private:
    int type_ = MAT_8U;
    std::vector<unsigned char> buf_;
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// [H][W][K] float32 descriptor tensor of one image (the reference keeps a vector<vector<VectorXf>>).
struct DescriptorTensor {
    uint64_t H = 0, W = 0, K = 0;
    std::vector<float> v;
};

// SGM::LoadRepresentationFromFile (sgm.h:378-410): int32 dims (must be 3, else the reference does
// `throw 1`), uint64 size[3] = {H, W, K}, then H*W*K float32.  A short file is an error here (the
// reference reads garbage).
inline void LoadRepresentationFromFile(const std::string &path, DescriptorTensor *t)
{
    std::ifstream f(path, std::ios::binary);
    if (!f)
        throw Error(SGMX_ERR_INVALID_ARG, "cannot open descriptor file " + path);
    int32_t dims = 0;
    f.read(reinterpret_cast<char *>(&dims), sizeof(dims));
    if (!f || dims != 3)
        throw Error(SGMX_ERR_INVALID_ARG, "descriptor file " + path + ": dims != 3");
    uint64_t size[3] = {0, 0, 0};
    f.read(reinterpret_cast<char *>(size), sizeof(size));
    if (!f || size[0] == 0 || size[1] == 0 || size[2] == 0 || size[0] > (1u << 16) || size[1] > (1u << 16) ||
        size[2] > (1u << 16))
        throw Error(SGMX_ERR_INVALID_ARG, "descriptor file " + path + ": bad sizes");
    t->H = size[0];
    t->W = size[1];
    t->K = size[2];
    t->v.resize(size[0] * size[1] * size[2]);
    f.read(reinterpret_cast<char *>(t->v.data()), (std::streamsize)(t->v.size() * sizeof(float)));
    if ((size_t)f.gcount() != t->v.size() * sizeof(float))
        throw Error(SGMX_ERR_INVALID_ARG, "descriptor file " + path + ": truncated");
}

// Owns one sgmx_ctx.
class Context {
  public:
    Context() {}
    explicit Context(const sgmx_params &p) { open(p); }
    ~Context() { sgmx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    void open(const sgmx_params &p)
    {
        sgmx_destroy(ctx_);
        ctx_ = nullptr;
        const int rc = sgmx_create(&p, &ctx_);
        if (rc != SGMX_OK)
            throw Error(rc, std::string("sgmx_create: ") + sgmx_last_error(nullptr));
    }
    sgmx_ctx *get() const { return ctx_; }
    void check(int rc, const char *what) const
    {
        if (rc != SGMX_OK)
            throw Error(rc, std::string(what) + ": " + sgmx_last_error(ctx_));
    }

  private:
    sgmx_ctx *ctx_ = nullptr;
};

} // namespace sgmx

namespace recon {

#ifdef SGMX_WITH_OPENCV
typedef cv::Mat DefaultMat;
#else
typedef sgmx::Mat DefaultMat;
#endif

// stereo_sgm.h:15-21
struct StereoSGMParams {
    int disp_range;
    int window_sz;
    int penalty1;
    int penalty2;
};

// recon::StereoSGM (stereo_sgm.h:53-57), COST_CENSUS build.
class StereoSGM {
  public:
    // The reference constructor takes the four-int struct only; device, num_paths (8, or the
    // axis-aligned 4) and max_batch are extensions with the reference's behaviour as default.
    explicit StereoSGM(StereoSGMParams &p, int device = 0, int num_paths = 8, int max_batch = 1) : params_(p)
    {
        sgmx_params q;
        std::memset(&q, 0, sizeof(q));
        q.disp_range = p.disp_range;
        q.window_sz = p.window_sz;
        q.penalty1 = p.penalty1;
        q.penalty2 = p.penalty2;
        q.num_paths = num_paths;
        q.device = device;
        q.max_batch = max_batch;
        ctx_.open(q);
    }

    // void compute(cv::Mat& left, cv::Mat& right, cv::Mat& disp) (stereo_sgm.h:57, stereo_sgm.cc:15-188):
    // 8-bit single-channel rectified pair in, CV_16U disparity out (256 * d, 0 = invalid).  The
    // reference allocates disp (stereo_sgm.h:280) and has undefined behaviour on other inputs; here
    // they are errors.
    template <class MatT> void compute(MatT &left, MatT &right, MatT &disp)
    {
        if (left.type() != sgmx::MAT_8U || right.type() != sgmx::MAT_8U || left.rows != right.rows ||
            left.cols != right.cols)
            throw sgmx::Error(SGMX_ERR_INVALID_ARG, "StereoSGM::compute: want two 8-bit single-channel images of equal size");
        disp.create(left.rows, left.cols, sgmx::MAT_16U);
        ctx_.check(sgmx_compute_census_u8(ctx_.get(), left.data, right.data, left.cols, left.rows, (int)(size_t)left.step,
                                          (int)(size_t)right.step, reinterpret_cast<uint16_t *>(disp.data),
                                          (int)(size_t)disp.step),
                   "sgmx_compute_census_u8");
    }

    // n dense pairs back to back (what scripts/run_sgm_batch_census.py:27-34 loops over), one call.
    void compute_batch(int n, const uint8_t *left, const uint8_t *right, int W, int H, uint16_t *disp)
    {
        ctx_.check(sgmx_compute_census_u8_batch(ctx_.get(), n, left, right, W, H, disp), "sgmx_compute_census_u8_batch");
    }

    // cv::GaussianBlur(img, img, cv::Size(3,3), sigma), the callers' pre-filter
    // (dense_stereo/main.cc:83-85, stereo_eval/main.cc:36-38); in place.
    template <class MatT> void gaussian_blur3(MatT &img, double sigma)
    {
        if (img.type() != sgmx::MAT_8U)
            throw sgmx::Error(SGMX_ERR_INVALID_ARG, "gaussian_blur3: 8-bit single channel only");
        ctx_.check(sgmx_gaussian_blur3_u8(ctx_.get(), img.data, img.cols, img.rows, (int)(size_t)img.step, sigma, img.data,
                                          (int)(size_t)img.step),
                   "sgmx_gaussian_blur3_u8");
    }

    // page-locked host memory for compute_batch buffers (sgmx_host_alloc / sgmx_host_free)
    void *host_alloc(size_t bytes)
    {
        void *p = nullptr;
        ctx_.check(sgmx_host_alloc(ctx_.get(), bytes, &p), "sgmx_host_alloc");
        return p;
    }
    void host_free(void *p) { (void)sgmx_host_free(ctx_.get(), p); }

    // The callers' cv::GaussianBlur(img, img, cv::Size(3,3), sigma) (dense_stereo/main.cc:83-85,
    // stereo_eval/main.cc:36-38) fused in front of the census transform of every compute call (extension).
    void set_preblur(double sigma, bool enabled = true)
    {
        ctx_.check(sgmx_set_preblur3(ctx_.get(), enabled ? 1 : 0, sigma), "sgmx_set_preblur3");
    }

    sgmx_ctx *handle() const { return ctx_.get(); }
    const StereoSGMParams &params() const { return params_; }

  private:
    StereoSGMParams params_;
    sgmx::Context ctx_;
};

// sgm.h:18-23
struct SGMParams {
    int disp_range;
    int window_sz;
    int penalty1;
    int penalty2;
};

// recon::SGM (sgm.h:61-65), COST_CNN build.
class SGM {
  public:
    explicit SGM(SGMParams &p, int device = 0, int num_paths = 8) : params_(p)
    {
        sgmx_params q;
        std::memset(&q, 0, sizeof(q));
        q.disp_range = p.disp_range;
        q.window_sz = 1; // unused by the float path (sgm.cc never reads it)
        q.penalty1 = p.penalty1;
        q.penalty2 = p.penalty2;
        q.num_paths = num_paths;
        q.device = device;
        q.max_batch = 1;
        ctx_.open(q);
    }

    // cv::Mat Compute(const std::string left_desc_path, const std::string right_desc_path)
    // (sgm.h:65, sgm.cc:14-174): H x W CV_16U, no padding (sgm.h:291).
    template <class MatT = DefaultMat> MatT Compute(const std::string left_desc_path, const std::string right_desc_path)
    {
        sgmx::DescriptorTensor l, r;
        sgmx::LoadRepresentationFromFile(left_desc_path, &l);
        sgmx::LoadRepresentationFromFile(right_desc_path, &r);
        return ComputeFromDescriptors<MatT>(l, r);
    }

    // Compute() for descriptor tensors the caller has already loaded (a batch driver reads the next pair's
    // files while this pair is on the GPU).
    template <class MatT = DefaultMat> MatT ComputeFromDescriptors(const sgmx::DescriptorTensor &l, const sgmx::DescriptorTensor &r)
    {
        if (l.H != r.H || l.W != r.W || l.K != r.K)
            throw sgmx::Error(SGMX_ERR_INVALID_ARG, "SGM::Compute: left and right descriptor tensors differ in size");
        MatT disp;
        disp.create((int)l.H, (int)l.W, sgmx::MAT_16U);
        std::vector<uint16_t> dense((size_t)l.H * l.W);
        ctx_.check(sgmx_compute_desc_f32(ctx_.get(), l.v.data(), r.v.data(), (int)l.W, (int)l.H, (int)l.K, dense.data()),
                   "sgmx_compute_desc_f32");
        for (uint64_t y = 0; y < l.H; y++)
            std::memcpy(disp.data + y * (size_t)disp.step, dense.data() + y * l.W, l.W * sizeof(uint16_t));
        return disp;
    }

    // The part of Compute after the data-cost loop (sgm.cc:76-84) for a caller that already holds
    // the [H][W][D] float volume (FLT_MAX where x < d, sgm.cc:55).
    template <class MatT = DefaultMat> MatT ComputeFromCost(const float *cost, int W, int H, int D)
    {
        MatT disp;
        disp.create(H, W, sgmx::MAT_16U);
        std::vector<uint16_t> dense((size_t)H * W);
        ctx_.check(sgmx_compute_cost_f32(ctx_.get(), cost, W, H, D, dense.data()), "sgmx_compute_cost_f32");
        for (int y = 0; y < H; y++)
            std::memcpy(disp.data + (size_t)y * (size_t)disp.step, dense.data() + (size_t)y * W, (size_t)W * sizeof(uint16_t));
        return disp;
    }

    sgmx_ctx *handle() const { return ctx_.get(); }

  private:
    SGMParams params_;
    sgmx::Context ctx_;
};

// recon::SGMStereo (reconstruction/base/new_sgm/sgm_stereo.h:11-81): the second float pipeline.  Same
// methods and defaults (kDisparityRange 256, kP1 3, kP2 40, kConsistencyThreshold 1; sgm_stereo.h:17-22).
// The disparity range of a data-cost volume comes from its file (sgm_stereo.cc:613-615), so the sgmx
// context is made on first use and remade when the range or the penalties change.
class SGMStereo {
  public:
    explicit SGMStereo(int device = 0) : device_(device) {}

    // sgm_stereo.cc:16-28
    void SetSmoothnessCostParameters(const double P1, const double P2)
    {
        if (P1 < 0 || P2 < 0)
            throw std::invalid_argument("[SGMStereo::SetSmoothnessCostParameters] smoothness penalty value is less than zero");
        if (P1 >= P2)
            throw std::invalid_argument("[SGMStereo::SetSmoothnessCostParameters] small value of smoothness penalty must be "
                                        "smaller than large penalty value");
        // the reference keeps them as float (sgm_stereo.h:63-64); sgmx_params holds ints, so fractional
        // penalties are refused instead of truncated
        // (NaN fails every comparison above and is caught here; the range check comes before any cast)
        if (!std::isfinite(P1) || !std::isfinite(P2) || P2 > 2147483647.0 || std::trunc(P1) != P1 || std::trunc(P2) != P2)
            throw std::invalid_argument("[SGMStereo::SetSmoothnessCostParameters] this engine takes whole-number penalties");
        P1_ = (int)P1;
        P2_ = (int)P2;
    }
    // sgm_stereo.cc:30-36
    void SetConsistencyThreshold(const int consistency_threshold)
    {
        if (consistency_threshold < 0)
            throw std::invalid_argument("[SGMStereo::SetConsistencyThreshold] threshold for LR consistency must be positive");
        consistency_threshold_ = consistency_threshold;
    }

    // void Compute(left_descriptors_path, right_descriptors_path, cv::Mat* disparity) (sgm_stereo.cc:38-50)
    template <class MatT>
    void Compute(const std::string left_descriptors_path, const std::string right_descriptors_path, MatT *disparity)
    {
        sgmx::DescriptorTensor l, r;
        sgmx::LoadRepresentationFromFile(left_descriptors_path, &l);
        sgmx::LoadRepresentationFromFile(right_descriptors_path, &r);
        if (l.H != r.H || l.W != r.W || l.K != r.K)
            throw std::invalid_argument("[SGMStereo::setImageSize] sizes of left and right images are different");
        sgmx::Context &c = context(disp_range_);
        std::vector<uint16_t> dense((size_t)l.H * l.W);
        c.check(sgmx_sgmstereo_desc_f32(c.get(), l.v.data(), r.v.data(), (int)l.W, (int)l.H, (int)l.K, consistency_threshold_,
                                        dense.data()),
                "sgmx_sgmstereo_desc_f32");
        fill(disparity, dense, (int)l.W, (int)l.H);
    }

    // void Compute(data_cost_path, cv::Mat* disparity) (sgm_stereo.cc:52-56): the file is the [H][W][D]
    // float left data cost in the descriptor container (LoadDataCostFromFile, :602-627).
    template <class MatT> void Compute(const std::string data_cost_path, MatT *disparity)
    {
        sgmx::DescriptorTensor t;
        sgmx::LoadRepresentationFromFile(data_cost_path, &t);
        sgmx::Context &c = context((int)t.K);
        std::vector<uint16_t> dense((size_t)t.H * t.W);
        c.check(sgmx_sgmstereo_cost_f32(c.get(), t.v.data(), (int)t.W, (int)t.H, (int)t.K, consistency_threshold_, dense.data()),
                "sgmx_sgmstereo_cost_f32");
        fill(disparity, dense, (int)t.W, (int)t.H);
    }

  private:
    sgmx::Context &context(int D)
    {
        if (!ctx_.get() || D != have_D_ || P1_ != have_P1_ || P2_ != have_P2_) {
            sgmx_params q;
            std::memset(&q, 0, sizeof(q));
            q.disp_range = D;
            q.window_sz = 1;
            q.penalty1 = P1_;
            q.penalty2 = P2_;
            q.device = device_;
            ctx_.open(q);
            have_D_ = D;
            have_P1_ = P1_;
            have_P2_ = P2_;
        }
        return ctx_;
    }
    template <class MatT> static void fill(MatT *disparity, const std::vector<uint16_t> &dense, int W, int H)
    {
        disparity->create(H, W, sgmx::MAT_16U);
        for (int y = 0; y < H; y++)
            std::memcpy(disparity->data + (size_t)y * (size_t)disparity->step, dense.data() + (size_t)y * W,
                        (size_t)W * sizeof(uint16_t));
    }

    int device_;
    int disp_range_ = 256, P1_ = 3, P2_ = 40, consistency_threshold_ = 1;
    int have_D_ = 0, have_P1_ = 0, have_P2_ = 0;
    sgmx::Context ctx_;
};

} // namespace recon
#endif // SGMX_RECON_HPP
