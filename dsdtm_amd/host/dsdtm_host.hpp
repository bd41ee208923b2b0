// dsdtm_host.hpp — C++ host side above the C ABI (include/dsdtm_amd.h), mirroring the reference's
// class interface for the hot path: same class names, constructor arguments, method names,
// argument order, return values and error behaviour as
//   DSDTM::Sprase_ImgAlign      (reference include/Sprase_ImageAlign.h:20-69)
//   DSDTM::Feature_Alignment    (reference include/Feature_alignment.h:23-99, the static
//                                Align2DGaussNewton and its batch form)
//   DSDTM::Feature_detector     (reference include/Feature_detection.h:35-75)
// over dependency-free stand-ins for the data the path reads (Frame / Feature / Camera / SE3): the
// reference's own types need OpenCV, Eigen and Sophus, which this build image does not have
// (INTEGRATION.md shows the adapter against the real types). All compute happens in
// libdsdtm_amd.so (HIP, gfx950); there is no CPU path here.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dsdtm_amd.h"

namespace DSDTM {

static const int mHalf_PatchSize = 4;   // include/Feature_alignment.h:21

// Config::Get<int>("Camera.Min_fts") etc. — only the keys the path reads
struct Config {
    static int& Min_fts() { static int v = 15; return v; }            // src/Sprase_ImageAlign.cpp:14
    static int& CellSize() { static int v = 25; return v; }           // src/Feature_detection.cpp:12
    static int& MaxPyraLevels() { static int v = 5; return v; }       // :13
    static int& Max_fts() { static int v = 200; return v; }           // :14
    static int& Min_dist() { static int v = 30; return v; }           // src/Frame.cpp:52
};

struct Image8 {                                   // cv::Mat CV_8UC1
    int cols = 0, rows = 0, step = 0;
    std::vector<uint8_t> data;
    Image8() = default;
    Image8(int w, int h) : cols(w), rows(h), step(w), data((size_t)w * h) {}
};

struct SE3 {                                      // Sophus::SE3 as [R|t] 3x4 row-major
    std::array<double, 12> m{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};
};

struct Camera {                                   // include/Camera.h:138-142 (float members)
    float mfx = 0, mfy = 0, mcx = 0, mcy = 0, mf = 0;
    int mwidth = 0, mheight = 0;
};
typedef std::shared_ptr<Camera> CameraPtr;

struct Feature {                                  // include/Feature.h:16-36 (+ the map point position)
    float mpx_x = 0, mpx_y = 0;
    int mlevel = 0;
    bool mbInitial = false;
    std::array<double, 3> mNormal{{0, 0, 0}};
    std::array<double, 3> mMptPose{{0, 0, 0}};    // Mpt->Get_Pose()
};

namespace detail { inline dsdtm_ctx* ctx(); }

struct Frame {                                    // include/Frame.h (what the path touches)
    CameraPtr mCamera;
    std::vector<Image8> mvImg_Pyr;
    std::vector<Feature> mvFeatures;
    SE3 mT_c2w;
    dsdtm_frame* mDev = nullptr;                  // the image pyramid, resident on the device (optional)
    const SE3& Get_Pose() const { return mT_c2w; }
    void Set_Pose(const SE3& T) { mT_c2w = T; }   // src/Frame.cpp:167-174
    // Frame::ComputeImagePyramid (src/Frame.cpp:74-81) on the device: level 0 crosses PCIe once, the
    // other levels are built by the library's bit-exact pyrDown and stay there for every Run.
    void ComputeImagePyramidOnDevice(int levels) {
        if (mDev || mvImg_Pyr.empty()) return;
        const Image8& l0 = mvImg_Pyr[0];
        if (dsdtm_frame_create_from_image(detail::ctx(), l0.data.data(), l0.cols, l0.rows, l0.step, levels, &mDev) != DSDTM_OK)
            throw std::runtime_error(std::string("dsdtm_frame_create_from_image: ") + dsdtm_last_error(detail::ctx()));
    }
    Frame() = default;
    Frame(const Frame&) = delete;
    Frame& operator=(const Frame&) = delete;
    ~Frame() { dsdtm_frame_destroy(nullptr, mDev); }
};
typedef std::shared_ptr<Frame> FramePtr;

namespace detail {
inline dsdtm_ctx* ctx() {                         // one context per calling thread (tracking thread)
    struct Holder {
        dsdtm_ctx* c = nullptr;
        Holder() {
            if (dsdtm_create(0, &c) != DSDTM_OK)
                throw std::runtime_error(std::string("dsdtm_create: ") + dsdtm_last_error(nullptr));
        }
        ~Holder() { dsdtm_destroy(c); }
    };
    static thread_local Holder h;
    return h.c;
}
inline dsdtm_pyramid to_pyr(const std::vector<Image8>& v) {
    dsdtm_pyramid p{};
    p.levels = (int)v.size();
    for (int l = 0; l < p.levels; ++l) {
        p.data[l] = v[l].data.data(); p.width[l] = v[l].cols; p.height[l] = v[l].rows; p.stride[l] = v[l].step;
    }
    return p;
}
}  // namespace detail

class Sprase_ImgAlign {
public:
    Sprase_ImgAlign(int tMaxLevel, int tMinLevel, int tMaxIterators)
        : mnMaxLevel(tMaxLevel), mnMinLevel(tMinLevel), mnMaxIterators(tMaxIterators),
          mnMinfts(Config::Min_fts()) {}

    // int Run(FramePtr tCurFrame, FramePtr tRefFrame) — include/Sprase_ImageAlign.h:29
    int Run(FramePtr tCurFrame, FramePtr tRefFrame) {
        const std::vector<Feature>& f = tRefFrame->mvFeatures;
        const int n = (int)f.size();
        std::vector<float> px(2 * (size_t)n);
        std::vector<double> bearing(3 * (size_t)n), pw(3 * (size_t)n);
        std::vector<uint8_t> ini((size_t)n);
        for (int i = 0; i < n; ++i) {
            px[2 * i] = f[i].mpx_x; px[2 * i + 1] = f[i].mpx_y; ini[i] = f[i].mbInitial ? 1 : 0;
            for (int k = 0; k < 3; ++k) { bearing[3 * i + k] = f[i].mNormal[k]; pw[3 * i + k] = f[i].mMptPose[k]; }
        }
        const Camera& c = *tRefFrame->mCamera;
        const dsdtm_camera cam{c.mfx, c.mfy, c.mcx, c.mcy, c.mf, c.mwidth, c.mheight};
        const dsdtm_pyramid ref = detail::to_pyr(tRefFrame->mvImg_Pyr), cur = detail::to_pyr(tCurFrame->mvImg_Pyr);
        SE3 Tc = tCurFrame->Get_Pose();
        const dsdtm_align_params prm{mnMaxLevel, mnMinLevel, mnMaxIterators, mnMinfts};
        int n_tracked = 0;
        const int rc = (tRefFrame->mDev && tCurFrame->mDev)
            ? dsdtm_sparse_align_frames(detail::ctx(), tRefFrame->mDev, tCurFrame->mDev, &cam, px.data(), bearing.data(),
                                        pw.data(), ini.data(), n, tRefFrame->Get_Pose().m.data(), Tc.m.data(), &prm,
                                        &n_tracked, &last_stats)
            : dsdtm_sparse_align(detail::ctx(), &ref, &cur, &cam, px.data(), bearing.data(), pw.data(),
                                 ini.data(), n, tRefFrame->Get_Pose().m.data(), Tc.m.data(), &prm,
                                 &n_tracked, &last_stats);
        if (rc != DSDTM_OK) throw std::runtime_error(std::string("dsdtm_sparse_align: ") + dsdtm_last_error(detail::ctx()));
        if (n < mnMinfts) {                                  // src/Sprase_ImageAlign.cpp:34-38
            std::fprintf(stderr, "Too few features to track\n");
            return 0;                                        // pose untouched
        }
        tCurFrame->Set_Pose(Tc);                             // :57
        return n_tracked;                                    // :59
    }

    dsdtm_align_stats last_stats{};

protected:
    int mnMaxLevel, mnMinLevel, mnMaxIterators, mnMinfts;
};

class Feature_Alignment {
public:
    explicit Feature_Alignment(CameraPtr camera) : mCam(camera) {}

    // static bool Align2DGaussNewton(const cv::Mat&, uchar*, uchar*, int, Eigen::Vector2d&)
    // — include/Feature_alignment.h:85. tCurPx is written back also on failure (:414).
    static bool Align2DGaussNewton(const Image8& tCurImg, uint8_t* tPatch_WithBoarder, uint8_t* tPatch,
                                   int MaxIters, double tCurPx[2]) {
        std::vector<Image8> one(1);
        dsdtm_pyramid cur{};
        cur.levels = 1; cur.data[0] = tCurImg.data.data(); cur.width[0] = tCurImg.cols; cur.height[0] = tCurImg.rows;
        cur.stride[0] = tCurImg.step;
        int32_t level = 0;
        uint8_t ok = 0;
        const int rc = dsdtm_align2d_batch(detail::ctx(), &cur, tPatch_WithBoarder, tPatch, &level, tCurPx, &ok, MaxIters, 1);
        if (rc != DSDTM_OK) throw std::runtime_error(std::string("dsdtm_align2d_batch: ") + dsdtm_last_error(detail::ctx()));
        return ok != 0;
    }

    // all candidates of a frame at once (the speculative form SearchLocalPoints uses)
    static void Align2DGaussNewtonBatch(const std::vector<Image8>& tCurPyr, const uint8_t* borders, const uint8_t* patches,
                                        const int32_t* levels, double* px_xy, uint8_t* converged, int MaxIters, int m) {
        const dsdtm_pyramid cur = detail::to_pyr(tCurPyr);
        const int rc = dsdtm_align2d_batch(detail::ctx(), &cur, borders, patches, levels, px_xy, converged, MaxIters, m);
        if (rc != DSDTM_OK) throw std::runtime_error(std::string("dsdtm_align2d_batch: ") + dsdtm_last_error(detail::ctx()));
    }

private:
    CameraPtr mCam;
};

// cv::circle(img, center, radius, 0, -1) for an 8-bit mask: OpenCV 2.4 drawing.cpp Circle() (midpoint
// algorithm, filled by horizontal spans, clipped to the image) — the detector's and the tracker's
// mask painting (src/Feature_detection.cpp:143, src/Frame.cpp:291)
inline void FillCircle(Image8& mask, int cx, int cy, int radius, uint8_t value) {
    auto hline = [&](int y, int x1, int x2) {
        if (y < 0 || y >= mask.rows) return;
        x1 = std::max(x1, 0); x2 = std::min(x2, mask.cols - 1);
        for (int x = x1; x <= x2; ++x) mask.data[(size_t)y * mask.step + x] = value;
    };
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hline(cy - dy, cx - dx, cx + dx); hline(cy + dy, cx - dx, cx + dx);
        hline(cy - dx, cx - dy, cx + dy); hline(cy + dx, cx - dy, cx + dy);
        dy++; err += plus; plus += 2;
        const int m = (err <= 0) - 1;
        err -= minus & m; dx += m; minus -= m & 2;
    }
}

struct Corner {                                   // include/Feature_detection.h:19-33
    int x, y, level;
    float score, angle;
    Corner(int _x, int _y, float _score, int _level, float _angle) : x(_x), y(_y), level(_level), score(_score), angle(_angle) {}
    bool operator<(const Corner& c) const { return c.score < score; }
};
typedef std::vector<Corner> Corners;

class Feature_detector {
public:
    Feature_detector(int width, int height)           // src/Feature_detection.cpp:10-21 (Camera.width / Camera.height)
        : mImg_height(height), mImg_width(width), mCell_size(Config::CellSize()), mPyr_levels(Config::MaxPyraLevels()),
          mMax_fts(Config::Max_fts()) {
        mGrid_rows = (int)std::ceil(1.0 * mImg_height / mCell_size);
        mGrid_cols = (int)std::ceil(1.0 * mImg_width / mCell_size);
        mvGrid_occupy.assign((size_t)mGrid_rows * mGrid_cols, 0);
    }

    void Set_ExistingFeatures(const std::vector<Feature>& features) {          // :40-47
        std::fill(mvGrid_occupy.begin(), mvGrid_occupy.end(), 0);
        for (const Feature& f : features)
            mvGrid_occupy[(size_t)((int)(f.mpx_y / mCell_size) * mGrid_cols + (int)(f.mpx_x / mCell_size))] = 1;
    }
    void ResetGrid() { std::fill(mvGrid_occupy.begin(), mvGrid_occupy.end(), 0); }   // :64-67

    // void detect(Frame* frame, const double detection_threshold, const bool tFirst = true) — :69-154.
    // The image work (:76-108) is one library call; the order-dependent rest follows the reference.
    void detect(Frame* frame, const double detection_threshold, const bool /*tFirst*/ = true) {
        if ((int)frame->mvFeatures.size() >= mMax_fts) return;                                   // :71-72
        const size_t G = mvGrid_occupy.size();
        std::vector<float> score(G);
        std::vector<int32_t> cx(G), cy(G), cl(G);
        const int levels = std::min<int>(mPyr_levels, (int)frame->mvImg_Pyr.size());
        const dsdtm_detect_params prm{mCell_size, mGrid_cols, mGrid_rows, levels, 20, (float)detection_threshold};
        int rc;
        if (frame->mDev) {
            rc = dsdtm_detect_cells_frame(detail::ctx(), frame->mDev, mvGrid_occupy.data(), &prm, score.data(), cx.data(), cy.data(), cl.data());
        } else {
            const dsdtm_pyramid pyr = detail::to_pyr(frame->mvImg_Pyr);
            rc = dsdtm_detect_cells(detail::ctx(), &pyr, mvGrid_occupy.data(), &prm, score.data(), cx.data(), cy.data(), cl.data());
        }
        if (rc != DSDTM_OK) throw std::runtime_error(std::string("dsdtm_detect_cells: ") + dsdtm_last_error(detail::ctx()));
        Corners corners;
        corners.reserve(G);
        for (size_t k = 0; k < G; ++k) corners.push_back(Corner(cx[k], cy[k], score[k], cl[k], 0.0f));
        std::stable_sort(corners.begin(), corners.end());                                        // :110
        Image8 mask(mImg_width, mImg_height);                                                    // src/Frame.cpp:64
        std::fill(mask.data.begin(), mask.data.end(), 255);
        if (!frame->mvFeatures.empty())                                                          // :119-122, Frame::Set_Mask
            for (const Feature& f : frame->mvFeatures)
                if (f.mbInitial) FillCircle(mask, (int)std::lround(f.mpx_x), (int)std::lround(f.mpx_y), Config::Min_dist(), 0);
        for (const Corner& c : corners) {                                                        // :124-150
            if (c.score > 20) {
                if (mask.data[(size_t)c.y * mask.step + c.x] != 255) continue;                   // :139
                Feature f;
                f.mpx_x = (float)c.x; f.mpx_y = (float)c.y; f.mlevel = c.level;                   // :142 (Add_Feature(.., 0): no bearing)
                frame->mvFeatures.push_back(f);
                FillCircle(mask, c.x, c.y, mCell_size, 0);                                       // :143
            }
            if ((int)frame->mvFeatures.size() >= mMax_fts) break;                                // :148-149
        }
        ResetGrid();                                                                             // :152
    }

    int mImg_height, mImg_width, mCell_size, mPyr_levels, mGrid_rows, mGrid_cols, mMax_fts;
    std::vector<uint8_t> mvGrid_occupy;
};

}  // namespace DSDTM
