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
#include <list>
#include <map>
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
    static float& LocalBAthreshhold() { static float v = 2.0f; return v; }   // src/Optimizer.cpp:22 (Config/default.yaml:94)
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

struct MapPoint;
struct Feature {                                  // include/Feature.h:16-36 (+ the map point position)
    float mpx_x = 0, mpx_y = 0;
    int mlevel = 0;
    bool mbInitial = false;
    std::array<double, 3> mNormal{{0, 0, 0}};
    std::array<double, 3> mMptPose{{0, 0, 0}};    // Mpt->Get_Pose() as Sprase_ImgAlign::Run snapshots it
    MapPoint* Mpt = nullptr;                      // include/Feature.h:33 (read by Optimizer::PoseOptimization)
};

namespace detail { inline dsdtm_ctx* ctx(); }

struct Frame {                                    // include/Frame.h (what the path touches)
    CameraPtr mCamera;
    std::vector<Image8> mvImg_Pyr;
    std::vector<Feature> mvFeatures;
    SE3 mT_c2w;
    Image8 mDynamicMask;                          // include/Frame.h mDynamicMask: moving-object mask (optional; empty = none)
    dsdtm_frame* mDev = nullptr;                  // the image pyramid, resident on the device (optional)
    const SE3& Get_Pose() const { return mT_c2w; }
    void Set_Pose(const SE3& T) { mT_c2w = T; }   // src/Frame.cpp:167-174
    std::array<double, 3> Get_CameraCnt() const { // mOw = -R^T t (src/Frame.cpp:170-173)
        const auto& m = mT_c2w.m;
        return {{-(m[0] * m[3] + m[4] * m[7] + m[8] * m[11]), -(m[1] * m[3] + m[5] * m[7] + m[9] * m[11]),
                 -(m[2] * m[3] + m[6] * m[7] + m[10] * m[11])}};
    }
    std::array<double, 2> World2Pixel(const std::array<double, 3>& P) const {   // src/Frame.cpp:318-323
        const auto& m = mT_c2w.m;
        const double x = m[0] * P[0] + m[1] * P[1] + m[2] * P[2] + m[3], y = m[4] * P[0] + m[5] * P[1] + m[6] * P[2] + m[7],
                     z = m[8] * P[0] + m[9] * P[1] + m[10] * P[2] + m[11];
        return {{(double)mCamera->mfx * x / z + (double)mCamera->mcx, (double)mCamera->mfy * y / z + (double)mCamera->mcy}};
    }
    // Frame::Add_Feature (src/Frame.cpp:83-92): mNormal = Pixel2Camera(mpx, 1.0).normalized(). The cv::Point2f
    // overload of Pixel2Camera (src/Camera.cpp:173-178) works in float; the Vector3d is normalised in double.
    void Add_Feature(Feature f, bool tbNormal = true) {
        if (tbNormal) {
            const float one = 1.0f;
            const double x = (double)((one * (f.mpx_x - mCamera->mcx)) / mCamera->mfx);
            const double y = (double)((one * (f.mpx_y - mCamera->mcy)) / mCamera->mfy);
            const double n = std::sqrt(x * x + y * y + 1.0 * 1.0);
            f.mNormal = {{x / n, y / n, 1.0 / n}};
        }
        mvFeatures.push_back(f);
    }
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
            for (int k = 0; k < 3; ++k) bearing[3 * i + k] = f[i].mNormal[k];
        }
        // :93 mvFeatures[i]->Mpt->Get_Pose() (features without a MapPoint object carry the position themselves)
        auto snapshot = [&]() {
            for (int i = 0; i < n; ++i) {
                const std::array<double, 3>& P = f[i].Mpt ? map_point_pose(f[i].Mpt) : f[i].mMptPose;
                for (int k = 0; k < 3; ++k) pw[3 * i + k] = P[k];
            }
        };
        const Camera& c = *tRefFrame->mCamera;
        const dsdtm_camera cam{c.mfx, c.mfy, c.mcx, c.mcy, c.mf, c.mwidth, c.mheight};
        const dsdtm_pyramid ref = detail::to_pyr(tRefFrame->mvImg_Pyr), cur = detail::to_pyr(tCurFrame->mvImg_Pyr);
        SE3 Tc = tCurFrame->Get_Pose();
        int n_tracked = 0;
        auto levels = [&](int max_level, int min_level, dsdtm_align_stats* st) {
            const dsdtm_align_params prm{max_level, min_level, mnMaxIterators, mnMinfts};
            const int rc = (tRefFrame->mDev && tCurFrame->mDev)
                ? dsdtm_sparse_align_frames(detail::ctx(), tRefFrame->mDev, tCurFrame->mDev, &cam, px.data(), bearing.data(),
                                            pw.data(), ini.data(), n, tRefFrame->Get_Pose().m.data(), Tc.m.data(), &prm,
                                            &n_tracked, st)
                : dsdtm_sparse_align(detail::ctx(), &ref, &cur, &cam, px.data(), bearing.data(), pw.data(),
                                     ini.data(), n, tRefFrame->Get_Pose().m.data(), Tc.m.data(), &prm,
                                     &n_tracked, st);
            if (rc != DSDTM_OK) throw std::runtime_error(std::string("dsdtm_sparse_align: ") + dsdtm_last_error(detail::ctx()));
        };
        if (!mbSnapshotPerLevel) {
            snapshot();                                          // once per Run: the whole Run is one device launch
            levels(mnMaxLevel, mnMinLevel, &last_stats);
        } else {
            // The reference reads Mpt->Get_Pose() at the top of EVERY level (src/Sprase_ImageAlign.cpp:84-103), so a
            // local-BA update of the map can land between two levels of one Run. This mode keeps that: one launch per
            // level, the map points read before each (the pose carried from level to level is the same SE(3) element up
            // to the rounding of T_c2r * T_ref and back: ~1e-16, same iterations).
            last_stats = dsdtm_align_stats{};
            for (int l = mnMaxLevel - 1; l >= mnMinLevel; --l) {
                dsdtm_align_stats st{};
                snapshot();
                levels(l + 1, l, &st);
                last_stats.iters[l] = st.iters[l]; last_stats.n_ref[l] = st.n_ref[l]; last_stats.n_vis[l] = st.n_vis[l];
                last_stats.exit_code[l] = st.exit_code[l]; last_stats.chi2[l] = st.chi2[l];
            }
        }
        if (n < mnMinfts) {                                  // src/Sprase_ImageAlign.cpp:34-38
            std::fprintf(stderr, "Too few features to track\n");
            return 0;                                        // pose untouched
        }
        tCurFrame->Set_Pose(Tc);                             // :57
        return n_tracked;                                    // :59
    }

    bool mbSnapshotPerLevel = false;   // true: map points re-read before every level, one launch per level (see Run)
    dsdtm_align_stats last_stats{};

protected:
    static const std::array<double, 3>& map_point_pose(const MapPoint* mp);   // MapPoint is complete below
    int mnMaxLevel, mnMinLevel, mnMaxIterators, mnMinfts;
};

struct MapPoint {                                 // include/MapPoint.h (what the search reads)
    std::array<double, 3> mPose{{0, 0, 0}};
    std::map<int, int> mObservations;             // keyframe index -> feature index (iterated in keyframe order)
    int mnFound = 1;
    bool mbBad = false;
    const std::array<double, 3>& Get_Pose() const { return mPose; }
    bool IsBad() const { return mbBad; }
    int Get_FoundNums() const { return mnFound; }
    void IncreaseFound(int n = 1) { mnFound += n; }
    void EraseFound(int n = 1) {                  // src/MapPoint.cpp:183-198; SetBadFlag :91-109 (map side not modelled)
        mnFound -= n;
        if (mnFound <= 0) { mbBad = true; mObservations.clear(); }
    }
};

inline const std::array<double, 3>& Sprase_ImgAlign::map_point_pose(const MapPoint* mp) { return mp->Get_Pose(); }

inline int cvRound(double v) { return (int)std::nearbyint(v); }      // OpenCV 2.4 cvRound: round half to even
inline bool IsInImage(const Camera& c, double x, double y, int boundary, int level = 0) {   // src/Camera.cpp:187-193
    return cvRound(x) >= boundary && cvRound(x) < c.mwidth / (1 << level) - boundary &&
           cvRound(y) >= boundary && cvRound(y) < c.mheight / (1 << level) - boundary;
}
inline void FillCircle(Image8& mask, int cx, int cy, int radius, uint8_t value);

class Feature_Alignment {
public:
    explicit Feature_Alignment(CameraPtr camera) : mCam(camera) {
        mGrid_Rows = (int)std::ceil(1.0 * mCam->mheight / Config::CellSize());      // src/Feature_alignment.cpp:29-30
        mGrid_Cols = (int)std::ceil(1.0 * mCam->mwidth / Config::CellSize());
        mCells.resize((size_t)mGrid_Rows * mGrid_Cols);
    }

    struct Candidate { MapPoint* mp; std::array<double, 2> px; };
    struct Match { int cell; MapPoint* mp; float px[2]; int level; };

    void ResetGrid() { for (auto& c : mCells) c.clear(); }                            // :46-52
    bool ReprojectPoint(const Frame& tFrame, MapPoint* tMPoint) {                     // :54-69
        const std::array<double, 2> px = tFrame.World2Pixel(tMPoint->Get_Pose());
        if (!(std::isfinite(px[0]) && std::isfinite(px[1]) && IsInImage(*mCam, px[0], px[1], 8))) return false;
        const int index = (int)(px[1] / Config::CellSize()) * mGrid_Cols + (int)(px[0] / Config::CellSize());
        mCells[(size_t)index].push_back(Candidate{tMPoint, px});
        return true;
    }

    // MapPoint::Get_ClosetObs (src/MapPoint.cpp:133-174): the observation whose viewing direction is
    // closest to the frame's; none when cos < 0.5
    static bool Get_ClosetObs(const MapPoint& mp, const Frame& frame, const std::vector<Frame*>& keyframes, int& kf, int& feat) {
        if (mp.mObservations.empty()) return false;
        auto unit = [](std::array<double, 3> v) { const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                                                  return std::array<double, 3>{{v[0] / n, v[1] / n, v[2] / n}}; };
        const std::array<double, 3> c = frame.Get_CameraCnt();
        const std::array<double, 3> v = unit({{c[0] - mp.mPose[0], c[1] - mp.mPose[1], c[2] - mp.mPose[2]}});
        int best = -1, first = -1;
        double best_cos = 0.0;
        for (const auto& o : mp.mObservations) {
            if (first < 0) first = o.first;
            const std::array<double, 3> k = keyframes[(size_t)o.first]->Get_CameraCnt();
            const std::array<double, 3> r = unit({{k[0] - mp.mPose[0], k[1] - mp.mPose[1], k[2] - mp.mPose[2]}});
            const double cs = r[0] * v[0] + r[1] * v[1] + r[2] * v[2];
            if (cs > best_cos) { best_cos = cs; best = o.first; }
        }
        if (best < 0) best = first;
        if (best_cos < 0.5) return false;
        kf = best; feat = mp.mObservations.at(best);
        return true;
    }

    // void SearchLocalPoints(FramePtr tFrame) (:71-83) with ReprojectCell (:85-121) and FindMatchDirect
    // (:128-158): every live candidate is warped and aligned speculatively in two library calls, then
    // the reference's order-dependent rules are replayed (cells in index order, candidates by found
    // count, mask, first success per cell, 200-match cap).
    std::vector<Match> SearchLocalPoints(Frame& tFrame, const std::vector<Frame*>& keyframes, Image8& img_mask) {
        struct Spec { int cell, pos, kf, feat; };
        std::vector<Spec> cand;
        std::vector<std::vector<Candidate>> cells(mCells.size());
        for (size_t ci = 0; ci < mCells.size(); ++ci) {
            cells[ci].assign(mCells[ci].begin(), mCells[ci].end());
            std::stable_sort(cells[ci].begin(), cells[ci].end(),                      // :88 (std::list::sort is stable)
                             [](const Candidate& a, const Candidate& b) { return a.mp->Get_FoundNums() > b.mp->Get_FoundNums(); });
            for (size_t pos = 0; pos < cells[ci].size(); ++pos) {
                const MapPoint& mp = *cells[ci][pos].mp;
                if (mp.IsBad()) continue;
                int kf, feat;
                if (!Get_ClosetObs(mp, tFrame, keyframes, kf, feat)) continue;         // :135
                const Feature& rf = keyframes[(size_t)kf]->mvFeatures[(size_t)feat];
                const int s = 1 << rf.mlevel;
                if (!IsInImage(*mCam, rf.mpx_x / s, rf.mpx_y / s, mHalf_PatchSize + 1, rf.mlevel)) continue;   // :138-140
                cand.push_back(Spec{(int)ci, (int)pos, kf, feat});
            }
        }
        const int m = (int)cand.size();
        std::vector<uint8_t> conv((size_t)m);
        std::vector<double> pxr(2 * (size_t)m);
        std::vector<int32_t> sl((size_t)m);
        if (m > 0) {
            std::vector<dsdtm_pyramid> pyrs;
            std::vector<double> Tk;
            for (Frame* k : keyframes) { pyrs.push_back(detail_to_pyr(k->mvImg_Pyr)); Tk.insert(Tk.end(), k->Get_Pose().m.begin(), k->Get_Pose().m.end()); }
            std::vector<int32_t> ck((size_t)m), rl((size_t)m);
            std::vector<float> rp(2 * (size_t)m);
            std::vector<double> rb(3 * (size_t)m), pw(3 * (size_t)m), aff(4 * (size_t)m);
            std::vector<uint8_t> pb(100 * (size_t)m), pp(64 * (size_t)m);
            for (int i = 0; i < m; ++i) {
                const Feature& rf = keyframes[(size_t)cand[i].kf]->mvFeatures[(size_t)cand[i].feat];
                const MapPoint& mp = *cells[(size_t)cand[i].cell][(size_t)cand[i].pos].mp;
                ck[i] = cand[i].kf; rl[i] = rf.mlevel; rp[2 * i] = rf.mpx_x; rp[2 * i + 1] = rf.mpx_y;
                for (int k = 0; k < 3; ++k) { rb[3 * i + k] = rf.mNormal[k]; pw[3 * i + k] = mp.mPose[k]; }   // :167 the same map point
            }
            const Camera& c = *mCam;
            const dsdtm_camera cam{c.mfx, c.mfy, c.mcx, c.mcy, c.mf, c.mwidth, c.mheight};
            bool resident = tFrame.mDev != nullptr;
            for (Frame* k : keyframes) resident = resident && k->mDev != nullptr;
            if (resident) {
                // device-resident frames: warp prelude + Align2D in ONE call, px in and out in level-0 pixels
                std::vector<const dsdtm_frame*> kd;
                for (Frame* k : keyframes) kd.push_back(k->mDev);
                for (int i = 0; i < m; ++i) {
                    const std::array<double, 2>& p0 = cells[(size_t)cand[i].cell][(size_t)cand[i].pos].px;
                    pxr[2 * i] = p0[0]; pxr[2 * i + 1] = p0[1];
                }
                const int rc = dsdtm_match_candidates_frames(ctx_(), tFrame.mDev, kd.data(), (int)kd.size(), &cam, Tk.data(),
                                                             tFrame.Get_Pose().m.data(), ck.data(), rp.data(), rl.data(), rb.data(),
                                                             pw.data(), Config::MaxPyraLevels() - 3, 10, m, pxr.data(), sl.data(),
                                                             conv.data());
                if (rc != DSDTM_OK) throw std::runtime_error(std::string("dsdtm_match_candidates_frames: ") + dsdtm_last_error(ctx_()));
                for (int i = 0; i < m; ++i) { pxr[2 * i] /= (1 << sl[i]); pxr[2 * i + 1] /= (1 << sl[i]); }   // (level units below)
            } else {
            int rc = dsdtm_warp_patches(ctx_(), pyrs.data(), (int)pyrs.size(), &cam, Tk.data(), tFrame.Get_Pose().m.data(), ck.data(),
                                        rp.data(), rl.data(), rb.data(), pw.data(), Config::MaxPyraLevels() - 3, m, aff.data(),
                                        sl.data(), pb.data(), pp.data());
            if (rc != DSDTM_OK) throw std::runtime_error(std::string("dsdtm_warp_patches: ") + dsdtm_last_error(ctx_()));
            for (int i = 0; i < m; ++i) {                                             // :150 px in the search level
                const std::array<double, 2>& p0 = cells[(size_t)cand[i].cell][(size_t)cand[i].pos].px;
                pxr[2 * i] = p0[0] / (1 << sl[i]); pxr[2 * i + 1] = p0[1] / (1 << sl[i]);
            }
            const dsdtm_pyramid cur = detail_to_pyr(tFrame.mvImg_Pyr);
            rc = dsdtm_align2d_batch(ctx_(), &cur, pb.data(), pp.data(), sl.data(), pxr.data(), conv.data(), 10, m);   // :152
            if (rc != DSDTM_OK) throw std::runtime_error(std::string("dsdtm_align2d_batch: ") + dsdtm_last_error(ctx_()));
            }
        }
        std::map<std::pair<int, int>, int> index;
        for (int i = 0; i < m; ++i) index[{cand[i].cell, cand[i].pos}] = i;
        std::vector<Match> matches;
        for (size_t ci = 0; ci < cells.size(); ++ci) {                                // :75 index order
            for (size_t pos = 0; pos < cells[ci].size(); ++pos) {
                MapPoint* mp = cells[ci][pos].mp;
                if (mp->IsBad()) continue;                                            // :93
                const std::array<double, 2>& px = cells[ci][pos].px;
                if (img_mask.data[(size_t)cvRound(px[1]) * img_mask.step + cvRound(px[0])] != 255) continue;   // :96
                const auto it = index.find({(int)ci, (int)pos});
                if (it == index.end() || !conv[(size_t)it->second]) continue;         // :101-104
                const int i = it->second;
                const double x = pxr[2 * i] * (1 << sl[i]), y = pxr[2 * i + 1] * (1 << sl[i]);   // :154-156
                mp->IncreaseFound();                                                  // :106
                FillCircle(img_mask, cvRound(x), cvRound(y), Config::CellSize(), 0);  // :111
                Match mt; mt.cell = (int)ci; mt.mp = mp; mt.px[0] = (float)x; mt.px[1] = (float)y; mt.level = sl[i];
                matches.push_back(mt);
                Feature f; f.mpx_x = mt.px[0]; f.mpx_y = mt.px[1]; f.mlevel = mt.level;   // :108 new Feature(frame, px, level)
                f.Mpt = mp; f.mbInitial = true; f.mMptPose = mp->Get_Pose();              // :109 SetPose (include/Feature.h:41-45)
                tFrame.Add_Feature(f);                                                // :113 (bearing); :114 Add_MapPoint = f.Mpt
                break;                                                                // :117 first success wins
            }
            if (matches.size() >= 200) break;                                         // :80
        }
        return matches;
    }

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

    int mGrid_Rows = 0, mGrid_Cols = 0;
    std::vector<std::list<Candidate>> mCells;

private:
    static dsdtm_ctx* ctx_();
    static dsdtm_pyramid detail_to_pyr(const std::vector<Image8>& v);
    CameraPtr mCam;
};

inline dsdtm_ctx* Feature_Alignment::ctx_() { return detail::ctx(); }
inline dsdtm_pyramid Feature_Alignment::detail_to_pyr(const std::vector<Image8>& v) { return detail::to_pyr(v); }

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
        std::stable_sort(corners.begin(), corners.end());                                        // :110 (std::sort there: ties unordered)
        Image8 mask(mImg_width, mImg_height);                                                    // src/Frame.cpp:64
        std::fill(mask.data.begin(), mask.data.end(), 255);
        if (!frame->mvFeatures.empty()) {                                                        // :119-122, Frame::Set_Mask
            for (const Feature& f : frame->mvFeatures)
                if (f.mbInitial) FillCircle(mask, (int)std::lround(f.mpx_x), (int)std::lround(f.mpx_y), Config::Min_dist(), 0);
            // src/Frame.cpp:294-296: threshold(mDynamicMask, 200) and the saturating mImgMask - mDynamicMask
            const Image8& dyn = frame->mDynamicMask;
            if (dyn.cols == mask.cols && dyn.rows == mask.rows)
                for (int y = 0; y < mask.rows; ++y)
                    for (int x = 0; x < mask.cols; ++x)
                        if (dyn.data[(size_t)y * dyn.step + x] > 200) mask.data[(size_t)y * mask.step + x] = 0;
        }
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

// ---- Optimizer::PoseOptimization (include/Optimizer.h:29, src/Optimizer.cpp:20-101) -----------------------
// Called by Tracking::TrackWithLocalMap right after SearchLocalPoints (src/Tracking.cpp:236). The Ceres solve and
// the residual norms are ONE library call (HIP, one wavefront); the EraseFound walk (:80-92) stays here.
class Optimizer {
public:
    static dsdtm_pose_opt_summary& LastSummary() { static thread_local dsdtm_pose_opt_summary s{}; return s; }
    static void PoseOptimization(FramePtr tCurFrame, int /*tIterations: never read by the reference*/ = 100) {
        std::vector<Feature>& fts = tCurFrame->mvFeatures;
        const int N = (int)fts.size();
        std::vector<double> bearing(3 * (size_t)N), pw(3 * (size_t)N, 0.0), rn((size_t)std::max(N, 1));
        std::vector<int32_t> level((size_t)N);
        std::vector<uint8_t> use((size_t)N, 0);
        std::map<int, MapPoint*> tvMpts;                                     // :43 keyed by FEATURE index (:57)
        for (int i = 0; i < N; ++i) {                                        // :45-65
            for (int k = 0; k < 3; ++k) bearing[3 * (size_t)i + k] = fts[(size_t)i].mNormal[k];
            level[(size_t)i] = fts[(size_t)i].mlevel;
            MapPoint* mp = fts[(size_t)i].Mpt;
            if (!mp || mp->IsBad() || !fts[(size_t)i].mbInitial) continue;
            use[(size_t)i] = 1;
            for (int k = 0; k < 3; ++k) pw[3 * (size_t)i + k] = mp->Get_Pose()[(size_t)k];
            tvMpts[i] = mp;
        }
        SE3 T = tCurFrame->Get_Pose();
        dsdtm_pose_opt_params prm;
        prm.max_iterations = 100; prm.reserved = 0;                          // :72
        dsdtm_pose_opt_summary& sm = LastSummary();
        const int rc = dsdtm_pose_optimization(detail::ctx(), bearing.data(), pw.data(), level.data(), use.data(), N, T.m.data(),
                                               &prm, rn.data(), &sm);
        if (rc != DSDTM_OK) throw std::runtime_error(std::string("dsdtm_pose_optimization: ") + dsdtm_last_error(detail::ctx()));
        tCurFrame->Set_Pose(T);                                              // :78
        double tOutlineThres = Config::LocalBAthreshhold();                  // :22-24: double(float) / float mf
        tOutlineThres = tOutlineThres / tCurFrame->mCamera->mf;
        for (int i = 0; i < sm.n_residual_blocks; ++i) {                     // :80-92: residual i is in BLOCK order, the
            if (rn[(size_t)i] > tOutlineThres) {                             // map is keyed by FEATURE index: kept as is
                auto it = tvMpts.find(i);
                if (it == tvMpts.end() || !it->second) continue;
                if (it->second->IsBad()) continue;
                it->second->EraseFound();
            }
        }
    }
};

// ---- one tracked frame in ONE library call (src/Tracking.cpp:199-256) --------------------------------------
// Tracking::TrackWithLastFrame + UpdateLocalMap + TrackWithLocalMap for a tracker whose frames live on the device:
// dsdtm_track_frame enqueues the new frame's upload and pyramid, Run, ReprojectPoint + Get_ClosetObs for every local map
// point, FindMatchDirect for all of them, the cell walk of SearchLocalPoints (replayed on the device) and
// PoseOptimization back to back and waits once. What stays here is what the reference's classes leave on its objects:
// the new Features (px, level, bearing, map point, mbInitial: src/Feature_alignment.cpp:108-114), IncreaseFound (:106),
// the mask discs (:111), Set_Pose and the EraseFound walk (src/Optimizer.cpp:78-92). Same results as the four calls
// above, bit for bit (tests/test_track_frame_gpu.py, tests/test_host_cpp.py).
class Tracking {
public:
    Tracking(CameraPtr camera, int tMaxLevel, int tMinLevel, int tMaxIterators, int tMinTracked = 20)   // src/Tracking.cpp:20-37, :208
        : mCam(camera), mMaxLevel(tMaxLevel), mMinLevel(tMinLevel), mMaxIters(tMaxIterators), mMinTracked(tMinTracked) {}

    struct Result { int n_tracked = 0; bool lost = false; std::vector<Feature_Alignment::Match> matches; dsdtm_align_stats stats{};
                    dsdtm_pose_opt_summary summary{}; SE3 T_run; };

    // level0: the new image. `last`, the keyframes: frames whose pyramids are resident (ComputeImagePyramidOnDevice).
    // local_points: the order UpdateLocalMap would hand them to ReprojectPoint (:283-299). img_mask: Frame::mImgMask or null.
    FramePtr TrackFrame(const Image8& level0, const FramePtr& last, const std::vector<Frame*>& keyframes,
                        const std::vector<MapPoint*>& local_points, Image8* img_mask, Result* out) {
        const int levels = mMaxLevel;
        const std::vector<Feature>& rf = last->mvFeatures;
        const size_t n = rf.size(), M = local_points.size();
        mPx.resize(2 * n); mBear.resize(3 * n); mPw.resize(3 * n); mIni.resize(n);
        for (size_t i = 0; i < n; ++i) {
            mPx[2 * i] = rf[i].mpx_x; mPx[2 * i + 1] = rf[i].mpx_y; mIni[i] = rf[i].mbInitial ? 1 : 0;
            const std::array<double, 3>& pw = rf[i].Mpt ? rf[i].Mpt->Get_Pose() : rf[i].mMptPose;   // Run snapshots Mpt->Get_Pose() (:93)
            for (int k = 0; k < 3; ++k) { mBear[3 * i + (size_t)k] = rf[i].mNormal[(size_t)k]; mPw[3 * i + (size_t)k] = pw[(size_t)k]; }
        }
        std::map<const Frame*, int> kf_index;
        std::vector<const dsdtm_frame*> kd;
        std::vector<double> Tk;
        for (size_t k = 0; k < keyframes.size(); ++k) {
            kf_index[keyframes[k]] = (int)k;
            if (!keyframes[k]->mDev) throw std::runtime_error("Tracking::TrackFrame: keyframe pyramids must be resident on the device");
            kd.push_back(keyframes[k]->mDev);
            Tk.insert(Tk.end(), keyframes[k]->Get_Pose().m.begin(), keyframes[k]->Get_Pose().m.end());
        }
        if (!last->mDev) throw std::runtime_error("Tracking::TrackFrame: the last frame's pyramid must be resident on the device");
        mMpw.resize(3 * M); mFound.resize(M); mBad.resize(M); mOff.assign(M + 1, 0);
        mOkf.clear(); mOpx.clear(); mOlv.clear(); mOb.clear();
        for (size_t i = 0; i < M; ++i) {
            const MapPoint& mp = *local_points[i];
            for (int k = 0; k < 3; ++k) mMpw[3 * i + (size_t)k] = mp.mPose[(size_t)k];
            mFound[i] = mp.Get_FoundNums(); mBad[i] = mp.IsBad() ? 1 : 0;
            for (const auto& o : mp.mObservations) {                       // iteration order of the map = Get_ClosetObs' order
                const Feature& f = keyframes[(size_t)o.first]->mvFeatures[(size_t)o.second];
                mOkf.push_back(o.first); mOpx.push_back(f.mpx_x); mOpx.push_back(f.mpx_y); mOlv.push_back(f.mlevel);
                for (int k = 0; k < 3; ++k) mOb.push_back(f.mNormal[(size_t)k]);
            }
            mOff[i + 1] = (int32_t)mOkf.size();
        }
        const Camera& c = *mCam;
        const dsdtm_camera cam{c.mfx, c.mfy, c.mcx, c.mcy, c.mf, c.mwidth, c.mheight};
        dsdtm_track_desc d{};
        d.image = level0.data.data(); d.width = level0.cols; d.height = level0.rows; d.stride = level0.step; d.levels = levels;
        d.ref = last->mDev; d.ref_px_xy = mPx.data(); d.ref_bearing = mBear.data(); d.ref_p_world = mPw.data(); d.ref_initial = mIni.data();
        d.n_ref_features = (int32_t)n; d.T_ref_w = last->Get_Pose().m.data(); d.T_seed = last->Get_Pose().m.data();     // :201
        d.align.max_level = mMaxLevel; d.align.min_level = mMinLevel; d.align.max_iters = mMaxIters; d.align.min_fts = Config::Min_fts();
        d.min_tracked = mMinTracked;
        d.kf = kd.data(); d.n_kf = (int32_t)kd.size(); d.T_kf_w = Tk.data();
        d.n_points = (int32_t)M; d.mp_world = mMpw.data(); d.mp_found = mFound.data(); d.mp_bad = mBad.data(); d.obs_offset = mOff.data();
        d.obs_kf = mOkf.data(); d.obs_px = mOpx.data(); d.obs_level = mOlv.data(); d.obs_bearing = mOb.data();
        if (img_mask) { d.mask = img_mask->data.data(); d.mask_stride = img_mask->step; }
        d.cell_size = Config::CellSize(); d.max_pyr_levels = Config::MaxPyraLevels(); d.max_matches = 200; d.align2d_iters = 10;
        d.pose_opt.max_iterations = 100; d.pose_opt.reserved = 0;
        dsdtm_track_result r;
        std::vector<dsdtm_track_match> ms(200);
        std::vector<double> rn(200);
        if (dsdtm_track_frame(detail::ctx(), &cam, &d, &r, ms.data(), rn.data()) != DSDTM_OK)
            throw std::runtime_error(std::string("dsdtm_track_frame: ") + dsdtm_last_error(detail::ctx()));
        FramePtr cur = std::make_shared<Frame>();
        cur->mCamera = mCam;
        cur->mvImg_Pyr.push_back(level0);                                  // (level 0 on the host; the pyramid lives on the device)
        cur->mDev = r.frame;
        SE3 T; std::copy(r.T_run, r.T_run + 12, T.m.begin());
        cur->Set_Pose(T);                                                  // src/Sprase_ImageAlign.cpp:57
        if (out) { out->n_tracked = r.n_tracked; out->lost = r.lost != 0; out->stats = r.stats; out->summary = r.summary; out->T_run = T; out->matches.clear(); }
        if (r.lost) return cur;                                            // src/Tracking.cpp:208-214
        for (int k = 0; k < r.n_matches; ++k) {
            MapPoint* mp = local_points[(size_t)ms[(size_t)k].point];
            mp->IncreaseFound();                                           // src/Feature_alignment.cpp:106
            if (img_mask) FillCircle(*img_mask, cvRound((double)ms[(size_t)k].px[0]), cvRound((double)ms[(size_t)k].px[1]), Config::CellSize(), 0);   // :111
            Feature f; f.mpx_x = ms[(size_t)k].px[0]; f.mpx_y = ms[(size_t)k].px[1]; f.mlevel = ms[(size_t)k].level;   // :108
            f.Mpt = mp; f.mbInitial = true; f.mMptPose = mp->Get_Pose();    // :109
            cur->Add_Feature(f);                                           // :113-114
            if (out) { Feature_Alignment::Match mt; mt.cell = ms[(size_t)k].cell; mt.mp = mp; mt.px[0] = f.mpx_x; mt.px[1] = f.mpx_y; mt.level = f.mlevel; out->matches.push_back(mt); }
        }
        std::copy(r.T_opt, r.T_opt + 12, T.m.begin());
        cur->Set_Pose(T);                                                  // src/Optimizer.cpp:78
        Optimizer::LastSummary() = r.summary;
        double thr = Config::LocalBAthreshhold();                          // :22-24
        thr = thr / mCam->mf;
        for (int i = 0; i < r.summary.n_residual_blocks && i < r.n_matches; ++i)     // :80-92 (every feature has a block: index = block)
            if (rn[(size_t)i] > thr) {
                MapPoint* mp = local_points[(size_t)ms[(size_t)i].point];
                if (!mp->IsBad()) mp->EraseFound();
            }
        return cur;
    }

private:
    CameraPtr mCam;
    int mMaxLevel, mMinLevel, mMaxIters, mMinTracked;
    std::vector<float> mPx, mOpx;
    std::vector<double> mBear, mPw, mMpw, mOb;
    std::vector<uint8_t> mIni, mBad;
    std::vector<int32_t> mFound, mOff, mOkf, mOlv;
};

}  // namespace DSDTM
