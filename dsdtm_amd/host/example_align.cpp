// example_align.cpp — C++ driver in the shape of the reference's stand-alone tests
// (Test/test_SpraseImg_alignment.cpp:85-168: set the seed pose, Run(cur, ref), print the result;
//  Test/test_Feature_alignment.cpp:47-86: Align2DGaussNewton on one patch), reading a scene dumped by
// tests/test_host_cpp.py and printing machine-readable results.
//   usage: example_align <scene.bin>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dsdtm_host.hpp"

using namespace DSDTM;

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
    if (fread(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s scene.bin\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 2; }
    int32_t hdr[8];      // levels, n_features, max_level, min_level, max_iters, min_fts, width, height
    rd(f, hdr, 8);
    const int levels = hdr[0], n = hdr[1];
    float camf[5];
    rd(f, camf, 5);
    CameraPtr cam = std::make_shared<Camera>();
    cam->mfx = camf[0]; cam->mfy = camf[1]; cam->mcx = camf[2]; cam->mcy = camf[3]; cam->mf = camf[4];
    cam->mwidth = hdr[6]; cam->mheight = hdr[7];
    FramePtr ref = std::make_shared<Frame>(), cur = std::make_shared<Frame>();
    ref->mCamera = cur->mCamera = cam;
    for (FramePtr fr : {ref, cur}) {
        int w = hdr[6], h = hdr[7];
        for (int l = 0; l < levels; ++l) {
            Image8 im(w, h);
            rd(f, im.data.data(), im.data.size());
            fr->mvImg_Pyr.push_back(std::move(im));
            w = (w + 1) / 2; h = (h + 1) / 2;
        }
    }
    ref->mvFeatures.resize(n);
    for (int i = 0; i < n; ++i) {
        Feature& ft = ref->mvFeatures[i];
        float p[2]; double b[3], w[3]; uint8_t ini;
        rd(f, p, 2); rd(f, b, 3); rd(f, w, 3); rd(f, &ini, 1);
        ft.mpx_x = p[0]; ft.mpx_y = p[1]; ft.mbInitial = ini != 0;
        for (int k = 0; k < 3; ++k) { ft.mNormal[k] = b[k]; ft.mMptPose[k] = w[k]; }
    }
    SE3 Tr, Tc;
    rd(f, Tr.m.data(), 12); rd(f, Tc.m.data(), 12);
    ref->Set_Pose(Tr);
    cur->Set_Pose(Tc);                                      // Test/test_SpraseImg_alignment.cpp:156
    // Align2D part: one bordered patch + patch + start pixel on cur level 0
    uint8_t border[100], patch[64]; double px[2];
    rd(f, border, 100); rd(f, patch, 64); rd(f, px, 2);
    std::fclose(f);

    Config::Min_fts() = hdr[5];
    Sprase_ImgAlign align(hdr[2], hdr[3], hdr[4]);
    const int n_tracked = align.Run(cur, ref);              // :157
    std::printf("n_tracked %d\npose", n_tracked);
    for (double v : cur->Get_Pose().m) std::printf(" %.17g", v);
    std::printf("\niters");
    for (int l = 0; l < levels; ++l) std::printf(" %d", align.last_stats.iters[l]);
    const bool ok = Feature_Alignment::Align2DGaussNewton(cur->mvImg_Pyr[0], border, patch, 10, px);   // Test/test_Feature_alignment.cpp:78
    std::printf("\nalign2d %d %.9g %.9g\n", ok ? 1 : 0, px[0], px[1]);
    // the same Run with both frames resident on the device (pyramids built there from level 0)
    ref->ComputeImagePyramidOnDevice(levels);
    cur->ComputeImagePyramidOnDevice(levels);
    cur->Set_Pose(Tc);
    const int n_resident = align.Run(cur, ref);
    std::printf("resident %d\npose_resident", n_resident);
    for (double v : cur->Get_Pose().m) std::printf(" %.17g", v);
    std::printf("\n");
    // the same Run with the map points re-read before every level, as the reference does (one launch per level)
    {
        cur->Set_Pose(Tc);
        align.mbSnapshotPerLevel = true;
        const int n_lv = align.Run(cur, ref);
        align.mbSnapshotPerLevel = false;
        std::printf("per_level %d\npose_per_level", n_lv);
        for (double v : cur->Get_Pose().m) std::printf(" %.17g", v);
        std::printf("\niters_per_level");
        for (int l = 0; l < levels; ++l) std::printf(" %d", align.last_stats.iters[l]);
        std::printf("\n");
    }
    // keyframe creation (src/Tracking.cpp:416): detect new features on the current frame, which holds none yet
    Config::MaxPyraLevels() = levels;
    Feature_detector detector(hdr[6], hdr[7]);
    detector.detect(cur.get(), 5.0);
    std::printf("detected %zu", cur->mvFeatures.size());
    for (const Feature& ft : cur->mvFeatures) std::printf(" %d %d %d", (int)ft.mpx_x, (int)ft.mpx_y, ft.mlevel);
    std::printf("\n");
    // wall time of Feature_detector::detect through the C++ layer on the resident frame (library call + the
    // sort / mask / cap bookkeeping of src/Feature_detection.cpp:110-150), median of 21 calls
    {
        std::vector<double> ms;
        for (int rep = 0; rep < 21; ++rep) {
            cur->mvFeatures.clear();
            const auto t0 = std::chrono::steady_clock::now();
            detector.detect(cur.get(), 5.0);
            ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        std::sort(ms.begin(), ms.end());
        std::printf("detect_ms %.4f features %zu\n", ms[ms.size() / 2], cur->mvFeatures.size());
    }
    // wall time of Sprase_ImgAlign::Run through the C++ layer on resident frames (what Tracking::TrackWithLastFrame waits for,
    // src/Tracking.cpp:199-217), median of 101 calls
    {
        cur->mvFeatures.clear();
        std::vector<double> ms;
        for (int rep = 0; rep < 101; ++rep) {
            cur->Set_Pose(Tc);
            const auto t0 = std::chrono::steady_clock::now();
            align.Run(cur, ref);
            ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        std::sort(ms.begin(), ms.end());
        std::printf("run_resident_ms %.4f min %.4f\n", ms[ms.size() / 2], ms[0]);
    }
    return 0;
}
