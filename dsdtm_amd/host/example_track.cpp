// example_track.cpp — the tracked-frame loop of the reference's Tracking through the C++ host layer, on
// device-resident frames (reference src/Tracking.cpp:199-246):
//   TrackWithLastFrame   cur->Set_Pose(last->Get_Pose()); Sprase_ImgAlign::Run(cur, last)              (:201-204)
//   UpdateLocalMap       Feature_Alignment::ResetGrid; ReprojectPoint for every good local map point    (:260-299)
//   TrackWithLocalMap    SearchLocalPoints(cur); Optimizer::PoseOptimization(cur)                       (:224, :236)
// and the frame becomes `last` of the next one — which needs the complete features SearchLocalPoints leaves
// behind (map point, mbInitial, bearing). Reads a world + frame sequence written by tests/test_host_cpp.py.
//   usage: example_track <world.bin> [onecall]     onecall: every frame through ONE dsdtm_track_frame (Tracking::TrackFrame of the
//                                                  host layer) instead of the four synchronous calls — the same lines are printed
#include <cstdio>
#include <cstdlib>
#include <string>

#include "dsdtm_host.hpp"

using namespace DSDTM;

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
    if (std::fread(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s world.bin\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 2; }
    int32_t hdr[8];      // levels, n_kf, n_points, width, height, cell size, max pyramid levels, n_frames
    rd(f, hdr, 8);
    const int levels = hdr[0], n_kf = hdr[1], n_pts = hdr[2], W = hdr[3], H = hdr[4], n_frames = hdr[7];
    float camf[5];
    rd(f, camf, 5);
    CameraPtr cam = std::make_shared<Camera>();
    cam->mfx = camf[0]; cam->mfy = camf[1]; cam->mcx = camf[2]; cam->mcy = camf[3]; cam->mf = camf[4];
    cam->mwidth = W; cam->mheight = H;
    Config::CellSize() = hdr[5];
    Config::MaxPyraLevels() = hdr[6];
    Config::Min_fts() = 15;
    std::vector<FramePtr> kfs;
    for (int k = 0; k < n_kf + 1; ++k) {                  // keyframes (+ one frame of the search test's format: skipped)
        FramePtr fr = std::make_shared<Frame>();
        fr->mCamera = cam;
        SE3 T; rd(f, T.m.data(), 12); fr->Set_Pose(T);
        int w = W, h = H;
        for (int l = 0; l < levels; ++l) {
            Image8 im(w, h);
            rd(f, im.data.data(), im.data.size());
            if (l == 0) fr->mvImg_Pyr.push_back(std::move(im));      // level 0 only: the pyramid is built on the device
            w = (w + 1) / 2; h = (h + 1) / 2;
        }
        int32_t nf; rd(f, &nf, 1);
        fr->mvFeatures.resize((size_t)nf);
        for (int i = 0; i < nf; ++i) {
            float p[2]; int32_t lv; double b[3];
            rd(f, p, 2); rd(f, &lv, 1); rd(f, b, 3);
            Feature& ft = fr->mvFeatures[(size_t)i];
            ft.mpx_x = p[0]; ft.mpx_y = p[1]; ft.mlevel = lv; ft.mbInitial = false;
            for (int j = 0; j < 3; ++j) ft.mNormal[j] = b[j];
        }
        if (k < n_kf) { fr->ComputeImagePyramidOnDevice(levels); kfs.push_back(fr); }
    }
    std::vector<MapPoint> mps((size_t)n_pts);
    for (int i = 0; i < n_pts; ++i) {
        int32_t meta[3];                                  // found, bad, n_obs
        rd(f, mps[(size_t)i].mPose.data(), 3); rd(f, meta, 3);
        mps[(size_t)i].mnFound = meta[0]; mps[(size_t)i].mbBad = meta[1] != 0;
        for (int o = 0; o < meta[2]; ++o) {
            int32_t kv[2]; rd(f, kv, 2);
            mps[(size_t)i].mObservations[kv[0]] = kv[1];
            Feature& ft = kfs[(size_t)kv[0]]->mvFeatures[(size_t)kv[1]];     // what keyframe creation leaves behind
            ft.Mpt = &mps[(size_t)i]; ft.mbInitial = true;
        }
    }
    Sprase_ImgAlign align(5, 0, 8);                       // src/Tracking.cpp:20-24,37
    Feature_Alignment fa(cam);
    std::vector<Frame*> kfp;
    for (auto& k : kfs) kfp.push_back(k.get());
    FramePtr last = kfs.back();
    const bool onecall = argc > 2 && std::string(argv[2]) == "onecall";
    Tracking tracker(cam, 5, 0, 8, 20);
    for (int k = 0; k < n_frames && onecall; ++k) {
        Image8 im(W, H);
        rd(f, im.data.data(), im.data.size());
        std::vector<MapPoint*> local;
        for (MapPoint& mp : mps) if (!mp.IsBad()) local.push_back(&mp);      // UpdateLocalMap skips bad points (:288)
        Tracking::Result res;
        FramePtr cur = tracker.TrackFrame(im, last, kfp, local, nullptr, &res);
        std::printf("frame %d run %d pose", k, res.n_tracked);
        for (double v : res.T_run.m) std::printf(" %.17g", v);
        std::printf("\nmatches %zu", res.matches.size());
        for (const auto& m : res.matches) std::printf(" %d %d %d %.9g %.9g", m.cell, (int)(m.mp - mps.data()), m.level, m.px[0], m.px[1]);
        std::printf("\nrefined %d %d pose", res.summary.iterations, res.summary.termination);
        for (double v : cur->Get_Pose().m) std::printf(" %.17g", v);
        std::printf("\nmap");
        for (const MapPoint& mp : mps) std::printf(" %d%s", mp.mnFound, mp.mbBad ? "b" : "");
        std::printf("\n");
        last = cur;
    }
    for (int k = 0; k < n_frames && !onecall; ++k) {
        FramePtr cur = std::make_shared<Frame>();
        cur->mCamera = cam;
        Image8 im(W, H);
        rd(f, im.data.data(), im.data.size());
        cur->mvImg_Pyr.push_back(std::move(im));
        cur->ComputeImagePyramidOnDevice(levels);
        cur->Set_Pose(last->Get_Pose());                                   // :201
        const int n = align.Run(cur, last);                                // :204
        std::printf("frame %d run %d pose", k, n);
        for (double v : cur->Get_Pose().m) std::printf(" %.17g", v);
        std::printf("\n");
        fa.ResetGrid();                                                    // :260
        for (MapPoint& mp : mps) if (!mp.IsBad()) fa.ReprojectPoint(*cur, &mp);   // :283-299
        Image8 mask(W, H);
        std::fill(mask.data.begin(), mask.data.end(), 255);
        const std::vector<Feature_Alignment::Match> ms = fa.SearchLocalPoints(*cur, kfp, mask);   // :224
        std::printf("matches %zu", ms.size());
        for (const auto& m : ms) std::printf(" %d %d %d %.9g %.9g", m.cell, (int)(m.mp - mps.data()), m.level, m.px[0], m.px[1]);
        std::printf("\n");
        Optimizer::PoseOptimization(cur, 10);                              // :236
        const dsdtm_pose_opt_summary& sm = Optimizer::LastSummary();
        std::printf("refined %d %d pose", sm.iterations, sm.termination);
        for (double v : cur->Get_Pose().m) std::printf(" %.17g", v);
        std::printf("\nmap");
        for (const MapPoint& mp : mps) std::printf(" %d%s", mp.mnFound, mp.mbBad ? "b" : "");
        std::printf("\n");
        last = cur;
    }
    std::fclose(f);
    return 0;
}
