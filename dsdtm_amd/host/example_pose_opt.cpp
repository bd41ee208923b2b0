// example_pose_opt.cpp — driver for Optimizer::PoseOptimization of the C++ host layer (reference
// src/Optimizer.cpp:20-101 as called from Tracking::TrackWithLocalMap, src/Tracking.cpp:236).
// Reads a frame written by tests/test_host_cpp.py, prints the refined pose and the map-point counters.
#include <cstdio>
#include <cstdlib>

#include "dsdtm_host.hpp"

using namespace DSDTM;

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
    if (std::fread(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s frame.bin\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 2; }
    int32_t hdr[2];                                   // n_features, n_map_points
    rd(f, hdr, 2);
    float mf; rd(f, &mf, 1);
    CameraPtr cam = std::make_shared<Camera>();
    cam->mf = mf;
    FramePtr fr = std::make_shared<Frame>();
    fr->mCamera = cam;
    SE3 T; rd(f, T.m.data(), 12); fr->Set_Pose(T);
    std::vector<MapPoint> mps((size_t)hdr[1]);
    for (MapPoint& mp : mps) {
        int32_t meta[2];                              // found, bad
        rd(f, mp.mPose.data(), 3); rd(f, meta, 2);
        mp.mnFound = meta[0]; mp.mbBad = meta[1] != 0;
    }
    fr->mvFeatures.resize((size_t)hdr[0]);
    for (Feature& ft : fr->mvFeatures) {
        int32_t meta[3];                              // level, initial, map point index or -1
        rd(f, ft.mNormal.data(), 3); rd(f, meta, 3);
        ft.mlevel = meta[0]; ft.mbInitial = meta[1] != 0;
        ft.Mpt = meta[2] >= 0 ? &mps[(size_t)meta[2]] : nullptr;
    }
    std::fclose(f);
    Optimizer::PoseOptimization(fr, 10);
    const dsdtm_pose_opt_summary& sm = Optimizer::LastSummary();
    std::printf("summary %d %d %d %d %.17g %.17g\n", sm.iterations, sm.successful_steps, sm.termination, sm.n_residual_blocks,
                sm.initial_cost, sm.final_cost);
    std::printf("pose");
    for (double v : fr->Get_Pose().m) std::printf(" %.17g", v);
    std::printf("\n");
    for (const MapPoint& mp : mps) std::printf("%d %d\n", mp.mnFound, mp.mbBad ? 1 : 0);
    return 0;
}
