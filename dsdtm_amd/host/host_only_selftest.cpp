// host_only_selftest.cpp — the parts of dsdtm_host.hpp that never reach the GPU library (cvRound / IsInImage, the
// mask discs of SearchLocalPoints, the reprojection grid, MapPoint::Get_ClosetObs, EraseFound, Frame::Add_Feature),
// exercised on edge cases. Built with -fsanitize=address,undefined by tests/test_sanitizers_cpu.py and run on the CPU:
// no context is ever created, no entry point of libdsdtm_amd.so is called.
#include <cassert>
#include <cmath>
#include <cstdio>

#include "dsdtm_host.hpp"

using namespace DSDTM;

int main() {
    // cvRound: round half to even; IsInImage with the level divisor (src/Camera.cpp:187-193)
    assert(cvRound(2.5) == 2 && cvRound(3.5) == 4 && cvRound(-0.5) == 0 && cvRound(7.49) == 7);
    Camera cam; cam.mfx = cam.mfy = 500.0f; cam.mcx = 320.0f; cam.mcy = 240.0f; cam.mf = 525.0f; cam.mwidth = 640; cam.mheight = 480;
    assert(IsInImage(cam, 8.0, 8.0, 8) && !IsInImage(cam, 7.4, 8.0, 8) && !IsInImage(cam, 632.0, 100.0, 8) && IsInImage(cam, 631.4, 471.4, 8));
    assert(IsInImage(cam, 150.0, 100.0, 5, 2) && !IsInImage(cam, 156.0, 100.0, 5, 2));
    // discs at every border and far outside: nothing is written beyond the mask
    Image8 mask(64, 48);
    std::fill(mask.data.begin(), mask.data.end(), 255);
    for (int cy : {-40, -3, 0, 24, 47, 50, 90})
        for (int cx : {-40, -3, 0, 32, 63, 66, 120})
            for (int r : {0, 1, 5, 25}) FillCircle(mask, cx, cy, r, 0);
    size_t zeros = 0;
    for (uint8_t v : mask.data) zeros += v == 0;
    assert(zeros > 0 && mask.data.size() == 64u * 48u);
    // reprojection grid: points on the border ring, behind the camera, at infinity
    CameraPtr camp = std::make_shared<Camera>(cam);
    Feature_Alignment fa(camp);
    Frame fr; fr.mCamera = camp;
    SE3 I; I.m = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};
    fr.Set_Pose(I);
    std::vector<MapPoint> mps(6);
    const double pts[6][3] = {{0, 0, 2}, {-1.24, -0.92, 2}, {1.27, 0.95, 2}, {0, 0, -2}, {0, 0, 0}, {1e300, 1e300, 1}};
    int in = 0;
    for (int i = 0; i < 6; ++i) { mps[i].mPose = {{pts[i][0], pts[i][1], pts[i][2]}}; in += fa.ReprojectPoint(fr, &mps[i]) ? 1 : 0; }
    assert(in >= 1 && in <= 3);
    fa.ResetGrid();
    // closest observation: no observations, one in front, one at a right angle (cos < 0.5)
    Frame kf0, kf1; kf0.mCamera = kf1.mCamera = camp; kf0.Set_Pose(I);
    SE3 side = I; side.m[3] = -10.0;                       // camera centre at x = +10
    kf1.Set_Pose(side);
    std::vector<Frame*> kfs{&kf0, &kf1};
    MapPoint mp; mp.mPose = {{0, 0, 2}};
    int k = -1, f = -1;
    assert(!Feature_Alignment::Get_ClosetObs(mp, fr, kfs, k, f));
    mp.mObservations[1] = 3;
    assert(!Feature_Alignment::Get_ClosetObs(mp, fr, kfs, k, f));        // only a sideways view: rejected
    mp.mObservations[0] = 7;
    assert(Feature_Alignment::Get_ClosetObs(mp, fr, kfs, k, f) && k == 0 && f == 7);
    // found counter never goes below the reference's floor; bearing of an added feature is a unit vector
    for (int i = 0; i < 5; ++i) mp.EraseFound();
    Feature ft; ft.mpx_x = 100.0f; ft.mpx_y = 50.0f;
    fr.Add_Feature(ft);
    const auto& n = fr.mvFeatures.back().mNormal;
    assert(std::fabs(n[0] * n[0] + n[1] * n[1] + n[2] * n[2] - 1.0) < 1e-12);
    std::printf("host-only selftest ok (%zu mask pixels cleared, %d points binned)\n", zeros, in);
    return 0;
}
