// example_search.cpp — driver for the reprojection search of the C++ host layer
// (Feature_Alignment::ResetGrid / ReprojectPoint / SearchLocalPoints, reference
// src/Feature_alignment.cpp:46-126 as called from Tracking::UpdateLocalMap / TrackWithLocalMap,
// src/Tracking.cpp:219-313). Reads a world written by tests/test_host_cpp.py, prints the matches.
#include <cstdio>
#include <cstdlib>

#include "dsdtm_host.hpp"

using namespace DSDTM;

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
    if (std::fread(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
}

static void read_pyr(FILE* f, Frame& fr, int levels, int w, int h) {
    for (int l = 0; l < levels; ++l) {
        Image8 im(w, h);
        rd(f, im.data.data(), im.data.size());
        fr.mvImg_Pyr.push_back(std::move(im));
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s world.bin\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 2; }
    int32_t hdr[8];      // levels, n_kf, n_points, width, height, cell size, max pyramid levels, -
    rd(f, hdr, 8);
    const int levels = hdr[0], n_kf = hdr[1], n_pts = hdr[2];
    float camf[5];
    rd(f, camf, 5);
    CameraPtr cam = std::make_shared<Camera>();
    cam->mfx = camf[0]; cam->mfy = camf[1]; cam->mcx = camf[2]; cam->mcy = camf[3]; cam->mf = camf[4];
    cam->mwidth = hdr[3]; cam->mheight = hdr[4];
    Config::CellSize() = hdr[5];
    Config::MaxPyraLevels() = hdr[6];
    std::vector<std::unique_ptr<Frame>> kfs;
    for (int k = 0; k <= n_kf; ++k) {                     // keyframes, then the current frame
        std::unique_ptr<Frame> fr(new Frame());
        fr->mCamera = cam;
        SE3 T; rd(f, T.m.data(), 12); fr->Set_Pose(T);
        read_pyr(f, *fr, levels, hdr[3], hdr[4]);
        int32_t nf; rd(f, &nf, 1);
        fr->mvFeatures.resize((size_t)nf);
        for (int i = 0; i < nf; ++i) {
            float p[2]; int32_t lv; double b[3];
            rd(f, p, 2); rd(f, &lv, 1); rd(f, b, 3);
            Feature& ft = fr->mvFeatures[(size_t)i];
            ft.mpx_x = p[0]; ft.mpx_y = p[1]; ft.mlevel = lv; ft.mbInitial = true;
            for (int j = 0; j < 3; ++j) ft.mNormal[j] = b[j];
        }
        kfs.push_back(std::move(fr));
    }
    std::unique_ptr<Frame> cur = std::move(kfs.back());
    kfs.pop_back();
    std::vector<MapPoint> mps((size_t)n_pts);
    for (int i = 0; i < n_pts; ++i) {
        int32_t meta[3];                                  // found, bad, n_obs
        rd(f, mps[(size_t)i].mPose.data(), 3); rd(f, meta, 3);
        mps[(size_t)i].mnFound = meta[0]; mps[(size_t)i].mbBad = meta[1] != 0;
        for (int o = 0; o < meta[2]; ++o) { int32_t kv[2]; rd(f, kv, 2); mps[(size_t)i].mObservations[kv[0]] = kv[1]; }
    }
    std::fclose(f);

    Feature_Alignment fa(cam);
    fa.ResetGrid();
    int n_in = 0;
    for (MapPoint& mp : mps) n_in += fa.ReprojectPoint(*cur, &mp) ? 1 : 0;       // src/Tracking.cpp:296-311
    std::vector<Frame*> kfp;
    for (auto& k : kfs) kfp.push_back(k.get());
    Image8 mask(cam->mwidth, cam->mheight);
    std::fill(mask.data.begin(), mask.data.end(), 255);
    const std::vector<Feature_Alignment::Match> ms = fa.SearchLocalPoints(*cur, kfp, mask);
    std::printf("reprojected %d\nmatches %zu\n", n_in, ms.size());
    for (const auto& m : ms) std::printf("%d %d %.9g %.9g %d\n", m.cell, (int)(m.mp - mps.data()), m.px[0], m.px[1], m.level);
    unsigned long long sum = 0;
    for (uint8_t v : mask.data) sum += v;
    std::printf("mask_sum %llu\n", sum);
    // the same search with every pyramid resident on the device (one library call for all candidates)
    for (auto& k : kfs) k->ComputeImagePyramidOnDevice(levels);
    cur->ComputeImagePyramidOnDevice(levels);
    cur->mvFeatures.clear();
    for (const auto& m : ms) m.mp->mnFound -= 1;                              // undo IncreaseFound of the first search
    Image8 mask2(cam->mwidth, cam->mheight);
    std::fill(mask2.data.begin(), mask2.data.end(), 255);
    const std::vector<Feature_Alignment::Match> mr = fa.SearchLocalPoints(*cur, kfp, mask2);
    bool same = mr.size() == ms.size() && mask2.data == mask.data;
    for (size_t i = 0; same && i < ms.size(); ++i)
        same = mr[i].cell == ms[i].cell && mr[i].mp == ms[i].mp && mr[i].px[0] == ms[i].px[0] && mr[i].px[1] == ms[i].px[1] && mr[i].level == ms[i].level;
    std::printf("resident_same %d\n", same ? 1 : 0);
    return 0;
}
