// example_rgbd.cpp — stand-in for the reference's Test/test_SpraseImg_alignment.cpp:85-168 (BASELINE config 1:
// an RGB-D sequence aligned frame by frame against its first frame) on data a test generates: the first frame
// becomes the reference (features detected on it, :127; 3-D points from the depth map, :130-137), every later
// frame is seeded with the previous pose (:147), aligned with Sprase_ImgAlign(4, 0, 30).Run(cur, ref) (:110,:150)
// and compared with its ground-truth pose (:153-157). TUM data is not in the image, so the frames come from
// tests/test_baseline_configs_gpu.py — or from a dataset in the TUM RGB-D layout converted by tools/tum_to_rgbd_bin.py;
// the flow, the class names and the printed quantities are the reference's.
//   usage: example_rgbd <sequence.bin> <features_out.bin>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dsdtm_host.hpp"

using namespace DSDTM;

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
    if (fread(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
}

// Camera::Pixel2Camera(px, 1.0) (src/Camera.cpp:173-185, no distortion), float intrinsics promoted at use
static std::array<double, 3> Pixel2Camera(const Camera& c, double u, double v) {
    return {{(u - (double)c.mcx) / (double)c.mfx, (v - (double)c.mcy) / (double)c.mfy, 1.0}};
}

// loadBlenderDepthmap (Test/test_SpraseImg_alignment.cpp:55-82): z-depth -> length of the viewing ray
static void BlenderDepthToRay(const Camera& c, std::vector<float>& depth) {
    for (int y = 0; y < c.mheight; ++y)
        for (int x = 0; x < c.mwidth; ++x) {
            std::array<double, 3> p = Pixel2Camera(c, x, y);
            const double n = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
            for (double& k : p) k /= n;
            const double ux = p[0] / p[2], uy = p[1] / p[2];
            float& d = depth[(size_t)y * c.mwidth + x];
            d = (float)(d * std::sqrt(ux * ux + uy * uy + 1.0));
        }
}

static SE3 inverse(const SE3& T) {
    SE3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i * 4 + j] = T.m[j * 4 + i];
    for (int i = 0; i < 3; ++i) r.m[i * 4 + 3] = -(r.m[i * 4] * T.m[3] + r.m[i * 4 + 1] * T.m[7] + r.m[i * 4 + 2] * T.m[11]);
    return r;
}
static SE3 mul(const SE3& A, const SE3& B) {
    SE3 r;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) r.m[i * 4 + j] = A.m[i * 4] * B.m[j] + A.m[i * 4 + 1] * B.m[4 + j] + A.m[i * 4 + 2] * B.m[8 + j];
        r.m[i * 4 + 3] = A.m[i * 4] * B.m[3] + A.m[i * 4 + 1] * B.m[7] + A.m[i * 4 + 2] * B.m[11] + A.m[i * 4 + 3];
    }
    return r;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s sequence.bin features_out.bin\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 2; }
    int32_t hdr[6];      // width, height, levels, n_frames, detection threshold, Max_fts
    rd(f, hdr, 6);
    const int W = hdr[0], H = hdr[1], levels = hdr[2], n_frames = hdr[3];
    float camf[5];
    rd(f, camf, 5);
    CameraPtr cam = std::make_shared<Camera>();
    cam->mfx = camf[0]; cam->mfy = camf[1]; cam->mcx = camf[2]; cam->mcy = camf[3]; cam->mf = camf[4];
    cam->mwidth = W; cam->mheight = H;
    Config::MaxPyraLevels() = levels;
    Config::Max_fts() = hdr[5];

    FramePtr frame_ref_, frame_cur_;
    Feature_detector feature_detector(W, H);
    Sprase_ImgAlign mSpraseAlign(4, 0, 30);                              // :110
    SE3 T_prev_w;
    for (int i = 0; i < n_frames; ++i) {
        Image8 img(W, H);
        rd(f, img.data.data(), img.data.size());
        SE3 T_w_g;                                                       // the frame's ground-truth pose (world -> camera)
        rd(f, T_w_g.m.data(), 12);
        if (i == 0) {
            frame_ref_ = std::make_shared<Frame>();
            frame_ref_->mCamera = cam;
            frame_ref_->mvImg_Pyr.push_back(std::move(img));
            frame_ref_->ComputeImagePyramidOnDevice(levels);             // Frame::ComputeImagePyramid on the device
            frame_ref_->Set_Pose(T_w_g);                                 // :122
            std::vector<float> depthmap((size_t)W * H);
            rd(f, depthmap.data(), depthmap.size());
            BlenderDepthToRay(*cam, depthmap);                           // :125
            feature_detector.detect(frame_ref_.get(), (double)hdr[4]);  // :127
            const SE3 T_g_w = inverse(frame_ref_->Get_Pose());
            // real depth maps have holes (TUM: 0 = no measurement; tools/tum_to_rgbd_bin.py): a feature without a depth cannot
            // carry a map point and is dropped here (the reference's Tracking leaves such features uninitialised, src/Frame.cpp:176-199)
            {
                std::vector<Feature> kept;
                for (const Feature& it : frame_ref_->mvFeatures)
                    if (depthmap[(size_t)(int)it.mpx_y * W + (int)it.mpx_x] > 0.0f) kept.push_back(it);
                frame_ref_->mvFeatures.swap(kept);
            }
            for (Feature& it : frame_ref_->mvFeatures) {                 // UndistortFeatures (:128) + :130-137
                std::array<double, 3> n = Pixel2Camera(*cam, it.mpx_x, it.mpx_y);
                const double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                for (int k = 0; k < 3; ++k) it.mNormal[k] = n[k] / nn;
                const float depth = depthmap[(size_t)(int)it.mpx_y * W + (int)it.mpx_x];
                const double pc[3] = {it.mNormal[0] * depth, it.mNormal[1] * depth, it.mNormal[2] * depth};
                for (int k = 0; k < 3; ++k)
                    it.mMptPose[k] = T_g_w.m[k * 4] * pc[0] + T_g_w.m[k * 4 + 1] * pc[1] + T_g_w.m[k * 4 + 2] * pc[2] + T_g_w.m[k * 4 + 3];
                it.mbInitial = true;
            }
            std::printf("features %zu\n", frame_ref_->mvFeatures.size());   // :139
            FILE* o = std::fopen(argv[2], "wb");                         // what the test hands to the CPU oracle
            if (!o) { std::perror("features_out"); return 2; }
            const int32_t n = (int32_t)frame_ref_->mvFeatures.size();
            std::fwrite(&n, 4, 1, o);
            for (const Feature& it : frame_ref_->mvFeatures) {
                const float p[2] = {it.mpx_x, it.mpx_y};
                std::fwrite(p, 4, 2, o); std::fwrite(it.mNormal.data(), 8, 3, o); std::fwrite(it.mMptPose.data(), 8, 3, o);
            }
            std::fclose(o);
            T_prev_w = frame_ref_->Get_Pose();                           // :140
            continue;
        }
        frame_cur_ = std::make_shared<Frame>();
        frame_cur_->mCamera = cam;
        frame_cur_->mvImg_Pyr.push_back(std::move(img));
        frame_cur_->ComputeImagePyramidOnDevice(levels);
        frame_cur_->Set_Pose(T_prev_w);                                  // :147
        const int n_tracked = mSpraseAlign.Run(frame_cur_, frame_ref_);  // :150
        const SE3 T_f_gt = mul(frame_cur_->Get_Pose(), inverse(T_w_g));  // :153
        const double te = std::sqrt(T_f_gt.m[3] * T_f_gt.m[3] + T_f_gt.m[7] * T_f_gt.m[7] + T_f_gt.m[11] * T_f_gt.m[11]);
        const double tr = T_f_gt.m[0] + T_f_gt.m[5] + T_f_gt.m[10];
        const double ang = std::acos(std::fmin(1.0, std::fmax(-1.0, (tr - 1.0) / 2.0)));
        std::printf("frame %d tracked %d translation_error %.9g angular_distance %.9g pose", i, n_tracked, te, ang);   // :156-157
        for (double v : frame_cur_->Get_Pose().m) std::printf(" %.17g", v);
        std::printf(" iters");
        for (int l = 0; l < 4; ++l) std::printf(" %d", mSpraseAlign.last_stats.iters[l]);
        std::printf("\n");
        T_prev_w = frame_cur_->Get_Pose();                               // :159
    }
    std::fclose(f);
    return 0;
}
