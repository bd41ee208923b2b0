// fast_ref_shim.cpp — C entry points over the REFERENCE's own FAST sources (Thirdparty/fast, compiled
// where they lie under /root/reference by `make -C oracle ref`, output only into oracle/_ref/).
// TEST INFRASTRUCTURE ONLY: pins oracle/dsdtm_oracle.c's restatement of the detector's third-party
// part (fast_corner_detect_10_sse2, fast_corner_score_10, fast_nonmax_3x3 — the calls of reference
// src/Feature_detection.cpp:79-92) to the code the reference actually runs. Nothing here is shipped.
#include <fast/fast.h>

#include <cstddef>
#include <cstdint>
#include <vector>

extern "C" {

// corners of one 8-bit image as the reference's detect() obtains them; returns the number of corners,
// writes at most `cap` (x, y, score, is_nonmax) quadruples in the order the reference sees them
int fast_ref_detect(const uint8_t* img, int width, int height, int stride, int barrier, int32_t* out, int cap) {
    std::vector<fast::fast_xy> corners;
    fast::fast_corner_detect_10_sse2((fast::fast_byte*)img, width, height, stride, (short)barrier, corners);
    std::vector<int> scores, nm;
    fast::fast_corner_score_10((fast::fast_byte*)img, stride, corners, barrier, scores);
    fast::fast_nonmax_3x3(corners, scores, nm);
    std::vector<char> keep(corners.size(), 0);
    for (int i : nm) keep[(std::size_t)i] = 1;
    const int n = (int)corners.size();
    for (int i = 0; i < n && i < cap; ++i) {
        out[4 * i] = corners[(std::size_t)i].x; out[4 * i + 1] = corners[(std::size_t)i].y;
        out[4 * i + 2] = scores[(std::size_t)i]; out[4 * i + 3] = keep[(std::size_t)i];
    }
    return n;
}

}  // extern "C"
