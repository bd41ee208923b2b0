// example_batch.cpp — a single-process C++ caller in the shape of the reference's process (src/System.cpp:39-55: one
// process, a few threads) spreading BASELINE config 4's independent frame pairs over several GPUs through the C ABI:
// one dsdtm_ctx per shard (device g % device_count); the host batch is cut into contiguous blocks, each run on its own
// host thread. Default: dsdtm_sparse_align_batch_streamed — only level 0 of every frame crosses the link (the packed
// host pyramids double as an image array: level 0 sits at offset 0, image_pitch = pyr_pitch), pyramids are built on the
// device, chunks of pairs are uploaded beside the alignment of the previous chunk. "sharded": the whole-pyramid upload of
// dsdtm_sparse_align_batch_sharded (same results, bit for bit). Reads a batch dumped by tests/test_sharded_gpu.py.
// (A production caller pins its arrays — hipHostMalloc / hipHostRegister — so that the copies run at link speed.)
//   usage: example_batch <batch.bin> <n_contexts> [sharded]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/dsdtm_amd.h"

template <typename T>
static void rd(FILE* f, std::vector<T>& v, size_t n) {
    v.resize(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s batch.bin n_contexts\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 2; }
    int32_t hdr[10];     // n_pairs, max_features, levels, width, height, max_level, min_level, max_iters, min_fts, pitch
    if (fread(hdr, 4, 10, f) != 10) return 2;
    float camf[5];
    if (fread(camf, 4, 5, f) != 5) return 2;
    const int P = hdr[0], N = hdr[1], L = hdr[2];
    dsdtm_batch_desc b;
    std::memset(&b, 0, sizeof b);
    b.n_pairs = P; b.max_features = N; b.levels = L; b.pyr_pitch = (size_t)hdr[9];
    size_t off = 0;
    for (int l = 0, w = hdr[3], h = hdr[4]; l < L; ++l, w = (w + 1) / 2, h = (h + 1) / 2) {
        b.width[l] = w; b.height[l] = h; b.stride[l] = w; b.level_offset[l] = off;
        off += ((size_t)w * h + 63) / 64 * 64;
    }
    std::vector<uint8_t> ref, cur, ini;
    std::vector<float> px;
    std::vector<double> be, pw, Tr, Tc;
    rd(f, ref, (size_t)P * b.pyr_pitch); rd(f, cur, (size_t)P * b.pyr_pitch);
    rd(f, px, (size_t)P * N * 2); rd(f, be, (size_t)P * N * 3); rd(f, pw, (size_t)P * N * 3); rd(f, ini, (size_t)P * N);
    rd(f, Tr, (size_t)P * 12); rd(f, Tc, (size_t)P * 12);
    std::fclose(f);
    std::vector<int32_t> nt(P, -1);
    std::vector<dsdtm_align_stats> st(P);
    b.ref_pyr = ref.data(); b.cur_pyr = cur.data(); b.px_xy = px.data(); b.bearing = be.data(); b.p_world = pw.data();
    b.initial = ini.data(); b.T_ref_w = Tr.data(); b.T_cur_w = Tc.data(); b.n_tracked = nt.data(); b.stats = st.data();
    const dsdtm_camera cam{camf[0], camf[1], camf[2], camf[3], camf[4], hdr[3], hdr[4]};
    const dsdtm_align_params prm{hdr[5], hdr[6], hdr[7], hdr[8]};

    const int G = std::atoi(argv[2]);
    const int ndev = dsdtm_device_count();
    if (G <= 0 || ndev <= 0) { std::fprintf(stderr, "no device / bad context count\n"); return 3; }
    std::vector<dsdtm_ctx*> ctx(G, nullptr);
    for (int g = 0; g < G; ++g)
        if (dsdtm_create(g % ndev, &ctx[g]) != DSDTM_OK) { std::fprintf(stderr, "dsdtm_create: %s\n", dsdtm_last_error(nullptr)); return 3; }
    int rc;
    if (argc > 3 && std::strcmp(argv[3], "sharded") == 0) {
        rc = dsdtm_sparse_align_batch_sharded(ctx.data(), G, &b, &cam, &prm);
    } else {
        dsdtm_stream_desc s;
        std::memset(&s, 0, sizeof s);
        s.n_pairs = P; s.max_features = N; s.levels = L; s.width = hdr[3]; s.height = hdr[4];
        s.row_stride = hdr[3]; s.image_pitch = b.pyr_pitch;
        s.ref_image = ref.data(); s.cur_image = cur.data();
        s.px_xy = px.data(); s.bearing = be.data(); s.p_world = pw.data(); s.initial = ini.data();
        s.T_ref_w = Tr.data(); s.T_cur_w = Tc.data(); s.n_tracked = nt.data(); s.stats = st.data();
        rc = dsdtm_sparse_align_batch_streamed(ctx.data(), G, &s, 2, &cam, &prm);      // chunks of 2 pairs: the pipeline at toy size
    }
    if (rc != DSDTM_OK) { std::fprintf(stderr, "batch failed: %d (%s)\n", rc, dsdtm_last_error(ctx[0])); return 4; }
    for (int g = 0; g < G; ++g) {
        int lo, hi;
        dsdtm_shard_range(P, G, g, &lo, &hi);
        std::printf("shard %d device %d pairs %d %d\n", g, g % ndev, lo, hi);
    }
    for (int i = 0; i < P; ++i) {
        std::printf("pair %d n_tracked %d iters", i, nt[i]);
        for (int l = 0; l < L; ++l) std::printf(" %d", st[i].iters[l]);
        std::printf(" pose");
        for (int k = 0; k < 12; ++k) std::printf(" %.17g", Tc[(size_t)i * 12 + k]);
        std::printf("\n");
    }
    for (dsdtm_ctx* c : ctx) dsdtm_destroy(c);
    return 0;
}
