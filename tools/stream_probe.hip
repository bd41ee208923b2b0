// Streaming ceilings of this box for the pyramid kernel's traffic shape (standalone: hipcc --offload-arch=gfx950 -O3
// tools/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe): read-only, 4:1 read:write, 1:1 copy, each with
// coalesced 16-byte accesses per lane and U independent loads in flight per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, int MODE>   // MODE 0: read only; 1: 4 reads : 1 write; 2: copy
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
    const size_t chunk = (size_t)blockDim.x * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n16; base += (size_t)gridDim.x * chunk) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[base + (size_t)u * blockDim.x + threadIdx.x];
        if (MODE == 0) {
            uint4 a = v[0];
#pragma unroll
            for (int u = 1; u < U; ++u) { a.x ^= v[u].x; a.y ^= v[u].y; a.z ^= v[u].z; a.w ^= v[u].w; }
            if (a.x == 0x12345678u && a.y == 0x9abcdef0u) out[0] = a;
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < U; u += 4) {
                uint4 a = v[u];
                a.x ^= v[u + 1].x ^ v[u + 2].x ^ v[u + 3].x; a.y ^= v[u + 1].y ^ v[u + 2].y ^ v[u + 3].y;
                a.z ^= v[u + 1].z ^ v[u + 2].z ^ v[u + 3].z; a.w ^= v[u + 1].w ^ v[u + 2].w ^ v[u + 3].w;
                out[(base >> 2) + (size_t)(u >> 2) * blockDim.x + threadIdx.x] = a;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) out[base + (size_t)u * blockDim.x + threadIdx.x] = v[u];
        }
    }
}

// The pyramid kernel's read pattern without its arithmetic: a 640-byte image row is 80 strips, a thread loads 16 bytes
// at 8-byte lane stride (neighbouring lanes overlap by half) from U consecutive rows; consecutive threads take
// consecutive strips, then the next chunk of RSTEP rows (U - RSTEP rows of halo are read twice, as 19 rows per 16).
template <int U, int RSTEP>
__global__ __launch_bounds__(256) void probe_rows(const uint8_t* __restrict__ in, uint4* __restrict__ out, int rows_total) {
    const int strips = 80, stride = 640;
    const int task = blockIdx.x * 256 + threadIdx.x;
    const int chunk = task / strips, strip = task - chunk * strips;
    const int r0 = chunk * RSTEP;
    if (r0 + U > rows_total) return;
    const uint8_t* p = in + (size_t)r0 * stride + (strip == 0 ? 0 : strip * 8 - 4);
    uint4 a = make_uint4(0, 0, 0, 0);
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const uint32_t* q = (const uint32_t*)(p + (size_t)u * stride); v[u] = make_uint4(q[0], q[1], q[2], q[3]); }
#pragma unroll
    for (int u = 0; u < U; ++u) { a.x ^= v[u].x; a.y ^= v[u].y; a.z ^= v[u].z; a.w ^= v[u].w; }
    if (a.x == 0x12345678u && a.y == 0x9abcdef0u) out[0] = a;
}
template <int U, int RSTEP>
static int run_rows(const uint8_t* in, uint4* out, size_t bytes) {
    const int rows_total = (int)(bytes / 640);
    const int chunks = (rows_total - U) / RSTEP + 1;
    const int grid = (chunks * 80 + 255) / 256;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe_rows<U, RSTEP>), dim3(grid), dim3(256), 0, 0, in, out, rows_total);
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe_rows<U, RSTEP>), dim3(grid), dim3(256), 0, 0, in, out, rows_total);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("rows pattern: %2d rows in flight, step %2d (16 B per lane at 8 B stride): %.3f ms  %.2f TB/s of distinct bytes\n", U, RSTEP, ms, (double)bytes / ms / 1e9);
    return 0;
}

template <int U, int MODE>
static int run(const char* name, const uint4* in, uint4* out, size_t bytes, int grid) {
    const size_t n16 = bytes / 16;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<U, MODE>), dim3(grid), dim3(256), 0, 0, in, out, n16);
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<U, MODE>), dim3(grid), dim3(256), 0, 0, in, out, n16);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double moved = MODE == 0 ? bytes : MODE == 1 ? bytes * 1.25 : bytes * 2.0;
    printf("%-28s U=%2d grid %6d: %.3f ms  %.2f TB/s (bytes read + written)\n", name, U, grid, ms, moved / ms / 1e9);
    return 0;
}

int main() {
    const size_t bytes = (size_t)2048 * 403200 / 4096 * 4096;      // what 2048 pyramids read
    uint4 *in, *out;
    CHECK(hipMalloc(&in, bytes)); CHECK(hipMalloc(&out, bytes)); CHECK(hipMemset(in, 1, bytes));
    run_rows<19, 16>((const uint8_t*)in, out, bytes); run_rows<16, 16>((const uint8_t*)in, out, bytes); run_rows<11, 8>((const uint8_t*)in, out, bytes);
    run_rows<8, 8>((const uint8_t*)in, out, bytes); run_rows<35, 32>((const uint8_t*)in, out, bytes);
    for (int grid : {8192}) {
        run<4, 0>("read only", in, out, bytes, grid); run<8, 0>("read only", in, out, bytes, grid); run<16, 0>("read only", in, out, bytes, grid);
        run<4, 1>("4 reads : 1 write", in, out, bytes, grid); run<8, 1>("4 reads : 1 write", in, out, bytes, grid); run<16, 1>("4 reads : 1 write", in, out, bytes, grid);
        run<4, 2>("copy", in, out, bytes, grid); run<8, 2>("copy", in, out, bytes, grid);
    }
    return 0;
}
