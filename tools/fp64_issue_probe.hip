// Micro-benchmark: cycles per v_fma_f64 for C independent dependency chains per wave, with W waves
// per SIMD (one workgroup per CU, 256*W threads). Build: hipcc --offload-arch=gfx950 -O2 -o probe fp64_issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int C>
__global__ void k(double* out, unsigned long long* cyc, int iters) {
    double acc[C];
    const double a = 1.0000001, b = 1e-9;
    for (int c = 0; c < C; ++c) acc[c] = threadIdx.x * 1e-3 + c;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = __builtin_fma(acc[c], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int c = 0; c < C; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int C> void run(int waves_per_simd) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * threads * blocks); hipMalloc(&cyc, 8 * blocks);
    hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    unsigned long long h[256]; hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += h[i]; m /= blocks;
    const double fmas_per_wave = (double)iters * 8 * C;
    printf("chains %d waves/SIMD %d: %.2f cycles per FMA per wave, %.2f cycles per FMA per SIMD\n", C, waves_per_simd,
           m / fmas_per_wave, m / (fmas_per_wave * waves_per_simd));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 3, 4}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
