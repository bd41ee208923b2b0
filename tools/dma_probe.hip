#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define DEFK(NAME, SZ) \
__global__ void NAME(const uint32_t* src, const int* idx, uint32_t* out) { \
    __shared__ uint32_t buf[64 * 8]; \
    const int lane = threadIdx.x; \
    for (int i = lane; i < 64 * 8; i += 64) buf[i] = 0xdeadbeefu; \
    __syncthreads(); \
    const uint32_t* g = src + idx[lane]; \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, \
                                     (__attribute__((address_space(3))) void*)&buf[0], SZ, 0, 0); \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    __syncthreads(); \
    for (int i = lane; i < 64 * 8; i += 64) out[i] = buf[i]; \
}
DEFK(k4, 4)
DEFK(k12, 12)
DEFK(k16, 16)

int main() {
    std::vector<uint32_t> h(4096); for (int i = 0; i < 4096; ++i) h[i] = i;
    std::vector<int> idx(64); for (int i = 0; i < 64; ++i) idx[i] = 1000 + 7 * i;   // lane l reads words 1000+7l ..
    uint32_t *d, *o; int* di;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 512 * 4); hipMalloc(&di, 64 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice); hipMemcpy(di, idx.data(), 64 * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> r(512);
    for (int sz : {4, 12, 16}) {
        if (sz == 4) hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, d, di, o);
        if (sz == 12) hipLaunchKernelGGL(k12, dim3(1), dim3(64), 0, 0, d, di, o);
        if (sz == 16) hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, d, di, o);
        hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
        printf("size %d: first 24 LDS dwords:", sz);
        for (int i = 0; i < 24; ++i) printf(" %u", r[i] == 0xdeadbeefu ? 0u : r[i]);
        printf("\n   dwords 64..75:"); for (int i = 64; i < 76; ++i) printf(" %u", r[i] == 0xdeadbeefu ? 0u : r[i]);
        printf("\n   dwords 128..139:"); for (int i = 128; i < 140; ++i) printf(" %u", r[i] == 0xdeadbeefu ? 0u : r[i]);
        printf("\n");
    }
    return 0;
}
