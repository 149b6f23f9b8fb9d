// which SIMD does wave w of a 512-thread workgroup land on?  (s_getreg_b32 HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] ...)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 1) void k(unsigned* out) {
    extern __shared__ float lds[];
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    lds[threadIdx.x] = 1.f;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 150 * 1024, 0, d);
    unsigned h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int pat[8][4] = {{0}};
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) pat[w][(h[b * 8 + w] >> 4) & 3]++;
    for (int w = 0; w < 8; ++w) printf("wave %d: simd histogram %d %d %d %d   (block 0: hw_id %08x simd %u wave_slot %u)\n", w, pat[w][0], pat[w][1], pat[w][2], pat[w][3], h[w], (h[w] >> 4) & 3, h[w] & 15);
    return 0;
}
