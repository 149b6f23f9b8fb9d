// issue rate of v_mfma_f32_16x16x32_bf16 (and friends) per SIMD: 1 or 2 waves per SIMD, distinct B operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int KIND, int NACC>
__global__ __launch_bounds__(512) void rate(float* out, const u32x4* w, int iters, long long* cyc) {
    u32x4 b[8];
    for (int i = 0; i < 8; ++i) b[i] = w[i * 64 + (threadIdx.x & 63)];
    u32x4 a0 = w[512 + (threadIdx.x & 63)], a1 = w[576 + (threadIdx.x & 63)];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x16 big[3] = {};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const u32x4 av = (r < 8) ? a0 : a1;
            const u32x4 bv = b[(r % 4) * 2 + ((r / 4) & 1)];
            if (KIND == 0) acc[r % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[r % NACC], 0, 0, 0);
            if (KIND == 1) acc[r % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[r & 3]), __uint_as_float(bv[r & 3]), acc[r % NACC], 0, 0, 0);
            if (KIND == 2) big[r % 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), big[r % 2], 0, 0, 0);
            if (KIND == 3) big[r % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av[r & 3]), __uint_as_float(bv[r & 3]), big[r % NACC], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][3];
    s += big[0][0] + big[1][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NACC>
void run(const char* name, int threads, float* out, u32x4* w, long long* cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate<KIND, NACC>), dim3(256), dim3(threads), 0, 0, out, w, 100, cyc);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate<KIND, NACC>), dim3(256), dim3(threads), 0, 0, out, w, iters, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double per = (double)c / iters / 12.0;
    const double ns = ms * 1e6 / iters / 12.0 / (threads / 256);
    printf("%-24s %d waves/SIMD, %d acc: %.1f memtime ticks per MFMA per wave; wall: %.2f ns per MFMA per SIMD (= %.1f cycles @2.4 GHz)\n", name, threads / 256, NACC, per, ns, ns * 2.4);
}

int main() {
    float* out; u32x4* w; long long* cyc;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&w, 1024 * 16)); CK(hipMalloc(&cyc, 8));
    CK(hipMemset(w, 0x3c, 1024 * 16));
    run<0, 4>("mfma_f32_16x16x32_bf16", 256, out, w, cyc);
    run<0, 4>("mfma_f32_16x16x32_bf16", 512, out, w, cyc);
    run<0, 2>("mfma_f32_16x16x32_bf16", 256, out, w, cyc);
    run<0, 1>("mfma_f32_16x16x32_bf16", 256, out, w, cyc);
    run<1, 4>("mfma_f32_16x16x4_f32", 256, out, w, cyc);
    run<1, 4>("mfma_f32_16x16x4_f32", 512, out, w, cyc);
    run<2, 2>("mfma_f32_32x32x16_bf16", 256, out, w, cyc);
    run<2, 2>("mfma_f32_32x32x16_bf16", 512, out, w, cyc);
    run<3, 2>("mfma_f32_32x32x2_f32", 256, out, w, cyc);         // (round 5: what the linear-layer kernels issue)
    run<3, 3>("mfma_f32_32x32x2_f32", 256, out, w, cyc);
    run<3, 2>("mfma_f32_32x32x2_f32", 512, out, w, cyc);
    return 0;
}
