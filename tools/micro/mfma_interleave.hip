// What does an independent instruction cost when it sits between two MFMAs of the same wave?  (gfx950, one wave per SIMD)
// burst of 64 v_mfma_f32_16x16x4_f32 on 4 rotating accumulators, with after every MFMA: nothing | 1 VALU | 2 VALU | 4 VALU |
// 1 ds_read_b128 | 1 global_load | 1 SALU.      hipcc --offload-arch=gfx950 -O3 -o mfma_interleave mfma_interleave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256, 1) void k(long long* out, float* sink, const float* g, int reps) {
    __shared__ f32x4 lds[256];
    lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f, v0 = 1.f, v1 = 2.f, v2 = 3.f, v3 = 4.f;
    f32x4 l = {0, 0, 0, 0}; float gl = 0.f; int sa = 0;
    long long tb = 0;
    for (int r = 0; r < reps; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        const long long t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (KIND == 1 || KIND == 2 || KIND == 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0));
                if (KIND == 2 || KIND == 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v1));
                if (KIND == 3) { asm volatile("v_add_f32 %0, %0, %0" : "+v"(v2)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(v3)); }
                if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"((unsigned)(threadIdx.x * 16)));
                if (KIND == 5) asm volatile("global_load_dword %0, %1, off" : "=v"(gl) : "v"(g + threadIdx.x));
                if (KIND == 6) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sa));
                if (KIND == 7) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(l) : "v"(g + threadIdx.x * 4));     // 1 KB per wave
                if (KIND == 8)     // the same 1 KB per wave straight into LDS (no VGPR destination)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + threadIdx.x * 4),
                                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        tb += __builtin_amdgcn_s_memtime() - t0;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (threadIdx.x == 0) out[0] = tb / reps;
    sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v0 + v1 + v2 + v3 + l[0] + gl + sa;
}
int main() {
    long long* d; float *s, *g; hipMalloc(&d, 64); hipMalloc(&s, 1024); hipMalloc(&g, 16384);
    const char* names[9] = {"nothing", "1 VALU", "2 VALU", "4 VALU", "1 ds_read_b128", "1 global_load", "1 SALU",
                            "1 global_load_dwordx4", "1 global_load_lds x4"};
    long long h;
#define RUN(K) hipLaunchKernelGGL(k<K>, dim3(1), dim3(256), 0, 0, d, s, g, 100); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); \
    printf("after every MFMA: %-22s -> %.1f cycles per MFMA\n", names[K], h / 64.0);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    return 0;
}
