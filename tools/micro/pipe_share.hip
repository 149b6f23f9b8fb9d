// How do two waves of one SIMD share the matrix pipe and the VALU issue port?  (gfx950, v_mfma_f32_16x16x4_f32 = 8 passes)
// Waves 0..3 ("A", one per SIMD): repeated { burst of 64 MFMAs, NOPA wait states after each; accumulators -> LDS;
// 4-wave barrier on an LDS counter }.  Waves 4..7 ("B", the second wave of each SIMD, tools/micro/hwid.hip):
// idle, or an endless MFMA stream with NOPB wait states after each MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o pipe_share pipe_share.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void nops() {
    if constexpr (N >= 8) { asm volatile("s_nop 7" :::); nops<N - 8>(); }
    else if constexpr (N > 0) { asm volatile("s_nop %0" :: "n"(N - 1)); }
}
template <int NOPA, int NOPB, bool BON>
__global__ __launch_bounds__(512, 1) void k(long long* out, float* sink, int reps) {
    __shared__ f32x4 lds[512 * 4];
    __shared__ volatile int stop;
    __shared__ unsigned cnt;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) { stop = 0; cnt = 0; }
    __syncthreads();
    long long nB = 0;
    const long long tB0 = __builtin_amdgcn_s_memtime();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = 1.0f + lane * 1e-3f, b = 0.5f;
    if (wave < 4) {
        long long tb = 0, tr = 0, tq = 0;
        for (int r = 0; r < reps; ++r) {
            const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) { acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0); nops<NOPA>(); }
            __builtin_amdgcn_sched_barrier(0);
            const long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int t = 0; t < 4; ++t) lds[threadIdx.x * 4 + t] = acc[t];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long long t2 = __builtin_amdgcn_s_memtime();
            if (lane == 0) __hip_atomic_fetch_add(&cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * (unsigned)(r + 1)) {}
            const long long t3 = __builtin_amdgcn_s_memtime();
            tb += t1 - t0; tr += t2 - t1; tq += t3 - t2;
            a += 1e-6f;
        }
        if (lane == 0 && wave == 0) { out[0] = tb / reps; out[1] = tr / reps; out[2] = tq / reps; }
        if (wave == 0) stop = 1;
    } else if (BON) {
        while (!stop) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) { acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0); nops<NOPB>(); }
            nB += 32;
        }
        if (wave == 4 && lane == 0) { out[3] = nB; out[4] = (long long)__builtin_amdgcn_s_memtime() - tB0; }
    }
    sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + a;
}
template <int NOPA, int NOPB, bool BON>
void run(long long* d, float* s) {
    hipLaunchKernelGGL((k<NOPA, NOPB, BON>), dim3(1), dim3(512), 0, 0, d, s, 200);
    long long h[5]; hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    printf("A: +%2d wait states per MFMA | B: %-22s -> A burst %5lld cycles (%.1f per MFMA), acc->LDS %3lld, 4-wave LDS barrier %5lld",
           NOPA, BON ? (NOPB ? "MFMA stream, spaced" : "MFMA stream, back-to-back") : "idle", h[0], h[0] / 64.0, h[1], h[2]);
    if (BON) printf(" | B: %.1f cycles per MFMA over the run (+%d wait states)", (double)h[4] / (double)h[3], NOPB);
    printf("\n");
}
int main() {
    long long* d; float* s; hipMalloc(&d, 128); hipMalloc(&s, 512 * 4);
    run<0, 0, false>(d, s); run<4, 0, false>(d, s); run<5, 0, false>(d, s); run<6, 0, false>(d, s); run<7, 0, false>(d, s);
    run<0, 0, true>(d, s); run<0, 4, true>(d, s); run<0, 5, true>(d, s); run<0, 6, true>(d, s); run<0, 7, true>(d, s);
    run<6, 6, true>(d, s); run<7, 7, true>(d, s);
    return 0;
}
