// Micro-test: does a VALU kernel compute different results when it shares CUs with a wave that issues
// v_mfma_f32_16x16x32_bf16 back to back?  (observed: mp_r6d_ik lanes 48..63 wrong beside mp_lstm_x3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256, 2) void hog(float* out, int iters) {
    u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
    u32x4 b = {0x3f803f80u, 0x3f003f00u + threadIdx.x, 0x3f803f80u, 0x3e803e80u};
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            if (KIND == 0) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[r & 3], 0, 0, 0);
            else if (KIND == 1) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[0]), __uint_as_float(b[0]), acc[r & 3], 0, 0, 0);
            else {
                typedef float f32x16 __attribute__((ext_vector_type(16)));
                f32x16 c16 = __builtin_shufflevector(acc[0], acc[1], 0,1,2,3,4,5,6,7,0,1,2,3,4,5,6,7);
                c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c16, 0, 0, 0);
                acc[r & 3] = f32x4{c16[0], c16[5], c16[10], c16[15]};
            }
        }
        a[0] ^= (unsigned)i;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

__global__ __launch_bounds__(256, 2) void hog_lds(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    u32x4* w = reinterpret_cast<u32x4*>(lds);
    for (int i = threadIdx.x; i < 4096; i += 256) w[i] = u32x4{0x3f803f80u + i, 0x3f003f00u, 0x3e803e80u, 0x3f803f00u};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 x0 = {0x3f803f80u + threadIdx.x, 0x3f813f80u, 0x3f003f00u, 0x3e803e80u}, x1 = x0;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int i = 0; i < iters; ++i) {
        u32x4 wl[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) wl[t] = w[(wave * 8 + t) * 64 + lane + ((i & 1) << 11)];
        u32x4 hi, lo;
        hi[0] = __builtin_amdgcn_perm(x0[1], x0[0], 0x07060302u); hi[1] = __builtin_amdgcn_perm(x0[3], x0[2], 0x07060302u);
        hi[2] = __builtin_amdgcn_perm(x1[1], x1[0], 0x07060302u); hi[3] = __builtin_amdgcn_perm(x1[3], x1[2], 0x07060302u);
        lo[0] = __builtin_amdgcn_perm(x0[1], x0[0], 0x05040100u); lo[1] = __builtin_amdgcn_perm(x0[3], x0[2], 0x05040100u);
        lo[2] = __builtin_amdgcn_perm(x1[1], x1[0], 0x05040100u); lo[3] = __builtin_amdgcn_perm(x1[3], x1[2], 0x05040100u);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi), __builtin_bit_cast(bf16x8, wl[2 * t]), acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi), __builtin_bit_cast(bf16x8, wl[2 * t + 1]), acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, lo), __builtin_bit_cast(bf16x8, wl[2 * t]), acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        x0[0] += 0x10000u; x1[2] ^= (unsigned)i;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

// victim kernels: out[i] = f(in[i], in2[i])
template <int OP>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ o, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y[i], r = 0.f;
#pragma unroll 1
    for (int rep = 0; rep < 8; ++rep) {
        if (OP == 0) r += a / (b + rep);                       // v_div_scale / v_div_fmas / v_div_fixup
        else if (OP == 1) r += sqrtf(fabsf(a) + rep);          // v_sqrt + refinement
        else if (OP == 2) r += __builtin_amdgcn_rcpf(b + rep); // trans
        else if (OP == 3) r += __builtin_amdgcn_exp2f(a * 0.1f + rep);
        else if (OP == 4) r = fmaf(a, b, r) + rep;             // plain VALU
        else if (OP == 5) { float t = a * b - b * (a + rep); r += (t != t) ? 0.f : t; }   // cmp + cndmask
        else if (OP == 6) { const float c0 = a * b, c1 = b * (a + rep), c2 = a * a; r += (c0 - c1) + c2; }
    }
    o[i] = r;
}
// packed fp32: cross products on float2 pairs
__global__ __launch_bounds__(256) void victim_pk(const float2* __restrict__ x, const float2* __restrict__ y, float2* __restrict__ o, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 a = x[i], b = y[i], r = {0.f, 0.f};
#pragma unroll 1
    for (int rep = 0; rep < 8; ++rep) {
        float2 t = {a.x * b.x, a.y * b.y};
        float2 u = {b.x * (a.x + rep), b.y * (a.y + rep)};
        r.x += t.x - u.x; r.y += t.y - u.y;
    }
    o[i] = r;
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class L>
void run_case(const char* name, L launch_victim, float* o_dev, size_t n_out, hipStream_t sv, hipStream_t sh, float* hogout) {
    std::vector<float> ref(n_out), got(n_out);
    launch_victim(sv);
    CK(hipStreamSynchronize(sv));
    CK(hipMemcpy(ref.data(), o_dev, n_out * 4, hipMemcpyDeviceToHost));
    for (int kind = 0; kind < 4; ++kind) {
        long bad = 0; unsigned long long lanes = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(o_dev, 0, n_out * 4, sv));
            CK(hipStreamSynchronize(sv));
            if (kind == 0) hipLaunchKernelGGL(hog<0>, dim3(512), dim3(256), 0, sh, hogout, 6000);
            else if (kind == 1) hipLaunchKernelGGL(hog<1>, dim3(512), dim3(256), 0, sh, hogout, 6000);
            else if (kind == 2) hipLaunchKernelGGL(hog<2>, dim3(512), dim3(256), 0, sh, hogout, 3000);
            else hipLaunchKernelGGL(hog_lds, dim3(512), dim3(256), 65536, sh, hogout, 6000);
            for (int k = 0; k < 10; ++k) launch_victim(sv);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), o_dev, n_out * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n_out; ++i)
                if (memcmp(&ref[i], &got[i], 4) != 0) { ++bad; lanes |= 1ull << (i & 63); }
        }
        printf("%-28s beside %-22s: %ld mismatching values, lane mask %016llx\n", name,
               kind == 0 ? "mfma_16x16x32_bf16" : (kind == 1 ? "mfma_16x16x4_f32" : (kind == 2 ? "mfma_32x32x16_bf16" : "lds+perm+mfma_bf16")), bad, lanes);
    }
}

int main() {
    const long n = 768 * 1024;
    std::vector<float> hx(2 * n), hy(2 * n);
    srand(1);
    for (long i = 0; i < 2 * n; ++i) { hx[i] = (rand() / (float)RAND_MAX) * 2 - 1; hy[i] = (rand() / (float)RAND_MAX) + 0.5f; }
    float *x, *y, *o, *hogout;
    CK(hipMalloc(&x, 2 * n * 4)); CK(hipMalloc(&y, 2 * n * 4)); CK(hipMalloc(&o, 2 * n * 4)); CK(hipMalloc(&hogout, 512 * 256 * 4));
    CK(hipMemcpy(x, hx.data(), 2 * n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(y, hy.data(), 2 * n * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)hog_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipStream_t sv, sh;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
    const dim3 g((n + 255) / 256), b(256);
    run_case("div (v_div_*)", [&](hipStream_t s) { hipLaunchKernelGGL(victim<0>, g, b, 0, s, x, y, o, n); }, o, n, sv, sh, hogout);
    run_case("sqrtf", [&](hipStream_t s) { hipLaunchKernelGGL(victim<1>, g, b, 0, s, x, y, o, n); }, o, n, sv, sh, hogout);
    run_case("v_rcp_f32", [&](hipStream_t s) { hipLaunchKernelGGL(victim<2>, g, b, 0, s, x, y, o, n); }, o, n, sv, sh, hogout);
    run_case("v_exp_f32", [&](hipStream_t s) { hipLaunchKernelGGL(victim<3>, g, b, 0, s, x, y, o, n); }, o, n, sv, sh, hogout);
    run_case("fma", [&](hipStream_t s) { hipLaunchKernelGGL(victim<4>, g, b, 0, s, x, y, o, n); }, o, n, sv, sh, hogout);
    run_case("cmp+cndmask", [&](hipStream_t s) { hipLaunchKernelGGL(victim<5>, g, b, 0, s, x, y, o, n); }, o, n, sv, sh, hogout);
    run_case("mul/sub scalar", [&](hipStream_t s) { hipLaunchKernelGGL(victim<6>, g, b, 0, s, x, y, o, n); }, o, n, sv, sh, hogout);
    run_case("packed fp32 (v_pk_*)", [&](hipStream_t s) { hipLaunchKernelGGL(victim_pk, g, b, 0, s, (const float2*)x, (const float2*)y, (float2*)o, n); }, o, 2 * n, sv, sh, hogout);
    return 0;
}
