// Standalone probe for the gfx950 co-execution glitch (profiles/NOTES_r01-r03.md 4.3): the library's own r6d -> rotation kernel, compiled
// WITH packed-fp32 instructions, beside synthetic MFMA hogs of varying register footprint.  No library involved.
//   hipcc --offload-arch=gfx950 -O3 -o corun2 corun2.hip && ./corun2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- victim: Gram-Schmidt 6D -> R, parent^T * child (same arithmetic as mp_r6d_ik; the cross product compiles to
// v_pk_mul_f32 with op_sel, the normalisations to v_sqrt / v_div_* sequences)
__device__ __forceinline__ float nan0(float x) { return x != x ? 0.f : x; }
__device__ __forceinline__ void gs(const float* __restrict__ p, float R[9]) {
#pragma clang fp contract(off)
    const float ax = p[0], ay = p[1], az = p[2], bx = p[3], by = p[4], bz = p[5];
    const float na = sqrtf((ax * ax + ay * ay) + az * az);
    const float c0x = ax / na, c0y = ay / na, c0z = az / na;
    const float d = (c0x * bx + c0y * by) + c0z * bz;
    const float ux = bx - d * c0x, uy = by - d * c0y, uz = bz - d * c0z;
    const float nu = sqrtf((ux * ux + uy * uy) + uz * uz);
    const float c1x = ux / nu, c1y = uy / nu, c1z = uz / nu;
    const float c2x = c0y * c1z - c0z * c1y, c2y = c0z * c1x - c0x * c1z, c2z = c0x * c1y - c0y * c1x;
    R[0] = nan0(c0x); R[1] = nan0(c1x); R[2] = nan0(c2x); R[3] = nan0(c0y); R[4] = nan0(c1y); R[5] = nan0(c2y);
    R[6] = nan0(c0z); R[7] = nan0(c1z); R[8] = nan0(c2z);
}
__global__ __launch_bounds__(256) void victim(const float* __restrict__ r6d, long N, float* __restrict__ pose) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N * 16) return;
    const long n = gid / 16;
    const int i = (int)(gid - n * 16);
    const float* row = r6d + n * 96;
    float G[9], P[9], out[9];
    gs(row + 6 * i, G);
    gs(row + 6 * ((i * 7 + 3) & 15), P);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[r * 3 + c] = P[r] * G[c] + P[3 + r] * G[3 + c] + P[6 + r] * G[6 + c];
    float* o = pose + gid * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = out[k];
}

// ---- aggressors: back-to-back bf16 MFMAs, NACC accumulator tiles + PAD dead-weight registers (so that 1..2 waves per SIMD
// leave different amounts of the 512-entry register file to the victim), optionally LDS traffic in between
template <int NACC, int LDSB>
__global__ __launch_bounds__(512) void hog(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
    u32x4 b = {0x3f803f80u, 0x3f003f00u + threadIdx.x, 0x3f803f80u, 0x3e803e80u};
    f32x4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, (float)t};
    if (LDSB) for (int i = threadIdx.x; i < LDSB / 4; i += blockDim.x) lds[i] = i;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[t], 0, 0, 0);
        if (LDSB) { a[1] ^= lds[(threadIdx.x * 4 + i) & (LDSB / 4 - 1)]; }
        a[0] ^= (unsigned)i;
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const long N = 32000;                       // frames; 16 "joints" each
    std::vector<float> h(N * 96);
    srand(3);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    float *r6d, *pose, *hogout;
    CK(hipMalloc(&r6d, h.size() * 4)); CK(hipMalloc(&pose, N * 16 * 9 * 4)); CK(hipMalloc(&hogout, 1024 * 512 * 4));
    CK(hipMemcpy(r6d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sh;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
    const size_t nout = (size_t)N * 16 * 9;
    std::vector<float> ref(nout), got(nout);
    const dim3 g((N * 16 + 255) / 256), b(256);
    hipLaunchKernelGGL(victim, g, b, 0, sv, r6d, N, pose);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), pose, nout * 4, hipMemcpyDeviceToHost));
    auto run = [&](const char* name, auto launch_hog) {
        long bad = 0; unsigned long long lanes = 0; long cols[9] = {0};
        for (int rep = 0; rep < 10; ++rep) {
            CK(hipMemsetAsync(pose, 0, nout * 4, sv));
            CK(hipDeviceSynchronize());
            launch_hog();
            for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(victim, g, b, 0, sv, r6d, N, pose);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), pose, nout * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nout; ++i)
                if (memcmp(&ref[i], &got[i], 4) != 0) { ++bad; lanes |= 1ull << ((i / 9) & 63); ++cols[i % 9]; }
        }
        printf("%-44s: %6ld wrong words, lane mask %016llx, by element %ld %ld %ld %ld %ld %ld %ld %ld %ld\n", name, bad, lanes,
               cols[0], cols[1], cols[2], cols[3], cols[4], cols[5], cols[6], cols[7], cols[8]);
    };
    run("alone", [&] {});
#define HOG(NACC, LDSB, WG, GRID) run("hog<" #NACC "," #LDSB "> " #WG " thr x " #GRID, [&] { \
        CK(hipFuncSetAttribute((const void*)hog<NACC, LDSB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB)); \
        hipLaunchKernelGGL((hog<NACC, LDSB>), dim3(GRID), dim3(WG), LDSB, sh, hogout, 40000); })
    HOG(4, 0, 256, 1024);
    HOG(16, 0, 256, 512);
    HOG(32, 0, 512, 256);
    HOG(40, 0, 512, 256);
    HOG(48, 0, 512, 256);
    HOG(32, 65536, 512, 256);
    HOG(48, 131072, 512, 256);
    HOG(48, 131072, 256, 256);
    HOG(56, 0, 256, 512);
    return 0;
}
