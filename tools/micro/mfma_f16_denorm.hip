// Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs on gfx950?  (round 4: can a two-piece fp16 split of an fp32
// operand keep its low piece -- which is subnormal in fp16 for |v| < 0.25 -- without scaling?)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f16_denorm.hip -o tools/micro/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float a, float b, float* out) {
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)A[0]; out[2] = (float)B[0]; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    const float cases[][2] = {{1.0f, 1.0f}, {9.5367431640625e-07f /*2^-20: subnormal fp16*/, 1.0f}, {1.0f, 9.5367431640625e-07f},
                              {5.9604644775390625e-08f /*2^-24: smallest subnormal*/, 1.0f}, {3.0517578125e-05f /*2^-15*/, 3.0517578125e-05f}};
    for (auto& c : cases) {
        k<<<1, 64>>>(c[0], c[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%.10e (as fp16 %.10e) b=%.10e (as fp16 %.10e): mfma sum over K=32: %.10e   expected %.10e\n", c[0], h[1], c[1], h[2], h[0], 32.0 * (double)h[1] * (double)h[2]);
    }
    return 0;
}
