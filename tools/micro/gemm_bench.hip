// Stand-alone timing of the exact-fp32 linear-layer GEMMs of the path (mp_gemm.hip is compiled into this file):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/micro/gemm_bench.hip -o tools/micro/gemm_bench [-DVARIANT=n]
// Shapes at B = 256, T = 125 (M = 32 000 rows): joints.linear1 (K 60 -> N 256), pose+velocity linear1 stacked (K 132 -> N 512,
// two outputs), foot.linear1 (K 132 -> N 64), joints.linear2 (K 512 -> N 72), velocity.linear2 (K 256 -> N 72).
#include "../../mobileposer_amd/csrc/mp_gemm.hip"
#include <cstdio>
#include <vector>

static float* dalloc(size_t n, float v) {
    float* p; hipMalloc(&p, n * sizeof(float));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = v * (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f * v;
    hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    return p;
}

int main() {
    const int B = 256, T = 125, M = B * T;
    float* joints = dalloc((size_t)M * 72, 1.f);      // [B][T][72]
    float* imu = dalloc((size_t)M * 60, 1.f);         // [B][T][60]
    float* act = dalloc((size_t)M * 512, 1.f);        // [T][B][512]
    float* out1 = dalloc((size_t)M * 256, 0.f);
    float* out2 = dalloc((size_t)M * 256, 0.f);
    float* W = dalloc((size_t)640 * 512, 0.1f);
    float* bias = dalloc(640, 0.1f);
    float* Wf = dalloc((size_t)640 * 512, 0.f);
    float* chk1 = dalloc((size_t)M * 256, 0.f);
    float* chk2 = dalloc((size_t)M * 256, 0.f);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Shape { const char* name; int K, N, relu, two, user_in; };
    const Shape shapes[] = {{"joints.linear1  K=60  N=256", 60, 256, 1, 0, 2}, {"pose+vel.linear1 K=132 N=512 (stacked)", 132, 512, 1, 1, 1},
                            {"pose.linear1    K=132 N=256", 132, 256, 1, 0, 1}, {"foot.linear1    K=132 N=64 ", 132, 64, 1, 0, 1},
                            {"joints.linear2  K=512 N=72 ", 512, 72, 0, 0, 0}, {"velocity.linear2 K=256 N=72 ", 256, 72, 0, 0, 0}};
    for (const Shape& sh : shapes) {
        GemmArgs g;
        const RowMap none{nullptr, 0, 0, 0};
        if (sh.user_in == 1) { g.a0 = RowMap{joints, (long)T * 72, 72, 72}; g.a1 = RowMap{imu, (long)T * 60, 60, 60}; }
        else if (sh.user_in == 2) { g.a0 = RowMap{imu, (long)T * 60, 60, 60}; g.a1 = none; }
        else { g.a0 = RowMap{act, (long)sh.K, (long)B * sh.K, sh.K}; g.a1 = none; }
        const int bn = mp_gemm_pick_bn(sh.N);
        g.W = W; g.bias = bias; g.C = out1; g.C2 = sh.two ? out2 : nullptr; g.nsplit = sh.two ? 256 : 0;
        const int ncols = sh.two ? 256 : sh.N;
        if (sh.relu) { g.cStrideB = ncols; g.cStrideT = (long)B * ncols; }          // internal time-major output
        else { g.cStrideB = (long)T * ncols; g.cStrideT = ncols; }                 // linear2 writes the caller's layout
        g.M = M; g.N = sh.N; g.K = sh.K; g.Kpad = (sh.K + 31) / 32 * 32; g.B = B; g.relu = sh.relu;
        const int npad = (sh.N + bn - 1) / bn * bn;
        for (int variant = 0; variant < 2; ++variant) {
            g.Wf = nullptr; g.NB = 0;
            if (variant == 1) {
                mp_launch_pack_wfrag(W, Wf, npad, g.Kpad, s);
                g.Wf = Wf; g.NB = npad / 32;
                g.C = chk1; if (sh.two) g.C2 = chk2;
            }
            for (int i = 0; i < 5; ++i) mp_launch_gemm(g, bn, s);
            hipStreamSynchronize(s);
            const int reps = 50;
            hipEventRecord(e0, s);
            for (int i = 0; i < reps; ++i) mp_launch_gemm(g, bn, s);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double us = 1e3 * ms / reps, gf = 2.0 * M * (double)sh.N * sh.K * 1e-9;
            printf("%-42s %-9s %7.1f us  %6.1f TFLOP/s (%.0f %% of 157.3)   err=%s\n", sh.name, variant ? "frag-W" : "current", us, gf / us * 1e-3 * 1e3, 100.0 * gf / us / 157.3,
                   hipGetErrorString(hipGetLastError()));
        }
        {   // bitwise comparison of the two variants' outputs
            const size_t nout = (size_t)M * (sh.two ? 256 : (sh.relu ? ncols : ncols));
            std::vector<float> a(nout), b(nout);
            hipMemcpy(a.data(), out1, nout * sizeof(float), hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), chk1, nout * sizeof(float), hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < nout; ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
            printf("    outputs differ in %zu of %zu values\n", bad, nout);
        }
    }
    return 0;
}
