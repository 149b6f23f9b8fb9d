// Does the workgroup dispatcher of MI355X map workgroup id -> XCD strictly (id mod 8), even when that XCD has no CU left
// for the workgroup while other XCDs stand empty?  40 workgroups that each need more than half a CU's LDS (one per CU)
// all have id = 0 mod 8; every one records its XCC id, its start time and spins for ~100 us.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_dispatch.hip -o tools/micro/xcd_dispatch && tools/micro/xcd_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <chrono>

__global__ void probe(unsigned long long* out, int spin_us, int modulo) {
    extern __shared__ float lds[];
    if ((blockIdx.x % 8) != (unsigned)modulo) return;
    const unsigned long long t0 = wall_clock64();                    // 100 MHz
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: wave/simd/cu/sh/se ...
    lds[threadIdx.x] = (float)t0;
    while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) {}
    if (threadIdx.x == 0) {
        out[3 * (blockIdx.x / 8) + 0] = xcc;
        out[3 * (blockIdx.x / 8) + 1] = t0;
        out[3 * (blockIdx.x / 8) + 2] = hwid;
    }
}

// mask bit x set: workgroups of XCD x record and spin, the others exit; slot = blockIdx (< 64 recorded)
__global__ void probe2(unsigned long long* out, int spin_us, int mask) {
    extern __shared__ float lds[];
    if (!((mask >> (blockIdx.x % 8)) & 1)) return;
    const unsigned long long t0 = wall_clock64();
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;
    lds[threadIdx.x] = (float)t0;
    while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) {}
    if (threadIdx.x == 0 && blockIdx.x < 64) { out[3 * blockIdx.x] = xcc; out[3 * blockIdx.x + 1] = t0; }
}

int main() {
    const int n = 40;
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 3 * n);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(d, 0xff, sizeof(unsigned long long) * 3 * n);
        hipLaunchKernelGGL(probe, dim3(8 * n), dim3(256), 84 * 1024, 0, d, 100, 0);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(3 * n);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 3 * n, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int i = 0; i < n; ++i) tmin = std::min(tmin, h[3 * i + 1]);
    int per_xcc[16] = {0}, late = 0;
    for (int i = 0; i < n; ++i) {
        per_xcc[h[3 * i] & 15]++;
        const double us = (double)(h[3 * i + 1] - tmin) / 100.0;
        if (us > 50) ++late;
        printf("wg %2d: xcc %llu  start +%.1f us  cu %llu se %llu\n", i, h[3 * i], us, (h[3 * i + 2] >> 8) & 15, (h[3 * i + 2] >> 13) & 7);
    }
    printf("workgroups per XCC:");
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\nstarted more than 50 us after the first: %d of %d\n", late, n);

    // ---- part 2: head-of-line blocking.  Kernel A fills XCD 0 (32 workgroups, 600 us).  Kernel B on another stream, started
    // while A runs: its workgroups 0, 8, 16, ... belong on XCD 0 (no room), 1, 9, 17, ... on XCD 1 (empty).  Do B's XCD-1
    // workgroups start at once, or only after its first XCD-0 workgroup has found a CU?
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    unsigned long long *da, *db;
    hipMalloc(&da, sizeof(unsigned long long) * 3 * 64);
    hipMalloc(&db, sizeof(unsigned long long) * 3 * 64);
    hipFuncSetAttribute((const void*)probe2, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(da, 0xff, sizeof(unsigned long long) * 3 * 64);
        hipMemset(db, 0xff, sizeof(unsigned long long) * 3 * 64);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe2, dim3(8 * 32), dim3(256), 84 * 1024, sa, da, 600, 1);        // mask 1: only XCD 0 works
        { const auto t = std::chrono::steady_clock::now();                                      // let A settle in first
          while (std::chrono::steady_clock::now() - t < std::chrono::microseconds(100)) {} }
        hipLaunchKernelGGL(probe2, dim3(8 * 8), dim3(256), 84 * 1024, sb, db, 20, 3);          // mask 3: XCD 0 and XCD 1 work
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> ha(3 * 64), hb(3 * 64);
    hipMemcpy(ha.data(), da, sizeof(unsigned long long) * 3 * 64, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), db, sizeof(unsigned long long) * 3 * 64, hipMemcpyDeviceToHost);
    // (A records its first 64 block ids only: blocks 0, 8, ..., 56 are 8 of its 32 XCD-0 workgroups)
    unsigned long long a0 = ~0ull, a1 = 0;
    int na = 0;
    for (int i = 0; i < 64; i += 8) if (ha[3 * i] != ~0ull) { a0 = std::min(a0, ha[3 * i + 1]); a1 = std::max(a1, ha[3 * i + 1]); ++na; }
    printf("\nkernel A: %d recorded XCD-0 workgroups, xcc of the first %llu, starts within %.1f us\n", na, ha[0], (double)(a1 - a0) / 100.0);
    printf("\nkernel B (8 workgroups per XCD, XCDs 0 and 1 work 20 us) beside kernel A (XCD 0 full for 600 us; B launched ~100 us after A):\n");
    for (int i = 0; i < 64; ++i)
        if (hb[3 * i] != ~0ull) printf("  B workgroup %2d: xcc %llu  start +%.1f us after A\n", i, hb[3 * i], (double)(hb[3 * i + 1] - a0) / 100.0);
    return 0;
}
