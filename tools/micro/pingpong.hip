// one-way hand-off latency between two workgroups through L2 granules (same XCD vs different XCD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int STORE_SCOPE_AGENT, int NGRAN>
__global__ void pp(u64* buf, int iters, int peer_block, long long* out, unsigned* xcc) {
    // participants: block 0 and block peer_block; everyone else exits
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == peer_block ? 1 : -1);
    if (me < 0) return;
    if (threadIdx.x == 0) xcc[me] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;
    u64* mine = buf + me * 1024;          // my outgoing granules [NGRAN][64 lanes]
    const u64* theirs = buf + (1 - me) * 1024;
    const int lane = threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 1; i <= iters; ++i) {
        if (me == 0) {
            for (int g = 0; g < NGRAN; ++g) {
                if (STORE_SCOPE_AGENT) __hip_atomic_store(mine + g * 64 + lane, ((u64)i << 32) | lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(mine + g * 64 + lane, ((u64)i << 32) | lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        // wait for the peer's epoch i (block 1 waits for block 0's, then answers; block 0 waits for the answer)
        bool ok;
        int spins = 0;
        do {
            if (++spins > 2000000) break;
            ok = true;
            for (int g = 0; g < NGRAN; ++g) {
                u64 v = __hip_atomic_load(theirs + g * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (unsigned)(v >> 32) == (unsigned)i;
            }
        } while (!__all(ok));
        if (me == 1) {
            for (int g = 0; g < NGRAN; ++g) {
                if (STORE_SCOPE_AGENT) __hip_atomic_store(mine + g * 64 + lane, ((u64)i << 32) | lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(mine + g * 64 + lane, ((u64)i << 32) | lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[me] = t1 - t0;
}

int main() {
    u64* buf; long long* out; unsigned* xcc;
    CK(hipMalloc(&buf, 2048 * 8)); CK(hipMalloc(&out, 16)); CK(hipMalloc(&xcc, 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 2000;
    for (int peer : {8, 1}) {
        for (int variant = 0; variant < 4; ++variant) {
            if (peer == 1 && !(variant & 1)) continue;      // workgroup-scope stores never become visible on another XCD
            CK(hipMemset(buf, 0, 2048 * 8));
            CK(hipEventRecord(a));
            if (variant == 0) hipLaunchKernelGGL((pp<0, 1>), dim3(16), dim3(64), 0, 0, buf, iters, peer, out, xcc);
            if (variant == 1) hipLaunchKernelGGL((pp<1, 1>), dim3(16), dim3(64), 0, 0, buf, iters, peer, out, xcc);
            if (variant == 2) hipLaunchKernelGGL((pp<0, 16>), dim3(16), dim3(64), 0, 0, buf, iters, peer, out, xcc);
            if (variant == 3) hipLaunchKernelGGL((pp<1, 16>), dim3(16), dim3(64), 0, 0, buf, iters, peer, out, xcc);
            CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            long long h[2]; unsigned x[2];
            CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
            printf("peer block %d (xcc %u vs %u)  %s stores, %2d granule rows: round trip %.0f ns (one way %.0f ns), %lld memtime ticks/iter\n",
                   peer, x[0], x[1], (variant & 1) ? "agent-scope" : "workgroup-scope", variant >= 2 ? 16 : 1,
                   ms * 1e6 / iters, ms * 1e6 / iters / 2, h[0] / iters);
            fflush(stdout);
        }
    }
    return 0;
}
