// Device helpers shared by the persistent LSTM kernels (mp_lstm_persist.hip: exact-fp32 MFMA operands,
// mp_lstm_x3.hip: split-bf16 operands): activation functions, the {epoch, value} granule hand-off of
// cdna_hip_programming.md Guideline 16 (form R2), the XCC id, and the bf16 hi/lo pair encoding.
#pragma once
#include "mp_common.h"

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// v_exp_f32 / v_rcp_f32 are 1-ulp instructions: sigma and tanh come out within ~2e-7 absolute of libm
static __device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// tanh = 1 - 2 / (exp(2x) + 1) cancels for small |x|: its ABSOLUTE error stays ~1e-7 where libm's is relative (1e-10 at
// |x| = 1e-3).  MP_TANH_POLY (round-5 experiment, verdict r4 item 4; tools/accuracy.py): the odd Taylor polynomial below
// |x| = 0.125 (next term 62/2835 x^9: 1.3e-9 relative there), selected branch-free.
#ifndef MP_TANH_POLY
#define MP_TANH_POLY 0
#endif
static __device__ __forceinline__ float tanhf_(float x) {
    const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * x);
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
#if MP_TANH_POLY
    const float x2 = x * x;
    const float p = x + x * x2 * (-0.33333333333f + x2 * (0.13333333333f + x2 * -0.05396825397f));
    return __builtin_fabsf(x) < 0.125f ? p : big;
#else
    return big;
#endif
}

// A bounded wait gave up: leave a code in the handle's error word.  The word lives in pinned, coherent HOST memory (every
// API entry reads it without synchronising), so this is a system-scope store, not a device atomic.
static __device__ __forceinline__ void mp_set_error(int* err, int code) {
    __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Every wait of a persistent kernel is bounded in TIME: a poll loop gives up once `max_ticks` of the constant 100 MHz clock
// (s_memrealtime) have passed since its first look at the clock (one look per 32 polls: the clock is a scalar memory read).
// budget == 0: an earlier wait of this wave already gave up -- never wait again.  Only ever reached on a slow path.
static __device__ __forceinline__ bool wait_over(unsigned& spins, unsigned budget, u64& t0, u64 max_ticks) {
    if (budget == 0) return true;
    if ((++spins & 31u) != 1u) return false;
    const u64 now = __builtin_amdgcn_s_memrealtime();
    if (spins == 1u) { t0 = now; return false; }
    return now - t0 > max_ticks;
}
// A wave that gave up a wait makes its cell state NaN: every h it publishes from now on, its rows of the layer output, the
// final state and -- through the peers that consume its h -- the whole slab's output turn NaN within a step.  A starved
// grid therefore never leaves plausible numbers behind (the error word says why).  Slow path only.
template <int N>
static __device__ __forceinline__ void poison_cells(float (&c)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = __builtin_nanf("");
}
static __device__ __forceinline__ void poison_cells(float& c) { c = __builtin_nanf(""); }

static __device__ __forceinline__ u64 granule_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void granule_store_bits(u64* p, unsigned epoch, unsigned v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// same granule as an ordinary store: through the write-through L1 into THIS XCD's L2, where it stays
static __device__ __forceinline__ void granule_store_l2_bits(u64* p, unsigned epoch, unsigned v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void granule_store(u64* p, unsigned epoch, float v) {
    granule_store_bits(p, epoch, __float_as_uint(v));
}
static __device__ __forceinline__ void granule_store_l2(u64* p, unsigned epoch, float v) {
    granule_store_l2_bits(p, epoch, __float_as_uint(v));
}
static __device__ __forceinline__ unsigned xcc_id() {
    return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;   // s_getreg_b32 hwreg(HW_REG_XCC_ID)
}
constexpr unsigned XCC_TAG = 0x7fffffffu;

// ---- split pair: one fp32 value v as two 16-bit floats in one 32-bit word, hi = rne16(v) in the upper half and
// lo = rne16(v - hi) in the lower half; the products hi*hi + hi*lo + lo*hi of two such pairs, accumulated in fp32 by the MFMA,
// stand in for the fp32 product.
// Round 4: the 16-bit format is IEEE fp16 (11 significand bits), not bf16 (8).  Rounds 1-3 used bf16 pairs: hi + lo carried
// 17 bits (2^-18 relative), ~100 x fp32's rounding, and on weights in the trained regime (saturated gates, recurrent gain > 1)
// the recurrence amplified that to 2e-3 .. 2e-2 on the network outputs -- outside the 1e-4 parity bound
// (profiles/r04_accuracy.json).  An fp16 pair carries 11 + 1 + 11 = 23 bits plus the hidden one: |v - hi - lo| <= 2^-24 |v|,
// fp32's own half ulp, as long as lo is a NORMAL fp16 number; lo is about 2^-12 |v| and fp16's normal range ends at 2^-14, so
// for |v| < 0.25 lo is subnormal (v_mfma_f32_16x16x32_f16 honours subnormal inputs: tools/micro/mfma_f16_denorm.hip) and the
// representation error becomes ABSOLUTE, 2^-25 = 3e-8 -- that of an fp32 value near 0.5.  Hidden states and layer inputs
// live with that (it is what fp32 gives the large terms of a dot product); WEIGHTS (|w| ~ 0.05 .. 0.5) do not -- they are
// multiplied by kPairWScale = 16 before they are split (exact) and the accumulated sum is multiplied by 1/16 (exact) where the
// bias is added.  CPU emulation (round 4, tools/experiments/x3h_emulation.py in the history): at or below plain fp32's distance from float64 on both
// weight profiles; without the weight scale 4 x above it.  Range: |value| <= 65504 (weights: 4094); beyond that an operand
// becomes inf and the output NaN -- loud, and far outside anything an LSTM pose network produces.
constexpr float kPairWScale = 16.0f;
constexpr float kPairWInv = 0.0625f;
static __device__ __forceinline__ unsigned pair_of(float x) {
#pragma clang fp contract(off)
    const _Float16 h = (_Float16)x;                              // v_cvt_f16_f32, round to nearest even
    const unsigned hi = __builtin_bit_cast(unsigned short, h);
    // (the difference is exact in fp32.  `fp contract(off)` above: with x = a * b the compiler would otherwise fuse the
    //  subtraction into fma(a, b, -hi), i.e. take the residual of the UNROUNDED product -- a lo part that differs in its
    //  last bit from the pair of the fp32 value, and only in those instantiations where the scheduler happens to see it)
    const _Float16 l = (_Float16)(x - (float)h);
    return (hi << 16) | (unsigned)__builtin_bit_cast(unsigned short, l);
}
static __device__ __forceinline__ unsigned wpair_of(float w) { return pair_of(w * kPairWScale); }   // weights (see above)

// Workgroup barrier that only waits for this wave's LDS traffic (lgkmcnt), not for its outstanding global loads
// and stores: __syncthreads() also drains vmcnt, which would put the latency of a prefetch that is in flight on
// every wave of the workgroup.  Use it where the barrier orders LDS accesses only.
static __device__ __forceinline__ void barrier_lds_only() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Re-arm the split-bf16 exchange area of `ncl` clusters (mp_lstm_x3.hip layout, H = 256): everything that is polled
// must read as "nothing published yet" -- the tag bit of every data word, the XCC table.  The area is simply zeroed
// (131 KB per cluster); the caller is share `part` of `nparts` equal shares, all threads of a workgroup take part.
static __device__ __forceinline__ void rearm_exchange(unsigned long long* hx, int ncl, int part, int nparts, int tid, int nthreads) {
    constexpr size_t SLABQ = ((size_t)4 * 16 * 256 + 16) / 2;            // 16-byte words per cluster
    const size_t total = (size_t)ncl * SLABQ;
    const size_t per = (total + nparts - 1) / nparts;
    const size_t lo = (size_t)part * per, hi = lo + per < total ? lo + per : total;
    u32x4* w = reinterpret_cast<u32x4*>(hx);
    for (size_t i = lo + tid; i < hi; i += nthreads) w[i] = u32x4{0u, 0u, 0u, 0u};
}

// ---- shared by the split-bf16 LSTM kernels (mp_lstm_x3.hip, mp_lstm_x3w.hip)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// (the name is from rounds 1-3, when the halves were bf16; they are fp16 now -- pair_of above)
static __device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// 8 pair words (k = e) -> hi fragment (8 halves, element e in the low/high half of dword e/2) and lo fragment
static __device__ __forceinline__ void split_pairs(u32x4 w0, u32x4 w1, u32x4& hi, u32x4& lo) {
    hi[0] = __builtin_amdgcn_perm(w0[1], w0[0], 0x07060302u);
    hi[1] = __builtin_amdgcn_perm(w0[3], w0[2], 0x07060302u);
    hi[2] = __builtin_amdgcn_perm(w1[1], w1[0], 0x07060302u);
    hi[3] = __builtin_amdgcn_perm(w1[3], w1[2], 0x07060302u);
    lo[0] = __builtin_amdgcn_perm(w0[1], w0[0], 0x05040100u);
    lo[1] = __builtin_amdgcn_perm(w0[3], w0[2], 0x05040100u);
    lo[2] = __builtin_amdgcn_perm(w1[1], w1[0], 0x05040100u);
    lo[3] = __builtin_amdgcn_perm(w1[3], w1[2], 0x05040100u);
}
// ---- the exchanged hidden-state word: the pair of h with the epoch tag of the exchange in bit 30 -- the top exponent bit of
// the fp16 hi half, which is 0 for every |h| < 2 and h = o * tanh(c) never leaves [-1, 1]; tag of the h written at `step`
// = ((step / 2) + 1) & 1 (see the kernel header).  (Rounds 1-3 took the last mantissa bit of lo for the tag; with fp16 halves
// that would cost the pair its 23rd bit -- 4 x fp32's rounding on every recurrent operand, measured 3 x on the trained-regime
// outputs.)  A NaN h -- a poisoned slab, mp_lstm_dev.h poison_cells -- would carry a set bit 30 of its own and lose its NaN-ness
// when the consumer clears the tag: it travels as hi = 0, lo = NaN instead, and the consumer's MFMAs turn it back into NaN gates.
constexpr unsigned kHTagBit = 1u << 30;
static __device__ __forceinline__ unsigned hpair_of(float x) { return x != x ? 0x00007e00u : pair_of(x); }
static __device__ __forceinline__ unsigned tag_of_step(int step) { return ((((unsigned)step >> 1) + 1u) & 1u) << 30; }
// stores of the exchange as inline asm: the compiler's s_waitcnt bookkeeping must not see a store in the loop (with loads
// AND stores pending it waits with vmcnt(0) everywhere); nothing ever waits for these stores -- the data is its own flag
static __device__ __forceinline__ void store_word_xcd(unsigned* p, unsigned v) {          // stays in this XCD's L2
    asm volatile("global_store_dword %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
static __device__ __forceinline__ void store_word_dev(unsigned* p, unsigned v) {          // write-through: visible device-wide
    asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
static __device__ __forceinline__ void store_word_plain(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
// ... and a granule {epoch, value} the same way (mp_lstm_v1: with a store among the pending loads the wait for x_t at the top
// of a step becomes vmcnt(0) and sits out the acknowledgement of the stores the step before ended with; replay chain 162 ->
// 158 ms).  A store the compiler does not see can only make a wait for a LOAD stricter, never laxer: loads return in order,
// so a load that is outstanding keeps every younger load outstanding, and the counter the wait was computed for is reached
// no earlier.
static __device__ __forceinline__ void store_granule_xcd(u64* p, unsigned epoch, float v) {
    const u64 w = ((u64)epoch << 32) | (u64)__float_as_uint(v);
    asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(w) : "memory");
}
static __device__ __forceinline__ void store_granule_dev(u64* p, unsigned epoch, float v) {
    const u64 w = ((u64)epoch << 32) | (u64)__float_as_uint(v);
    asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
}

