// K6 -- the translation solver that lives inline in MobilePoserNet.forward_offline (models/net.py:130-154)
// and forward_online (models/net.py:186-208): foot-contact velocity (+) network velocity lerp,
// floor-penetration clamp, integration to the root position.  (north_star calls this the "physics
// optimizer inner loop"; the reference's PhysicsOptimizer module itself is absent -- SURVEY F3-F5.)
//
// Offline: one wavefront per sequence.  The per-frame part (foot deltas, arg-max foot, sigmoid/clamp
// weight, lerp) is computed by all 64 lanes into LDS; the integration is a genuinely serial
// recurrence  y_t = max(y_{t-1} + v_t, floor - foot_t)  which three lanes (x, y, z) walk in double
// precision exactly as the reference's Python loop does (net.py:148-153 uses Python floats),
// so the "<=" decisions match; ~20 cycles per frame out of LDS, i.e. ~1 us for T = 125.  Results go back
// through LDS and leave as one coalesced [T,3] run.  HBM-bound: 72+... B in, 12 B out per frame.
#include "mp_common.h"

namespace {

constexpr int TCHUNK = 2048;
constexpr float GRAVITY_VELOCITY = -0.018f;   // config.py:131
constexpr float VEL_DIVISOR = 15.0f;          // datasets.fps / amass.vel_scale, net.py:141,196

__device__ __forceinline__ float prob_to_weight(float p) {   // net.py:90-91
    return (fminf(fmaxf(p, 0.5f), 0.9f) - 0.5f) / 0.4f;
}

MP_KERNEL __launch_bounds__(64) void mp_translate_offline(const float* __restrict__ joints,
                                                            const float* __restrict__ vel,
                                                            const float* __restrict__ contact,
                                                            const int* __restrict__ lengths, int B, int T,
                                                            float floor_y_f, float* __restrict__ tran) {
    __shared__ float sv[TCHUNK * 3];
    __shared__ float sfoot[TCHUNK];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int len = min(lengths[b], T);
    const float* J = joints + (size_t)b * T * 72;
    const float* V = vel + (size_t)b * T * 72;
    const float* C = contact + (size_t)b * T * 2;
    float* O = tran + (size_t)b * T * 3;
    const double floor_y = (double)floor_y_f;
    double acc = 0.0;                                  // lane c < 3: running root position component
    for (int t0 = 0; t0 < len; t0 += TCHUNK) {
        const int n = min(TCHUNK, len - t0);
        for (int k = lane; k < n; k += 64) {
            const int t = t0 + k;
            const float* jt = J + (size_t)t * 72;
            const float lx = jt[30], ly = jt[31], lz = jt[32], rx = jt[33], ry = jt[34], rz = jt[35];
            float dx = 0.f, dy = 0.f, dz = 0.f;
            const float c0 = C[t * 2], c1 = C[t * 2 + 1];
            if (t > 0) {                               // net.py:134-135: zero row prepended
                const float* jp = jt - 72;
                if (c1 > c0) { dx = jp[33] - rx; dy = jp[34] - ry; dz = jp[35] - rz; }   // arg-max foot, ties -> left
                else         { dx = jp[30] - lx; dy = jp[31] - ly; dz = jp[32] - lz; }
            }
            const float cvx = 0.f + dx, cvy = GRAVITY_VELOCITY + dy, cvz = 0.f + dz;     // net.py:133
            const float pvx = V[(size_t)t * 72] / VEL_DIVISOR, pvy = V[(size_t)t * 72 + 1] / VEL_DIVISOR,
                        pvz = V[(size_t)t * 72 + 2] / VEL_DIVISOR;                         // net.py:140-141
            const float m = fmaxf(c0, c1);
            const float w = prob_to_weight(1.0f / (1.0f + expf(-m)));                      // net.py:144
            const float omw = 1.0f - w;
            sv[k * 3 + 0] = pvx * omw + cvx * w;                                           // net.py:145
            sv[k * 3 + 1] = pvy * omw + cvy * w;
            sv[k * 3 + 2] = pvz * omw + cvz * w;
            sfoot[k] = fminf(ly, ry);
        }
        __syncthreads();
        if (lane < 3) {
            if (lane == 1) {
                for (int k = 0; k < n; ++k) {          // net.py:149-153
                    float vy = sv[k * 3 + 1];
                    const double cur_foot = acc + (double)sfoot[k];
                    if (cur_foot + (double)vy <= floor_y) vy = (float)(floor_y - cur_foot);
                    acc += (double)vy;
                    sv[k * 3 + 1] = (float)acc;
                }
            } else {
                for (int k = 0; k < n; ++k) {          // net.py:154 (prefix sum of the velocities)
                    acc += (double)sv[k * 3 + lane];
                    sv[k * 3 + lane] = (float)acc;
                }
            }
        }
        __syncthreads();
        for (int k = lane; k < n * 3; k += 64) O[(size_t)t0 * 3 + k] = sv[k];
        __syncthreads();
    }
    // frames past the sequence end hold the last translation
    const float facc = (float)acc;
    const float l0 = __shfl(facc, 0, 64), l1 = __shfl(facc, 1, 64), l2 = __shfl(facc, 2, 64);
    for (int k = len * 3 + lane; k < T * 3; k += 64) O[k] = (k % 3 == 0) ? l0 : ((k % 3 == 1) ? l1 : l2);
}

// one thread per stream: forward_online's solver on window index `idx` (= num_past_frames = 40)
MP_KERNEL void mp_translate_online(const float* __restrict__ joints, const float* __restrict__ vel,
                                    const float* __restrict__ contact, int S, int T, int idx, float floor_y_f,
                                    OnlineState st, float* __restrict__ root_pos_out,
                                    float* __restrict__ contact_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const float* jt = joints + ((size_t)s * T + idx) * 72;
    const float* vt = vel + ((size_t)s * T + idx) * 72;
    const float c0 = contact[((size_t)s * T + idx) * 2], c1 = contact[((size_t)s * T + idx) * 2 + 1];
    const float lx = jt[30], ly = jt[31], lz = jt[32], rx = jt[33], ry = jt[34], rz = jt[35];
    float* lf = st.last_foot + (size_t)s * 6;
    float cvx, cvy, cvz;
    if (c0 > c1) { cvx = lf[0] - lx + 0.f; cvy = lf[1] - ly + GRAVITY_VELOCITY; cvz = lf[2] - lz + 0.f; }   // net.py:189-192
    else         { cvx = lf[3] - rx + 0.f; cvy = lf[4] - ry + GRAVITY_VELOCITY; cvz = lf[5] - rz + 0.f; }
    const float pvx = vt[0] / VEL_DIVISOR, pvy = vt[1] / VEL_DIVISOR, pvz = vt[2] / VEL_DIVISOR;             // net.py:196
    const float w = prob_to_weight(fmaxf(c0, c1));      // raw logit, no sigmoid (net.py:197)
    const float omw = 1.0f - w;
    const float vx = pvx * omw + cvx * w;
    float vy = pvy * omw + cvy * w;
    const float vz = pvz * omw + cvz * w;
    const double floor_y = (double)floor_y_f;
    double root_y = st.root_y[s];
    const double cur_foot = root_y + (double)fminf(ly, ry);                 // net.py:201
    if (cur_foot + (double)vy <= floor_y) vy = (float)(floor_y - cur_foot); // net.py:202-203
    root_y += (double)vy;                                                    // net.py:205
    st.root_y[s] = root_y;
    lf[0] = lx; lf[1] = ly; lf[2] = lz; lf[3] = rx; lf[4] = ry; lf[5] = rz;  // net.py:206
    float* rp = st.root_pos + (size_t)s * 3;
    rp[0] += vx; rp[1] += vy; rp[2] += vz;                                   // net.py:208
    root_pos_out[s * 3 + 0] = rp[0]; root_pos_out[s * 3 + 1] = rp[1]; root_pos_out[s * 3 + 2] = rp[2];
    contact_out[s * 2 + 0] = c0; contact_out[s * 2 + 1] = c1;
}

}  // namespace

void mp_launch_translate_offline(const float* joints, const float* vel, const float* contact, const int* lengths,
                                 int B, int T, float floor_y, float* tran, hipStream_t s) {
    hipLaunchKernelGGL(mp_translate_offline, dim3(B), dim3(64), 0, s, joints, vel, contact, lengths, B, T, floor_y,
                       tran);
}

void mp_launch_translate_online(const float* joints, const float* vel, const float* contact, int S, int T, int idx,
                                float floor_y, OnlineState st, float* root_pos_out, float* contact_out,
                                hipStream_t s) {
    hipLaunchKernelGGL(mp_translate_online, dim3((S + 63) / 64), dim3(64), 0, s, joints, vel, contact, S, T, idx,
                       floor_y, st, root_pos_out, contact_out);
}

namespace {

// one block per stream; the 45 x 60 window is shifted in place through registers
MP_KERNEL __launch_bounds__(256) void mp_window_push(float* __restrict__ window, const float* __restrict__ frames,
                                                       uint8_t* __restrict__ fresh, int S, int W) {
    const int s = blockIdx.x;
    float* win = window + (size_t)s * W * 60;
    const float* f = frames + (size_t)s * 60;
    const bool isFresh = fresh[s] != 0;
    const int n = W * 60;
    constexpr int PER = 12;                       // 256 * 12 = 3072 >= 45 * 60
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = threadIdx.x + i * 256;
        v[i] = 0.f;
        if (k < n) {
            const int t = k / 60, c = k - t * 60;
            if (isFresh || t == W - 1) v[i] = f[c];
            else v[i] = win[k + 60];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = threadIdx.x + i * 256;
        if (k < n) win[k] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) fresh[s] = 0;
}

MP_KERNEL void mp_stream_reset(const uint8_t* __restrict__ mask, uint8_t* __restrict__ fresh,
                                double* __restrict__ root_y, float* __restrict__ root_pos, float* __restrict__ velH,
                                float* __restrict__ velC, int S) {
    const int s = blockIdx.x;
    if (mask && !mask[s]) return;
    if (threadIdx.x == 0) {
        fresh[s] = 1;
        root_y[s] = 0.0;
        root_pos[s * 3 + 0] = 0.f; root_pos[s * 3 + 1] = 0.f; root_pos[s * 3 + 2] = 0.f;
    }
    if (velH) {
        for (int layer = 0; layer < 2; ++layer) {
            velH[((size_t)layer * S + s) * 256 + threadIdx.x] = 0.f;
            velC[((size_t)layer * S + s) * 256 + threadIdx.x] = 0.f;
        }
    }
}

}  // namespace

namespace {

// ---- replay of N consecutive forward_online calls of ONE stream (mp_stream_replay; evaluate.py:62-64) ----
// history = the 45 frames the stream's window held before the first call (a fresh stream: 45 copies of the first frame,
// net.py:175) followed by the N new frames: window k of the replay is history[k + 1 .. k + 45]
MP_KERNEL void mp_replay_history(const float* __restrict__ window, const uint8_t* __restrict__ fresh,
                                  const float* __restrict__ frames, int N, int W, float* __restrict__ hist) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)(W + N) * 60;
    if (i >= total) return;
    const long row = i / 60;
    const int c = (int)(i - row * 60);
    if (row < W) hist[i] = fresh[0] ? frames[c] : window[i];
    else hist[i] = frames[(row - W) * 60 + c];
}
// ... and the window the stream holds afterwards: the last 45 frames of the history
MP_KERNEL void mp_replay_window(const float* __restrict__ hist, int N, int W, float* __restrict__ window, uint8_t* __restrict__ fresh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W * 60) window[i] = hist[(long)N * 60 + i];
    if (i == 0) fresh[0] = 0;
}

// the solver of forward_online (net.py:186-208) for frames k = 0 .. N-1 in order, on window index `idx` of every window:
// joints [N][T][72], contact [N][T][2] (batch layout), vel [N][72] (row idx of every window only).  One thread: the chain
// through last foot positions / root height / root position is serial (same arithmetic as mp_translate_online)
MP_KERNEL void mp_translate_replay(const float* __restrict__ joints, const float* __restrict__ vel, const float* __restrict__ contact,
                                    int N, int T, int idx, float floor_y_f, OnlineState st, float* __restrict__ root_pos_out,
                                    float* __restrict__ contact_out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float lf[6];
    for (int i = 0; i < 6; ++i) lf[i] = st.last_foot[i];
    double root_y = st.root_y[0];
    float rp0 = st.root_pos[0], rp1 = st.root_pos[1], rp2 = st.root_pos[2];
    const double floor_y = (double)floor_y_f;
    for (int k = 0; k < N; ++k) {
        const float* jt = joints + ((size_t)k * T + idx) * 72;
        const float* vt = vel + (size_t)k * 72;
        const float c0 = contact[((size_t)k * T + idx) * 2], c1 = contact[((size_t)k * T + idx) * 2 + 1];
        const float lx = jt[30], ly = jt[31], lz = jt[32], rx = jt[33], ry = jt[34], rz = jt[35];
        float cvx, cvy, cvz;
        if (c0 > c1) { cvx = lf[0] - lx + 0.f; cvy = lf[1] - ly + GRAVITY_VELOCITY; cvz = lf[2] - lz + 0.f; }   // net.py:189-192
        else         { cvx = lf[3] - rx + 0.f; cvy = lf[4] - ry + GRAVITY_VELOCITY; cvz = lf[5] - rz + 0.f; }
        const float pvx = vt[0] / VEL_DIVISOR, pvy = vt[1] / VEL_DIVISOR, pvz = vt[2] / VEL_DIVISOR;             // net.py:196
        const float w = prob_to_weight(fmaxf(c0, c1));      // raw logit, no sigmoid (net.py:197)
        const float omw = 1.0f - w;
        const float vx = pvx * omw + cvx * w;
        float vy = pvy * omw + cvy * w;
        const float vz = pvz * omw + cvz * w;
        const double cur_foot = root_y + (double)fminf(ly, ry);                 // net.py:201
        if (cur_foot + (double)vy <= floor_y) vy = (float)(floor_y - cur_foot); // net.py:202-203
        root_y += (double)vy;                                                    // net.py:205
        lf[0] = lx; lf[1] = ly; lf[2] = lz; lf[3] = rx; lf[4] = ry; lf[5] = rz;  // net.py:206
        rp0 += vx; rp1 += vy; rp2 += vz;                                         // net.py:208
        root_pos_out[k * 3 + 0] = rp0; root_pos_out[k * 3 + 1] = rp1; root_pos_out[k * 3 + 2] = rp2;
        contact_out[k * 2 + 0] = c0; contact_out[k * 2 + 1] = c1;
    }
    for (int i = 0; i < 6; ++i) st.last_foot[i] = lf[i];
    st.root_y[0] = root_y;
    st.root_pos[0] = rp0; st.root_pos[1] = rp1; st.root_pos[2] = rp2;
}

// up to six small device-to-device copies as ONE launch (the state a recoverable call starts from: five hipMemcpyAsync calls
// were 60 us of a 400 us one-stream tick -- every copy is a kernel of its own with ~10 us between two of them)
MP_KERNEL void mp_copy_words(CopyJobs js) {
    const unsigned stride = gridDim.x * blockDim.x;
#pragma unroll
    for (int k = 0; k < CopyJobs::kMax; ++k) {
        const unsigned* __restrict__ src = static_cast<const unsigned*>(js.j[k].src);
        unsigned* __restrict__ dst = static_cast<unsigned*>(js.j[k].dst);
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < js.j[k].words; i += stride) dst[i] = src[i];
    }
}

}  // namespace

void mp_launch_copy_words(const CopyJobs& js, hipStream_t s) {
    unsigned most = 0;
    for (int k = 0; k < CopyJobs::kMax; ++k) most = js.j[k].words > most ? js.j[k].words : most;
    if (!most) return;
    const unsigned blocks = (most + 255) / 256 < 512 ? (most + 255) / 256 : 512;
    hipLaunchKernelGGL(mp_copy_words, dim3(blocks), dim3(256), 0, s, js);
}

void mp_launch_replay_history(const float* window, const uint8_t* fresh, const float* frames, int N, int W, float* hist, hipStream_t s) {
    const long total = (long)(W + N) * 60;
    hipLaunchKernelGGL(mp_replay_history, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, window, fresh, frames, N, W, hist);
}
void mp_launch_replay_window(const float* hist, int N, int W, float* window, uint8_t* fresh, hipStream_t s) {
    hipLaunchKernelGGL(mp_replay_window, dim3((W * 60 + 255) / 256), dim3(256), 0, s, hist, N, W, window, fresh);
}
void mp_launch_translate_replay(const float* joints, const float* vel, const float* contact, int N, int T, int idx, float floor_y,
                                OnlineState st, float* root_pos_out, float* contact_out, hipStream_t s) {
    hipLaunchKernelGGL(mp_translate_replay, dim3(1), dim3(64), 0, s, joints, vel, contact, N, T, idx, floor_y, st, root_pos_out, contact_out);
}

void mp_launch_window_push(float* window, const float* frames, uint8_t* fresh, int S, int W, hipStream_t s) {
    hipLaunchKernelGGL(mp_window_push, dim3(S), dim3(256), 0, s, window, frames, fresh, S, W);
}

void mp_launch_stream_reset(const uint8_t* mask, uint8_t* fresh, double* root_y, float* root_pos, float* velH,
                            float* velC, int S, hipStream_t s) {
    hipLaunchKernelGGL(mp_stream_reset, dim3(S), dim3(256), 0, s, mask, fresh, root_y, root_pos, velH, velC, S);
}
