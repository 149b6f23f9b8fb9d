// Error words of the persistent kernels, what a reported error invalidates, snapshot / restore of carried state around a call
// (the templates finish_or_recover / run_maybe_graph are in mp_host.h), and the test / probe hooks of
// include/mobileposer_hip_internal.h.
#include "mp_host.h"

namespace mph {

// A bounded wait inside a persistent kernel of an EARLIER call timed out (the grid was starved of CUs -- e.g. the GPU is
// shared with another process): the results of that call are invalid.  Reported once, by the next API entry.
// The side-by-side schedules address clusters by the XCD a workgroup really lands on, which rests on a probed but
// undocumented dispatcher order.  After any device error the handle stops relying on it: launches fall back to the
// blockIdx % 8 round robin (placement then only affects speed, never which (cluster, slice) a workgroup takes).
// ... and the exchange areas of every plan are zeroed before their next use: a launch that lost a workgroup leaves tagged
// words behind that no later launch's bookkeeping (ModuleWS::hx_flip) describes.
void disable_xcd_tables(mp_handle* h) {
    h->xcd_rr = false;
    for (Plan* q : h->plans)
        for (ModuleWS& w : q->ws) w.hx_epoch = 0;
}

// A failed call without recovery has poisoned what it carries forward: the velocity LSTM state it updated in place is NaN for
// the starved slab, and a streaming tick derived root height / root position / last foot positions from NaN outputs.  Once
// the error has been REPORTED the handle must not keep feeding that state into later calls (they would return NaN with
// MP_OK): the carried velocity state is dropped (as `model.velocity.rnn_state = None`) and every stream is put back to its
// state after construction + reset() (fresh window, root height / position 0, last foot positions = rest pose, net.py:59-64).
void invalidate_carried_state(mp_handle* h) {
    h->vstate.B = 0;
    StreamCtx& c = h->sc;
    if (!c.S) return;
    // (on s_main, which is a non-blocking stream: null-stream memsets are not ordered against its later work -- ADVICE r4 --
    //  and the host buffer must outlive the asynchronous copy: wait for it)
    (void)hipStreamSynchronize(h->s_main);
    std::vector<float> lf((size_t)c.S * 6);
    for (int s = 0; s < c.S; ++s) memcpy(&lf[(size_t)s * 6], h->feet_pos, sizeof(h->feet_pos));
    (void)hipMemcpyAsync(c.st.last_foot, lf.data(), lf.size() * sizeof(float), hipMemcpyHostToDevice, h->s_main);
    (void)hipMemsetAsync(c.fresh, 1, c.S, h->s_main);
    (void)hipMemsetAsync(c.st.root_y, 0, (size_t)c.S * sizeof(double), h->s_main);
    (void)hipMemsetAsync(c.st.root_pos, 0, (size_t)c.S * 3 * sizeof(float), h->s_main);
    (void)hipStreamSynchronize(h->s_main);
}

// The handle's error words (pinned host memory the kernels store to): [0] = a bounded wait gave up (1 + step, or 1000000 =
// the start-up handshake) -- STARVATION: a workgroup may be missing, the exchange areas are in an unknown state and the
// physical-XCD placement is no longer trusted; [1] = 2000000, an initial hidden state the tagged words cannot carry -- a
// property of the caller's STATE: every workgroup ran, nothing about placement or the exchange areas is wrong (round 5: the
// two used to share one word, and a state code paid the starvation remedy -- tables off for the handle's lifetime).
// Returns the code (starvation first) and clears both words; *starved = whether word [0] was set.
int take_device_error(mp_handle* h, bool* starved) {
    if (starved) *starved = false;
    if (!h->err_host) return 0;
    volatile int* e = (volatile int*)h->err_host;
    const int c0 = e[0], c1 = e[1];
    if (!c0 && !c1) return 0;
    e[0] = 0; e[1] = 0;
    if (starved) *starved = c0 != 0;
    return c0 ? c0 : c1;
}
bool device_error_pending(const mp_handle* h) {
    if (!h->err_host) return false;
    const volatile int* e = (const volatile int*)h->err_host;
    return e[0] != 0 || e[1] != 0;
}

int pending_device_error(mp_handle* h, const char* where) {
    bool starved = false;
    const int code = take_device_error(h, &starved);
    if (!code) return MP_OK;
    if (starved) disable_xcd_tables(h);
    invalidate_carried_state(h);
    return fail(h, MP_ERR_DEVICE, "%s: a previous call's persistent LSTM kernel gave up a wait for another workgroup's "
                "hidden state (code %d: 1+step, or 1000000 = start-up handshake; the GPU was shared?  2000000 = an initial hidden state "
                "outside (-2, 2) or NaN, which only the per-step kernels take: recovery on handles it); the affected "
                "outputs of that call are NaN and the state it carried forward is lost: the velocity LSTM state has been "
                "dropped and all streams reset.  Physical-XCD placement tables are now off for this handle", where, code);
}

int need_weights(mp_handle* h, const char* what) {
    if (h->has_weights) return MP_OK;
    return fail(h, MP_ERR_INVALID, "%s: this is a body-only handle (mp_create_body): it has no network weights", what);
}

int enter(mp_handle* h, void* stream) {
    if (int rc = pending_device_error(h, "mobileposer")) return rc;
    HIPCHK(h, hipEventRecord(h->ev_in, (hipStream_t)stream));
    HIPCHK(h, hipStreamWaitEvent(h->s_main, h->ev_in, 0));
    return MP_OK;
}
int leave(mp_handle* h, void* stream) {
    HIPCHK(h, hipEventRecord(h->ev_out, h->s_main));
    HIPCHK(h, hipStreamWaitEvent((hipStream_t)stream, h->ev_out, 0));
    return MP_OK;
}


// ---- recovery (mp_set_recovery) ------------------------------------------------------------------------------------
// snapshot / restore of the carried velocity state around a call (the fused kernels update it in place)
// (`more`: further copies of the same snapshot -- the solver state of a streaming tick -- that go out in the same launch)
int snapshot_vstate(mp_handle* h, int B, bool has_state, CopyJobs* more) {
    CopyJobs js;
    if (more) js = *more;
    if (h->recovery && has_state) {
        if (int rc = ensure_vstate(h, h->vsnap, B)) return rc;
        const size_t n = (size_t)2 * B * 256 * sizeof(float);
        js.add(h->vsnap.h, h->vstate.h, n);
        js.add(h->vsnap.c, h->vstate.c, n);
    }
    mp_launch_copy_words(js, h->s_main);
    HIPCHK(h, hipGetLastError());
    return MP_OK;
}
int restore_vstate(mp_handle* h, int B, bool has_state) {
    if (!has_state) return MP_OK;                      // the call started from zero state: nothing to restore
    const size_t n = (size_t)2 * B * 256 * sizeof(float);
    HIPCHK(h, hipMemcpyAsync(h->vstate.h, h->vsnap.h, n, hipMemcpyDeviceToDevice, h->s_main));
    HIPCHK(h, hipMemcpyAsync(h->vstate.c, h->vsnap.c, n, hipMemcpyDeviceToDevice, h->s_main));
    return MP_OK;
}


}  // namespace mph

// ================================================================================================ C ABI
extern "C" {

int mp_device_error(mp_handle* h, int* code) {
    if (!h || !code) return MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    bool starved = false;
    *code = take_device_error(h, &starved);
    if (starved) disable_xcd_tables(h);
    if (*code) invalidate_carried_state(h);
    return MP_OK;
}

int mp_finish(mp_handle* h) {
    if (!h) return MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    return pending_device_error(h, "mp_finish");
}

int mp_set_recovery(mp_handle* h, int on) {
    if (!h) return MP_ERR_INVALID;
    h->recovery = on != 0;
    return MP_OK;
}

int mp_recovery_count(const mp_handle* h) { return h ? h->recoveries : 0; }

namespace {
MP_KERNEL void mp_poke_error(int* err, int code) { mp_set_error(code == 2000000 ? err + 1 : err, code); }
// one wave, a dependent FMA chain between two looks at both clocks: ticks of the constant 100 MHz clock (s_memrealtime) and
// of the shader clock (s_memtime) -- their ratio is the frequency the CU ran at during the probe
MP_KERNEL void mp_clock_probe(unsigned long long* out, int spin) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    asm volatile("" :: "v"(x));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = r1 - r0; out[1] = c1 - c0; }
}
}

int mp_debug_clock_probe(mp_handle* h, double* shader_mhz, double* probe_us) {
    if (!h || !shader_mhz) return MP_ERR_INVALID;
    ON_DEVICE(h);
    unsigned long long* buf = nullptr;
    HIPCHK(h, hipHostMalloc((void**)&buf, 16, hipHostMallocDefault));
    buf[0] = buf[1] = 0;
    hipLaunchKernelGGL(mp_clock_probe, dim3(1), dim3(64), 0, h->s_main, buf, 2000);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->s_main);
    const double real = (double)buf[0], shader = (double)buf[1];
    (void)hipHostFree(buf);
    if (e != hipSuccess) return fail(h, MP_ERR_HIP, "mp_debug_clock_probe: %s", hipGetErrorString(e));
    *shader_mhz = real > 0 ? shader / real * 100.0 : 0.0;
    if (probe_us) *probe_us = real / 100.0;
    return MP_OK;
}

namespace {
// The same two clocks under LOAD: every wave of a grid that fills the chip (n_cu workgroups x 4 waves) issues a stream of
// independent fp32 MFMAs -- what the layer kernels do -- between its two looks at them.  The one-wave probe above runs on an
// otherwise idle chip and cannot see what power management does to a chip that has just been handed 1 024 busy matrix pipes.
MP_KERNEL __launch_bounds__(256) void mp_clock_probe_loaded(unsigned long long* out, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float a = 1e-3f * (float)(threadIdx.x & 63), b = 0.5f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[j]));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = r1 - r0; out[2 * w + 1] = c1 - c0;
    }
}
}

int mp_debug_clock_probe_loaded(mp_handle* h, int iters, double* mhz_mean, double* mhz_min, double* us_mean, double* us_max) {
    if (!h || iters < 1 || iters > (1 << 20) || !mhz_mean) return MP_ERR_INVALID;
    ON_DEVICE(h);
    const int nw = h->n_cu * 4;
    unsigned long long* buf = nullptr;
    HIPCHK(h, hipHostMalloc((void**)&buf, (size_t)nw * 16, hipHostMallocDefault));
    memset(buf, 0, (size_t)nw * 16);
    hipLaunchKernelGGL(mp_clock_probe_loaded, dim3(h->n_cu), dim3(256), 0, h->s_main, buf, iters);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->s_main);
    double sum = 0.0, mn = 1e30, us = 0.0, usmax = 0.0;
    for (int w = 0; w < nw; ++w) {
        const double real = (double)buf[2 * w], shader = (double)buf[2 * w + 1];
        const double mhz = real > 0 ? shader / real * 100.0 : 0.0;
        sum += mhz; mn = mhz < mn ? mhz : mn; us += real / 100.0; usmax = real / 100.0 > usmax ? real / 100.0 : usmax;
    }
    (void)hipHostFree(buf);
    if (e != hipSuccess) return fail(h, MP_ERR_HIP, "mp_debug_clock_probe_loaded: %s", hipGetErrorString(e));
    *mhz_mean = sum / nw;
    if (mhz_min) *mhz_min = mn;
    if (us_mean) *us_mean = us / nw;
    if (us_max) *us_max = usmax;
    return MP_OK;
}

int mp_debug_poke_error(mp_handle* h, int code) {
    if (!h) return MP_ERR_INVALID;
    ON_DEVICE(h);
    hipLaunchKernelGGL(mp_poke_error, dim3(1), dim3(1), 0, h->s_main, h->err_dev, code);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    return MP_OK;
}

int mp_debug_drop_workgroup(mp_handle* h, int block, int skip, int launches) {
    if (!h || block < 0 || skip < 0 || launches < 0) return MP_ERR_INVALID;
    h->dbg_drop_block = block;
    h->dbg_drop_skip = skip;
    h->dbg_drop_left = launches;
    return MP_OK;
}

int mp_debug_read_prof(mp_handle* h, long long* out, int n_words) {
    if (!h || !out || !h->prof_dev || n_words > (int)kProfWords) return MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out, h->prof_dev, (size_t)n_words * sizeof(long long), hipMemcpyDeviceToHost));
    return MP_OK;
}

int mp_set_transport(mp_handle* h, int force_remote) {
    if (!h) return MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    h->force_remote = force_remote != 0;
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second.exec);   // captured launches carry the old setting
    h->graphs.clear();
    return MP_OK;
}


}  // extern "C"
