// mp_host.h -- what the host-side translation units of libmobileposer_hip.so share: the handle, its plans / workspaces / stream
// context, the error plumbing (fail, HIPCHK, DeviceScope), the per-call scopes and the two templates that wrap a call (graph
// capture / replay, finish-or-recover).  Round 6 split the former 2 700-line mp_api.hip into
//   mp_handle.hip    weight manifest, packing at mp_create, handle lifetime                      (mp_create*, mp_destroy, ...)
//   mp_plans.hip     workspaces by capacity class, lengths upload, event-timed segments          (mp_timing_*, mp_debug_plan_stats)
//   mp_schedule.hip  one RNN block as launches, XCD placement, the schedules of MobilePoserNet.forward (forward_body)
//   mp_recovery.hip  error words, what a reported error invalidates, snapshot / restore, test and probe hooks
//   mp_api.hip       the forward / kinematics / evaluator / state entry points of the C ABI
//   mp_stream.hip    forward_online as a service: mp_stream_* (ticks, replay, state)
// Everything internal lives in namespace mph (the C ABI and struct mp_handle are global).
#pragma once
#include "../../include/mobileposer_hip_internal.h"
#include "mp_common.h"
#include "mp_lstm_dev.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace mph {

extern std::string g_create_error;       // the error text of a failed mp_create (no handle to carry it)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct Packed { float* W = nullptr; float* bias = nullptr; int N = 0, K = 0, Kpad = 0, Npad = 0, bn = 0;
                float* Wp = nullptr;      // Wp: the same padded matrix as split-bf16 pair words (linear layers only)
                float* Wf = nullptr; };   // Wf: the same padded matrix in MFMA B-fragment order (mp_gemm_f32_frag; linear layers only)
struct ModuleW {
    int n_in = 0, n_out = 0, H = 0, dirs = 0, nslice = 0, nsliceX = 0;   // slices per slab: fp32 kernels | split-bf16 kernels
    Packed lin1, ih[2], lin2;
    float* whh[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};    // per-step kernel layout
    float* whhP[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // persistent kernel layout for `nslice` slices per slab
    float* wihP[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // W_ih, the same
    float* whhX[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // split-bf16 kernel layout (H = 256 modules)
    float* wihX[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    float* whhP16[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // 16-slice packing of the bidirectional H = 256 blocks (small
    float* wihP16[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  //  batches; a unidirectional block's whhP / wihP already is it)
    float* whhP8[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // 8-slice packing of the unidirectional H = 256 block (the
    float* wihP8[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   //  two-layer wavefront launch; a bidirectional block's whhP / wihP already is it)
    float* whhU8[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // 32 slices of 8 units (mp_lstm_u8): small batches, H = 256 blocks
    float* wihU8[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    float* whhR[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};    // one sequence: W_hh / W_ih in mp_lstm_v1's per-lane order (H = 256)
    float* wihR[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};    //  or as torch has them (H = 64, mp_lstm_v1s)
    float* wVF[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};     // H = 64 block: rider fragments of mp_lstm_fused<..., FK> ("VF")
};
struct ModuleWS {
    float *xproj = nullptr, *out0 = nullptr, *out1 = nullptr;   // X1 (linear1 output) aliases out1 ...
    float* x1 = nullptr;                // ... except in the unidirectional H = 256 block: its two layers run as a wavefront (layer 1
                                        // writes out1 while layer 0 still reads X1), so X1 has a buffer of its own
    float* hbuf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    float* cbuf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    unsigned long long* hx = nullptr;   // hidden-state exchange buffer of the persistent kernels (split-bf16 mode: of layer 0)
    unsigned long long* hx2 = nullptr;  // split-bf16 mode: exchange buffer of layer 1 (re-armed by the layer-0 launch)
    size_t hx_bytes = 0;
    unsigned hx_epoch = 0;              // next epoch base of `hx` (mp_lstm_fused launches); 0 = must be zeroed first
    unsigned hx_flip = 3;               // tagged-word launches (LstmPersistArgs::tag_flip): first tags of the next launch
    unsigned hx_flipF = 3;              // ... of a rider's words in the same area (tag_flip_f): only launches that carry one write them
    bool hx_tagged = false;             // the area holds tagged words (else: granules / the 32-slice kernels' flagged words, the epoch family)
};
struct VelState { float* h = nullptr; float* c = nullptr; int B = 0; int cap = 0; };   // [2][B][256] each

struct GraphKey {
    int kind, B, T, flags;
    const void* p[8];
    bool operator<(const GraphKey& o) const { return memcmp(this, &o, sizeof(GraphKey)) < 0; }
};

struct Plan {
    int B = 0, T = 0;                  // the shape of the call that is using the plan (set by get_plan)
    int capB = 0;                      // capacity: batches of the class `capB` (plan_batch_class) ...
    size_t capRows = 0;                // ... with B * T <= capRows rows
    int lastB = 0;                     // B of the call before: another batch's words in the exchange areas
    bool streaming = false;            // the (S, 45) plan of mp_stream_create: never evicted (its graphs are keyed by its buffers)
    unsigned long long last_use = 0;   // LRU stamp (plans and their graphs are evicted when shapes keep changing)
    ModuleWS ws[4];
    float* r6d = nullptr;            // [B,T,96] when the caller does not ask for it
    int* lengths_dev = nullptr;
    int* lengths_pin = nullptr;      // pinned staging
    std::vector<int> lengths_cache;
    std::vector<void*> allocs;
};

constexpr size_t kProfWords = 512 * 8 + 2048 * 32 * 8;   // per-workgroup phase sums + (debug builds) a 32-step trace
struct Seg { int cls; hipEvent_t a, b; int launches; double flop; };

struct StreamCtx {
    int S = 0;
    float* window = nullptr;         // [S,45,60]
    uint8_t* fresh = nullptr;        // [S]
    uint8_t* mask_dev = nullptr;     // [S]
    OnlineState st;
    float *joints = nullptr, *vel = nullptr, *contact = nullptr;
    float* replay_ws = nullptr;      // mp_stream_replay: frame history | index-40 velocity rows | joints / contact of the batch
    size_t replay_bytes = 0;
};

}  // namespace mph
using namespace mph;

struct mp_handle {
    int device = 0;
    std::string err;
    bool has_weights = true;         // false: body-only handle (mp_create_body) -- kinematics entry points only
    ModuleW mod[4];
    int* parent_dev = nullptr;
    int* depth_dev = nullptr;
    float* bone_dev = nullptr;
    float* jrest_dev = nullptr;      // root-aligned rest joints [24,3]
    float* vrest_dev = nullptr;      // root-aligned template vertices [V,3] (mp_set_mesh)
    float* skinw_dev = nullptr;      // skinning weights [V,24]
    float* vtpl_dev = nullptr;       // raw template vertices [V,3] (shape blending starts from these, model.py:86)
    float* shapedirs_dev = nullptr;  // [V,3,10] (mp_set_shape_space)
    float* jreg_dev = nullptr;       // dense J_regressor [24,V]
    float* posedirsT_dev = nullptr;  // pose blend shapes, transposed [207][3V] (mp_set_pose_blendshape; nullptr = off)
    float* eval_ws = nullptr;        // mp_eval_metrics workspace: masked poses, FK outputs of prediction and truth, partials
    size_t eval_ws_bytes = 0;
    float* shape_ws = nullptr;       // mp_fk_shape workspace: vrest [ns][V][3] | jraw | jrest | bone [ns][72] each
    size_t shape_ws_floats = 0;
    int n_vertex = 0;
    float J0[3] = {0, 0, 0};
    float floor_y = 0.f;
    float feet_pos[6] = {0, 0, 0, 0, 0, 0};
    hipStream_t s_main = nullptr, s_vel = nullptr, s_foot = nullptr, s_gp = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_j = nullptr, ev_v = nullptr, ev_f = nullptr;
    hipEvent_t ev_x[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int* err_host = nullptr;         // error word of the persistent kernels: pinned, coherent host memory that the kernels
    int* err_dev = nullptr;          // store to directly (err_dev = its device address), so that every API entry can
                                     // look at it without synchronising anything
    long long* prof_dev = nullptr;   // debug: per-workgroup phase cycle sums of the last persistent launch
    bool force_remote = false;       // test hook (mp_set_transport)
    unsigned long long wait_ticks = 25000000ull;   // bound of every wait inside a persistent kernel: 0.25 s of the 100 MHz
                                     // constant clock (env MP_WAIT_MS: tests of the starvation path shorten it)
    int n_cu = 256;                  // compute units of this device: bounds the co-resident persistent grids
    bool persist = true;
    bool x3 = false;                 // false (default, mode 1): H = 256 layers on exact-fp32 MFMA operands -- the reference's
                                     // arithmetic; true (mode 3, mp_set_lstm_mode(h, 3) / MP_LSTM_MODE=x3): the opt-in fast
                                     // mode, split-bf16 MFMA operands (mp_lstm_x3.hip)
    unsigned epoch_start = 1;        // first epoch base after a zeroing (MP_VARIANT epoch_start: start close to the wrap guard)
    bool epoch_tags = true;          // MP_VARIANT epoch_tags=0: zero the exchange area before every fp32 layer launch (as round 1 did)
    bool slices16_ok = true;         // MP_VARIANT slices16=0: bidirectional fp32 layers always on 8 slices
    bool vec_ok = true;              // MP_VARIANT vec=0: B = 1 on the 32-slice MFMA kernel (mp_lstm_u8), not on the matrix-vector kernel (mp_lstm_v1)
    bool slices32_ok = true;         // MP_VARIANT slices32=0: no 32-slice kernels for batches of one or two slabs
    bool wide_ok = true;             // MP_VARIANT wide=0: never run pose / velocity / foot-contact side by side (small batches)
    bool exclusive_ok = true;        // MP_VARIANT exclusive=0: never pad the LDS request of concurrent persistent launches (below)
    int excl_lds = 0;                // forward_body -> rnn_rec: LstmPersistArgs::min_lds of the launches being issued
    bool pose_slices8 = false;       // forward_body -> fp32_slices: this call runs the pose layers on 8 slices per slab (below)
    bool xcd_rr = false;             // probed at create: workgroups are dealt round robin over 8 XCDs
    bool xcd_probe = false;          // ... what the probe said (xcd_rr is switched off after a starvation error; this is not)
    bool xcd_plan_on[4] = {false, false, false, false};   // forward_body -> rnn_rec: clusters per XCD of module id's layer launches
    unsigned char xcd_plan[4][8] = {};
    bool half_ok = true;             // MP_VARIANT half=0: no pose-on-half-the-chip schedule for 64 < B <= 128
    Packed lin1_pv;                  // pose.linear1 and velocity.linear1 stacked (split-bf16 mode: one GEMM over the shared rows)
    Packed lin1_pvf;                 // ... with foot_contact.linear1 on top (exact-fp32 mode, B > 128: one GEMM, three outputs)
    int x3w_mask = 2;                // split-bf16 layers run by the 4-wave kernel mp_lstm_x3w: bit 0 K_in = 256, bit 1 K_in = 512
                                     // (default: the K_in = 512 layers, measured 373 vs 391 us; K_in = 256: 300 vs 285 us; MP_VARIANT x3w)
    std::vector<Plan*> plans;        // workspaces by capacity class (get_plan)
    int plan_allocs = 0;             // plans allocated so far (mp_debug_plan_stats)
    struct GraphEntry { hipGraphExec_t exec; unsigned long long last_use; };
    std::map<GraphKey, GraphEntry> graphs;
    unsigned long long use_clock = 0;
    VelState vstate;
    VelState vsnap;                  // recovery: the carried velocity state a call started from
    float* rnn_snap = nullptr;       // recovery: mp_rnn_forward's state when the caller passes state_in == state_out
    size_t rnn_snap_bytes = 0;
    StreamCtx sc;
    OnlineState st_snap;             // recovery: per-stream solver state a streaming tick started from
    bool vf_ok = true;               // MP_VARIANT vf=0: the foot-contact layers always run as launches of their own
    bool one_stream_ok = true;       // MP_VARIANT one_stream=0: full batches on the round-3 three-stream schedule (forward_body's last branch)
    bool wf_ok = true;               // MP_VARIANT wf=0: velocity as two 16-slice layer launches (rounds 3-4), not as ONE two-layer wavefront launch
    bool late_pair_ok = true;        // MP_VARIANT late_pair=0: no schedule 4 (pose layer 0 alone, then pose layer 1 beside velocity + rider) for 64 < B <= 128
    const void* vf_foot = nullptr;   // forward_body -> rnn_rec: the foot-contact job that rides in this call's velocity launches
    int dbg_drop_block = 0, dbg_drop_left = 0, dbg_drop_skip = 0;   // mp_debug_drop_workgroup
    bool recovery = true;            // mp_set_recovery: calls wait for themselves and repair a starved run in LSTM mode 0
    int recoveries = 0;
    bool use_graph = false;          // opt-in (mp_set_graph_mode / MP_GRAPH=1): see the note at the top of this file
    bool graph_serial = false;       // graph mode 2: every launch captured on s_main -- a single-branch graph
    bool timing = false;
    std::vector<Seg> segs;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    bool capturing = false;
};


namespace mph {

inline int fail(mp_handle* h, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(h, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail(h, MP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline int dev_alloc(mp_handle* h, void** p, size_t bytes, std::vector<void*>* track = nullptr) {
    HIPCHK(h, hipMalloc(p, bytes ? bytes : 16));
    if (track) track->push_back(*p);
    return MP_OK;
}

// The calling thread's current device is put back when an entry point returns: torch (and any other HIP user in the process)
// reads hipGetDevice() as ITS current device, so a library call on a handle of another GPU must not move it.
struct DeviceScope {
    int prev = -1, dev;
    bool ok = true;
    explicit DeviceScope(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
#define ON_DEVICE(h) DeviceScope dev_scope_((h)->device); \
    if (!dev_scope_.ok) return fail((h), MP_ERR_HIP, "hipSetDevice(%d) failed", (h)->device)

hipEvent_t next_event(mp_handle* h);     // mp_plans.hip: the next event of the handle's pool (timing on)
struct SegScope {
    mp_handle* h; hipStream_t s; bool on; size_t idx;
    SegScope(mp_handle* h_, hipStream_t s_, int cls, int launches, double flop = 0.0) : h(h_), s(s_), on(false), idx(0) {
        if (!h->timing || h->capturing) return;
        hipEvent_t a = next_event(h), b = next_event(h);
        if (!a || !b) return;
        on = true;
        idx = h->segs.size();
        h->segs.push_back({cls, a, b, launches, flop});
        (void)hipEventRecord(a, s);
    }
    ~SegScope() { if (on) (void)hipEventRecord(h->segs[idx].b, s); }
};

constexpr int kSeqClusterMax = 4;     // batches of up to this many sequences run on the one-sequence kernels (mp_schedule.hip seq_clusters)
constexpr int kExclusiveLdsBytes = 84 * 1024;   // LstmPersistArgs::min_lds: more than half of a CU's 160 KB

// The per-call schedule fields forward_body hands to rnn_rec through the handle (excl_lds, pose_slices8, xcd_plan_on[],
// vf_foot).  A ScheduleScope sets them and its destructor puts ALL of them back to the defaults -- whichever way the
// enclosing block is left, early error returns included -- so a handle can never carry one call's schedule into the next
// (round 3 reset them by hand after collecting return codes).
struct ScheduleScope {
    mp_handle* h;
    explicit ScheduleScope(mp_handle* h_) : h(h_) {}
    ScheduleScope& exclusive_lds(int bytes) { h->excl_lds = bytes; return *this; }
    ScheduleScope& pose_on_8_slices(bool on) { h->pose_slices8 = on; return *this; }
    ScheduleScope& tables(int module, bool on) { h->xcd_plan_on[module] = on; return *this; }
    ScheduleScope& rider(const void* foot_job) { h->vf_foot = foot_job; return *this; }
    ~ScheduleScope() {
        h->excl_lds = 0;
        h->pose_slices8 = false;
        for (bool& b : h->xcd_plan_on) b = false;
        h->vf_foot = nullptr;
    }
    ScheduleScope(const ScheduleScope&) = delete;
    ScheduleScope& operator=(const ScheduleScope&) = delete;
};

enum StateMode { STATE_ZERO, STATE_FROM };

// One RNN block (models/rnn.py:20-33) as five phases so that the orchestrator can put the GEMM phases and the
// recurrences of different modules on different streams:
//   g0: linear1+ReLU, W_ih(l0) projection, initial (h,c) of both layers      rec(0): layer-0 recurrence
//   g1: W_ih(l1) projection                                                   rec(1): layer-1 recurrence
//   g2: final (h,c) copy-out, linear2 into the caller's layout
// x = [a0 | a1] rows (b,t); y rows (b,t) at y + b*yStrideB + t*yStrideT.
// in_h/in_c, out_h/out_c: [layers*dirs][B][H] carried state (read when mode == STATE_FROM, written when out != null)
struct RnnJob {
    mp_handle* h; Plan* p; int id;
    RowMap a0, a1;
    float* y; long yStrideB, yStrideT;
    StateMode mode;
    const float *in_h, *in_c;
    float *out_h, *out_c;
};

// ---- mp_handle.hip
bool multibranch_graphs_allowed();
// ---- mp_plans.hip
int get_plan(mp_handle* h, int B, int T, Plan** out, const Plan* keep = nullptr);
int ensure_step_ws(mp_handle* h, Plan* p);
int upload_lengths(mp_handle* h, Plan* p, const int32_t* lengths);
// ---- mp_schedule.hip
RowMap internal_map(const float* base, int B, int width);
RowMap user_map(const float* base, int T, int width);
int run_gemm(mp_handle* h, hipStream_t s, RowMap a0, RowMap a1, const Packed& w, float* C, long cStrideB,
             long cStrideT, int M, int B, int relu, bool pair_out = false, bool a_pairs = false, bool x3_gemm = false,
             unsigned long long* zero_hx = nullptr, int zero_ncl = 0);
bool use_x3(const mp_handle* h, const ModuleW& m);
float* x1_buffer(const mp_handle* h, const ModuleW& m, ModuleWS& w);
int rnn_rec(const RnnJob& j, int l, hipStream_t s);
int rnn_g1(const RnnJob& j, hipStream_t s);
int run_rnn(const RnnJob& j, hipStream_t s);
int ensure_vstate(mp_handle* h, VelState& v, int B);
int forward_body(mp_handle* h, Plan* p, const float* imu, float* pose, long poseRows, long poseRowStride,
                 long poseRowOffset, float* joints, float* vel, float* contact, float* r6d, VelState& vs,
                 bool has_state, float* fk_rglobal = nullptr, float* fk_joint = nullptr, bool* tail_pending = nullptr);
// ---- mp_recovery.hip
void disable_xcd_tables(mp_handle* h);
int take_device_error(mp_handle* h, bool* starved);
bool device_error_pending(const mp_handle* h);
int need_weights(mp_handle* h, const char* what);
int enter(mp_handle* h, void* stream);
int leave(mp_handle* h, void* stream);
int snapshot_vstate(mp_handle* h, int B, bool has_state, CopyJobs* more = nullptr);
int restore_vstate(mp_handle* h, int B, bool has_state);

constexpr size_t kMaxGraphs = 64;

template <class Body>
int run_maybe_graph(mp_handle* h, GraphKey key, Body body) {
    if (!h->use_graph || h->timing || h->dbg_drop_left > 0) return body();       // (the drop hook edits launch arguments: eager)
    key.flags |= h->graph_serial ? 16 : 0;
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        if (h->graphs.size() >= kMaxGraphs) {              // a caller that keeps changing buffers: drop the least recently used one
            auto victim = h->graphs.begin();
            for (auto jt = h->graphs.begin(); jt != h->graphs.end(); ++jt)
                if (jt->second.last_use < victim->second.last_use) victim = jt;
            HIPCHK(h, hipStreamSynchronize(h->s_main));    // (it may still be executing)
            (void)hipGraphExecDestroy(victim->second.exec);
            h->graphs.erase(victim);
        }
        hipGraph_t graph = nullptr;
        HIPCHK(h, hipStreamBeginCapture(h->s_main, hipStreamCaptureModeThreadLocal));
        h->capturing = true;
        int rc = body();
        h->capturing = false;
        hipError_t e = hipStreamEndCapture(h->s_main, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return fail(h, MP_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) return fail(h, MP_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
        it = h->graphs.emplace(key, mp_handle::GraphEntry{exec, 0}).first;
    }
    it->second.last_use = ++h->use_clock;
    HIPCHK(h, hipGraphLaunch(it->second.exec, h->s_main));
    return MP_OK;
}

// After `first` has been enqueued: wait for it; if a persistent kernel gave up a wait, `restore()` puts back the state
// the call started from and `again()` runs the call with per-step kernels (eager).  MP_OK + a warning when repaired.
template <class Restore, class Again>
int finish_or_recover(mp_handle* h, Plan* p, const char* what, Restore restore, Again again) {
    if (!h->recovery || h->capturing) return MP_OK;
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    bool starved = false;
    const int code = take_device_error(h, &starved);
    if (!code) return MP_OK;
    if (starved) disable_xcd_tables(h);
    const bool persist = h->persist, x3 = h->x3, graph = h->use_graph;
    h->persist = false; h->x3 = false; h->use_graph = false;
    int rc = p ? ensure_step_ws(h, p) : MP_OK;
    if (!rc) rc = restore();
    if (!rc) rc = again();
    h->persist = persist; h->x3 = x3; h->use_graph = graph;
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    ++h->recoveries;
    char buf[512];
    snprintf(buf, sizeof(buf), "warning: %s: a fused LSTM layer grid was starved of compute units (code %d; is the GPU shared?  2000000 = an initial hidden state the fused kernels do not take); "
             "the call was run again with per-step kernels and its results are valid (recovery #%d)", what, code, h->recoveries);
    h->err = buf;
    return MP_OK;
}


}  // namespace mph
