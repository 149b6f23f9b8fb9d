// Host runtime + C ABI of libmobileposer_hip.so (see include/mobileposer_hip.h).
//
// Orchestrates MobilePoserNet.forward (models/net.py:101-119) on one MI355X:
//   joints RNN -> [pose RNN + r6d/IK] || velocity RNN || foot-contact RNN
// on three library-owned HIP streams (fork/join by events), each RNN being
//   GEMM(linear1+ReLU) -> GEMM(W_ih l0) -> T x lstm_step -> GEMM(W_ih l1) -> T x lstm_step -> GEMM(linear2),
// launched eagerly (default) or, opt-in, captured once per (shape, buffer set) into a hipGraph and replayed.
// Weights are re-laid-out once at load time into MFMA fragment order; workspaces are sized per (B, T) plan and
// kept (288 GB of HBM: no reuse games).
//
// Why eager is the default: the multi-branch graph executor of the HIP runtime this image ships (libamdhip64 of
// ROCm 7.0, hip::Graph::UpdateStreams) picks max_streams-1 of the max_streams internal streams a hipGraphExec_t owns,
// skipping those that share a hardware queue with the launch stream, WITHOUT a bounds check: when two of them map to
// the launch stream's queue (GPU_MAX_HW_QUEUES = 4 by default, so this depends on every stream the process has ever
// created) it reads past the end of the vector and the process dies with SIGSEGV inside hipGraphLaunch
// (profiles/r02_hipgraph_segv.md: backtrace, disassembly, and the GPU_MAX_HW_QUEUES experiment).  Eager launches on
// the library's streams measure 1.755 vs 1.716 ms (split-bf16 mode) and 4.46 vs 4.57 ms (fp32 mode) per
// 256 x 125 batch, i.e. nothing is lost.
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

namespace {

std::string g_create_error;

struct ModSpec { const char* prefix; int n_in, n_out, H, bi, id; };
// registration order of the reference's state_dict (models/net.py:40-43)
const ModSpec kSpecs[4] = {
    {"pose.pose.", 132, 96, 256, 1, MP_MOD_POSE},
    {"joints.joints.", 60, 72, 256, 1, MP_MOD_JOINTS},
    {"foot_contact.footcontact.", 132, 2, 64, 1, MP_MOD_FOOT_CONTACT},
    {"velocity.vel.", 132, 72, 256, 0, MP_MOD_VELOCITY},
};

enum Kind { K_WIH, K_WHH, K_BIH, K_BHH, K_L1W, K_L1B, K_L2W, K_L2B };
struct Entry { std::string name; int ndim; int64_t shape[2]; size_t offset; int mod, kind, layer, dir; };

std::vector<Entry> build_manifest() {
    std::vector<Entry> v;
    size_t off = 0;
    auto add = [&](const std::string& name, int ndim, int64_t s0, int64_t s1, int mod, int kind, int layer, int dir) {
        Entry e{name, ndim, {s0, s1}, off, mod, kind, layer, dir};
        v.push_back(e);
        off += (size_t)s0 * (ndim == 2 ? (size_t)s1 : 1);
    };
    for (const ModSpec& m : kSpecs) {
        const int dirs = m.bi ? 2 : 1;
        for (int l = 0; l < 2; ++l) {
            const int in_l = l == 0 ? m.H : m.H * dirs;
            for (int d = 0; d < dirs; ++d) {
                const std::string sfx = "_l" + std::to_string(l) + (d ? "_reverse" : "");
                add(std::string(m.prefix) + "rnn.weight_ih" + sfx, 2, 4 * m.H, in_l, m.id, K_WIH, l, d);
                add(std::string(m.prefix) + "rnn.weight_hh" + sfx, 2, 4 * m.H, m.H, m.id, K_WHH, l, d);
                add(std::string(m.prefix) + "rnn.bias_ih" + sfx, 1, 4 * m.H, 1, m.id, K_BIH, l, d);
                add(std::string(m.prefix) + "rnn.bias_hh" + sfx, 1, 4 * m.H, 1, m.id, K_BHH, l, d);
            }
        }
        add(std::string(m.prefix) + "linear1.weight", 2, m.H, m.n_in, m.id, K_L1W, 0, 0);
        add(std::string(m.prefix) + "linear1.bias", 1, m.H, 1, m.id, K_L1B, 0, 0);
        add(std::string(m.prefix) + "linear2.weight", 2, m.n_out, m.H * dirs, m.id, K_L2W, 0, 0);
        add(std::string(m.prefix) + "linear2.bias", 1, m.n_out, 1, m.id, K_L2B, 0, 0);
    }
    return v;
}
const std::vector<Entry>& manifest() {
    static const std::vector<Entry> m = build_manifest();
    return m;
}
size_t manifest_floats() {
    const Entry& e = manifest().back();
    return e.offset + (size_t)e.shape[0] * (e.ndim == 2 ? (size_t)e.shape[1] : 1);
}

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

}  // namespace

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

namespace {

int fail(mp_handle* h, int code, const char* fmt, ...) {
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

int dev_alloc(mp_handle* h, void** p, size_t bytes, std::vector<void*>* track = nullptr) {
    HIPCHK(h, hipMalloc(p, bytes ? bytes : 16));
    if (track) track->push_back(*p);
    return MP_OK;
}

// ------------------------------------------------------------------------------------------ weights
int alloc_packed(mp_handle* h, Packed& p, int N, int K) {
    p.N = N; p.K = K; p.Kpad = round_up(K, 32); p.bn = mp_gemm_pick_bn(N); p.Npad = round_up(N, p.bn);
    if (int rc = dev_alloc(h, (void**)&p.W, (size_t)p.Npad * p.Kpad * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&p.bias, (size_t)p.Npad * sizeof(float))) return rc;
    HIPCHK(h, hipMemsetAsync(p.W, 0, (size_t)p.Npad * p.Kpad * sizeof(float), h->s_main));
    HIPCHK(h, hipMemsetAsync(p.bias, 0, (size_t)p.Npad * sizeof(float), h->s_main));
    return MP_OK;
}

int pack_weights(mp_handle* h, const float* blob) {
    for (const ModSpec& s : kSpecs) {
        ModuleW& m = h->mod[s.id];
        m.n_in = s.n_in; m.n_out = s.n_out; m.H = s.H; m.dirs = s.bi ? 2 : 1;
        // B = 256 bidirectional = 2 x 16 slabs x 8 slices = 256 workgroups (one per CU); a unidirectional layer
        // reaches the same 256 with 16 slices.  (Two 4-wave workgroups per CU were measured slower: the
        // lock-step of a cluster turns any contention between co-resident workgroups into waiting for everyone.)
        m.nslice = m.H != 256 ? 4 : (m.dirs == 2 ? 8 : 16);       // 8 slices / four 512-register waves for bidirectional layers that fill the chip, 16 for unidirectional ones
        // split-bf16 kernels: 8 slices (8-wave workgroups) for every H = 256 layer -- the unidirectional velocity
        // layers then occupy 128 CUs and leave the other half of the chip to the foot-contact block (measured:
        // 312 vs 326 us per velocity layer, foot-contact layers 200 vs 265 us)
        m.nsliceX = 8;
        if (int rc = alloc_packed(h, m.lin1, m.H, m.n_in)) return rc;
        if (int rc = alloc_packed(h, m.ih[0], m.dirs * 4 * m.H, m.H)) return rc;
        if (int rc = alloc_packed(h, m.ih[1], m.dirs * 4 * m.H, m.dirs * m.H)) return rc;
        if (int rc = alloc_packed(h, m.lin2, m.n_out, m.dirs * m.H)) return rc;
        for (int l = 0; l < 2; ++l)
            for (int d = 0; d < m.dirs; ++d)
            {
                if (int rc = dev_alloc(h, (void**)&m.whh[l][d], mp_whh_pack_floats(m.H) * sizeof(float))) return rc;
                if (int rc = dev_alloc(h, (void**)&m.whhP[l][d], mp_whh_pack_floats(m.H) * sizeof(float))) return rc;
                const int kin = l == 0 ? m.H : m.dirs * m.H;
                if (int rc = dev_alloc(h, (void**)&m.wihP[l][d], (size_t)4 * m.H * kin * sizeof(float))) return rc;
                if (m.H == 256) {
                    if (int rc = dev_alloc(h, (void**)&m.whhU8[l][d], (size_t)4 * m.H * m.H * sizeof(float))) return rc;
                    if (int rc = dev_alloc(h, (void**)&m.wihU8[l][d], (size_t)4 * m.H * kin * sizeof(float))) return rc;
                }
                if (int rc = dev_alloc(h, (void**)&m.whhR[l][d], (size_t)4 * m.H * m.H * sizeof(float))) return rc;
                if (int rc = dev_alloc(h, (void**)&m.wihR[l][d], (size_t)4 * m.H * kin * sizeof(float))) return rc;
                if (m.H == 256 && m.nslice != 16) {
                    if (int rc = dev_alloc(h, (void**)&m.whhP16[l][d], mp_whh_pack_floats(m.H) * sizeof(float))) return rc;
                    if (int rc = dev_alloc(h, (void**)&m.wihP16[l][d], (size_t)4 * m.H * kin * sizeof(float))) return rc;
                }
                if (m.H == 256 && m.nslice != 8 && m.dirs == 1) {
                    if (int rc = dev_alloc(h, (void**)&m.whhP8[l][d], mp_whh_pack_floats(m.H) * sizeof(float))) return rc;
                    if (int rc = dev_alloc(h, (void**)&m.wihP8[l][d], (size_t)4 * m.H * kin * sizeof(float))) return rc;
                }
                if (m.H == 64)
                    if (int rc = dev_alloc(h, (void**)&m.wVF[l][d], mp_foot_vf_floats(kin) * sizeof(float))) return rc;
                if (m.H == 256) {
                    if (int rc = dev_alloc(h, (void**)&m.whhX[l][d], (size_t)4 * m.H * m.H * sizeof(float))) return rc;
                    if (int rc = dev_alloc(h, (void**)&m.wihX[l][d], (size_t)4 * m.H * kin * sizeof(float))) return rc;
                }
            }
    }
    const std::vector<Entry>& man = manifest();
    auto find = [&](int mod, int kind, int layer, int dir) -> const float* {
        for (const Entry& e : man)
            if (e.mod == mod && e.kind == kind && e.layer == layer && e.dir == dir) return blob + e.offset;
        return nullptr;
    };
    for (const ModSpec& s : kSpecs) {
        ModuleW& m = h->mod[s.id];
        mp_launch_pack_linear(find(s.id, K_L1W, 0, 0), find(s.id, K_L1B, 0, 0), m.lin1.W, m.lin1.bias, m.lin1.N,
                              m.lin1.K, m.lin1.Kpad, h->s_main);
        mp_launch_pack_linear(find(s.id, K_L2W, 0, 0), find(s.id, K_L2B, 0, 0), m.lin2.W, m.lin2.bias, m.lin2.N,
                              m.lin2.K, m.lin2.Kpad, h->s_main);
        for (Packed* pk : {&m.lin1, &m.lin2}) {
            const size_t n = (size_t)pk->Npad * pk->Kpad;
            if (int rc = dev_alloc(h, (void**)&pk->Wp, n * sizeof(float))) return rc;
            mp_launch_pairs(pk->W, pk->Wp, n, h->s_main);
            if (int rc = dev_alloc(h, (void**)&pk->Wf, n * sizeof(float))) return rc;
            mp_launch_pack_wfrag(pk->W, pk->Wf, pk->Npad, pk->Kpad, h->s_main);
        }
        for (int l = 0; l < 2; ++l)
            for (int d = 0; d < m.dirs; ++d) {
                mp_launch_pack_wih(find(s.id, K_WIH, l, d), find(s.id, K_BIH, l, d), find(s.id, K_BHH, l, d),
                                   m.ih[l].W, m.ih[l].bias, m.H, m.ih[l].K, m.ih[l].Kpad, d * 4 * m.H, h->s_main);
                mp_launch_pack_whh(find(s.id, K_WHH, l, d), m.whh[l][d], m.H, h->s_main);
                mp_launch_pack_whh_persist(find(s.id, K_WHH, l, d), m.whhP[l][d], m.H, m.nslice, h->s_main);
                mp_launch_pack_wih_persist(find(s.id, K_WIH, l, d), m.wihP[l][d], m.H, m.ih[l].K, m.nslice, h->s_main);
                if (m.wVF[l][d]) mp_launch_pack_foot_vf(find(s.id, K_WIH, l, d), find(s.id, K_WHH, l, d), m.wVF[l][d], m.ih[l].K, h->s_main);
                if (m.whhU8[l][d]) {
                    mp_launch_pack_w_u8(find(s.id, K_WHH, l, d), m.whhU8[l][d], m.H, h->s_main);
                    mp_launch_pack_w_u8(find(s.id, K_WIH, l, d), m.wihU8[l][d], m.ih[l].K, h->s_main);
                }
                if (m.whhR[l][d] && m.H == 256) {                  // mp_lstm_v1: per-lane order
                    mp_launch_pack_w_v1(find(s.id, K_WHH, l, d), m.whhR[l][d], m.H, h->s_main);
                    mp_launch_pack_w_v1(find(s.id, K_WIH, l, d), m.wihR[l][d], m.ih[l].K, h->s_main);
                } else if (m.whhR[l][d]) {                         // mp_lstm_v1s: the matrices as they are
                    HIPCHK(h, hipMemcpyAsync(m.whhR[l][d], find(s.id, K_WHH, l, d), (size_t)4 * m.H * m.H * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
                    HIPCHK(h, hipMemcpyAsync(m.wihR[l][d], find(s.id, K_WIH, l, d), (size_t)4 * m.H * m.ih[l].K * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
                }
                if (m.whhP8[l][d]) {
                    mp_launch_pack_whh_persist(find(s.id, K_WHH, l, d), m.whhP8[l][d], m.H, 8, h->s_main);
                    mp_launch_pack_wih_persist(find(s.id, K_WIH, l, d), m.wihP8[l][d], m.H, m.ih[l].K, 8, h->s_main);
                }
                if (m.whhP16[l][d]) {
                    mp_launch_pack_whh_persist(find(s.id, K_WHH, l, d), m.whhP16[l][d], m.H, 16, h->s_main);
                    mp_launch_pack_wih_persist(find(s.id, K_WIH, l, d), m.wihP16[l][d], m.H, m.ih[l].K, 16, h->s_main);
                }
                if (m.H == 256) {
                    mp_launch_pack_w_x3(find(s.id, K_WHH, l, d), m.whhX[l][d], m.H, m.nsliceX, h->s_main);
                    mp_launch_pack_w_x3(find(s.id, K_WIH, l, d), m.wihX[l][d], m.ih[l].K, m.nsliceX, h->s_main);
                }
            }
    }
    {   // pose.linear1 on top of velocity.linear1 (same inputs: cat(joints, imu), net.py:106,113): pair words and bias
        const Packed& a = h->mod[MP_MOD_POSE].lin1;
        const Packed& b = h->mod[MP_MOD_VELOCITY].lin1;
        Packed& pv = h->lin1_pv;
        if (a.K == b.K && a.Kpad == b.Kpad && a.N == a.Npad && a.bn == b.bn && a.N % a.bn == 0) {
            pv.N = a.N + b.N; pv.K = a.K; pv.Kpad = a.Kpad; pv.bn = a.bn; pv.Npad = a.Npad + b.Npad;
            if (int rc = dev_alloc(h, (void**)&pv.Wp, (size_t)pv.Npad * pv.Kpad * sizeof(float))) return rc;
            if (int rc = dev_alloc(h, (void**)&pv.bias, (size_t)pv.Npad * sizeof(float))) return rc;
            const size_t na = (size_t)a.Npad * a.Kpad, nb = (size_t)b.Npad * b.Kpad;
            if (int rc = dev_alloc(h, (void**)&pv.W, (size_t)pv.Npad * pv.Kpad * sizeof(float))) return rc;   // the fp32 image of the stack
            HIPCHK(h, hipMemcpyAsync(pv.W, a.W, na * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(pv.W + na, b.W, nb * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(pv.Wp, a.Wp, na * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(pv.Wp + na, b.Wp, nb * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(pv.bias, a.bias, (size_t)a.Npad * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(pv.bias + a.Npad, b.bias, (size_t)b.Npad * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            if (int rc = dev_alloc(h, (void**)&pv.Wf, (size_t)pv.Npad * pv.Kpad * sizeof(float))) return rc;
            mp_launch_pack_wfrag(pv.W, pv.Wf, pv.Npad, pv.Kpad, h->s_main);
            // the foot-contact block reads the same rows too (net.py:113): its 64 linear1 rows under the other 512
            const Packed& f = h->mod[MP_MOD_FOOT_CONTACT].lin1;
            Packed& pvf = h->lin1_pvf;
            if (f.K == a.K && f.Kpad == a.Kpad && f.N % 64 == 0 && f.N <= f.Npad) {
                pvf.N = pv.N + f.N; pvf.K = a.K; pvf.Kpad = a.Kpad; pvf.bn = a.bn; pvf.Npad = pv.Npad + f.N;
                const size_t nf = (size_t)f.N * f.Kpad;
                if (int rc = dev_alloc(h, (void**)&pvf.W, (size_t)pvf.Npad * pvf.Kpad * sizeof(float))) return rc;
                if (int rc = dev_alloc(h, (void**)&pvf.Wf, (size_t)pvf.Npad * pvf.Kpad * sizeof(float))) return rc;
                if (int rc = dev_alloc(h, (void**)&pvf.bias, (size_t)pvf.Npad * sizeof(float))) return rc;
                HIPCHK(h, hipMemcpyAsync(pvf.W, pv.W, (na + nb) * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
                HIPCHK(h, hipMemcpyAsync(pvf.W + na + nb, f.W, nf * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
                HIPCHK(h, hipMemcpyAsync(pvf.bias, pv.bias, (size_t)pv.Npad * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
                HIPCHK(h, hipMemcpyAsync(pvf.bias + pv.Npad, f.bias, (size_t)f.N * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
                mp_launch_pack_wfrag(pvf.W, pvf.Wf, pvf.Npad, pvf.Kpad, h->s_main);
            }
        }
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    return MP_OK;
}

int setup_smpl(mp_handle* h, const int32_t parent[24], const float J[72]) {
    int par[24], depth[24];
    float bone[72], j[72];
    for (int i = 0; i < 24; ++i) {
        par[i] = i == 0 ? -1 : parent[i];
        if (i > 0 && (par[i] < 0 || par[i] >= i)) return fail(h, MP_ERR_INVALID, "parent[%d] = %d must be in [0,%d)", i, par[i], i);
        for (int c = 0; c < 3; ++c) j[i * 3 + c] = J[i * 3 + c] - J[c];          // model.py:87
    }
    depth[0] = 0;
    for (int c = 0; c < 3; ++c) bone[c] = j[c];
    for (int i = 1; i < 24; ++i) {
        depth[i] = depth[par[i]] + 1;
        if (depth[i] > 8) return fail(h, MP_ERR_INVALID, "kinematic tree deeper than 8 levels");
        for (int c = 0; c < 3; ++c) bone[i * 3 + c] = j[i * 3 + c] - j[par[i] * 3 + c];   // spatial.py:148-167
    }
    for (int c = 0; c < 6; ++c) h->feet_pos[c] = j[30 + c];                         // net.py:48
    h->floor_y = j[31] < j[34] ? j[31] : j[34];                                     // net.py:49
    if (int rc = dev_alloc(h, (void**)&h->parent_dev, sizeof(par))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->depth_dev, sizeof(depth))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->bone_dev, sizeof(bone))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->jrest_dev, sizeof(j))) return rc;
    HIPCHK(h, hipMemcpy(h->jrest_dev, j, sizeof(j), hipMemcpyHostToDevice));
    for (int c = 0; c < 3; ++c) h->J0[c] = J[c];
    HIPCHK(h, hipMemcpy(h->parent_dev, par, sizeof(par), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->depth_dev, depth, sizeof(depth), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->bone_dev, bone, sizeof(bone), hipMemcpyHostToDevice));
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

// Multi-branch graphs (graph mode 1) can SIGSEGV inside hipGraphLaunch of this ROCm, depending on the process's stream history
// (profiles/r02_hipgraph_segv.md): an option that can crash the host process is not one `int` away -- mode 1 means mode 2
// (single-branch: bitwise-equal results, same speed) unless the environment asks for the real thing.
bool multibranch_graphs_allowed() {
    const char* e = getenv("MP_GRAPH_MULTIBRANCH");
    return e && e[0] == '1';
}

int create_common(mp_handle** out, int device, const float* blob, bool blob_on_device, size_t n_floats,
                  const int32_t parent[24], const float J[72]) {
    if (!out || !parent || !J) return fail(nullptr, MP_ERR_INVALID, "mp_create: NULL argument");
    const bool body_only = blob == nullptr && n_floats == 0;
    if (!body_only && (!blob || n_floats != manifest_floats()))
        return fail(nullptr, MP_ERR_INVALID, "mp_create: weight blob has %zu floats, expected %zu", n_floats, manifest_floats());
    {   // a device index this process cannot see is the caller's mistake, not a runtime failure (round 5: a clear MP_ERR_INVALID
        // instead of whatever hipSetDevice says)
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess) { (void)hipGetLastError(); n_dev = 0; }
        if (device < 0 || device >= n_dev)
            return fail(nullptr, MP_ERR_INVALID, "mp_create: device index %d, but this process sees %d device(s) "
                        "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumber them from 0)", device, n_dev);
    }
    mp_handle* h = new mp_handle();
    h->device = device;
    h->has_weights = !body_only;
    auto bail = [&](int rc) { g_create_error = h->err; mp_destroy(h); return rc; };
    DeviceScope on_device(device);                      // (the caller's current device is restored on every return path)
    if (!on_device.ok) { h->err = "hipSetDevice failed"; return bail(MP_ERR_HIP); }
    if (const char* e = getenv("MP_GRAPH")) { h->use_graph = e[0] && e[0] != '0'; h->graph_serial = e[0] == '2' || !multibranch_graphs_allowed(); }
    {   // dynamic-LDS limits of the persistent kernels are per-device attributes (and must not be set under capture)
        hipError_t ea = mp_lstm_persist_device_attrs();
        if (ea == hipSuccess) ea = mp_lstm_u8_device_attrs();
        if (ea == hipSuccess) ea = mp_lstm_v1_device_attrs();
        if (ea == hipSuccess) ea = mp_lstm_x3_device_attrs();
        if (ea == hipSuccess) ea = mp_lstm_x3w_device_attrs();
        if (ea != hipSuccess) { h->err = std::string("hipFuncSetAttribute failed: ") + hipGetErrorString(ea); return bail(MP_ERR_HIP); }
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
        // a persistent layer needs at least one cluster (dirs x 16 workgroups) resident at one workgroup per CU
        if (h->n_cu < 32) h->persist = false;
    }
    {
        // side-by-side schedules place clusters XCD by XCD (place_clusters): only on a device that deals workgroups round
        // robin over 8 XCDs of n_cu / 8 CUs each -- probed, not assumed
        int* probe = nullptr;
        h->xcd_rr = false;
        if (h->n_cu % 8 == 0 && hipHostMalloc((void**)&probe, 64 * sizeof(int), hipHostMallocDefault) == hipSuccess) {
            for (int i = 0; i < 64; ++i) probe[i] = -1;
            mp_launch_xcc_probe(probe, h->s_main);
            if (hipStreamSynchronize(h->s_main) == hipSuccess) {
                unsigned seen = 0;
                bool ok = true;
                for (int b = 0; b < 64; ++b) ok = ok && probe[b] >= 0 && probe[b] < 8 && probe[b] == probe[b & 7];
                for (int b = 0; b < 8 && ok; ++b) seen |= 1u << probe[b];
                h->xcd_rr = h->xcd_probe = ok && seen == 0xffu;
            }
            (void)hipHostFree(probe);
        }
        (void)hipGetLastError();
    }
    if (const char* e = getenv("MP_LSTM_MODE")) {
        h->persist = strcmp(e, "step") != 0;
        h->x3 = h->persist && strcmp(e, "x3") == 0;            // "fp32" (default) | "x3" | "step"
    }
    if (const char* e = getenv("MP_WAIT_MS")) { const double ms = atof(e); if (ms > 0.0 && ms < 60000.0) h->wait_ticks = (unsigned long long)(ms * 1e5); }
    // MP_VARIANT: ONE debug switch for the kernel / schedule variants kept for cross-checks and A/B runs -- a comma-separated
    // list of key=value (tests/test_gpu_parity.py exercises them; nothing here changes results beyond summation order):
    //   x3w=0..3       split-bf16 layers on the four-wave kernel: bit 0 K_in=256, bit 1 K_in=512 (2)
    //   slices16=0 / slices32=0: no 16- / 32-slice kernels (bidirectional fp32 layers always on 8 slices per slab)
    //   wide=0         never run pose / velocity / foot contact side by side;  half=0: no pose-on-half-the-chip schedule
    //   exclusive=0    no LDS padding / XCD tables for concurrent persistent launches
    //   epoch_tags=0   zero the exchange area before every fp32 layer launch;  epoch_start=N: first epoch base (wrap tests)
    //   vf=0           foot-contact layers as launches of their own beside velocity (B > 128), not as riders in its workgroups
    //   wf=0           velocity layers as two 16-slice launches (rounds 3-4), not as one two-layer wavefront launch (B > 128)
    //   late_pair=0    64 < B <= 128: both pose layers on 8 slices beside velocity (schedules 2 / 3) instead of schedule 4
    if (const char* e = getenv("MP_VARIANT")) {
        std::string all(e);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string tok = all.substr(pos, end - pos);
            pos = end + 1;
            const size_t eq = tok.find('=');
            if (eq == std::string::npos) continue;
            const std::string key = tok.substr(0, eq);
            const unsigned long v = strtoul(tok.c_str() + eq + 1, nullptr, 0);
            if (key == "x3w") h->x3w_mask = (int)(v & 3);
            else if (key == "slices16") h->slices16_ok = v != 0;
            else if (key == "slices32") h->slices32_ok = v != 0;
            else if (key == "vec") h->vec_ok = v != 0;
            else if (key == "wide") h->wide_ok = v != 0;
            else if (key == "half") h->half_ok = v != 0;
            else if (key == "exclusive") h->exclusive_ok = v != 0;
            else if (key == "epoch_tags") h->epoch_tags = v != 0;
            else if (key == "epoch_start") { if (v >= 1 && v < 0xf0000000ul) h->epoch_start = (unsigned)v; }
            else if (key == "kin_scalar") {}                   // read by mp_kin.hip
            else if (key == "one_stream") h->one_stream_ok = v != 0;   // 0 = the round-3 three-stream serial schedule (a cross-check)
            else if (key == "vf") h->vf_ok = v != 0;
            else if (key == "wf") h->wf_ok = v != 0;
            else if (key == "late_pair") h->late_pair_ok = v != 0;
            else { h->err = "MP_VARIANT: unknown key '" + key + "'"; return bail(MP_ERR_INVALID); }
        }
    }
    if (getenv("MP_PERSIST_PROF")) {
        if (hipMalloc((void**)&h->prof_dev, kProfWords * sizeof(long long)) != hipSuccess) h->prof_dev = nullptr;
        else (void)hipMemset(h->prof_dev, 0, kProfWords * sizeof(long long));
    }
    hipError_t e = hipSuccess;
    e = e ? e : hipStreamCreateWithFlags(&h->s_main, hipStreamNonBlocking);
    e = e ? e : hipStreamCreateWithFlags(&h->s_vel, hipStreamNonBlocking);
    e = e ? e : hipStreamCreateWithFlags(&h->s_foot, hipStreamNonBlocking);
    // (three streams, not four: with the caller's own stream that makes four -- the number of hardware queues the HIP runtime
    //  multiplexes a process's streams onto by default (GPU_MAX_HW_QUEUES).  Two streams on one queue are serialised: with a
    //  fourth library stream the foot-contact chain was seen queued behind pose's linear2 / IK for 110 us,
    //  profiles/r02_timeline_256x125.txt.  The pose tail (serial schedule) and the velocity chain (side-by-side schedules)
    //  never run in the same call, so they share s_vel.)
    h->s_gp = h->s_vel;
    for (hipEvent_t& ev : h->ev_x) e = e ? e : hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (!e) e = hipHostMalloc((void**)&h->err_host, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (!e) { memset(h->err_host, 0, 64); e = hipHostGetDevicePointer((void**)&h->err_dev, h->err_host, 0); }
    hipEvent_t* evs[5] = {&h->ev_in, &h->ev_out, &h->ev_j, &h->ev_v, &h->ev_f};
    for (hipEvent_t* ev : evs) e = e ? e : hipEventCreateWithFlags(ev, hipEventDisableTiming);
    if (e != hipSuccess) { h->err = std::string("stream/event creation failed: ") + hipGetErrorString(e); return bail(MP_ERR_HIP); }
    float* staging = nullptr;
    const float* dev_blob = blob;
    if (body_only) {
        int rc_b = setup_smpl(h, parent, J);
        if (rc_b) return bail(rc_b);
        *out = h;
        return MP_OK;
    }
    if (!blob_on_device) {
        if (hipMalloc((void**)&staging, n_floats * sizeof(float)) != hipSuccess ||
            hipMemcpy(staging, blob, n_floats * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            h->err = "weight upload failed";
            if (staging) (void)hipFree(staging);
            return bail(MP_ERR_HIP);
        }
        dev_blob = staging;
    }
    int rc = pack_weights(h, dev_blob);
    if (staging) (void)hipFree(staging);
    if (rc) return bail(rc);
    rc = setup_smpl(h, parent, J);
    if (rc) return bail(rc);
    *out = h;
    return MP_OK;
}

// ------------------------------------------------------------------------------------------ plans
// Workspaces by CAPACITY (round 6; rounds 1-5 kept one plan per exact (B, T), 0.4 GB at 256 x 125 and 60-110 ms to map, and
// the facade cut a replayed sequence into power-of-two chunks so that a service would not thrash).  Internal activations are
// time-major [T][B][C] with the batch as a run-time stride, per-sequence buffers are indexed by b alone and the exchange areas by
// slab: a plan allocated for (capB, capRows) serves every call with B <= capB sequences and B * T <= capRows rows.  A call takes
// the smallest plan of ITS batch class that has the rows; batch classes are exact up to 64 sequences (a handful of shapes: ticks,
// evaluate.py's single sequence, small batches) and {2^k, 1.5 * 2^k} above; a class whose plan is too short gets a new one of at
// least twice the rows, so a caller that walks through sequence lengths (evaluate.py) allocates a few times, not per length.
void free_plan(mp_handle* h, Plan* p) {
    // captured graphs reference the workspaces of the plan they were captured on (and are few): all of them go
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second.exec);
    h->graphs.clear();
    for (void* q : p->allocs) (void)hipFree(q);
    if (p->lengths_pin) (void)hipHostFree(p->lengths_pin);
    delete p;
}

constexpr size_t kMaxPlans = 24;
constexpr size_t kMaxRows = (size_t)4 << 20;      // rows (B * T, ~12.5 KB of workspace each) of all plans together: 50 GB

size_t round_class(size_t n) {                    // the next of {2^k, 1.5 * 2^k}
    size_t p = 64;
    while (true) {
        if (n <= p) return p;
        if (n <= p + p / 2) return p + p / 2;
        p *= 2;
    }
}
int plan_batch_class(int B) { return B <= 64 ? B : (int)round_class((size_t)B); }

// `keep`: a plan the caller is still using (mp_stream_replay holds two): never the victim of this call's eviction
int get_plan(mp_handle* h, int B, int T, Plan** out, const Plan* keep = nullptr) {
    // (the layer kernels step through their output with a 32-bit row pitch: B * 512 floats must stay below 4 GB)
    if ((size_t)B * 512 * sizeof(float) > 0xffffffffull)
        return fail(h, MP_ERR_INVALID, "batch of %d sequences is beyond the supported 2^21 - 1; split it", B);
    const int cls = plan_batch_class(B);
    const size_t rows = (size_t)B * T;
    Plan* best = nullptr;
    size_t class_rows = 0;                         // the longest plan this class has so far
    for (Plan* q : h->plans) {
        if (q->capB != cls || q == keep) continue;
        class_rows = q->capRows > class_rows ? q->capRows : class_rows;
        if (q->capRows >= rows && (!best || q->capRows < best->capRows)) best = q;
    }
    if (best) {
        if (best->lastB != B)                      // another batch size wrote the exchange areas last: start from zeroed ones
            for (ModuleWS& w : best->ws) w.hx_epoch = 0;
        best->B = B; best->T = T; best->lastB = B; best->last_use = ++h->use_clock;
        *out = best;
        return MP_OK;
    }
    size_t cap_rows = round_class((size_t)cls * T);      // (a plan serves its whole batch class at this length: 100 x T and 128 x T share one)
    if (class_rows && cap_rows < 2 * class_rows) cap_rows = round_class(2 * class_rows);
    while (true) {
        size_t total = cap_rows;
        for (const Plan* q : h->plans) total += q->capRows;
        if (h->plans.size() < kMaxPlans && total <= kMaxRows) break;
        size_t victim = h->plans.size();
        for (size_t i = 0; i < h->plans.size(); ++i) {
            const Plan* q = h->plans[i];
            if (q->streaming || q == keep) continue;
            if (victim == h->plans.size() || q->last_use < h->plans[victim]->last_use) victim = i;
        }
        if (victim == h->plans.size()) break;
        HIPCHK(h, hipDeviceSynchronize());
        free_plan(h, h->plans[victim]);
        h->plans.erase(h->plans.begin() + (long)victim);
    }
    Plan* p = new Plan();
    p->B = B; p->T = T; p->lastB = B; p->capB = cls; p->capRows = cap_rows; p->last_use = ++h->use_clock;
    h->plans.push_back(p);
    ++h->plan_allocs;
    const size_t M = cap_rows, CB = (size_t)cls;
    for (int id = 0; id < 4; ++id) {
        const ModuleW& m = h->mod[id];
        ModuleWS& w = p->ws[id];
        w.xproj = nullptr;                                   // gate pre-activations: per-step mode only, allocated on demand
        if (int rc = dev_alloc(h, (void**)&w.out0, M * m.dirs * m.H * sizeof(float), &p->allocs)) return rc;
        if (int rc = dev_alloc(h, (void**)&w.out1, M * m.dirs * m.H * sizeof(float), &p->allocs)) return rc;
        if (m.H == 256 && m.dirs == 1)
            if (int rc = dev_alloc(h, (void**)&w.x1, M * m.H * sizeof(float), &p->allocs)) return rc;
        for (int l = 0; l < 2; ++l)
            for (int d = 0; d < m.dirs; ++d) {
                if (int rc = dev_alloc(h, (void**)&w.hbuf[l][d], (size_t)2 * CB * m.H * sizeof(float), &p->allocs)) return rc;
                if (int rc = dev_alloc(h, (void**)&w.cbuf[l][d], CB * m.H * sizeof(float), &p->allocs)) return rc;
            }
        w.hx_bytes = (size_t)2 * ((CB + 15) / 16) * ((size_t)4 * 16 * m.H + 16) * sizeof(unsigned long long);
        if (int rc = dev_alloc(h, (void**)&w.hx, w.hx_bytes, &p->allocs)) return rc;
        if (m.H == 256)
            if (int rc = dev_alloc(h, (void**)&w.hx2, w.hx_bytes, &p->allocs)) return rc;
    }
    if (int rc = dev_alloc(h, (void**)&p->r6d, M * 96 * sizeof(float), &p->allocs)) return rc;
    if (int rc = dev_alloc(h, (void**)&p->lengths_dev, CB * sizeof(int), &p->allocs)) return rc;
    HIPCHK(h, hipHostMalloc((void**)&p->lengths_pin, CB * sizeof(int), hipHostMallocDefault));
    *out = p;
    return MP_OK;
}

// per-step mode keeps the [B*T, dirs*4H] gate pre-activations in HBM; allocate them outside of any graph capture
int ensure_step_ws(mp_handle* h, Plan* p) {
    if (h->persist) return MP_OK;
    for (int id = 0; id < 4; ++id) {
        const ModuleW& m = h->mod[id];
        ModuleWS& w = p->ws[id];
        if (!w.xproj)
            if (int rc = dev_alloc(h, (void**)&w.xproj, p->capRows * m.dirs * 4 * m.H * sizeof(float), &p->allocs)) return rc;
    }
    return MP_OK;
}

int upload_lengths(mp_handle* h, Plan* p, const int32_t* lengths) {
    if (int rc = ensure_step_ws(h, p)) return rc;
    int mx = 0;
    for (int b = 0; b < p->B; ++b) {
        if (lengths[b] < 1 || lengths[b] > p->T) return fail(h, MP_ERR_LENGTHS, "lengths[%d] = %d outside 1..%d", b, lengths[b], p->T);
        mx = lengths[b] > mx ? lengths[b] : mx;
    }
    if (mx != p->T)
        return fail(h, MP_ERR_LENGTHS, "max(lengths) = %d but T = %d (the reference's torch.cat at net.py:106 fails)", mx, p->T);
    if ((int)p->lengths_cache.size() == p->B && memcmp(p->lengths_cache.data(), lengths, p->B * sizeof(int)) == 0)
        return MP_OK;
    HIPCHK(h, hipStreamSynchronize(h->s_main));      // the staging buffer may still be in flight
    memcpy(p->lengths_pin, lengths, p->B * sizeof(int));
    HIPCHK(h, hipMemcpyAsync(p->lengths_dev, p->lengths_pin, p->B * sizeof(int), hipMemcpyHostToDevice, h->s_main));
    p->lengths_cache.assign(lengths, lengths + p->B);
    return MP_OK;
}

// ------------------------------------------------------------------------------------------ timing
hipEvent_t next_event(mp_handle* h) {
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_used++];
}
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

// ------------------------------------------------------------------------------------------ one RNN block
enum StateMode { STATE_ZERO, STATE_FROM };

RowMap internal_map(const float* base, int B, int width) { return RowMap{base, (long)width, (long)B * width, width}; }
RowMap user_map(const float* base, int T, int width) { return RowMap{base, (long)T * width, (long)width, width}; }

int run_gemm(mp_handle* h, hipStream_t s, RowMap a0, RowMap a1, const Packed& w, float* C, long cStrideB,
             long cStrideT, int M, int B, int relu, bool pair_out = false, bool a_pairs = false, bool x3_gemm = false,
             unsigned long long* zero_hx = nullptr, int zero_ncl = 0) {
    SegScope seg(h, s, 0, 1, 2.0 * M * (double)w.N * w.K);
    GemmArgs g;
    g.a0 = a0; g.a1 = a1; g.W = w.W; g.bias = w.bias; g.C = C; g.cStrideB = cStrideB; g.cStrideT = cStrideT;
    g.M = M; g.N = w.N; g.K = w.K; g.Kpad = w.Kpad; g.B = B; g.relu = relu; g.pairOut = pair_out ? 1 : 0;
    g.Wf = w.Wf; g.NB = w.Wf ? w.Npad / 32 : 0;
    if (x3_gemm && w.Wp) {                                    // split-bf16 mode: the H = 256 blocks' linear layers run on bf16 MFMAs as well
        g.W = w.Wp; g.aPairs = a_pairs ? 1 : 0; g.zero_hx = zero_hx; g.zero_ncl = zero_hx ? zero_ncl : 0;
        mp_launch_gemm_x3(g, w.bn, s);
    } else {
        mp_launch_gemm(g, w.bn, s);
    }
    return MP_OK;
}

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

// linear1's output X1 normally lives in out1's memory (dead until layer 1 writes it); the two-layer wavefront
// kernel writes out1 while layer 0 is still reading X1, so there X1 goes to the (otherwise unused) out0
// split-bf16 operands for this module's LSTM layers?  (X1 and the layer-0 output are then stored as pairs)
bool use_x3(const mp_handle* h, const ModuleW& m) { return h->persist && h->x3 && m.H == 256; }

// Slices per slab of an exact-fp32 layer launch: a bidirectional H = 256 layer normally uses 8 slices (four 512-register waves, one
// per CU at B = 256); when the batch is small enough that 16 slices still fit the chip (B <= 128), the 16-slice / 4-wave
// decomposition halves the matrix work per CU and step (7 600 instead of 11 900 cycles per step).
int fp32_slices(const mp_handle* h, const ModuleW& m, int B) {
    const int nslab = (B + 15) / 16;
    const int cus = h->n_cu < 256 ? h->n_cu : 256;
    if (h->pose_slices8 && &m == &h->mod[MP_MOD_POSE]) return m.nslice;
    // one or two slabs (B <= 32): 32 slices of 8 units, every (direction, slab) cluster on an XCD of its own (mp_lstm_u8.hip) --
    // at most 4 + 2 clusters of pose and velocity side by side, foot contact on the two XCDs that are left
    // (the joints block always has the chip to itself: 32 slices while its 2 * nslab clusters find an XCD each, B <= 64)
    // (without placement tables only blocks that have the chip to themselves use them: joints, and pose in the serial schedule)
    const int max32 = &m == &h->mod[MP_MOD_JOINTS] ? 4 : ((!h->xcd_rr && &m == &h->mod[MP_MOD_VELOCITY]) ? 0 : 2);
    if (m.H == 256 && m.whhU8[0][0] && h->slices32_ok && h->slices16_ok && nslab <= max32 && cus == 256) return 32;
    if (m.H == 256 && m.nslice == 8 && m.whhP16[0][0] && h->slices16_ok && m.dirs * nslab * 16 <= cus) return 16;
    return m.nslice;
}

int layer_workgroups(const mp_handle* h, const ModuleW& m, int B) {
    const int nslab = (B + 15) / 16;
    return m.dirs * nslab * (use_x3(h, m) ? m.nsliceX : fp32_slices(h, m, B));
}

float* x1_buffer(const mp_handle* h, const ModuleW& m, ModuleWS& w) {
    (void)h; (void)m;
    return w.x1 ? w.x1 : w.out1;
}

int rnn_g0(const RnnJob& j, hipStream_t s) {
    mp_handle* h = j.h;
    const ModuleW& m = h->mod[j.id];
    ModuleWS& w = j.p->ws[j.id];
    const int B = j.p->B, T = j.p->T, M = B * T, H = m.H, dirs = m.dirs;
    const RowMap none{nullptr, 0, 0, 0};
    float* X1 = x1_buffer(h, m, w);
    // (split-bf16 mode: this GEMM also re-arms the exchange area of the layer-0 launch that follows it)
    run_gemm(h, s, j.a0, j.a1, m.lin1, X1, H, (long)B * H, M, B, 1, use_x3(h, m), false, use_x3(h, m),
             use_x3(h, m) ? w.hx : nullptr, dirs * ((B + 15) / 16));                               // rnn.py:22
    // the per-step kernels take the input projection from a GEMM; the persistent kernel computes it itself
    if (!h->persist && !w.xproj) return fail(h, MP_ERR_INVALID, "internal: per-step workspace missing");
    if (!h->persist)
        run_gemm(h, s, internal_map(X1, B, H), none, m.ih[0], w.xproj, dirs * 4 * H, (long)B * dirs * 4 * H, M, B, 0);
    for (int l = 0; l < 2; ++l)
        for (int d = 0; d < dirs; ++d) {
            const size_t n = (size_t)B * H * sizeof(float);
            const int k = l * dirs + d;
            if (h->persist && j.out_h == j.in_h && j.out_h) {
                // the persistent kernel reads its initial and writes its final (h,c) in place: carried state
                // (velocity.rnn_state) needs no staging copies at all
            } else if (j.mode == STATE_FROM) {
                HIPCHK(h, hipMemcpyAsync(w.hbuf[l][d], j.in_h + (size_t)k * B * H, n, hipMemcpyDeviceToDevice, s));
                HIPCHK(h, hipMemcpyAsync(w.cbuf[l][d], j.in_c + (size_t)k * B * H, n, hipMemcpyDeviceToDevice, s));
            } else if (!h->persist) {                                      // persistent kernel: zero_state flag
                HIPCHK(h, hipMemsetAsync(w.hbuf[l][d], 0, n, s));
                HIPCHK(h, hipMemsetAsync(w.cbuf[l][d], 0, n, s));
            }
        }
    HIPCHK(h, hipGetLastError());
    return MP_OK;
}

// linear1 of the pose and the velocity block in ONE GEMM (same rows cat(joints, imu); stacked weights; either operand mode): one launch
// instead of two on two streams, and no cross-stream edge into the velocity layers later.  Only what rnn_g0 does for
// the persistent path with zero / in-place state; returns false when that does not apply.
bool rnn_g0_pose_velocity(const RnnJob& jp, const RnnJob& jv, hipStream_t s, int* rc, const RnnJob* jf = nullptr) {
    mp_handle* h = jp.h;
    const ModuleW& mp = h->mod[jp.id];
    const ModuleW& mv = h->mod[jv.id];
    *rc = MP_OK;
    if (!h->lin1_pv.Wp || !h->persist || use_x3(h, mp) != use_x3(h, mv)) return false;
    const bool x3 = use_x3(h, mp);
    if (jp.mode != STATE_ZERO || !(jv.mode == STATE_ZERO || (jv.out_h == jv.in_h && jv.out_h))) return false;
    if (jp.a0.base != jv.a0.base || jp.a1.base != jv.a1.base) return false;
    ModuleWS& wp = jp.p->ws[jp.id];
    ModuleWS& wv = jv.p->ws[jv.id];
    const int B = jp.p->B, T = jp.p->T, M = B * T, H = mp.H;
    // (jf: the foot-contact block's linear1 as a third output of the same launch -- exact-fp32 operands, fragment-ordered W)
    const bool three = jf != nullptr && !x3 && h->lin1_pvf.Wf != nullptr && jf->mode == STATE_ZERO && jf->a0.base == jp.a0.base &&
                       jf->a1.base == jp.a1.base;
    if (jf != nullptr && !three) return false;
    const Packed& w = three ? h->lin1_pvf : h->lin1_pv;
    SegScope seg(h, s, 0, 1, 2.0 * M * (double)w.N * w.K);
    GemmArgs g;
    if (three) {
        const ModuleW& mf = h->mod[jf->id];
        g.C3 = x1_buffer(h, mf, jf->p->ws[jf->id]); g.nsplit3 = h->lin1_pv.Npad; g.c3StrideB = mf.H; g.c3StrideT = (long)B * mf.H;
    }
    g.a0 = jp.a0; g.a1 = jp.a1; g.W = x3 ? w.Wp : w.W; g.bias = w.bias; g.C = x1_buffer(h, mp, wp); g.C2 = x1_buffer(h, mv, wv);
    g.nsplit = mp.lin1.Npad; g.cStrideB = H; g.cStrideT = (long)B * H;
    g.M = M; g.N = w.N; g.K = w.K; g.Kpad = w.Kpad; g.B = B; g.relu = 1; g.pairOut = x3 ? 1 : 0; g.aPairs = 0;
    g.Wf = w.Wf; g.NB = w.Wf ? w.Npad / 32 : 0;
    if (x3) {
        const int nslab = (B + 15) / 16;
        g.zero_hx = wp.hx; g.zero_ncl = mp.dirs * nslab; g.zero_hx2 = wv.hx; g.zero_ncl2 = mv.dirs * nslab;
        mp_launch_gemm_x3(g, w.bn, s);
    } else {
        mp_launch_gemm(g, w.bn, s);              // exact-fp32 operands: the same stacked launch (round 3)
    }
    if (hipGetLastError() != hipSuccess) *rc = fail(h, MP_ERR_HIP, "fused linear1 launch failed");
    return true;
}

// joints.linear2 and the stacked linear1 of pose | velocity | foot contact as ONE launch (mp_gemm_l2l1): what rnn_g2(J) and
// rnn_g0_pose_velocity(P, V, F) do for the full-batch exact-fp32 schedule.  false = not applicable (nothing launched).
bool rnn_g2_g0_fused(const RnnJob& jj, const RnnJob& jp, const RnnJob& jv, const RnnJob& jf, hipStream_t s, int* rc) {
    mp_handle* h = jj.h;
    *rc = MP_OK;
    const ModuleW& mj = h->mod[jj.id];
    const ModuleW& mp = h->mod[jp.id];
    const ModuleW& mv = h->mod[jv.id];
    const ModuleW& mf = h->mod[jf.id];
    if (!h->persist || use_x3(h, mj) || use_x3(h, mp) || use_x3(h, mv) || !h->lin1_pvf.Wf || !mj.lin2.Wf) return false;
    if (jj.out_h || jp.mode != STATE_ZERO || jf.mode != STATE_ZERO || !(jv.mode == STATE_ZERO || (jv.out_h == jv.in_h && jv.out_h))) return false;
    // the stacked GEMM must read exactly what linear2 writes: cat(pred_joints, imu) with pred_joints = this call's output
    if (jp.a0.base != jj.y || jv.a0.base != jj.y || jf.a0.base != jj.y || jp.a1.base != jv.a1.base || jp.a1.base != jf.a1.base) return false;
    if (jp.a0.strideB != jj.yStrideB || jp.a0.strideT != jj.yStrideT) return false;
    const int B = jj.p->B, T = jj.p->T, M = B * T, H = mp.H;
    const Packed& w2 = mj.lin2;
    const Packed& w1 = h->lin1_pvf;
    GemmArgs g2, g1;
    g2.a0 = internal_map(jj.p->ws[jj.id].out1, B, mj.dirs * mj.H); g2.a1 = RowMap{nullptr, 0, 0, 0};
    g2.W = w2.W; g2.Wf = w2.Wf; g2.NB = w2.Npad / 32; g2.bias = w2.bias; g2.C = jj.y; g2.cStrideB = jj.yStrideB; g2.cStrideT = jj.yStrideT;
    g2.M = M; g2.N = w2.N; g2.K = w2.K; g2.Kpad = w2.Kpad; g2.B = B; g2.relu = 0;
    g1.a0 = jp.a0; g1.a1 = jp.a1; g1.W = w1.W; g1.Wf = w1.Wf; g1.NB = w1.Npad / 32; g1.bias = w1.bias;
    g1.C = x1_buffer(h, mp, jp.p->ws[jp.id]); g1.C2 = x1_buffer(h, mv, jv.p->ws[jv.id]); g1.C3 = x1_buffer(h, mf, jf.p->ws[jf.id]);
    g1.nsplit = mp.lin1.Npad; g1.nsplit3 = h->lin1_pv.Npad; g1.cStrideB = H; g1.cStrideT = (long)B * H; g1.c3StrideB = mf.H; g1.c3StrideT = (long)B * mf.H;
    g1.M = M; g1.N = w1.N; g1.K = w1.K; g1.Kpad = w1.Kpad; g1.B = B; g1.relu = 1;
    if (!mp_gemm_l2l1_applicable(g2, g1)) return false;
    SegScope seg(h, s, 0, 1, 2.0 * M * ((double)w2.N * w2.K + (double)w1.N * w1.K));
    (void)mp_launch_gemm_l2l1(g2, g1, s);
    if (hipGetLastError() != hipSuccess) *rc = fail(h, MP_ERR_HIP, "fused linear2 / linear1 launch failed");
    return true;
}

inline int fm_kin0(const mp_handle* h) { return h->mod[MP_MOD_FOOT_CONTACT].H; }   // K_in of the rider's layer 0 (= its H)

// Do the two layers of module m (the unidirectional H = 256 block) run as ONE two-layer wavefront launch of the 8-slice kernel
// (mp_lstm_fused<256,8,256,*,*,WF>) at this shape?  Full batches only (B > 128: the schedules of smaller batches place 16-slice
// velocity clusters beside pose clusters with XCD tables), exact-fp32 operands, and a layer-0 output the kernel can address
// with 32-bit byte offsets.
bool wavefront_applies(const mp_handle* h, const ModuleW& m, int B, int T) {
    return h->persist && h->wf_ok && !use_x3(h, m) && m.H == 256 && m.dirs == 1 && m.whhP8[0][0] != nullptr && B > 128 &&
           fp32_slices(h, m, B) == 16 && (size_t)B * T * m.H * sizeof(float) < 0x7fffffffull;
}

// The same for ONE sequence on the matrix-vector kernel (mp_lstm_v1<256,*,true>): the chain of mp_stream_replay and the velocity
// block of a one-stream tick.  Both clusters (32 workgroups each) on ONE XCD, two workgroups per CU -- the schedules count the
// block as one cluster, as without the wavefront.
bool wavefront1_applies(const mp_handle* h, const ModuleW& m, int B) {
    return h->persist && h->wf_ok && h->vec_ok && !use_x3(h, m) && m.H == 256 && m.dirs == 1 && m.whhR[0][0] != nullptr && B == 1 &&
           fp32_slices(h, m, B) == 32;
}

int rnn_rec(const RnnJob& j, int l, hipStream_t s) {
    mp_handle* h = j.h;
    const ModuleW& m = h->mod[j.id];
    ModuleWS& w = j.p->ws[j.id];
    const int B = j.p->B, T = j.p->T, H = m.H, dirs = m.dirs;
    float* out = l == 0 ? w.out0 : w.out1;
    const bool wf32 = wavefront1_applies(h, m, B);
    const bool wf = wf32 || (wavefront_applies(h, m, B, T) && !h->xcd_plan_on[j.id]);
    if (wf && l == 1) return MP_OK;                          // both layers went out with the layer-0 call (below)
    if (h->persist) {
        const int nslab = (B + 15) / 16;
        // every polled word is re-zeroed before every launch: all granules (fp32 kernels) or the flags (split-bf16 kernels)
        // (split-bf16 kernels: the flags of layer 0's area were zeroed by the linear1 GEMM, those of layer 1's area by the layer-0 launch)
        // exact-fp32 kernels: mp_lstm_fused tags its granules with a per-launch epoch base, so the area is zeroed only when
        // something else may have written to it (first use, another kernel family, graph capture -- replays repeat the same
        // base -- or an imminent wrap of the 32-bit tag); the split-bf16 kernels re-arm themselves
        // (not for a 16-slice launch that fills the chip: its 4-wave workgroups can start on CUs where workgroups of the
        //  previous layer launch are still finishing, and their start-up polling slows those down -- measured 3.05 -> 3.28 ms
        //  at 128 x 125; the memset between the launches is the boundary that prevents it.  The 8-slice kernels own their CU.)
        const bool crowded16 = !use_x3(h, m) && dirs == 2 && fp32_slices(h, m, B) == 16 && dirs * nslab * 16 > 128;
        const bool epoch_ok = !use_x3(h, m) && !h->capturing && h->epoch_tags && !crowded16;
        const int nsl = wf32 ? 32 : wf ? 8 : (use_x3(h, m) ? m.nsliceX : fp32_slices(h, m, B));
        const bool p16 = !use_x3(h, m) && nsl == 16 && m.nslice != 16;      // 16-slice packing of a bidirectional block
        const bool p8 = !use_x3(h, m) && nsl == 8 && m.nslice != 8;         // 8-slice packing of the unidirectional block
        // (tagged-word kernels: one tag bit per word, so the area is also zeroed when the other kernel family -- granules with
        //  32-bit epochs -- wrote to it last; the tags a launch starts with follow from what the previous one left: hx_flip)
        const bool tagged = !use_x3(h, m) && H == 256 && (nsl == 8 || nsl == 16);
        unsigned epoch_base = 0;
        // (recovery off: calls are enqueued without a sync, so a launch that lost a workgroup may already have reported it while
        //  this one is being issued -- the words it left behind are not what hx_flip describes: start from a zeroed area.  ADVICE r4)
        if (h->err_host && *(volatile int*)h->err_host) w.hx_epoch = 0;
        if (!use_x3(h, m)) {
            if (!epoch_ok || w.hx_epoch == 0 || w.hx_epoch > 0xf0000000u || w.hx_tagged != tagged) {
                // (a plan's areas are sized for its capacity: this call's slabs are what the launches below can touch)
                const size_t hx_need = (size_t)2 * nslab * ((size_t)4 * 16 * H + 16) * sizeof(unsigned long long);
                HIPCHK(h, hipMemsetAsync(w.hx, 0, hx_need < w.hx_bytes ? hx_need : w.hx_bytes, s));
                w.hx_epoch = epoch_ok ? h->epoch_start : 0u;
                w.hx_flip = 3u;
                w.hx_flipF = 3u;
            }
            w.hx_tagged = tagged;
            epoch_base = epoch_ok ? w.hx_epoch : 0u;
        } else {
            w.hx_epoch = 0;                                   // split-bf16 words in there now
        }
        unsigned long long* hx_l = (use_x3(h, m) && l == 1) ? w.hx2 : w.hx;
        const bool u8 = !use_x3(h, m) && H == 256 && nsl == 32;
        const bool v1 = u8 && B == 1 && h->vec_ok && m.whhR[0][0] != nullptr;   // one sequence: matrix-vector steps (mp_lstm_v1)
        // ... and the H = 64 block of one sequence: a whole direction per workgroup (mp_lstm_v1s).  Only where the H = 256 blocks
        // of this batch run on the 32-slice family too (fp32_slices: 256 CUs, round-robin dispatch where blocks run side by side)
        const bool v1s = !use_x3(h, m) && H == 64 && B == 1 && h->vec_ok && m.whhR[0][0] != nullptr &&
                         fp32_slices(h, h->mod[MP_MOD_VELOCITY], B) == 32;
        const int cus = h->n_cu < 256 ? h->n_cu : 256;
        // slabs per launch: grid <= #CUs, one workgroup per CU
        const int chunk = cus / ((wf ? 2 : dirs) * nsl) > 0 ? cus / ((wf ? 2 : dirs) * nsl) : 1;
        const int kin = l == 0 ? H : dirs * H;
        // timing classes: 1 = H256 bidirectional K_in=256, 4 = H256 bidirectional K_in=512, 5 = H256 unidirectional
        const int cls = H != 256 ? 6 : (dirs == 1 ? 5 : (kin == 256 ? 1 : 4));
        // (a velocity launch that carries the foot-contact layer as a rider is credited with that layer's FLOPs as well)
        // the foot-contact layer that rides in this launch (forward_body decides that one does -- ScheduleScope::rider -- this
        // function which): 16-slice velocity layer l carries foot-contact layer l (rounds 3-4; B <= 128 today); the velocity
        // wavefront carries layer 1, and layer 0 rides in pose layer 0 (8 slices) in front of it (round 5)
        const RnnJob* fj = nullptr;
        int f_layer = 0;
        if (h->vf_foot && !use_x3(h, m) && kin == 256) {
            if (j.id == MP_MOD_VELOCITY && wf && !wf32) { fj = static_cast<const RnnJob*>(h->vf_foot); f_layer = 1; }
            else if (j.id == MP_MOD_VELOCITY && nsl == 16 && !p16) { fj = static_cast<const RnnJob*>(h->vf_foot); f_layer = l; }
            else if (j.id == MP_MOD_POSE && nsl == 8 && l == 0) { fj = static_cast<const RnnJob*>(h->vf_foot); f_layer = 0; }
        }
        const double rider_flop = fj ? 2.0 * 2 * (double)B * T * 4.0 * 64 * ((f_layer == 0 ? 64 : 128) + 64) : 0.0;
        SegScope seg(h, s, cls, (nslab + chunk - 1) / chunk, (wf ? 2.0 : 1.0) * 2.0 * dirs * (double)B * T * 4.0 * H * (kin + H) + rider_flop);
        const float* xin = l == 0 ? x1_buffer(h, m, w) /* X1 */ : w.out0;
        float* outp = l == 0 ? w.out0 : w.out1;
        // layer 1 overwrites out1, which still holds X1 while layer 0 runs -- layer 0 has finished by then
        for (int s0 = 0; s0 < nslab; s0 += chunk) {
            LstmPersistArgs a;
            a.lengths = j.p->lengths_dev; a.ndir = wf ? 2 : dirs; a.B = B; a.T = T;
            a.slab0 = s0; a.nslab = nslab - s0 < chunk ? nslab - s0 : chunk;
            a.hx = hx_l + (size_t)(wf ? 2 : dirs) * s0 * ((size_t)4 * 16 * H + 16);
            a.hx_next = (use_x3(h, m) && l == 0) ? w.hx2 + (size_t)dirs * s0 * ((size_t)4 * 16 * H + 16) : nullptr;
            static const int prof_layer = getenv("MP_PERSIST_PROF_LAYER") ? atoi(getenv("MP_PERSIST_PROF_LAYER")) : -1;
            a.err = h->err_dev; a.max_spin = 1u; a.max_ticks = h->wait_ticks;
            static const int prof_mod = getenv("MP_PERSIST_PROF_MODULE") ? atoi(getenv("MP_PERSIST_PROF_MODULE")) : -1;
            a.prof = ((prof_layer < 0 || prof_layer == l) && (prof_mod < 0 || prof_mod == j.id)) ? h->prof_dev : nullptr;
            a.zero_state = j.mode == STATE_ZERO ? 1 : 0; a.force_remote = h->force_remote ? 1 : 0;
            if (h->dbg_drop_skip > 0) --h->dbg_drop_skip;
            else if (h->dbg_drop_left > 0) { a.debug_drop = h->dbg_drop_block + 1; --h->dbg_drop_left; }
            const bool x3 = use_x3(h, m);
            a.epoch_base = epoch_base;
            a.tag_flip = w.hx_flip;
            a.tag_flip_f = w.hx_flipF;
            a.min_lds = x3 ? 0 : h->excl_lds;
            if (!x3 && h->xcd_plan_on[j.id] && a.nslab == nslab) { mp_fill_xcd_table(a, h->xcd_plan[j.id]); a.xcd_physical = 1; }
            a.out_pairs = x3 ? 1 : 0;                          // both layers feed split-bf16 consumers (layer 1 / linear2)
            for (int d = 0; d < dirs; ++d) {
                LstmDir& dd = a.d[d];
                dd.wpack = x3 ? m.whhX[l][d] : ((v1 || v1s) ? m.whhR[l][d] : u8 ? m.whhU8[l][d] : p8 ? m.whhP8[l][d] : (p16 ? m.whhP16[l][d] : m.whhP[l][d]));
                dd.xproj = nullptr; dd.out = outp + (size_t)d * H;
                const bool inplace = j.out_h == j.in_h && j.out_h;
                dd.hbuf = inplace ? j.out_h + (size_t)(l * dirs + d) * B * H : w.hbuf[l][d];
                dd.hin = dd.hbuf;
                // T = 1 on a carried state: every workgroup reads the whole initial h, nobody waits for anybody in a one-step
                // launch, and the final state goes where the initial one was -- the step-0 operand comes from a copy (the
                // second half of the plan's buffer, which only the per-step kernels use)
                if (T == 1 && j.mode != STATE_ZERO) {
                    HIPCHK(h, hipMemcpyAsync(w.hbuf[l][d] + (size_t)B * H, dd.hbuf, (size_t)B * H * sizeof(float), hipMemcpyDeviceToDevice, s));
                    dd.hin = w.hbuf[l][d] + (size_t)B * H;
                }
                dd.cbuf = inplace ? j.out_c + (size_t)(l * dirs + d) * B * H : w.cbuf[l][d];
                dd.xprojStride = 0; dd.outStride = dirs * H; dd.reverse = d;
                dd.wihpack = x3 ? m.wihX[l][d] : ((v1 || v1s) ? m.wihR[l][d] : u8 ? m.wihU8[l][d] : p8 ? m.wihP8[l][d] : (p16 ? m.wihP16[l][d] : m.wihP[l][d])); dd.bias = m.ih[l].bias + (size_t)d * 4 * H; dd.xin = xin;
            }
            if (dirs == 1) a.d[1] = a.d[0];
            if (wf) {
                // the wavefront: "direction" 1 = layer 1, fed by layer 0's output; in-place state of both layers; clusters
                // (slab, layer) are dealt to the XCDs slab by slab, so that the two layers of a slab share an L2
                LstmDir& d1 = a.d[1];
                const bool inplace = j.out_h == j.in_h && j.out_h;
                d1.wpack = wf32 ? m.whhR[1][0] : m.whhP8[1][0]; d1.wihpack = wf32 ? m.wihR[1][0] : m.wihP8[1][0];
                d1.bias = m.ih[1].bias; d1.xin = w.out0; d1.out = w.out1;
                d1.hbuf = inplace ? j.out_h + (size_t)1 * B * H : w.hbuf[1][0];
                d1.hin = d1.hbuf;
                if (T == 1 && j.mode != STATE_ZERO) {
                    HIPCHK(h, hipMemcpyAsync(w.hbuf[1][0] + (size_t)B * H, d1.hbuf, (size_t)B * H * sizeof(float), hipMemcpyDeviceToDevice, s));
                    d1.hin = w.hbuf[1][0] + (size_t)B * H;
                }
                d1.cbuf = inplace ? j.out_c + (size_t)1 * B * H : w.cbuf[1][0];
                d1.xproj = nullptr; d1.xprojStride = 0; d1.outStride = H; d1.reverse = 0;
                unsigned char cnt[8];
                // (one sequence: both clusters where forward_body's table has the block's one cluster, else on XCD 0)
                for (int x = 0; x < 8; ++x)
                    cnt[x] = (unsigned char)(2 * (wf32 ? (a.xcd_physical ? h->xcd_plan[j.id][x] : (x == 0 ? 1 : 0)) : (a.nslab + 7 - x) / 8));
                mp_fill_xcd_table(a, cnt);
                if (wf32) a.min_lds = 0;                      // (two workgroups per CU are the point)
            }
            if (fj) {
                const ModuleW& fm = h->mod[MP_MOD_FOOT_CONTACT];
                ModuleWS& fws = fj->p->ws[MP_MOD_FOOT_CONTACT];
                for (int fd = 0; fd < 2; ++fd) { a.f_w[fd] = fm.wVF[f_layer][fd]; a.f_bias[fd] = fm.ih[f_layer].bias + (size_t)fd * 4 * fm.H; }
                a.f_xin = f_layer == 0 ? x1_buffer(h, fm, fws) /* X1 */ : fws.out0;
                a.f_out = f_layer == 0 ? fws.out0 : fws.out1;
            }
            const int fk = fj ? (f_layer == 0 ? fm_kin0(h) : 2 * h->mod[MP_MOD_FOOT_CONTACT].H) : 0;
            if (v1) mp_launch_lstm_v1(a, kin, wf32, s);
            else if (v1s) mp_launch_lstm_v1s(a, kin, s);
            else if (wf || (fj && nsl == 8)) {
                if (!mp_launch_lstm_persist8(a, fk, wf, s)) return fail(h, MP_ERR_INVALID, "internal: 8-slice launch (rider %d, wavefront %d) not built", fk, (int)wf);
            } else if (fj) mp_launch_lstm_vf(a, fk, s);
            else if (x3 && nsl == 8 && (h->x3w_mask & (kin == 256 ? 1 : 2))) mp_launch_lstm_x3w(a, kin, s);
            else if (x3) mp_launch_lstm_x3(a, kin, nsl, s);
            else if (u8) mp_launch_lstm_u8(a, kin, s);
            else mp_launch_lstm_persist(a, H, kin, nsl, s);
        }
        if (epoch_base) w.hx_epoch += (unsigned)T + 1u;       // tags base .. base + T are used up
        // parity slot 0 was written (T + 1) / 2 times (steps 0, 2, ...), slot 1 1 + T / 2 times (the initial state as "step -1",
        // then steps 1, 3, ...), tags alternating: an odd count turns the slot's next first tag around
        if (tagged && epoch_ok) {
            const unsigned turn = (unsigned)(((T + 1) / 2) & 1) | ((unsigned)((1 + T / 2) & 1) << 1);
            w.hx_flip ^= turn;
            if (fj) w.hx_flipF ^= turn;                       // (the rider's words: written by this launch only if it carried one)
        }
    } else {
        SegScope seg(h, s, 7, T, 2.0 * dirs * (double)B * T * 4.0 * H * H);
        LstmStepArgs a;
        a.lengths = j.p->lengths_dev; a.ndir = dirs; a.B = B; a.T = T;
        for (int d = 0; d < dirs; ++d)
            a.d[d] = LstmDir{m.whh[l][d], w.xproj + (size_t)d * 4 * H, out + (size_t)d * H, w.hbuf[l][d], w.cbuf[l][d],
                             dirs * 4 * H, dirs * H, d, nullptr, nullptr, nullptr};
        if (dirs == 1) a.d[1] = a.d[0];
        for (int step = 0; step < T; ++step) {
            a.step = step;
            mp_launch_lstm_step(a, H, s);
        }
    }
    HIPCHK(h, hipGetLastError());
    return MP_OK;
}

int rnn_g1(const RnnJob& j, hipStream_t s) {
    mp_handle* h = j.h;
    const ModuleW& m = h->mod[j.id];
    ModuleWS& w = j.p->ws[j.id];
    const int B = j.p->B, T = j.p->T, M = B * T, H = m.H, dirs = m.dirs;
    const RowMap none{nullptr, 0, 0, 0};
    if (!h->persist)
        run_gemm(h, s, internal_map(w.out0, B, dirs * H), none, m.ih[1], w.xproj, dirs * 4 * H, (long)B * dirs * 4 * H, M, B, 0);
    HIPCHK(h, hipGetLastError());
    return MP_OK;
}

int rnn_g2(const RnnJob& j, hipStream_t s) {
    mp_handle* h = j.h;
    const ModuleW& m = h->mod[j.id];
    ModuleWS& w = j.p->ws[j.id];
    const int B = j.p->B, T = j.p->T, M = B * T, H = m.H, dirs = m.dirs;
    const RowMap none{nullptr, 0, 0, 0};
    if (j.out_h && !(h->persist && j.out_h == j.in_h)) {
        const size_t fin = h->persist ? 0 : (size_t)(T & 1) * B * H;   // where the recurrence left h_n
        for (int l = 0; l < 2; ++l)
            for (int d = 0; d < dirs; ++d) {
                const size_t n = (size_t)B * H * sizeof(float);
                const int k = l * dirs + d;
                HIPCHK(h, hipMemcpyAsync(j.out_h + (size_t)k * B * H, w.hbuf[l][d] + fin, n, hipMemcpyDeviceToDevice, s));
                HIPCHK(h, hipMemcpyAsync(j.out_c + (size_t)k * B * H, w.cbuf[l][d], n, hipMemcpyDeviceToDevice, s));
            }
    }
    run_gemm(h, s, internal_map(w.out1, B, dirs * H), none, m.lin2, j.y, j.yStrideB, j.yStrideT, M, B, 0, false,
             use_x3(h, m), use_x3(h, m));                                                          // rnn.py:32
    HIPCHK(h, hipGetLastError());
    return MP_OK;
}

// linear2 of two blocks in ONE launch (mp_launch_gemm_pair: velocity + foot contact at the end of a forward); neither job
// copies state out (in-place / no carried state).  Falls back to two launches when the pair form does not cover the shapes.
int rnn_g2_pair(const RnnJob& j1, const RnnJob& j2, hipStream_t s) {
    mp_handle* h = j1.h;
    auto copies_state = [&](const RnnJob& j) { return j.out_h && !(h->persist && j.out_h == j.in_h); };
    if (!copies_state(j1) && !copies_state(j2) && !use_x3(h, h->mod[j1.id]) && !use_x3(h, h->mod[j2.id])) {
        const int B = j1.p->B, T = j1.p->T, M = B * T;
        auto args = [&](const RnnJob& j) {
            const ModuleW& m = h->mod[j.id];
            GemmArgs g;
            g.a0 = internal_map(j.p->ws[j.id].out1, B, m.dirs * m.H); g.a1 = RowMap{nullptr, 0, 0, 0};
            g.W = m.lin2.W; g.bias = m.lin2.bias; g.C = j.y; g.cStrideB = j.yStrideB; g.cStrideT = j.yStrideT;
            g.M = M; g.N = m.lin2.N; g.K = m.lin2.K; g.Kpad = m.lin2.Kpad; g.B = B; g.relu = 0;
            g.Wf = m.lin2.Wf; g.NB = m.lin2.Wf ? m.lin2.Npad / 32 : 0;
            return g;
        };
        const GemmArgs g1 = args(j1), g2 = args(j2);
        SegScope seg(h, s, 0, 1, 2.0 * M * ((double)g1.N * g1.K + (double)g2.N * g2.K));
        if (mp_launch_gemm_pair(g1, g2, s)) { HIPCHK(h, hipGetLastError()); return MP_OK; }
    }
    if (int rc = rnn_g2(j1, s)) return rc;
    return rnn_g2(j2, s);
}

int run_rnn(const RnnJob& j, hipStream_t s) {
    if (int rc = rnn_g0(j, s)) return rc;
    if (int rc = rnn_rec(j, 0, s)) return rc;
    if (int rc = rnn_g1(j, s)) return rc;
    if (int rc = rnn_rec(j, 1, s)) return rc;
    return rnn_g2(j, s);
}

int ensure_vstate(mp_handle* h, VelState& v, int B) {
    if (v.cap >= B) return MP_OK;
    // captured graphs hold the old buffers' addresses in their kernel arguments; the new `h` buffer can land on the old one's
    // address (the two freed blocks coalesce), which made a stale graph match its key again and write through the freed `c`
    // pointer (found in round 3 by running the whole suite under MP_GRAPH=2): every graph goes when these buffers go
    if (!h->graphs.empty()) {
        HIPCHK(h, hipStreamSynchronize(h->s_main));
        for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second.exec);
        h->graphs.clear();
    }
    if (v.h) (void)hipFree(v.h);
    if (v.c) (void)hipFree(v.c);
    v.h = v.c = nullptr; v.cap = 0;
    if (int rc = dev_alloc(h, (void**)&v.h, (size_t)2 * B * 256 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&v.c, (size_t)2 * B * 256 * sizeof(float))) return rc;
    v.cap = B;
    return MP_OK;
}

// Clusters per XCD for persistent launches that run at the same time.  The dispatcher sends workgroup b to XCD b % 8 and
// lets it wait there when no CU has room, whatever the other XCDs are doing (tools/micro/xcd_dispatch.hip), so "fewer
// workgroups than CUs" is not enough: every XCD must hold its share.  Greedy: widest clusters first, each cluster to the
// XCD with the most CUs left.  `load` (CUs taken per XCD) is updated; false = does not fit (nothing is assigned then).
struct XcdJob { int id, ncl, wgs; };
bool place_clusters(const mp_handle* h, const XcdJob* jobs, int njobs, int load[8], unsigned char cnt[4][8]) {
    const int cap = h->n_cu / 8;
    int ld[8]; unsigned char c[4][8] = {};
    for (int x = 0; x < 8; ++x) ld[x] = load[x];
    int order[4] = {0, 1, 2, 3};
    for (int a = 0; a < njobs; ++a)
        for (int b = a + 1; b < njobs; ++b)
            if (jobs[order[b]].wgs > jobs[order[a]].wgs) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
    for (int a = 0; a < njobs; ++a) {
        const XcdJob& jb = jobs[order[a]];
        for (int k = 0; k < jb.ncl; ++k) {
            int best = 0;
            for (int x = 1; x < 8; ++x) if (ld[x] < ld[best]) best = x;
            if (ld[best] + jb.wgs > cap || c[jb.id][best] == 255) return false;
            ld[best] += jb.wgs; ++c[jb.id][best];
        }
    }
    for (int x = 0; x < 8; ++x) load[x] = ld[x];
    for (int a = 0; a < njobs; ++a) memcpy(cnt[jobs[a].id], c[jobs[a].id], 8);
    return true;
}

// Which side-by-side schedule (forward_body) fits batch B: 0 = none, 1 = pose / velocity / foot contact at once, 2 = the same
// with the pose layers on 8 slices per slab, 3 = pose (8 slices) beside velocity, foot contact after velocity, 4 = pose layer 0
// on 16 slices with the chip to itself, then pose layer 1 on 8 slices beside the velocity layers that carry the foot-contact
// layers as riders.  Fills the per-XCD cluster tables of the three blocks (h->xcd_plan).
int side_by_side_plan(mp_handle* h, int B) {
    if (!h->wide_ok) return 0;
    const ModuleW& pm = h->mod[MP_MOD_POSE];
    const ModuleW& vm = h->mod[MP_MOD_VELOCITY];
    const ModuleW& fm = h->mod[MP_MOD_FOOT_CONTACT];
    const int nslab = (B + 15) / 16;
    const bool any_x3 = use_x3(h, pm) || use_x3(h, vm);
    auto job = [&](int id, const ModuleW& m, int slices) { return XcdJob{id, m.dirs * nslab, slices}; };
    const int pslices = use_x3(h, pm) ? pm.nsliceX : fp32_slices(h, pm, B);
    const int vslices = use_x3(h, vm) ? vm.nsliceX : fp32_slices(h, vm, B);
    XcdJob all[3] = {job(MP_MOD_POSE, pm, pslices), job(MP_MOD_VELOCITY, vm, vslices), job(MP_MOD_FOOT_CONTACT, fm, fm.nslice)};
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (any_x3)                                     // (no tables: the split-bf16 kernels spread their clusters themselves)
        return layer_workgroups(h, pm, B) + layer_workgroups(h, vm, B) + layer_workgroups(h, fm, B) <= h->n_cu ? 1 : 0;
    // exact-fp32 kernels side by side need every cluster placed on an XCD with room for ALL its workgroups (two grids that
    // are each partly resident wait for CUs the other holds until their waits time out): no tables, no side-by-side schedule
    if (!h->xcd_rr || !h->exclusive_ok) return 0;
    if (place_clusters(h, all, 3, load, h->xcd_plan)) return 1;
    if (!h->half_ok || pm.nslice != 8 || pslices != 16) return 0;
    all[0].wgs = pm.nslice;                   // pose on 8 slices per slab (four-wave kernels)
    for (int x = 0; x < 8; ++x) load[x] = 0;
    // schedule 4 wherever it applies (every B that does not fit schedule 1): its two halves take 797 and 2 x 404 us, after 396 us
    // of layer 0 -- against a 560 + 797 us pose chain in schedules 2 and 3
    if (h->late_pair_ok && h->vf_ok && vm.nslice == 16 && vslices == 16 && fm.wVF[0][0] && place_clusters(h, all, 2, load, h->xcd_plan))
        return 4;
    for (int x = 0; x < 8; ++x) load[x] = 0;
    if (place_clusters(h, all, 3, load, h->xcd_plan)) return 2;
    for (int x = 0; x < 8; ++x) load[x] = 0;
    if (!place_clusters(h, all, 2, load, h->xcd_plan)) return 0;          // pose + velocity
    for (int x = 0; x < 8; ++x) load[x] -= h->xcd_plan[MP_MOD_VELOCITY][x] * vslices;   // foot contact takes over velocity's CUs
    return place_clusters(h, all + 2, 1, load, h->xcd_plan) ? 3 : 0;
}

// Workgroups of one persistent layer launch of module m at batch B (every one of them fits a CU of its own)
// models/net.py:101-119 on the library's streams (eager or under capture).
// Stream plan (persistent mode).  The persistent layer kernels are grids of clusters of workgroups that wait on each
// other every step, so two such grids may only run concurrently when ALL their workgroups are resident at once;
// otherwise two partly-resident grids could starve one another (the waits are bounded, so that would end in
// MP_ERR_DEVICE rather than a hang, but it must not happen).
//  * Batches whose pose + velocity + foot-contact launches together need no more workgroups than the device has CUs
//    (B <= 64 with fp32 operands, B <= 128 split-bf16): net.py:106-117 makes the three blocks independent given the
//    joints, so each runs whole on its own stream -- four dependent layer launches deep instead of six
//    (16 x 125: 1.7 -> 1.2 ms; evaluate.py's [1, 3000, 60] call: 33 -> 22 ms).
//  * Larger batches: the joints / pose layers fill the chip (one workgroup per CU, 160 KB of LDS), so all H = 256
//    recurrences are serialised on s_main; the H = 64 foot-contact layers (4 slices per slab, 48 KB of LDS: they fit on
//    a CU beside a velocity workgroup, LDS 80 + 48 KB, or on the half of the chip the split-bf16 velocity layers leave
//    free) run on s_foot beside the velocity layers, the linear2 / IK / FK tail of pose on s_gp (= s_vel, idle by then).
int forward_body(mp_handle* h, Plan* p, const float* imu, float* pose, long poseRows, long poseRowStride,
                 long poseRowOffset, float* joints, float* vel, float* contact, float* r6d, VelState& vs,
                 bool has_state, float* fk_rglobal = nullptr, float* fk_joint = nullptr, bool* tail_pending = nullptr) {
    const int T = p->T;
    if (tail_pending) *tail_pending = false;
    const RowMap none{nullptr, 0, 0, 0};
    const RowMap xj = user_map(joints, T, 72), xi = user_map(imu, T, 60);
    RnnJob J{h, p, MP_MOD_JOINTS, xi, none, joints, (long)T * 72, 72, STATE_ZERO, nullptr, nullptr, nullptr, nullptr};
    RnnJob P{h, p, MP_MOD_POSE, xj, xi, r6d, (long)T * 96, 96, STATE_ZERO, nullptr, nullptr, nullptr, nullptr};
    RnnJob V{h, p, MP_MOD_VELOCITY, xj, xi, vel, (long)T * 72, 72, has_state ? STATE_FROM : STATE_ZERO, vs.h, vs.c, vs.h, vs.c};
    RnnJob F{h, p, MP_MOD_FOOT_CONTACT, xj, xi, contact, (long)T * 2, 2, STATE_ZERO, nullptr, nullptr, nullptr, nullptr};
    // (graph mode 2: a single-branch graph -- every launch is captured on s_main, in an order that respects all the
    //  dependencies below; the event record / wait pairs between "streams" become same-stream no-ops)
    const bool one_branch = h->capturing && h->graph_serial;
    hipStream_t sm = h->s_main, sp = one_branch ? sm : h->s_gp, sv = one_branch ? sm : h->s_vel, sf = one_branch ? sm : h->s_foot;
#define RC(x) do { if (int rc_ = (x)) return rc_; } while (0)
    // ---- the default schedule of full batches (B > 128, exact-fp32 operands), round 4: ONE stream for everything but pose's
    // linear2 / IK / FK tail.  joints block -> linear1 of pose | velocity | foot contact as ONE GEMM with three outputs -> pose
    // layers -> velocity layers with the foot-contact layers riding in their workgroups -> linear2 of velocity and foot contact
    // as ONE launch.  Round 3 ran foot contact's two linear layers on a stream of their own: four cross-stream edges on the
    // critical chain (9-16 us of barrier packets each in the rocprof timeline: 45 us per forward) and a linear1 that ran beside
    // the stacked one and slowed it down (90 vs 79 us).  The tail's join is left to the caller when it asks for that
    // (tail_pending): the translation solver does not read the pose.
    {
        const ModuleW& vmod = h->mod[MP_MOD_VELOCITY];
        const bool vf = h->vf_ok && p->B > 128 && h->persist && !use_x3(h, vmod) && vmod.nslice == 16 &&
                        fp32_slices(h, vmod, p->B) == 16 && h->mod[MP_MOD_FOOT_CONTACT].wVF[0][0] != nullptr;
        if (vf && h->one_stream_ok && h->lin1_pvf.Wf && !use_x3(h, h->mod[MP_MOD_POSE]) &&
            side_by_side_plan(h, p->B) == 0) {
            RC(rnn_g0(J, sm)); RC(rnn_rec(J, 0, sm)); RC(rnn_g1(J, sm)); RC(rnn_rec(J, 1, sm));   // net.py:103
            int rc_pv = MP_OK;
            if (!rnn_g2_g0_fused(J, P, V, F, sm, &rc_pv)) {         // (linear2 of joints + the stacked linear1: one launch, else two)
                RC(rnn_g2(J, sm));
                if (!rnn_g0_pose_velocity(P, V, sm, &rc_pv, &F)) return fail(h, MP_ERR_INVALID, "internal: stacked linear1 refused");
            }
            RC(rc_pv);
            // round 5: the velocity layers as ONE two-layer wavefront launch of the 8-slice kernel (wavefront_applies); the
            // foot-contact layers ride in pose layer 0 (its layer 0: both directions must be complete before its layer 1
            // starts) and in the wavefront launch (its layer 1).  Every CU is then taken by four 512-register waves from the
            // joints block to the end of the velocity block, so pose's linear2 / IK / FK tail can no longer run beside the
            // velocity layers: it forks off BEHIND them and runs beside velocity's / foot contact's linear2 and the solver.
            const bool wfv = wavefront_applies(h, vmod, p->B, p->T);
            {
                ScheduleScope sched(h);
                if (wfv) sched.rider(&F);
                RC(rnn_rec(P, 0, sm));                                                      // net.py:106-107
            }
            RC(rnn_rec(P, 1, sm));
            if (!wfv) HIPCHK(h, hipEventRecord(h->ev_x[2], sm));
            {
                ScheduleScope sched(h);
                sched.rider(&F);
                RC(rnn_rec(V, 0, sm));                                                      // net.py:113-117
                RC(rnn_rec(V, 1, sm));
            }
            // (measured and dropped in round 5: linear2 of all three blocks as ONE launch with K split over wave pairs, two waves per
            //  SIMD -- 85 us against ~65 us for the two launches side by side on two streams, 3.662 vs 3.636 ms per step;
            //  profiles/NOTES_r05.md)
            if (wfv) HIPCHK(h, hipEventRecord(h->ev_x[2], sm));
            RC(rnn_g2_pair(V, F, sm));
            HIPCHK(h, hipStreamWaitEvent(sp, h->ev_x[2], 0));
            RC(rnn_g2(P, sp));
            {   // net.py:110 (+ articulate/model.py:208-232 when the caller wants the FK outputs: one launch for both)
                SegScope seg(h, sp, 2, 1);
                if (!(fk_rglobal && mp_launch_r6d_ik_fk(r6d, poseRows, poseRowStride, poseRowOffset, pose, h->bone_dev, h->parent_dev,
                                                        fk_rglobal, fk_joint, sp))) {
                    mp_launch_r6d_ik_strided(r6d, poseRows, poseRowStride, poseRowOffset, pose, h->parent_dev, sp);
                    if (fk_rglobal) mp_launch_fk(pose, nullptr, poseRows, h->bone_dev, h->parent_dev, h->depth_dev, fk_rglobal, fk_joint, sp);
                }
            }
            HIPCHK(h, hipEventRecord(h->ev_x[3], sp));
            if (tail_pending) *tail_pending = true;
            else HIPCHK(h, hipStreamWaitEvent(sm, h->ev_x[3], 0));
            HIPCHK(h, hipGetLastError());
            return MP_OK;
        }
    }
    // joints(batch)                                                                       net.py:103
    // (round 6: schedule 4 -- 64 < B <= 128, the share of one GPU in eight of configs[3] -- takes the joints -> pose seam as the
    //  ONE launch the full-batch schedule uses, joints.linear2 + the stacked linear1 of pose | velocity | foot contact, instead of
    //  three launches on two streams)
    bool seam_fused = false;
    if (h->persist && side_by_side_plan(h, p->B) == 4) {
        RC(rnn_g0(J, sm)); RC(rnn_rec(J, 0, sm)); RC(rnn_g1(J, sm)); RC(rnn_rec(J, 1, sm));
        int rc_f = MP_OK;
        seam_fused = rnn_g2_g0_fused(J, P, V, F, sm, &rc_f);
        RC(rc_f);
        if (!seam_fused) RC(rnn_g2(J, sm));
    } else {
        RC(run_rnn(J, sm));
    }
    HIPCHK(h, hipEventRecord(h->ev_j, sm));
    HIPCHK(h, hipStreamWaitEvent(sf, h->ev_j, 0));
    if (!h->persist) {
        HIPCHK(h, hipStreamWaitEvent(sv, h->ev_j, 0));
        // per-step kernels have no cross-workgroup waits: the three remaining blocks simply run side by side
        RC(run_rnn(F, sf));                                                               // net.py:113-114
        HIPCHK(h, hipEventRecord(h->ev_f, sf));
        RC(run_rnn(P, sm));                                                               // net.py:106-107
        { SegScope seg(h, sm, 2, 1);
          mp_launch_r6d_ik_strided(r6d, poseRows, poseRowStride, poseRowOffset, pose, h->parent_dev, sm); }   // net.py:110
        if (fk_rglobal) mp_launch_fk(pose, nullptr, poseRows, h->bone_dev, h->parent_dev, h->depth_dev, fk_rglobal, fk_joint, sm);
        RC(run_rnn(V, sv));                                                               // net.py:117
        HIPCHK(h, hipEventRecord(h->ev_v, sv));
    } else if (const int side = side_by_side_plan(h, p->B)) {
        // the blocks side by side: every workgroup of the concurrent layer launches has a CU of its own -- and gets one: the
        // exact-fp32 launches ask for more than half a CU's LDS, so the dispatcher cannot put two persistent workgroups on
        // one CU while others stand empty (it spreads every launch on its own, and a workgroup that shares its SIMDs slows
        // its whole lock-stepped cluster).
        //   side 1 (B <= 64 fp32, <= 128 split-bf16): pose, velocity and foot contact at once;
        //   side 2 (B <= 96, fp32): the same with the pose layers on 8 slices per slab (the four-wave kernels, 16 CUs per
        //           slab and direction instead of 32): a longer pose chain (1.38 instead of 0.85 ms), but nothing after it;
        //   side 3 (B <= 128, fp32): pose on 8 slices beside velocity; foot contact follows velocity on the CUs it vacates.
        HIPCHK(h, hipStreamWaitEvent(sv, h->ev_j, 0));
        if (side == 4) {
            //   side 4 (64 < B <= 128, fp32): pose layer 0 on 16 slices with the chip to itself (as the joints layers), then pose
            //           layer 1 on 8 slices (the four-wave kernel: half of the CUs) on s_main beside velocity layer 0 -> 1 on s_vel,
            //           the foot-contact layers riding in the velocity workgroups ("VF"); every cluster placed by the tables
            auto rec = [&](int i, hipStream_t on) -> int { HIPCHK(h, hipEventRecord(h->ev_x[i], on)); return MP_OK; };
            auto wait = [&](int i, hipStream_t on) -> int { HIPCHK(h, hipStreamWaitEvent(on, h->ev_x[i], 0)); return MP_OK; };
            // (two streams only: foot contact's linear layers go where its recurrent layers run, on s_vel)
            if (!seam_fused) {
                RC(rnn_g0(F, sv));                                                        // linear1 of foot contact
                int rc_pv = MP_OK;
                if (!rnn_g0_pose_velocity(P, V, sm, &rc_pv)) { RC(rnn_g0(V, sm)); RC(rnn_g0(P, sm)); }
                RC(rc_pv);
            }
            RC(rnn_rec(P, 0, sm));                                                        // 16 slices, every CU
            RC(rec(1, sm));
            RC(wait(1, sv));                                                              // (the velocity grid must not start under it)
            {
                ScheduleScope sched(h);
                sched.exclusive_lds(kExclusiveLdsBytes).pose_on_8_slices(true).tables(MP_MOD_POSE, true).tables(MP_MOD_VELOCITY, true).rider(&F);
                RC(rnn_rec(V, 0, sv));                                                    // net.py:113-117
                RC(rnn_rec(P, 1, sm));                                                    // net.py:106-107
                RC(rnn_rec(V, 1, sv));
            }
            RC(rnn_g2_pair(V, F, sv));                                                    // net.py:117, 113-114: one launch
            HIPCHK(h, hipEventRecord(h->ev_v, sv));
            HIPCHK(h, hipEventRecord(h->ev_f, sv));
            RC(rec(4, sf)); RC(wait(4, sm));                // (s_foot was forked into the call above and gets no work here: join it)
            RC(rnn_g2(P, sm));
            { SegScope seg(h, sm, 2, 1);
              mp_launch_r6d_ik_strided(r6d, poseRows, poseRowStride, poseRowOffset, pose, h->parent_dev, sm); }   // net.py:110
            if (fk_rglobal) mp_launch_fk(pose, nullptr, poseRows, h->bone_dev, h->parent_dev, h->depth_dev, fk_rglobal, fk_joint, sm);
        } else {
        {
        ScheduleScope sched(h);
        const bool tables = h->exclusive_ok && h->xcd_rr && !use_x3(h, h->mod[MP_MOD_POSE]) && !use_x3(h, h->mod[MP_MOD_VELOCITY]);
        sched.exclusive_lds(h->exclusive_ok ? kExclusiveLdsBytes : 0).pose_on_8_slices(side >= 2)
             .tables(MP_MOD_POSE, tables).tables(MP_MOD_VELOCITY, tables).tables(MP_MOD_FOOT_CONTACT, tables);
        if (side == 3) {
            RC(rnn_g0(F, sf));                                                            // linear1 right away
            RC(run_rnn(V, sv));                                                           // net.py:117
            HIPCHK(h, hipEventRecord(h->ev_v, sv));
            HIPCHK(h, hipStreamWaitEvent(sf, h->ev_v, 0));
            RC(rnn_rec(F, 0, sf));                                                        // net.py:113-114
            RC(rnn_g1(F, sf));
            RC(rnn_rec(F, 1, sf));
            RC(rnn_g2(F, sf));
            HIPCHK(h, hipEventRecord(h->ev_f, sf));
        } else {
            RC(run_rnn(F, sf));                                                           // net.py:113-114
            HIPCHK(h, hipEventRecord(h->ev_f, sf));
            RC(run_rnn(V, sv));                                                           // net.py:117
            HIPCHK(h, hipEventRecord(h->ev_v, sv));
        }
        RC(run_rnn(P, sm));                                                               // net.py:106-107
        }
        { SegScope seg(h, sm, 2, 1);
          mp_launch_r6d_ik_strided(r6d, poseRows, poseRowStride, poseRowOffset, pose, h->parent_dev, sm); }   // net.py:110
        if (fk_rglobal) mp_launch_fk(pose, nullptr, poseRows, h->bone_dev, h->parent_dev, h->depth_dev, fk_rglobal, fk_joint, sm);
        }
    } else {
        auto rec = [&](int i, hipStream_t on) -> int { HIPCHK(h, hipEventRecord(h->ev_x[i], on)); return MP_OK; };
        auto wait = [&](int i, hipStream_t on) -> int { HIPCHK(h, hipStreamWaitEvent(on, h->ev_x[i], 0)); return MP_OK; };
        // (every cross-stream edge into a node of the critical chain costs 8-20 us of graph dependency resolution, so the
        //  chain joints -> pose linear1 -> pose layers -> velocity layers -> velocity linear2 stays on s_main and the one
        //  edge it needs from a side stream -- velocity's linear1 -- is taken early, in front of the pose layers)
        RC(rnn_g0(F, sf));                                       // linear1 of the three blocks, concurrently
        // "VF": the foot-contact layers ride in the workgroups of the velocity layer launches (mp_lstm_fused<256,16,256,1,*,FK>)
        // instead of running as launches of their own beside them -- exact-fp32 16-slice velocity kernel, zero initial state
        const ModuleW& vmod = h->mod[MP_MOD_VELOCITY];
        const bool fuse_vf = h->vf_ok && p->B > 128 && h->persist && !use_x3(h, vmod) && vmod.nslice == 16 &&
                             fp32_slices(h, vmod, p->B) == 16 && h->mod[MP_MOD_FOOT_CONTACT].wVF[0][0] != nullptr;
        if (fuse_vf) RC(rec(4, sf));                             // linear1 of foot contact is done
        int rc_pv = MP_OK;
        const bool fused_pv = rnn_g0_pose_velocity(P, V, sm, &rc_pv);   // pose + velocity: one GEMM on the main stream
        RC(rc_pv);
        if (!fused_pv) {                                         // (s_vel is only forked into the call when it gets work)
            HIPCHK(h, hipStreamWaitEvent(sv, h->ev_j, 0));
            RC(rnn_g0(V, sv)); RC(rec(1, sv));
            RC(rnn_g0(P, sm));
        }
        // (when the velocity block below goes out as the two-layer wavefront, that launch carries foot-contact layer 1
        //  only -- layer 0 has to ride in pose layer 0, as in the one-stream schedule; before round 6 this branch never asked for
        //  it and foot contact's layer 1 read a stale out0)
        const bool wfv = fuse_vf && wavefront_applies(h, vmod, p->B, p->T);
        if (wfv) RC(wait(4, sm));                                // the rider reads foot contact's X1
        {
            ScheduleScope sched(h);
            if (wfv) sched.rider(&F);
            RC(rnn_rec(P, 0, sm));                                                          // net.py:106-107
        }
        RC(rnn_rec(P, 1, sm)); RC(rec(2, sm));
        // (captured BEFORE the side-stream work that hangs off the same event: the graph launches the successors of a
        //  node in creation order, and the velocity layers are the critical chain)
        if (!fused_pv) RC(wait(1, sm));
        // velocity and foot contact run side by side: when together they need no more workgroups than there are CUs
        // (B <= 128) each workgroup gets a CU of its own (see the side-by-side schedule above)
        int excl_vf = 0;
        bool vf_tables = false;
        // (without placement tables velocity and foot contact share CUs -- 80 + 48 KB of LDS, registers to match: a
        //  velocity and a foot-contact workgroup fit on one CU together, so both grids are always fully resident)
        if (h->exclusive_ok && h->xcd_rr && !use_x3(h, h->mod[MP_MOD_VELOCITY])) {
            const int nslab = (p->B + 15) / 16;
            const XcdJob vf[2] = {{MP_MOD_VELOCITY, h->mod[MP_MOD_VELOCITY].dirs * nslab, fp32_slices(h, h->mod[MP_MOD_VELOCITY], p->B)},
                                  {MP_MOD_FOOT_CONTACT, h->mod[MP_MOD_FOOT_CONTACT].dirs * nslab, h->mod[MP_MOD_FOOT_CONTACT].nslice}};
            int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (place_clusters(h, vf, 2, load, h->xcd_plan)) { excl_vf = kExclusiveLdsBytes; vf_tables = true; }
        }
        if (fuse_vf) { excl_vf = 0; vf_tables = false; if (!wfv) RC(wait(4, sm)); }
        {
            ScheduleScope sched(h);
            sched.exclusive_lds(excl_vf).tables(MP_MOD_VELOCITY, vf_tables).rider(fuse_vf ? &F : nullptr);
            RC(rnn_rec(V, 0, sm));
            RC(rnn_rec(V, 1, sm));
        }
        if (fuse_vf) RC(rec(5, sm));
        RC(rnn_g2(V, sm));                                                                  // net.py:117
        HIPCHK(h, hipEventRecord(h->ev_v, sm));
        RC(wait(2, sp)); RC(rnn_g2(P, sp));
        { SegScope seg(h, sp, 2, 1);
          mp_launch_r6d_ik_strided(r6d, poseRows, poseRowStride, poseRowOffset, pose, h->parent_dev, sp); }   // net.py:110
        if (fk_rglobal) mp_launch_fk(pose, nullptr, poseRows, h->bone_dev, h->parent_dev, h->depth_dev, fk_rglobal, fk_joint, sp);
        RC(rec(3, sp));
        if (fuse_vf) {                                   // both foot-contact layers ran inside the velocity launches
            RC(wait(5, sf));
            RC(rnn_g2(F, sf));                                                              // net.py:113-114
        } else {
            RC(wait(2, sf));
            {
                ScheduleScope sched(h);
                sched.exclusive_lds(excl_vf).tables(MP_MOD_FOOT_CONTACT, vf_tables);
                RC(rnn_rec(F, 0, sf));
                RC(rnn_rec(F, 1, sf));
            }
            RC(rnn_g2(F, sf));                                                              // net.py:113-114
        }
        HIPCHK(h, hipEventRecord(h->ev_f, sf));
        RC(wait(3, sm));
    }
#undef RC
    HIPCHK(h, hipStreamWaitEvent(sm, h->ev_v, 0));
    HIPCHK(h, hipStreamWaitEvent(sm, h->ev_f, 0));
    HIPCHK(h, hipGetLastError());
    return MP_OK;
}

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
int snapshot_vstate(mp_handle* h, int B, bool has_state, CopyJobs* more = nullptr) {
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

}  // namespace

// ================================================================================================ C ABI
extern "C" {

size_t mp_weight_count(void) { return manifest_floats(); }

#ifndef MP_SRC_MD5
#define MP_SRC_MD5 "unknown"
#endif
// (the marker in front lets __graft_entry__._needs_build find the id in the file without loading it)
const char* mp_build_id(void) { static const char id[] = "MP_BUILD_ID=" MP_SRC_MD5; return id + 12; }

int mp_manifest_entry(int i, char* name, size_t name_cap, int* ndim, int64_t shape[2], size_t* offset) {
    const std::vector<Entry>& m = manifest();
    if (i < 0 || i >= (int)m.size()) return MP_ERR_INVALID;
    const Entry& e = m[i];
    if (name && name_cap) { strncpy(name, e.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (ndim) *ndim = e.ndim;
    if (shape) { shape[0] = e.shape[0]; shape[1] = e.ndim == 2 ? e.shape[1] : 0; }
    if (offset) *offset = e.offset;
    return MP_OK;
}

int mp_create(mp_handle** out, int device, const float* weights_host, size_t n_floats, const int32_t parent[24],
              const float J[72]) {
    return create_common(out, device, weights_host, false, n_floats, parent, J);
}

int mp_create_from_device(mp_handle** out, int device, const float* weights_dev, size_t n_floats,
                          const int32_t parent[24], const float J[72]) {
    return create_common(out, device, weights_dev, true, n_floats, parent, J);
}

int mp_create_body(mp_handle** out, int device, const int32_t parent[24], const float J[72]) {
    return create_common(out, device, nullptr, false, 0, parent, J);
}

void mp_destroy(mp_handle* h) {
    if (!h) return;
    DeviceScope on_device(h->device);
    (void)hipDeviceSynchronize();
    if (device_error_pending(h))                        // nobody asked (mp_finish / mp_device_error / a later call): say it
        fprintf(stderr, "libmobileposer_hip: mp_destroy: an unreported device error was pending (code %d): a persistent LSTM "
                        "kernel gave up a wait; the affected outputs of that call were NaN\n", take_device_error(h, nullptr));
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second.exec);
    for (Plan* q : h->plans) {
        for (void* p : q->allocs) (void)hipFree(p);
        if (q->lengths_pin) (void)hipHostFree(q->lengths_pin);
        delete q;
    }
    h->plans.clear();
    for (ModuleW& m : h->mod) {
        Packed* ps[4] = {&m.lin1, &m.ih[0], &m.ih[1], &m.lin2};
        for (Packed* p : ps) { if (p->W) (void)hipFree(p->W); if (p->bias) (void)hipFree(p->bias); if (p->Wp) (void)hipFree(p->Wp); if (p->Wf) (void)hipFree(p->Wf); }
        for (int l = 0; l < 2; ++l) for (int d = 0; d < 2; ++d) {
            if (m.whh[l][d]) (void)hipFree(m.whh[l][d]);
            if (m.whhP[l][d]) (void)hipFree(m.whhP[l][d]);
            if (m.wihP[l][d]) (void)hipFree(m.wihP[l][d]);
            if (m.whhP8[l][d]) (void)hipFree(m.whhP8[l][d]);
            if (m.wihP8[l][d]) (void)hipFree(m.wihP8[l][d]);
            if (m.whhU8[l][d]) (void)hipFree(m.whhU8[l][d]);
            if (m.wihU8[l][d]) (void)hipFree(m.wihU8[l][d]);
            if (m.whhR[l][d]) (void)hipFree(m.whhR[l][d]);
            if (m.wihR[l][d]) (void)hipFree(m.wihR[l][d]);
            if (m.whhP16[l][d]) (void)hipFree(m.whhP16[l][d]);
            if (m.wihP16[l][d]) (void)hipFree(m.wihP16[l][d]);
            if (m.whhX[l][d]) (void)hipFree(m.whhX[l][d]);
            if (m.wVF[l][d]) (void)hipFree(m.wVF[l][d]);
            if (m.wihX[l][d]) (void)hipFree(m.wihX[l][d]);
        }
    }
    void* misc[] = {h->parent_dev, h->depth_dev, h->bone_dev, h->vstate.h, h->vstate.c, h->sc.window, h->sc.replay_ws, h->sc.fresh,
                    h->sc.mask_dev, h->sc.st.last_foot, h->sc.st.root_y, h->sc.st.root_pos, h->sc.joints, h->sc.vel,
                    h->sc.contact, h->jrest_dev, h->vrest_dev, h->skinw_dev, h->lin1_pv.Wp, h->lin1_pv.W, h->lin1_pv.Wf, h->lin1_pv.bias, h->lin1_pvf.W, h->lin1_pvf.Wf, h->lin1_pvf.bias, h->prof_dev,
                    h->vtpl_dev, h->shapedirs_dev, h->jreg_dev, h->shape_ws, h->posedirsT_dev, h->rnn_snap,
                    h->vsnap.h, h->vsnap.c, h->st_snap.last_foot, h->st_snap.root_y, h->st_snap.root_pos, h->eval_ws};
    for (void* p : misc) if (p) (void)hipFree(p);
    if (h->err_host) (void)hipHostFree(h->err_host);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    hipEvent_t evs[5] = {h->ev_in, h->ev_out, h->ev_j, h->ev_v, h->ev_f};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_x) if (e) (void)hipEventDestroy(e);
    hipStream_t ss[3] = {h->s_main, h->s_vel, h->s_foot};             // (s_gp is s_vel)
    for (hipStream_t s : ss) if (s) (void)hipStreamDestroy(s);
    delete h;
}

const char* mp_last_error(const mp_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mp_device_info(const mp_handle* h, int* device, int* n_cu, int* xcd_round_robin) {
    if (!h) return MP_ERR_INVALID;
    if (device) *device = h->device;
    if (n_cu) *n_cu = h->n_cu;
    if (xcd_round_robin) *xcd_round_robin = (h->xcd_probe ? 1 : 0) | (h->xcd_rr ? 2 : 0);   // bit 0: probed at creation, bit 1: tables still in use
    return MP_OK;
}

int mp_get_constants(const mp_handle* h, float* floor_y, float feet_pos[6]) {
    if (!h) return MP_ERR_INVALID;
    if (floor_y) *floor_y = h->floor_y;
    if (feet_pos) memcpy(feet_pos, h->feet_pos, sizeof(h->feet_pos));
    return MP_OK;
}

int mp_forward(mp_handle* h, const float* imu_dev, const int32_t* lengths_host, int B, int T, float* pose_dev,
               float* joints_dev, float* vel_dev, float* contact_dev, float* r6d_dev, void* stream) {
    if (!h) return MP_ERR_INVALID;
    if (int rc = need_weights(h, "mp_forward")) return rc;
    if (!imu_dev || !lengths_host || !pose_dev || !joints_dev || !vel_dev || !contact_dev || B < 1 || T < 1)
        return fail(h, MP_ERR_INVALID, "mp_forward: NULL buffer or non-positive shape");
    if (h->vstate.B != 0 && h->vstate.B != B)
        return fail(h, MP_ERR_STATE_SHAPE, "carried velocity state has batch %d, call has batch %d (reset it first; "
                    "the reference raises here too, velocity.py:45-48)", h->vstate.B, B);
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, B, T, &p)) return rc;
    if (int rc = upload_lengths(h, p, lengths_host)) return rc;
    if (int rc = ensure_vstate(h, h->vstate, B)) return rc;
    const bool has_state = h->vstate.B == B;
    float* r6d = r6d_dev ? r6d_dev : p->r6d;
    h->segs.clear(); h->ev_used = 0;
    if (int rc = snapshot_vstate(h, B, has_state)) return rc;
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.kind = 0; key.B = B; key.T = T; key.flags = (has_state ? 1 : 0) | (h->persist ? 2 : 0) | (h->x3 ? 8 : 0);
    key.p[0] = imu_dev; key.p[1] = pose_dev; key.p[2] = joints_dev; key.p[3] = vel_dev; key.p[4] = contact_dev;
    key.p[5] = r6d; key.p[6] = h->vstate.h;
    auto body = [&]() {
        return forward_body(h, p, imu_dev, pose_dev, (long)B * T, 96, 0, joints_dev, vel_dev, contact_dev, r6d,
                            h->vstate, has_state);
    };
    int rc;
    {
        SegScope whole(h, h->s_main, 3, 1);
        rc = run_maybe_graph(h, key, body);
    }
    if (rc) return rc;
    if (int rc2 = finish_or_recover(h, p, "mp_forward", [&]() { return restore_vstate(h, B, has_state); }, body)) return rc2;
    h->vstate.B = B;
    return leave(h, stream);
}

int mp_forward_offline(mp_handle* h, const float* imu_dev, const int32_t* lengths_host, int B, int T, float* pose_dev,
                       float* joints_dev, float* vel_dev, float* contact_dev, float* tran_dev, float* rglobal_dev,
                       float* joint_dev, void* stream) {
    if (!h) return MP_ERR_INVALID;
    if (int rc = need_weights(h, "mp_forward_offline")) return rc;
    if (!imu_dev || !lengths_host || !pose_dev || !joints_dev || !vel_dev || !contact_dev || !tran_dev || B < 1 || T < 1 ||
        ((rglobal_dev == nullptr) != (joint_dev == nullptr)))
        return fail(h, MP_ERR_INVALID, "mp_forward_offline: NULL buffer or non-positive shape");
    if (h->vstate.B != 0 && h->vstate.B != B)
        return fail(h, MP_ERR_STATE_SHAPE, "carried velocity state has batch %d, call has batch %d", h->vstate.B, B);
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, B, T, &p)) return rc;
    if (int rc = upload_lengths(h, p, lengths_host)) return rc;
    if (int rc = ensure_vstate(h, h->vstate, B)) return rc;
    const bool has_state = h->vstate.B == B;
    h->segs.clear(); h->ev_used = 0;
    if (int rc = snapshot_vstate(h, B, has_state)) return rc;
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.kind = 2; key.B = B; key.T = T; key.flags = (has_state ? 1 : 0) | (h->persist ? 2 : 0) | (h->x3 ? 8 : 0);
    key.p[0] = imu_dev; key.p[1] = pose_dev; key.p[2] = joints_dev; key.p[3] = vel_dev; key.p[4] = contact_dev;
    key.p[5] = tran_dev; key.p[6] = h->vstate.h; key.p[7] = rglobal_dev;
    auto body = [&]() {
        bool tail = false;                 // pose's linear2 / IK / FK still running on the side stream (the solver does not read them)
        if (int r = forward_body(h, p, imu_dev, pose_dev, (long)B * T, 96, 0, joints_dev, vel_dev, contact_dev, p->r6d,
                                 h->vstate, has_state, rglobal_dev, joint_dev, &tail)) return r;
        mp_launch_translate_offline(joints_dev, vel_dev, contact_dev, p->lengths_dev, B, T, h->floor_y, tran_dev,
                                    h->s_main);                                       // net.py:130-154
        if (tail) HIPCHK(h, hipStreamWaitEvent(h->s_main, h->ev_x[3], 0));
        HIPCHK(h, hipGetLastError());
        return (int)MP_OK;
    };
    int rc;
    {
        SegScope whole(h, h->s_main, 3, 1);
        rc = run_maybe_graph(h, key, body);
    }
    if (rc) return rc;
    if (int rc2 = finish_or_recover(h, p, "mp_forward_offline", [&]() { return restore_vstate(h, B, has_state); }, body)) return rc2;
    h->vstate.B = B;
    return leave(h, stream);
}

int mp_rnn_forward(mp_handle* h, int module, const float* x_dev, const int32_t* lengths_host, int B, int T,
                   float* y_dev, const float* state_in_dev, float* state_out_dev, void* stream) {
    if (!h) return MP_ERR_INVALID;
    if (int rc = need_weights(h, "mp_rnn_forward")) return rc;
    if (module < 0 || module > 3 || !x_dev || !y_dev || !lengths_host || B < 1 || T < 1)
        return fail(h, MP_ERR_INVALID, "mp_rnn_forward: bad argument");
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, B, T, &p)) return rc;
    if (int rc = upload_lengths(h, p, lengths_host)) return rc;
    const ModuleW& m = h->mod[module];
    const size_t half = (size_t)2 * m.dirs * B * m.H;
    const RowMap none{nullptr, 0, 0, 0};
    h->segs.clear(); h->ev_used = 0;
    RnnJob job{h, p, module, user_map(x_dev, T, m.n_in), none, y_dev, (long)T * m.n_out, m.n_out,
               state_in_dev ? STATE_FROM : STATE_ZERO, state_in_dev, state_in_dev ? state_in_dev + half : nullptr,
               state_out_dev, state_out_dev ? state_out_dev + half : nullptr};
    // state_in_dev == state_out_dev: the fused kernels update the state in place, so a starved run would leave NaN where the
    // repair has to start from -- keep a copy for the restore (recovery on only)
    const bool aliased = state_in_dev && state_in_dev == state_out_dev;
    const size_t state_bytes = 2 * half * sizeof(float);
    if (aliased && h->recovery) {
        if (state_bytes > h->rnn_snap_bytes) {
            HIPCHK(h, hipStreamSynchronize(h->s_main));
            if (h->rnn_snap) (void)hipFree(h->rnn_snap);
            h->rnn_snap = nullptr; h->rnn_snap_bytes = 0;
            if (int rc = dev_alloc(h, (void**)&h->rnn_snap, state_bytes)) return rc;
            h->rnn_snap_bytes = state_bytes;
        }
        HIPCHK(h, hipMemcpyAsync(h->rnn_snap, state_in_dev, state_bytes, hipMemcpyDeviceToDevice, h->s_main));
    }
    int rc = run_rnn(job, h->s_main);
    if (rc) return rc;
    if (int rc2 = finish_or_recover(h, p, "mp_rnn_forward", [&]() {
            if (aliased) HIPCHK(h, hipMemcpyAsync(state_out_dev, h->rnn_snap, state_bytes, hipMemcpyDeviceToDevice, h->s_main));
            return (int)MP_OK;
        }, [&]() { return run_rnn(job, h->s_main); })) return rc2;
    return leave(h, stream);
}

int mp_reduced_global_to_full(mp_handle* h, const float* r6d_dev, int64_t N, float* pose_dev, void* stream) {
    if (!h || !r6d_dev || !pose_dev || N < 0) return h ? fail(h, MP_ERR_INVALID, "mp_reduced_global_to_full: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    mp_launch_r6d_ik(r6d_dev, (long)N, pose_dev, h->parent_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_r6d_to_rotation_matrix(mp_handle* h, const float* r6d_dev, int64_t n, float* rot_dev, void* stream) {
    if (!h || !r6d_dev || !rot_dev || n < 0) return h ? fail(h, MP_ERR_INVALID, "mp_r6d_to_rotation_matrix: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    mp_launch_r6d_to_rot(r6d_dev, (long)n, rot_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_translate_offline(mp_handle* h, const float* joints_dev, const float* vel_dev, const float* contact_dev,
                         const int32_t* lengths_host, int B, int T, float* tran_dev, void* stream) {
    if (!h || !joints_dev || !vel_dev || !contact_dev || !lengths_host || !tran_dev || B < 1 || T < 1)
        return h ? fail(h, MP_ERR_INVALID, "mp_translate_offline: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, B, T, &p)) return rc;
    if (int rc = upload_lengths(h, p, lengths_host)) return rc;
    mp_launch_translate_offline(joints_dev, vel_dev, contact_dev, p->lengths_dev, B, T, h->floor_y, tran_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_fk(mp_handle* h, const float* pose_dev, const float* tran_dev, int64_t N, float* rglobal_dev, float* joint_dev,
          void* stream) {
    if (!h || !pose_dev || !rglobal_dev || !joint_dev || N < 0) return h ? fail(h, MP_ERR_INVALID, "mp_fk: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    mp_launch_fk(pose_dev, tran_dev, (long)N, h->bone_dev, h->parent_dev, h->depth_dev, rglobal_dev, joint_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_set_mesh(mp_handle* h, const float* v_template_host, const float* weights_host, int n_vertex) {
    if (!h || !v_template_host || !weights_host || n_vertex < 1) return h ? fail(h, MP_ERR_INVALID, "mp_set_mesh: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    for (float** q : {&h->vrest_dev, &h->skinw_dev, &h->vtpl_dev, &h->shapedirs_dev, &h->jreg_dev, &h->posedirsT_dev}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    h->n_vertex = 0;
    std::vector<float> v((size_t)n_vertex * 3);
    for (int i = 0; i < n_vertex; ++i)
        for (int c = 0; c < 3; ++c) v[(size_t)i * 3 + c] = v_template_host[(size_t)i * 3 + c] - h->J0[c];   // model.py:87
    if (int rc = dev_alloc(h, (void**)&h->vrest_dev, v.size() * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->skinw_dev, (size_t)n_vertex * 24 * sizeof(float))) return rc;
    HIPCHK(h, hipMemcpy(h->vrest_dev, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->skinw_dev, weights_host, (size_t)n_vertex * 24 * sizeof(float), hipMemcpyHostToDevice));
    if (int rc = dev_alloc(h, (void**)&h->vtpl_dev, v.size() * sizeof(float))) return rc;
    HIPCHK(h, hipMemcpy(h->vtpl_dev, v_template_host, v.size() * sizeof(float), hipMemcpyHostToDevice));
    h->n_vertex = n_vertex;
    return MP_OK;
}

int mp_set_shape_space(mp_handle* h, const float* shapedirs_host, const float* j_regressor_host) {
    if (!h || !shapedirs_host || !j_regressor_host) return h ? fail(h, MP_ERR_INVALID, "mp_set_shape_space: bad argument") : MP_ERR_INVALID;
    if (!h->n_vertex) return fail(h, MP_ERR_INVALID, "mp_set_shape_space before mp_set_mesh");
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    for (float** q : {&h->shapedirs_dev, &h->jreg_dev}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    const size_t V = (size_t)h->n_vertex;
    if (int rc = dev_alloc(h, (void**)&h->shapedirs_dev, V * 30 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->jreg_dev, V * 24 * sizeof(float))) return rc;
    HIPCHK(h, hipMemcpy(h->shapedirs_dev, shapedirs_host, V * 30 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->jreg_dev, j_regressor_host, V * 24 * sizeof(float), hipMemcpyHostToDevice));
    return MP_OK;
}

// forward_kinematics in all its forms (articulate/model.py:208-240): optional shape (n_shape bodies: 1 or N), optional
// mesh, optional pose blend shapes (the handle's, mp_set_pose_blendshape).  Workspace layout (floats):
//   vrest [ns][V][3] | jraw, jrest, bone [ns][72] each | vposed [N][V][3] (pose blend shapes only)
namespace {
int ensure_shape_ws(mp_handle* h, size_t need) {
    if (need <= h->shape_ws_floats) return MP_OK;
    HIPCHK(h, hipDeviceSynchronize());
    if (h->shape_ws) (void)hipFree(h->shape_ws);
    h->shape_ws = nullptr; h->shape_ws_floats = 0;
    if (int rc = dev_alloc(h, (void**)&h->shape_ws, need * sizeof(float))) return rc;
    h->shape_ws_floats = need;
    return MP_OK;
}

int fk_general(mp_handle* h, const char* what, const float* pose_dev, const float* shape_dev, int n_shape,
               const float* tran_dev, int64_t N, float* rglobal_dev, float* joint_dev, float* vert_dev, void* stream) {
    if (N == 0) return MP_OK;
    if (vert_dev && !h->n_vertex) return fail(h, MP_ERR_INVALID, "%s before mp_set_mesh", what);
    if (shape_dev && !h->shapedirs_dev) return fail(h, MP_ERR_INVALID, "%s before mp_set_shape_space", what);
    ON_DEVICE(h);
    const size_t V = (size_t)h->n_vertex, ns = shape_dev ? (size_t)n_shape : 0;
    const bool blend = vert_dev && h->posedirsT_dev;
    const size_t f_body = ns * (V * 3 + 3 * 72);
    if (int rc = ensure_shape_ws(h, f_body + (blend ? (size_t)N * V * 3 : 0))) return rc;
    if (int rc = enter(h, stream)) return rc;
    const float *bone = h->bone_dev, *jrest = h->jrest_dev, *vrest = h->vrest_dev;
    long bstride = 0, vstride = 0;
    if (shape_dev) {
        float* vr = h->shape_ws;
        float* jraw = vr + ns * V * 3;
        float* jr = jraw + ns * 72;
        float* bn = jr + ns * 72;
        mp_launch_shape_body(shape_dev, n_shape, h->shapedirs_dev, h->vtpl_dev, h->jreg_dev, h->parent_dev, h->n_vertex, vr,
                             jraw, jr, bn, h->s_main);                                        // model.py:84-89
        bone = bn; jrest = jr; vrest = vr;
        if (n_shape != 1) { bstride = 72; vstride = (long)V * 3; }
    }
    mp_launch_fk(pose_dev, tran_dev, (long)N, bone, h->parent_dev, h->depth_dev, rglobal_dev, joint_dev, h->s_main, bstride);
    if (vert_dev) {
        if (blend) {                                                                          // model.py:236-238
            float* vposed = h->shape_ws + f_body;
            mp_launch_pose_blend(pose_dev, (long)N, vrest, vstride, h->posedirsT_dev, h->n_vertex, vposed, h->s_main);
            vrest = vposed; vstride = (long)V * 3;
        }
        mp_launch_lbs(rglobal_dev, joint_dev, tran_dev, (long)N, jrest, bstride, vrest, vstride, h->skinw_dev, h->n_vertex,
                      vert_dev, h->s_main);
    }
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}
}  // namespace

int mp_fk_shape(mp_handle* h, const float* pose_dev, const float* shape_dev, int n_shape, const float* tran_dev, int64_t N,
                float* rglobal_dev, float* joint_dev, float* vert_dev, void* stream) {
    if (!h || !pose_dev || !shape_dev || !rglobal_dev || !joint_dev || N < 0 || !(n_shape == 1 || n_shape == N))
        return h ? fail(h, MP_ERR_INVALID, "mp_fk_shape: bad argument (n_shape must be 1 or N)") : MP_ERR_INVALID;
    return fk_general(h, "mp_fk_shape", pose_dev, shape_dev, n_shape, tran_dev, N, rglobal_dev, joint_dev, vert_dev, stream);
}

int mp_fk_mesh(mp_handle* h, const float* pose_dev, const float* tran_dev, int64_t N, float* rglobal_dev, float* joint_dev,
               float* vert_dev, void* stream) {
    if (!h || !pose_dev || !rglobal_dev || !joint_dev || !vert_dev || N < 0) return h ? fail(h, MP_ERR_INVALID, "mp_fk_mesh: bad argument") : MP_ERR_INVALID;
    return fk_general(h, "mp_fk_mesh", pose_dev, nullptr, 0, tran_dev, N, rglobal_dev, joint_dev, vert_dev, stream);
}

int mp_set_pose_blendshape(mp_handle* h, const float* posedirs_host) {
    if (!h) return MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    if (h->posedirsT_dev) { (void)hipFree(h->posedirsT_dev); h->posedirsT_dev = nullptr; }
    if (!posedirs_host) return MP_OK;                                          // use_pose_blendshape = False
    if (!h->n_vertex) return fail(h, MP_ERR_INVALID, "mp_set_pose_blendshape before mp_set_mesh");
    const size_t n = (size_t)h->n_vertex * 3 * 207;
    float* staging = nullptr;
    if (int rc = dev_alloc(h, (void**)&staging, n * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->posedirsT_dev, n * sizeof(float))) { (void)hipFree(staging); return rc; }
    hipError_t e = hipMemcpy(staging, posedirs_host, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) { mp_launch_transpose_posedirs(staging, h->n_vertex, h->posedirsT_dev, h->s_main); e = hipStreamSynchronize(h->s_main); }
    (void)hipFree(staging);
    if (e != hipSuccess) return fail(h, MP_ERR_HIP, "mp_set_pose_blendshape: %s", hipGetErrorString(e));
    return MP_OK;
}

int mp_zero_pose_body(mp_handle* h, const float* shape_dev, int n_shape, float* joint_dev, float* vert_dev, void* stream) {
    if (!h || !shape_dev || n_shape < 1 || !joint_dev || !vert_dev)
        return h ? fail(h, MP_ERR_INVALID, "mp_zero_pose_body: bad argument") : MP_ERR_INVALID;
    if (!h->shapedirs_dev) return fail(h, MP_ERR_INVALID, "mp_zero_pose_body before mp_set_shape_space");
    ON_DEVICE(h);
    const size_t ns = (size_t)n_shape;
    if (int rc = ensure_shape_ws(h, ns * 2 * 72)) return rc;
    if (int rc = enter(h, stream)) return rc;
    // vertices and joints go straight into the caller's buffers; the raw joints and the bone vectors are scratch
    mp_launch_shape_body(shape_dev, n_shape, h->shapedirs_dev, h->vtpl_dev, h->jreg_dev, h->parent_dev, h->n_vertex, vert_dev,
                         h->shape_ws, joint_dev, h->shape_ws + ns * 72, h->s_main);           // model.py:84-89
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_eval_metrics(mp_handle* h, const float* pose_p_dev, const float* pose_t_dev, const float* tran_p_dev,
                    const float* tran_t_dev, int64_t N, int fps, int align_joint, unsigned joint_mask, unsigned ignored_mask,
                    int use_mesh, float* table_dev, void* stream) {
    if (!h || !pose_p_dev || !pose_t_dev || !table_dev || N < 1 || fps < 1 || align_joint < 0 || align_joint > 23)
        return h ? fail(h, MP_ERR_INVALID, "mp_eval_metrics: bad argument") : MP_ERR_INVALID;
    if (use_mesh && !h->n_vertex) return fail(h, MP_ERR_INVALID, "mp_eval_metrics: use_mesh without mp_set_mesh");
    ON_DEVICE(h);
    const size_t V = use_mesh ? (size_t)h->n_vertex : 0, n = (size_t)N;
    const size_t f_pose = n * 216, f_j = n * 72, f_v = n * V * 3;
    const size_t floats = 2 * (2 * f_pose + f_j + f_v);
    const size_t bytes = floats * sizeof(float) + mp_eval_partial_doubles((int)V) * sizeof(double) + 64;
    if (bytes > h->eval_ws_bytes) {
        HIPCHK(h, hipDeviceSynchronize());
        if (h->eval_ws) (void)hipFree(h->eval_ws);
        h->eval_ws = nullptr; h->eval_ws_bytes = 0;
        if (int rc = dev_alloc(h, (void**)&h->eval_ws, bytes)) return rc;
        h->eval_ws_bytes = bytes;
    }
    if (int rc = enter(h, stream)) return rc;
    float* w = h->eval_ws;
    float* mp_ = w;            float* mt_ = mp_ + f_pose;        // masked local poses
    float* rp = mt_ + f_pose;  float* rt = rp + f_pose;          // global rotations
    float* jp = rt + f_pose;   float* jt = jp + f_j;             // joints
    float* vp = jt + f_j;      float* vt = vp + f_v;             // vertices
    double* part = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(vt + f_v) + 63) & ~(uintptr_t)63);
    hipStream_t s = h->s_main;
    mp_launch_mask_pose(pose_p_dev, mp_, (long)N, ignored_mask, s);                                  // evaluate.py:25-26
    mp_launch_mask_pose(pose_t_dev, mt_, (long)N, ignored_mask, s);
    const float* pose[2] = {mp_, mt_};
    const float* tran[2] = {tran_p_dev, tran_t_dev};
    float* rg[2] = {rp, rt}; float* jj[2] = {jp, jt}; float* vv[2] = {vp, vt};
    for (int k = 0; k < 2; ++k) {                                                                    // evaluator.py:319-320
        mp_launch_fk(pose[k], tran[k], (long)N, h->bone_dev, h->parent_dev, h->depth_dev, rg[k], jj[k], s);
        if (V) mp_launch_lbs(rg[k], jj[k], tran[k], (long)N, h->jrest_dev, 0, h->vrest_dev, 0, h->skinw_dev, h->n_vertex, vv[k], s);
    }
    mp_launch_eval_metrics(mp_, mt_, rp, rt, jp, jt, V ? vp : nullptr, V ? vt : nullptr, (long)N, (int)V, fps, align_joint,
                           joint_mask, part, table_dev, s);                                          // evaluator.py:321-343
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_reset_state(mp_handle* h, int clear_velocity) {
    if (!h) return MP_ERR_INVALID;
    if (clear_velocity) h->vstate.B = 0;
    return MP_OK;
}

int mp_get_velocity_state(mp_handle* h, float* state_dev, int* batch) {
    if (!h || !batch) return MP_ERR_INVALID;
    ON_DEVICE(h);
    *batch = h->vstate.B;
    if (h->vstate.B && state_dev) {
        const size_t n = (size_t)2 * h->vstate.B * 256;
        HIPCHK(h, hipStreamSynchronize(h->s_main));
        HIPCHK(h, hipMemcpy(state_dev, h->vstate.h, n * sizeof(float), hipMemcpyDeviceToDevice));
        HIPCHK(h, hipMemcpy(state_dev + n, h->vstate.c, n * sizeof(float), hipMemcpyDeviceToDevice));
    }
    return MP_OK;
}

int mp_set_velocity_state(mp_handle* h, const float* state_dev, int batch) {
    if (!h || batch < 0 || (batch > 0 && !state_dev)) return MP_ERR_INVALID;
    ON_DEVICE(h);
    if (batch == 0) { h->vstate.B = 0; return MP_OK; }
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    if (int rc = ensure_vstate(h, h->vstate, batch)) return rc;
    const size_t n = (size_t)2 * batch * 256;
    HIPCHK(h, hipMemcpy(h->vstate.h, state_dev, n * sizeof(float), hipMemcpyDeviceToDevice));
    HIPCHK(h, hipMemcpy(h->vstate.c, state_dev + n, n * sizeof(float), hipMemcpyDeviceToDevice));
    h->vstate.B = batch;
    return MP_OK;
}

// ------------------------------------------------------------------------------------------ streaming
int mp_stream_create(mp_handle* h, int S) {
    if (!h || S < 1) return h ? fail(h, MP_ERR_INVALID, "mp_stream_create: S must be positive") : MP_ERR_INVALID;
    if (int rc = need_weights(h, "mp_stream_create")) return rc;
    ON_DEVICE(h);
    StreamCtx& c = h->sc;
    if (c.S) return fail(h, MP_ERR_INVALID, "streams already created (S = %d)", c.S);
    const int W = 45;
    if (int rc = dev_alloc(h, (void**)&c.window, (size_t)S * W * 60 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.fresh, S)) return rc;
    if (int rc = dev_alloc(h, (void**)&c.mask_dev, S)) return rc;
    if (int rc = dev_alloc(h, (void**)&c.st.last_foot, (size_t)S * 6 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.st.root_y, (size_t)S * sizeof(double))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.st.root_pos, (size_t)S * 3 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->st_snap.last_foot, (size_t)S * 6 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->st_snap.root_y, (size_t)S * sizeof(double))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->st_snap.root_pos, (size_t)S * 3 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.joints, (size_t)S * W * 72 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.vel, (size_t)S * W * 72 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.contact, (size_t)S * W * 2 * sizeof(float))) return rc;
    std::vector<float> lf((size_t)S * 6);
    for (int s = 0; s < S; ++s) memcpy(&lf[(size_t)s * 6], h->feet_pos, sizeof(h->feet_pos));   // net.py:59
    HIPCHK(h, hipMemcpy(c.st.last_foot, lf.data(), lf.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemset(c.fresh, 1, S));
    HIPCHK(h, hipMemset(c.st.root_y, 0, (size_t)S * sizeof(double)));
    HIPCHK(h, hipMemset(c.st.root_pos, 0, (size_t)S * 3 * sizeof(float)));
    c.S = S;
    Plan* p = nullptr;
    if (int rc = get_plan(h, S, W, &p)) return rc;
    p->streaming = true;
    std::vector<int32_t> len(S, W);
    if (int rc = upload_lengths(h, p, len.data())) return rc;
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    return MP_OK;
}

int mp_stream_step(mp_handle* h, const float* frames_dev, float* pose_dev, float* joints_dev, float* root_pos_dev,
                   float* contact_dev, void* stream) {
    if (!h) return MP_ERR_INVALID;
    StreamCtx& c = h->sc;
    if (!c.S) return fail(h, MP_ERR_NO_STREAMS, "mp_stream_step before mp_stream_create");
    if (!frames_dev || !pose_dev || !root_pos_dev || !contact_dev) return fail(h, MP_ERR_INVALID, "mp_stream_step: NULL buffer");
    const int S = c.S, W = 45, PAST = 40;
    // one velocity.rnn_state per model, shared by the batch and the online path (velocity.py:30)
    if (h->vstate.B != 0 && h->vstate.B != S)
        return fail(h, MP_ERR_STATE_SHAPE, "carried velocity state has batch %d, streaming has %d streams", h->vstate.B, S);
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    if (int rc = ensure_vstate(h, h->vstate, S)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, S, W, &p)) return rc;
    {   // the (S,45) plan may have been used by mp_forward with other lengths in between
        std::vector<int32_t> len(S, W);
        if (int rc = upload_lengths(h, p, len.data())) return rc;
    }
    float* joints = joints_dev ? joints_dev : c.joints;
    const bool has_state = h->vstate.B == S;
    h->segs.clear(); h->ev_used = 0;
    {
        CopyJobs js;
        if (h->recovery) {      // the solver state of the tick (net.py:59-64): last foot positions, root height, root position
            js.add(h->st_snap.last_foot, c.st.last_foot, (size_t)S * 6 * sizeof(float));
            js.add(h->st_snap.root_y, c.st.root_y, (size_t)S * sizeof(double));
            js.add(h->st_snap.root_pos, c.st.root_pos, (size_t)S * 3 * sizeof(float));
        }
        if (int rc = snapshot_vstate(h, S, has_state, &js)) return rc;      // (+ the velocity state: one launch)
    }
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.kind = 1; key.B = S; key.T = W; key.flags = (has_state ? 1 : 0) | (h->persist ? 2 : 0) | (h->x3 ? 8 : 0);
    key.p[0] = frames_dev; key.p[1] = pose_dev; key.p[2] = joints; key.p[3] = root_pos_dev; key.p[4] = contact_dev;
    key.p[6] = h->vstate.h;
    auto net_and_solver = [&]() {
        // forward on the 45-frame window (net.py:178); pose only for index 40 (net.py:181)
        bool tail = false;
        if (int r = forward_body(h, p, c.window, pose_dev, S, (long)W * 96, (long)PAST * 96, joints, c.vel, c.contact,
                                 p->r6d, h->vstate, has_state, nullptr, nullptr, &tail)) return r;
        mp_launch_translate_online(joints, c.vel, c.contact, S, W, PAST, h->floor_y, c.st, root_pos_dev, contact_dev,
                                   h->s_main);                                                // net.py:186-208
        if (tail) HIPCHK(h, hipStreamWaitEvent(h->s_main, h->ev_x[3], 0));
        HIPCHK(h, hipGetLastError());
        return (int)MP_OK;
    };
    int rc;
    {
        SegScope whole(h, h->s_main, 3, 1);
        rc = run_maybe_graph(h, key, [&]() {
            mp_launch_window_push(c.window, frames_dev, c.fresh, S, W, h->s_main);               // net.py:175
            return net_and_solver();
        });
    }
    if (rc) return rc;
    // (a repaired tick does not push the frame again: the window already holds it -- only network and solver are redone)
    if (int rc2 = finish_or_recover(h, p, "mp_stream_step", [&]() {
            if (int r = restore_vstate(h, S, has_state)) return r;
            HIPCHK(h, hipMemcpyAsync(c.st.last_foot, h->st_snap.last_foot, (size_t)S * 6 * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(c.st.root_y, h->st_snap.root_y, (size_t)S * sizeof(double), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(c.st.root_pos, h->st_snap.root_pos, (size_t)S * 3 * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            return (int)MP_OK;
        }, net_and_solver)) return rc2;
    h->vstate.B = S;
    return leave(h, stream);
}

// N consecutive forward_online calls of a single stream as ONE call (round 5; evaluate.py:62-64 runs
// `[model.forward_online(f) for f in ...]`, T + 5 of them per sequence, each a full launch chain on a 1 x 45 batch).
// Three of the four blocks are stateless per window (net.py:103-114): joints, pose and foot contact of all N windows run as ONE
// N x 45 batch -- the windows are never materialised, window k is rows k+1 .. k+45 of the frame history, which a RowMap with
// strideB = strideT = 60 addresses in place.  The velocity block is not: every call runs its 45 steps ON the state the previous
// call left (velocity.py:45-48), i.e. the N calls together are one 2-layer LSTM over a single sequence of N * 45
// steps whose input is the stacked linear1 of the N windows -- computed in the batch, written in sequence order, then two layer
// launches at B = 1, T = N * 45.  Only index 40 of every window is needed behind the layers (net.py:181-187): pose's and
// velocity's linear2 / IK run on N rows.  The solver chain over the N frames is one serial kernel.
int mp_stream_replay(mp_handle* h, const float* frames_dev, int N, float* pose_dev, float* joints_dev, float* root_pos_dev,
                     float* contact_dev, void* stream) {
    if (!h) return MP_ERR_INVALID;
    StreamCtx& c = h->sc;
    if (!c.S) return fail(h, MP_ERR_NO_STREAMS, "mp_stream_replay before mp_stream_create");
    if (c.S != 1) return fail(h, MP_ERR_INVALID, "mp_stream_replay drives a single stream (S = %d)", c.S);
    if (!frames_dev || !pose_dev || !root_pos_dev || !contact_dev || N < 1) return fail(h, MP_ERR_INVALID, "mp_stream_replay: NULL buffer or N < 1");
    const int W = 45, PAST = 40;
    if ((long)N * W > 0x3fffffffL / 256) return fail(h, MP_ERR_INVALID, "mp_stream_replay: %d frames in one call is beyond the supported size; split it", N);
    if (h->vstate.B != 0 && h->vstate.B != 1)
        return fail(h, MP_ERR_STATE_SHAPE, "carried velocity state has batch %d, the replayed stream has 1", h->vstate.B);
    if (use_x3(h, h->mod[MP_MOD_VELOCITY]))      // (before anything is enqueued; the Python facade feeds the frames tick by tick in mode 3)
        return fail(h, MP_ERR_INVALID, "mp_stream_replay runs on exact-fp32 operands (LSTM mode 1 or 0)");
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    if (int rc = ensure_vstate(h, h->vstate, 1)) return rc;
    const bool has_state = h->vstate.B == 1;
    // workspaces: the batch plan (N windows x 45), the chain plan (1 sequence x N*45), history / index-40 rows
    Plan *pb = nullptr, *pc = nullptr;
    if (int rc = get_plan(h, N, W, &pb)) return rc;
    if (int rc = get_plan(h, 1, N * W, &pc, pb)) return rc;      // (pb is in use: neither the victim of this acquisition nor its result)
    {
        std::vector<int32_t> len(N, W);
        if (int rc = upload_lengths(h, pb, len.data())) return rc;
        const int32_t one = N * W;
        if (int rc = upload_lengths(h, pc, &one)) return rc;
    }
    const size_t need = ((size_t)(W + N) * 60 + (size_t)N * 72 + (size_t)N * W * 72 + (size_t)N * W * 2) * sizeof(float);
    if (need > c.replay_bytes) {
        HIPCHK(h, hipStreamSynchronize(h->s_main));
        if (c.replay_ws) (void)hipFree(c.replay_ws);
        c.replay_ws = nullptr; c.replay_bytes = 0;
        if (int rc = dev_alloc(h, (void**)&c.replay_ws, need)) return rc;
        c.replay_bytes = need;
    }
    float* hist = c.replay_ws;
    float* vel40 = hist + (size_t)(W + N) * 60;
    float* joints_own = vel40 + (size_t)N * 72;
    float* contact_b = joints_own + (size_t)N * W * 72;
    float* joints = joints_dev ? joints_dev : joints_own;
    h->segs.clear(); h->ev_used = 0;
    {
        CopyJobs js;
        if (h->recovery) {
            js.add(h->st_snap.last_foot, c.st.last_foot, 6 * sizeof(float));
            js.add(h->st_snap.root_y, c.st.root_y, sizeof(double));
            js.add(h->st_snap.root_pos, c.st.root_pos, 3 * sizeof(float));
        }
        if (int rc = snapshot_vstate(h, 1, has_state, &js)) return rc;
    }
    mp_launch_replay_history(c.window, c.fresh, frames_dev, N, W, hist, h->s_main);
    const RowMap none{nullptr, 0, 0, 0};
    auto body = [&]() -> int {
        hipStream_t sm = h->s_main;
        const RowMap xi{hist + 60, 60, 60, 60};                            // window k, frame i = history row k + 1 + i
        const RowMap xj = user_map(joints, W, 72);
        RnnJob J{h, pb, MP_MOD_JOINTS, xi, none, joints, (long)W * 72, 72, STATE_ZERO, nullptr, nullptr, nullptr, nullptr};
        RnnJob P{h, pb, MP_MOD_POSE, xj, xi, pb->r6d, (long)W * 96, 96, STATE_ZERO, nullptr, nullptr, nullptr, nullptr};
        RnnJob F{h, pb, MP_MOD_FOOT_CONTACT, xj, xi, contact_b, (long)W * 2, 2, STATE_ZERO, nullptr, nullptr, nullptr, nullptr};
        if (int r = run_rnn(J, sm)) return r;                              // net.py:103
        if (int r = run_rnn(P, sm)) return r;                              // net.py:106-107
        mp_launch_r6d_ik_strided(pb->r6d, N, (long)W * 96, (long)PAST * 96, pose_dev, h->parent_dev, sm);   // net.py:110,181
        if (int r = run_rnn(F, sm)) return r;                              // net.py:113-114
        // velocity (net.py:117): linear1 of every window in the batch, rows written in (window, frame) order = the chain's time order
        const ModuleW& mv = h->mod[MP_MOD_VELOCITY];
        ModuleWS& wc = pc->ws[MP_MOD_VELOCITY];
        float* X1 = x1_buffer(h, mv, wc);
        run_gemm(h, sm, xj, xi, mv.lin1, X1, (long)W * mv.H, mv.H, N * W, N, 1);
        if (!h->persist && !wc.xproj) return fail(h, MP_ERR_INVALID, "internal: per-step workspace missing");
        if (!h->persist) run_gemm(h, sm, internal_map(X1, 1, mv.H), none, mv.ih[0], wc.xproj, 4 * mv.H, (long)4 * mv.H, N * W, 1, 0);
        RnnJob V{h, pc, MP_MOD_VELOCITY, none, none, nullptr, 0, 0, has_state ? STATE_FROM : STATE_ZERO, h->vstate.h, h->vstate.c, h->vstate.h, h->vstate.c};
        if (!h->persist) {       // per-step kernels keep their state in the plan's buffers: stage it in and out
            for (int l = 0; l < 2; ++l) {
                const size_t n = (size_t)mv.H * sizeof(float);
                if (has_state) {
                    HIPCHK(h, hipMemcpyAsync(wc.hbuf[l][0], h->vstate.h + (size_t)l * mv.H, n, hipMemcpyDeviceToDevice, sm));
                    HIPCHK(h, hipMemcpyAsync(wc.cbuf[l][0], h->vstate.c + (size_t)l * mv.H, n, hipMemcpyDeviceToDevice, sm));
                } else {
                    HIPCHK(h, hipMemsetAsync(wc.hbuf[l][0], 0, n, sm));
                    HIPCHK(h, hipMemsetAsync(wc.cbuf[l][0], 0, n, sm));
                }
            }
        }
        if (int r = rnn_rec(V, 0, sm)) return r;
        if (int r = rnn_g1(V, sm)) return r;
        if (int r = rnn_rec(V, 1, sm)) return r;
        if (!h->persist) {
            const size_t fin = (size_t)((N * W) & 1) * mv.H;
            for (int l = 0; l < 2; ++l) {
                const size_t n = (size_t)mv.H * sizeof(float);
                HIPCHK(h, hipMemcpyAsync(h->vstate.h + (size_t)l * mv.H, wc.hbuf[l][0] + fin, n, hipMemcpyDeviceToDevice, sm));
                HIPCHK(h, hipMemcpyAsync(h->vstate.c + (size_t)l * mv.H, wc.cbuf[l][0], n, hipMemcpyDeviceToDevice, sm));
            }
        }
        // linear2 on row 40 of every window only (net.py:196 reads nothing else)
        run_gemm(h, sm, RowMap{wc.out1 + (size_t)PAST * mv.H, (long)W * mv.H, 0, mv.H}, none, mv.lin2, vel40, 72, 0, N, N, 0);
        mp_launch_translate_replay(joints, vel40, contact_b, N, W, PAST, h->floor_y, c.st, root_pos_dev, contact_dev, sm);   // net.py:186-208
        HIPCHK(h, hipGetLastError());
        return (int)MP_OK;
    };
    int rc;
    {
        SegScope whole(h, h->s_main, 3, 1);
        rc = body();
    }
    if (rc) return rc;
    if (int rc2 = finish_or_recover(h, pb, "mp_stream_replay", [&]() {
            if (int r = ensure_step_ws(h, pc)) return r;
            if (int r = restore_vstate(h, 1, has_state)) return r;
            HIPCHK(h, hipMemcpyAsync(c.st.last_foot, h->st_snap.last_foot, 6 * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(c.st.root_y, h->st_snap.root_y, sizeof(double), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(c.st.root_pos, h->st_snap.root_pos, 3 * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            return (int)MP_OK;
        }, body)) return rc2;
    mp_launch_replay_window(hist, N, W, c.window, c.fresh, h->s_main);   // net.py:175: the stream's window after the last call
    HIPCHK(h, hipGetLastError());
    h->vstate.B = 1;
    return leave(h, stream);
}

int mp_stream_reset(mp_handle* h, const uint8_t* mask_host, int clear_velocity) {
    if (!h) return MP_ERR_INVALID;
    StreamCtx& c = h->sc;
    if (!c.S) return fail(h, MP_ERR_NO_STREAMS, "mp_stream_reset before mp_stream_create");
    ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    if (mask_host) HIPCHK(h, hipMemcpy(c.mask_dev, mask_host, c.S, hipMemcpyHostToDevice));
    const bool vel = clear_velocity && h->vstate.B == c.S;
    mp_launch_stream_reset(mask_host ? c.mask_dev : nullptr, c.fresh, c.st.root_y, c.st.root_pos, vel ? h->vstate.h : nullptr,
                           vel ? h->vstate.c : nullptr, c.S, h->s_main);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    return MP_OK;
}

int mp_live_form_frames(mp_handle* h, const float* quat_dev, const float* acc_dev, const float* smpl2imu_dev,
                        const float* device2bone_dev, const float* acc_offsets_dev, unsigned keep_mask, int S,
                        float* frames_dev, void* stream) {
    if (!h || !quat_dev || !acc_dev || !smpl2imu_dev || !device2bone_dev || !acc_offsets_dev || !frames_dev || S < 1)
        return h ? fail(h, MP_ERR_INVALID, "mp_live_form_frames: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    mp_launch_live_frames(quat_dev, acc_dev, smpl2imu_dev, device2bone_dev, acc_offsets_dev, keep_mask, 30.0f /* config.py:74 */,
                          S, frames_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_stream_get_state(mp_handle* h, int s, float* window_dev, float last_foot_host[6], double* root_y_host,
                        float root_pos_host[3], int* fresh_host) {
    if (!h) return MP_ERR_INVALID;
    StreamCtx& c = h->sc;
    if (!c.S) return fail(h, MP_ERR_NO_STREAMS, "mp_stream_get_state before mp_stream_create");
    if (s < 0 || s >= c.S) return fail(h, MP_ERR_INVALID, "mp_stream_get_state: stream %d outside 0..%d", s, c.S - 1);
    ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    if (window_dev)
        HIPCHK(h, hipMemcpy(window_dev, c.window + (size_t)s * 45 * 60, (size_t)45 * 60 * sizeof(float), hipMemcpyDeviceToDevice));
    if (last_foot_host) HIPCHK(h, hipMemcpy(last_foot_host, c.st.last_foot + (size_t)s * 6, 6 * sizeof(float), hipMemcpyDeviceToHost));
    if (root_y_host) HIPCHK(h, hipMemcpy(root_y_host, c.st.root_y + s, sizeof(double), hipMemcpyDeviceToHost));
    if (root_pos_host) HIPCHK(h, hipMemcpy(root_pos_host, c.st.root_pos + (size_t)s * 3, 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (fresh_host) {
        uint8_t f = 0;
        HIPCHK(h, hipMemcpy(&f, c.fresh + s, 1, hipMemcpyDeviceToHost));
        *fresh_host = f;
    }
    return MP_OK;
}

int mp_stream_set_state(mp_handle* h, int s, const float* window_dev, const float last_foot_host[6], const double* root_y_host,
                        const float root_pos_host[3], const int* fresh_host) {
    if (!h) return MP_ERR_INVALID;
    StreamCtx& c = h->sc;
    if (!c.S) return fail(h, MP_ERR_NO_STREAMS, "mp_stream_set_state before mp_stream_create");
    if (s < 0 || s >= c.S) return fail(h, MP_ERR_INVALID, "mp_stream_set_state: stream %d outside 0..%d", s, c.S - 1);
    ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    if (window_dev)
        HIPCHK(h, hipMemcpy(c.window + (size_t)s * 45 * 60, window_dev, (size_t)45 * 60 * sizeof(float), hipMemcpyDeviceToDevice));
    if (last_foot_host) HIPCHK(h, hipMemcpy(c.st.last_foot + (size_t)s * 6, last_foot_host, 6 * sizeof(float), hipMemcpyHostToDevice));
    if (root_y_host) HIPCHK(h, hipMemcpy(c.st.root_y + s, root_y_host, sizeof(double), hipMemcpyHostToDevice));
    if (root_pos_host) HIPCHK(h, hipMemcpy(c.st.root_pos + (size_t)s * 3, root_pos_host, 3 * sizeof(float), hipMemcpyHostToDevice));
    if (fresh_host) {
        const uint8_t f = *fresh_host ? 1 : 0;
        HIPCHK(h, hipMemcpy(c.fresh + s, &f, 1, hipMemcpyHostToDevice));
    }
    return MP_OK;
}

// ------------------------------------------------------------------------------------------ measurement
int mp_timing_enable(mp_handle* h, int on) {
    if (!h) return MP_ERR_INVALID;
    ON_DEVICE(h);
    h->timing = on != 0;
    h->segs.clear(); h->ev_used = 0;
    return MP_OK;
}

int mp_timing_read(mp_handle* h, int cls, int* launches, float* ms, double* gflop) {
    if (!h || !launches || !ms) return MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    *launches = 0; *ms = 0.f;
    double fl = 0.0;
    for (const Seg& s : h->segs) {
        if (s.cls != cls) continue;
        float t = 0.f;
        HIPCHK(h, hipEventElapsedTime(&t, s.a, s.b));
        *ms += t;
        *launches += s.launches;
        fl += s.flop;
    }
    if (gflop) *gflop = fl * 1e-9;
    return MP_OK;
}

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

int mp_debug_plan_stats(mp_handle* h, int* n_plans, int* n_allocs, long long* cap_rows) {
    if (!h) return MP_ERR_INVALID;
    if (n_plans) *n_plans = (int)h->plans.size();
    if (n_allocs) *n_allocs = h->plan_allocs;
    if (cap_rows) { long long r = 0; for (const Plan* q : h->plans) r += (long long)q->capRows; *cap_rows = r; }
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

int mp_set_lstm_mode(mp_handle* h, int mode) {
    if (!h || mode < 0 || mode > 3) return MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    h->persist = mode != 0;              // (mode 2, rounds 1-4's separate two-layer wavefront kernel: since round 5 the velocity
    h->x3 = mode == 3;                   //  block of mode 1's full-batch schedule IS a two-layer wavefront -- 2 means 1)
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

int mp_set_graph_mode(mp_handle* h, int on) {
    if (!h || on < 0 || on > 2) return MP_ERR_INVALID;
    ON_DEVICE(h);
    h->use_graph = on != 0;
    h->graph_serial = on == 2 || (on == 1 && !multibranch_graphs_allowed());
    return MP_OK;
}

}  // extern "C"
