// Handle lifetime (include/mobileposer_hip.h: mp_create, mp_create_from_device, mp_create_body, mp_destroy) and everything that
// happens once per handle: the weight manifest of the reference's state_dict (models/net.py:40-43), the re-layout of every
// matrix into the fragment order of the kernel family that reads it (mp_create), the SMPL constants, the device probe
// (CU count, XCD round robin), MP_VARIANT / MP_LSTM_MODE / MP_GRAPH.
#include "mp_host.h"

namespace mph { std::string g_create_error; }

namespace mph {

namespace {

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

}  // namespace

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


}  // namespace mph

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


}  // extern "C"
