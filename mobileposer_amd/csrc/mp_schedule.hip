// MobilePoserNet.forward (models/net.py:101-119) as launches: one RNN block (models/rnn.py:20-33) in five phases, the choice of
// kernel family and slices per slab, cluster placement on XCDs, and the schedules of forward_body (which blocks run side by
// side, what rides in whose workgroups, which seam is one launch).
#include "mp_host.h"

namespace mph {


RowMap internal_map(const float* base, int B, int width) { return RowMap{base, (long)width, (long)B * width, width}; }
RowMap user_map(const float* base, int T, int width) { return RowMap{base, (long)T * width, (long)width, width}; }

int run_gemm(mp_handle* h, hipStream_t s, RowMap a0, RowMap a1, const Packed& w, float* C, long cStrideB,
             long cStrideT, int M, int B, int relu, bool pair_out, bool a_pairs, bool x3_gemm,
             unsigned long long* zero_hx, int zero_ncl) {
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

// The one-sequence kernels (mp_lstm_v1 / mp_lstm_v1s) take a cluster per (direction, SEQUENCE) -- not per 16-sequence slab -- so
// a handful of sequences are a handful of clusters: up to kSeqClusterMax sequences (2 directions x 4 sequences = 8 clusters, one
// XCD each) run on them (round 6; round 5 routed B = 1 only).  A block gets them where its batch runs on the 32-slice family
// (fp32_slices == 32: 256 CUs, and round-robin dispatch wherever blocks run side by side).
bool seq_clusters(const mp_handle* h, const ModuleW& m, int B) {
    if (!h->persist || !h->vec_ok || use_x3(h, m) || B > kSeqClusterMax || m.whhR[0][0] == nullptr) return false;
    if (m.H == 256) return fp32_slices(h, m, B) == 32;
    return m.H == 64 && fp32_slices(h, h->mod[MP_MOD_VELOCITY], B) == 32;
}
// units of a layer launch of module m: sequences on the one-sequence kernels, 16-sequence slabs everywhere else
int launch_units(const mp_handle* h, const ModuleW& m, int B) { return seq_clusters(h, m, B) ? B : (B + 15) / 16; }

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
    return h->persist && h->wf_ok && h->vec_ok && !use_x3(h, m) && m.H == 256 && m.dirs == 1 && m.whhR[0][0] != nullptr && B <= kSeqClusterMax &&
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
        const int nslab = launch_units(h, m, B);               // (sequences on the one-sequence kernels, else 16-sequence slabs)
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
        const bool v1 = u8 && seq_clusters(h, m, B);           // a few sequences: matrix-vector steps, a cluster per sequence (mp_lstm_v1)
        // ... and the H = 64 block: a whole (direction, sequence) per workgroup (mp_lstm_v1s).  Only where the H = 256 blocks
        // of this batch run on the 32-slice family too (fp32_slices: 256 CUs, round-robin dispatch where blocks run side by side)
        const bool v1s = H == 64 && seq_clusters(h, m, B);
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
                // (a few sequences: both clusters of a sequence where forward_body's table has the sequence's one cluster, else
                //  sequence k on XCD k)
                for (int x = 0; x < 8; ++x)
                    cnt[x] = (unsigned char)(2 * (wf32 && a.xcd_physical ? h->xcd_plan[j.id][x] : (a.nslab + 7 - x) / 8));
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
    const bool any_x3 = use_x3(h, pm) || use_x3(h, vm);
    auto job = [&](int id, const ModuleW& m, int slices) { return XcdJob{id, m.dirs * launch_units(h, m, B), slices}; };
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
                 bool has_state, float* fk_rglobal, float* fk_joint, bool* tail_pending) {
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
            const ModuleW& vm_ = h->mod[MP_MOD_VELOCITY];
            const ModuleW& fm_ = h->mod[MP_MOD_FOOT_CONTACT];
            const XcdJob vf[2] = {{MP_MOD_VELOCITY, vm_.dirs * launch_units(h, vm_, p->B), fp32_slices(h, vm_, p->B)},
                                  {MP_MOD_FOOT_CONTACT, fm_.dirs * launch_units(h, fm_, p->B), fm_.nslice}};
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


}  // namespace mph
