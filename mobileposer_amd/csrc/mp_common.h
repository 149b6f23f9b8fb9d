// Shared declarations of the gfx950 kernels behind libmobileposer_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// No kernel of this library contains packed-fp32 VALU instructions (v_pk_mul/add/fma_f32): the build passes
// `-Xclang -target-feature -Xclang -packed-fp32-ops` for all device code (__graft_entry__.build) and tests/test_cabi_cpu.py
// disassembles the library to check that none slipped in.  Insurance, not a claimed hardware erratum: some historical
// revisions of the split-bf16 LSTM kernels disturbed such results (lanes 48..63) of a kernel running beside them, the
// current ones do not, and no trigger could be named (profiles/NOTES_r01-r03.md 4.3, profiles/r02_coexec_glitch.md).
#define MP_KERNEL __global__

// ReLU as torch computes it (rnn.py:22, F.relu): a NaN stays a NaN.  fmaxf(x, 0) returns 0 for a NaN (IEEE maxNum), which
// turned a NaN input sample -- or a poisoned slab's NaN joints -- into plausible numbers behind every linear1 (round 4).
static __device__ __forceinline__ float relu_(float x) { return x < 0.f ? 0.f : x; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Address of logical row m = t*B + b (time-major row index) of an activation matrix:
// base + b*strideB + t*strideT.  Internal buffers are time-major ([T][B][C]: strideB = C,
// strideT = B*C); caller buffers are batch-major ([B][T][C]: strideB = T*C, strideT = C).
struct RowMap {
    const float* base;
    long strideB;
    long strideT;
    int width;       // number of valid K columns supplied by this segment
};

// ---------------------------------------------------------------- K1/K3: fp32 MFMA GEMM
// C[m][n] = act( sum_k A[m][k] * W[n][k] + bias[n] ),  A = [a0 | a1] (concat fused), W packed
// [Npad][Kpad] zero padded (Kpad % 32 == 0, Npad % BN == 0).
struct GemmArgs {
    RowMap a0, a1;
    const float* W;
    const float* bias;
    float* C;
    long cStrideB, cStrideT;
    int M, N, K, Kpad, B, relu;
    int pairOut = 0;     // 1: store C as split-bf16 pairs (mp_lstm_dev.h pair_of) -- input format of mp_lstm_x3.hip
    int aPairs = 0;      // mp_gemm_x3 only: the A segments already hold pair words
    // mp_gemm_x3 only: re-arm (zero the polled words of) `zero_ncl` clusters of a split-bf16 exchange area on the way -- the
    // linear1 GEMM of a block does it for the layer-0 launch that follows, which saves a kernel boundary on the critical path
    unsigned long long* zero_hx = nullptr;
    int zero_ncl = 0;
    // mp_gemm_x3 only: two linear layers over the same rows in one launch -- output columns [nsplit, N) go to C2 (as its
    // columns [0, N - nsplit), same row strides); a second exchange area to re-arm
    float* C2 = nullptr;
    int nsplit = 0;
    unsigned long long* zero_hx2 = nullptr;
    int zero_ncl2 = 0;
    // mp_gemm_f32 only: the same padded W in MFMA B-fragment order (mp_launch_pack_wfrag) -- piece (k-tile kt, quarter q,
    // 32-column tile b) is 64 lanes x 16 bytes contiguous -- and its number of 32-column tiles; nullptr: row-major W only
    const float* Wf = nullptr;
    int NB = 0;
    // mp_gemm_f32_frag only: a third stacked output -- columns [nsplit3, N) go to C3 (as its columns [0, N - nsplit3)) with
    // row strides of its own (the foot-contact block's linear1 is 64 wide, pose's and velocity's 256)
    float* C3 = nullptr;
    int nsplit3 = 0;
    long c3StrideB = 0, c3StrideT = 0;
};
// two independent GEMMs in ONE launch (workgroups [0, n1) run g1, the rest g2): linear2 of the velocity block and of the
// foot-contact block at the end of a forward -- the second one is 10 us of work that would otherwise cost a launch of its own on
// the critical chain or a cross-stream join in front of the solver.  Both need W in fragment order; false = shapes not covered
bool mp_launch_gemm_pair(const GemmArgs& g1, const GemmArgs& g2, hipStream_t s);
// linear2 of one block and the stacked linear1 that reads its output (plus a second K segment) as ONE launch
// (mp_gemm_l2l1: joints.linear2 -> pose | velocity | foot-contact linear1); g1.a0 is not read (y comes from LDS), its width is
// the number of y columns.  false = shapes not covered, nothing launched
bool mp_launch_gemm_l2l1(const GemmArgs& g2, const GemmArgs& g1, hipStream_t s);
bool mp_gemm_l2l1_applicable(const GemmArgs& g2, const GemmArgs& g1);
// W [Npad][Kpad] row-major -> fragment order [Kpad/32][4][Npad/32][64 lanes][4]: lane (li = lane & 31, lh = lane >> 5) of piece
// (kt, q, b) holds W[b*32 + li][kt*32 + lh*16 + q*4 .. +3] -- what a lane of mp_gemm_f32_rows feeds its MFMAs of quarter q
void mp_launch_pack_wfrag(const float* W, float* Wf, int Npad, int Kpad, hipStream_t s);
// bn: 128, 96 or 32 (chosen by the caller from N)
void mp_launch_gemm(const GemmArgs& g, int bn, hipStream_t s);
int mp_gemm_pick_bn(int N);
// the same GEMM on split-bf16 MFMA operands (mp_gemm_x3.hip); g.W = pair words made by mp_launch_pairs
void mp_launch_gemm_x3(const GemmArgs& g, int bn, hipStream_t s);
void mp_launch_pairs(const float* src, float* dst, size_t n, hipStream_t s);

// ---------------------------------------------------------------- K2: LSTM recurrence step
struct LstmDir {
    const float* wpack;   // W_hh in MFMA B-fragment order (per-step layout: mp_lstm.hip; persistent: mp_lstm_persist.hip)
    const float* xproj;   // per-step kernel: [T*B][xprojStride] gate-interleaved (4*j+g) pre-activations, bias folded in
    float* out;           // [T*B][outStride], column offset applied
    float* hbuf;          // per-step: [2][B][H] ping-pong hidden state; persistent: [B][H] initial/final state
    float* cbuf;          // [B][H]
    int xprojStride, outStride, reverse;
    // fused persistent kernel only
    const float* wihpack; // W_ih in B-fragment order (mp_pack_wih_persist)
    const float* bias;    // [4H] gate-interleaved b_ih + b_hh of this direction
    const float* xin;     // layer input, time-major [T][B][K_in]
    // persistent kernels: where the A operand of step 0 -- the initial h of ALL units, read by every workgroup of the cluster --
    // comes from.  Normally hbuf itself; for T = 1 with a carried state the host passes a copy: nobody waits for anybody in a
    // one-step launch, and a workgroup that starts late must not find the final state of a neighbour that has already finished
    // where the initial state was (the streaming paths and the carried velocity state keep their state in place).
    const float* hin;
};
struct LstmStepArgs {
    LstmDir d[2];
    const int* lengths;   // [B] device
    int ndir, B, T, step;
};
void mp_launch_lstm_step(const LstmStepArgs& a, int H, hipStream_t s);
size_t mp_whh_pack_floats(int H);   // H*4H
void mp_launch_pack_whh(const float* whh, float* dst, int H, hipStream_t s);
// W_ih rows -> gate-interleaved rows of a [Npad][Kpad] matrix at row offset dirOff; bias = b_ih+b_hh
void mp_launch_pack_wih(const float* wih, const float* bih, const float* bhh, float* dstW, float* dstBias,
                        int H, int K, int Kpad, int dirOff, hipStream_t s);
void mp_launch_pack_linear(const float* w, const float* b, float* dstW, float* dstBias, int N, int K, int Kpad,
                           hipStream_t s);

// persistent variant (mp_lstm_persist.hip): one launch = all T steps of <= 2 directions for `nslab` slabs
// of 16 sequences starting at slab `slab0`; grid = nslab * NSLICE x ndir workgroups, all co-resident.
struct LstmPersistArgs {
    LstmDir d[2];
    const int* lengths;
    unsigned long long* hx;       // granules [ndir][nslab][4*16*H + 16], zeroed before every launch
    int* err;                     // device error word (0 = ok, 1+step = a gather timed out)
    int ndir, B, T, slab0, nslab;
    int zero_state;               // 1: start from h = c = 0 without reading hbuf / cbuf
    int force_remote;             // test hook (PROF instantiation): use the any-placement (sc1) transport even inside one XCD
    unsigned max_spin;            // 0: never wait (a test hook); otherwise waits are allowed, bounded by max_ticks
    unsigned long long max_ticks; // bound of every wait in ticks of the constant 100 MHz clock (s_memrealtime); mp_api: 0.25 s
    long long* prof;              // optional [grid][6] cycle sums per phase (debug), else nullptr
    // mp_lstm_fused<256,16|8,256,*,FK> only: a bidirectional H = 64 layer -- the foot-contact block -- rides along in the
    // workgroups of this H = 256 launch (zero initial state, same lengths).  16 slices ("VF", unidirectional carrier): slice j
    // of a slab also carries units 8*(j & 7) .. +7 of direction j >> 3 of that slab's H = 64 layer; 8 slices (two clusters per
    // slab): slice j of cluster direction d carries units 8*j .. +7 of direction d.  f_w: fragments from mp_launch_pack_foot_vf.
    const float* f_w[2] = {nullptr, nullptr};
    const float* f_bias[2] = {nullptr, nullptr};   // per direction: gate-interleaved b_ih + b_hh [4 * 64]
    const float* f_xin = nullptr;                  // rider's layer input, time-major [T][B][FK]
    float* f_out = nullptr;                        // rider's layer output, time-major [T][B][128] (direction d at column 64 d)
    int debug_drop = 0;           // test hook (mp_debug_drop_workgroup): block debug_drop - 1 exits at once, as if it never
                                  // became resident -- its cluster's waits run into their time bound
    int out_pairs = 0;            // split-bf16 kernel only: write the layer output as pairs (it feeds another layer)
    // mp_lstm_fused only: 0 = the area was zeroed before this launch (tags = step numbers); otherwise the area may hold
    // anything earlier launches of THIS kernel family left there, all with tags < epoch_base: the launch tags its granules
    // epoch_base + step (and its XCC table entries epoch_base), so nothing stale can match and no memset sits between two
    // layers on the critical path.  The host advances the base by T + 1 per launch and re-zeroes before a wrap-around.
    unsigned epoch_base = 0;
    // mp_lstm_fused, the kernels that exchange tagged words ("TAGX": H = 256 on 16 slices or on 8 slices with four waves; see
    // mp_persist_tagged): bit p = the tag of the FIRST write of this launch to parity slot p (slot 0: steps 0, 2, ...; slot 1:
    // the initial state as "step -1", then steps 1, 3, ...), the opposite of what the slot's words were left with -- 3 after
    // the area was zeroed; the host keeps the two bits per area (every launch rewrites every word of a slot, with alternating
    // tags) and zeroes the area when another kernel family wrote to it, after a device error, and where launches are
    // replayed (graphs).
    unsigned tag_flip = 3;
    // ... and the same two bits for the words of a riding H = 64 layer (f_w != nullptr): a bookkeeping of their own, because only
    // launches that carry a rider write them (round 5: pose layer 0 carries foot-contact layer 0, pose layer 1 nothing)
    unsigned tag_flip_f = 3;
    // mp_lstm_fused, host side only: ask for at least this much dynamic LDS (bytes) although the kernel uses less.  With
    // more than half a CU's LDS per workgroup no CU takes two persistent workgroups -- of this launch or of a launch
    // running beside it -- as long as CUs are free: a workgroup that shares its SIMDs slows its whole lock-stepped cluster.
    int min_lds = 0;
    // mp_lstm_fused: which clusters ((direction, slab) pairs, 0 .. ndir*nslab-1) run on which XCD.  The dispatcher sends
    // workgroup b to XCD b % 8 -- strictly: a workgroup whose XCD has no CU left for it WAITS, even while other XCDs stand
    // empty (tools/micro/xcd_dispatch.hip) -- and a cluster's workgroups share an XCD (their hand-off lives in its L2).  XCD x
    // runs clusters xcd_base[x] .. xcd_base[x] + xcd_cnt[x] - 1; all zero = spread round robin (mp_fill_xcd_table).
    alignas(8) unsigned char xcd_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    alignas(8) unsigned short xcd_base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // 1: the table is indexed by the XCD a workgroup really runs on (its XCC id), not by blockIdx % 8 -- the round robin
    // does not start at XCD 0 for every launch (another stream's launch of the same micro-benchmark started at XCD 7), and
    // tables of launches that run side by side must mean the same XCDs.  Only when the device was probed (mp_create) to
    // deal workgroups b, b + 8, ... onto one XCD and 8 consecutive ones onto 8 different XCDs.
    int xcd_physical = 0;
    unsigned long long* hx_next = nullptr;   // split-bf16 kernel only: exchange area of the NEXT layer's launch (same cluster
                                              // indexing), re-armed by this launch at its start
};
// Entry x of the two tables by shifts on whole words (round 4).  `a.xcd_cnt[xcd]` with a run-time index is a LOAD from the kernel
// argument segment -- two dependent memory round trips (count, then base) in front of the first weight load of every layer
// launch; the words themselves arrive with the other arguments through the scalar cache.
static __device__ __forceinline__ int mp_xcd_count(const LstmPersistArgs& a, int xcd) {
    unsigned long long w;
    __builtin_memcpy(&w, a.xcd_cnt, 8);
    return (int)((w >> (8 * xcd)) & 0xffu);
}
static __device__ __forceinline__ int mp_xcd_first(const LstmPersistArgs& a, int xcd) {
    unsigned long long w[2];
    __builtin_memcpy(w, a.xcd_base, 16);
    return (int)(((xcd & 4 ? w[1] : w[0]) >> (16 * (xcd & 3))) & 0xffffu);
}
// nslice: workgroups sharing one slab -- H = 256: 16 (four 256-register waves, two workgroups per CU) or 8 (four 512-register
// waves with AccVGPR-resident weights, one per CU); H = 64: 4.  The H = 256 kernels exchange tagged words (LstmPersistArgs::tag_flip)
void mp_launch_lstm_persist(const LstmPersistArgs& a, int H, int KIN, int nslice, hipStream_t s);
// the H = 256, K_in = 256 kernel on 8 slices with a riding H = 64 layer (fk = its K_in: 0 | 64 | 128) and / or as the two-layer
// wavefront of a unidirectional block (wf: ndir = 2, d[0] = layer 0, d[1] = layer 1 with d[1].xin = d[0].out, cluster index =
// slab * 2 + layer; mp_lstm_persist.hip).  false = combination not built, nothing launched
bool mp_launch_lstm_persist8(const LstmPersistArgs& a, int fk, bool wf, hipStream_t s);
// the unidirectional H = 256, K_in = 256 layer on 16 slices with an H = 64 bidirectional layer riding along (fk = its K_in: 64 | 128)
void mp_launch_lstm_vf(const LstmPersistArgs& a, int fk, hipStream_t s);
size_t mp_foot_vf_floats(int fk);     // per direction
void mp_launch_pack_foot_vf(const float* wih, const float* whh, float* dst, int fk, hipStream_t s);
void mp_fill_xcd_table(LstmPersistArgs& a, const unsigned char* cnt);   // LstmPersistArgs::xcd_cnt / xcd_base
void mp_launch_xcc_probe(int* out64, hipStream_t s);                    // 64 workgroups -> their XCC ids
// 32 slices of 8 units per slab (mp_lstm_u8.hip): small batches; wpack / wihpack from mp_launch_pack_w_u8 (K = 256 / K_in)
void mp_launch_lstm_u8(const LstmPersistArgs& a, int KIN, hipStream_t s);
void mp_launch_pack_w_u8(const float* w, float* dst, int K, hipStream_t s);
hipError_t mp_lstm_u8_device_attrs();
// mp_lstm_v1.hip: one sequence per cluster (B = 1), matrix-vector steps on the vector ALU; d[].wpack / wihpack from mp_launch_pack_w_v1
// (H = 256) or torch's row-major matrices (mp_lstm_v1s, H = 64)
void mp_launch_lstm_v1(const LstmPersistArgs& a, int KIN, bool wavefront, hipStream_t s);
void mp_launch_pack_w_v1(const float* w, float* dst, int K, hipStream_t s);   // W [1024][K] -> the per-lane order of mp_lstm_v1
void mp_launch_lstm_v1s(const LstmPersistArgs& a, int KIN, hipStream_t s);   // H = 64: one workgroup per (direction, sequence)
hipError_t mp_lstm_v1_device_attrs();
void mp_launch_pack_whh_persist(const float* whh, float* dst, int H, int nslice, hipStream_t s);
void mp_launch_pack_wih_persist(const float* wih, float* dst, int H, int KIN, int nslice, hipStream_t s);
int mp_persist_max_wg(int H, int nslice);    // largest grid that is co-resident

// split-bf16 variant of the persistent layer (mp_lstm_x3.hip), H = 256 only: xin and (out_pairs) out hold pairs,
// wpack / wihpack come from mp_launch_pack_w_x3 (W_hh: K = 256; W_ih: K = K_in)
void mp_launch_lstm_x3(const LstmPersistArgs& a, int KIN, int nslice, hipStream_t s);
void mp_launch_lstm_x3w(const LstmPersistArgs& a, int KIN, hipStream_t s);   // four 512-register waves per workgroup (8 slices)
void mp_launch_pack_w_x3(const float* w, float* dst, int K, int nslice, hipStream_t s);
// Dynamic-LDS limits (80-160 KB) of the persistent kernels are per-DEVICE function attributes: mp_create sets them for
// the handle's device after hipSetDevice, outside of any stream capture.
hipError_t mp_lstm_persist_device_attrs();
hipError_t mp_lstm_x3_device_attrs();
hipError_t mp_lstm_x3w_device_attrs();

// ---------------------------------------------------------------- K4/K5: kinematics
void mp_launch_r6d_ik(const float* r6d, long N, float* pose, const int* parent_dev, hipStream_t s);
void mp_launch_r6d_to_rot(const float* r6d, long n, float* out, hipStream_t s);   // n six-vectors -> n 3x3 matrices
// ParametricModel.inverse_kinematics_R (articulate/model.py:146-164): rglobal [N,24,3,3] -> rlocal [N,24,3,3] (distinct buffers)
void mp_launch_global_to_local(const float* rglobal, long N, float* rlocal, const int* parent_dev, hipStream_t s);
// frame n reads its 96 numbers at r6d + n*rowStride + rowOffset
void mp_launch_r6d_ik_strided(const float* r6d, long N, long rowStride, long rowOffset, float* pose,
                              const int* parent_dev, hipStream_t s);
// mp_launch_r6d_ik_strided + mp_launch_fk (one body, no translation) as ONE launch (mp_r6d_ik_fk); false = not applicable,
// nothing launched
bool mp_launch_r6d_ik_fk(const float* r6d, long N, long rowStride, long rowOffset, float* pose, const float* bone_dev,
                         const int* parent_dev, float* rglobal, float* joint, hipStream_t s);
// boneStride: 0 = one body (bone_dev [24][3]) for all frames, 72 = frame n uses bone_dev + n*72 (per-frame shapes)
void mp_launch_fk(const float* pose, const float* tran, long N, const float* bone_dev, const int* parent_dev,
                  const int* depth_dev, float* rglobal, float* joint, hipStream_t s, long boneStride = 0);

// linear blend skinning on mp_fk's outputs (joint already translated by tran); any N (chunked inside);
// jrestStride / vrestStride: floats between the rest joints / rest vertices of consecutive frames (0 = shared body)
void mp_launch_lbs(const float* rglobal, const float* joint, const float* tran, long N, const float* jrest_dev,
                   long jrestStride, const float* vrest_dev, long vrestStride, const float* weights_dev, int V,
                   float* vert, hipStream_t s);
// zero-pose bodies of ns shapes (articulate/model.py:84-89): vrest [ns][V][3] and jrest [ns][24][3] root-aligned,
// bone [ns][24][3]; jraw [ns][24][3] is scratch
void mp_launch_shape_body(const float* shape, int ns, const float* shapedirs, const float* vtemplate_raw,
                          const float* jreg, const int* parent_dev, int V, float* vrest, float* jraw, float* jrest,
                          float* bone, hipStream_t s);

// pose blend shapes (articulate/model.py:236-238): vposed [N][V][3] = vrest[(n)] + posedirs . (pose[1:] - I); posedirsT is
// the [207][3V] transpose made by mp_launch_transpose_posedirs from the pickle's [V][3][207]
void mp_launch_pose_blend(const float* pose, long N, const float* vrest, long vrestStride, const float* posedirsT, int V,
                          float* vposed, hipStream_t s);
void mp_launch_transpose_posedirs(const float* src, int V, float* dst, hipStream_t s);

// ---------------------------------------------------------------- evaluator metrics (mp_eval.hip)
// FullMotionEvaluator.__call__ (articulate/evaluator.py:292-343) on FK outputs; v_p / v_t may be nullptr (no mesh);
// part: mp_eval_partial_doubles(V) doubles of scratch; table: [10][2] floats
size_t mp_eval_partial_doubles(int V);
void mp_launch_mask_pose(const float* in, float* out, long N, unsigned ignored, hipStream_t s);
void mp_launch_eval_metrics(const float* pose_p, const float* pose_t, const float* rg_p, const float* rg_t, const float* j_p,
                            const float* j_t, const float* v_p, const float* v_t, long N, int V, int fps, int align,
                            unsigned mask, double* part, float* table, hipStream_t s);

// ---------------------------------------------------------------- live front-end (mp_live.hip)
// raw sensor samples of S streams -> network input frames [S,60] (live_demo.py:213-236)
void mp_launch_live_frames(const float* quat, const float* acc, const float* smpl2imu, const float* device2bone,
                           const float* acc_off, unsigned keep, float acc_scale, int S, float* frames, hipStream_t s);

// ---------------------------------------------------------------- K6: translation solver
void mp_launch_translate_offline(const float* joints, const float* vel, const float* contact, const int* lengths,
                                 int B, int T, float floor_y, float* tran, hipStream_t s);
struct OnlineState {       // per-stream state of forward_online (net.py:59-64,205-208)
    float* last_foot;      // [S][2][3]
    double* root_y;        // [S]
    float* root_pos;       // [S][3]
};
// streaming window maintenance (net.py:175): window[s] = fresh[s] ? frame x45 : cat(window[s][1:], frame)
void mp_launch_window_push(float* window, const float* frames, uint8_t* fresh, int S, int W, hipStream_t s);
// several small device-to-device copies (4-byte words, 4-byte aligned) as one launch: mp_tran.hip
struct CopyJobs {
    static constexpr int kMax = 6;
    struct Job { void* dst; const void* src; unsigned words; } j[kMax] = {};
    int n = 0;
    void add(void* dst, const void* src, size_t bytes) { if (n < kMax) j[n++] = Job{dst, src, (unsigned)(bytes / 4)}; }
};
void mp_launch_copy_words(const CopyJobs& js, hipStream_t s);
// reset() for the masked streams (mask == nullptr: all): fresh = 1, root_y = 0, root_pos = 0 (net.py:84-88);
// velH/velC (optional) rows of the carried velocity state [2][S][256] are zeroed as well
void mp_launch_stream_reset(const uint8_t* mask, uint8_t* fresh, double* root_y, float* root_pos, float* velH,
                            float* velC, int S, hipStream_t s);
// replay of N consecutive forward_online calls of one stream (mp_stream_replay): the frame history (the stream's 45-frame window
// or, for a fresh stream, 45 copies of the first frame, then the N frames: window k = hist[k+1 .. k+45]), the window the stream
// holds afterwards, and the serial solver over the N frames (joints / contact in batch layout [N][T][.], vel [N][72])
void mp_launch_replay_history(const float* window, const uint8_t* fresh, const float* frames, int N, int W, float* hist, hipStream_t s);
void mp_launch_replay_window(const float* hist, int N, int W, float* window, uint8_t* fresh, hipStream_t s);
void mp_launch_translate_replay(const float* joints, const float* vel, const float* contact, int N, int T, int idx, float floor_y,
                                OnlineState st, float* root_pos_out, float* contact_out, hipStream_t s);
void mp_launch_translate_online(const float* joints, const float* vel, const float* contact, int S, int T, int idx,
                                float floor_y, OnlineState st, float* root_pos_out, float* contact_out,
                                hipStream_t s);
