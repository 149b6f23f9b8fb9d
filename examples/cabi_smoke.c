/* Plain C consumer of libmobileposer_hip.so: no Python, no torch -- the drop-in boundary of INTEGRATION.md used from C.
 *
 *   gcc -O2 -std=c11 examples/cabi_smoke.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -Lmobileposer_amd -lmobileposer_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/mobileposer_amd -o cabi_smoke
 *   ./cabi_smoke            (needs an MI355X)
 *
 * Builds a model from pseudo-random weights of the manifest's size, runs MobilePoserNet.forward_offline on a ragged
 * batch through the C ABI, and checks the structural properties the reference guarantees: unit-norm orthonormal
 * local rotations, identity on the ignored joints, translation constant after a sequence's end, error codes.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mobileposer_hip.h"

#define CHECK_HIP(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_MP(e) do { int rc_ = (e); if (rc_ != MP_OK) { fprintf(stderr, "mp error %d: %s (%s:%d)\n", rc_, mp_last_error(h), __FILE__, __LINE__); return 3; } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float frand(void) {                       /* xorshift64*, uniform in [-1, 1) */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (float)((rng_state * 0x2545F4914F6CDD1Dull) >> 40) / 8388608.0f - 1.0f;
}

int main(void) {
    mp_handle* h = NULL;
    const size_t nw = mp_weight_count();
    float* w = (float*)malloc(nw * sizeof(float));
    for (size_t i = 0; i < nw; ++i) w[i] = 0.06f * frand();       /* ~ U(-1/sqrt(H), 1/sqrt(H)) like nn.LSTM's init */
    const int32_t parent[24] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};
    float J[72];
    for (int i = 0; i < 24; ++i) { J[3 * i] = 0.1f * frand(); J[3 * i + 1] = 0.9f - 0.04f * i; J[3 * i + 2] = 0.05f * frand(); }
    J[31] = -0.95f; J[34] = -0.96f;                                /* feet are the lowest joints */

    /* error behaviour first: wrong blob size */
    if (mp_create(&h, 0, w, nw - 1, parent, J) != MP_ERR_INVALID) { fprintf(stderr, "expected MP_ERR_INVALID\n"); return 1; }
    printf("mp_create(bad size) -> MP_ERR_INVALID: %s\n", mp_last_error(NULL));
    CHECK_MP(mp_create(&h, 0, w, nw, parent, J));

    enum { B = 20, T = 40 };
    int32_t lengths[B];
    for (int b = 0; b < B; ++b) lengths[b] = T;
    lengths[3] = 7; lengths[11] = 1; lengths[19] = 39;
    const size_t n_imu = (size_t)B * T * 60, n_pose = (size_t)B * T * 216, n72 = (size_t)B * T * 72;
    float* imu_h = (float*)malloc(n_imu * sizeof(float));
    for (size_t i = 0; i < n_imu; ++i) imu_h[i] = 0.5f * frand();
    float *imu, *pose, *joints, *vel, *contact, *tran;
    CHECK_HIP(hipMalloc((void**)&imu, n_imu * 4)); CHECK_HIP(hipMalloc((void**)&pose, n_pose * 4));
    CHECK_HIP(hipMalloc((void**)&joints, n72 * 4)); CHECK_HIP(hipMalloc((void**)&vel, n72 * 4));
    CHECK_HIP(hipMalloc((void**)&contact, (size_t)B * T * 2 * 4)); CHECK_HIP(hipMalloc((void**)&tran, (size_t)B * T * 3 * 4));
    CHECK_HIP(hipMemcpy(imu, imu_h, n_imu * 4, hipMemcpyHostToDevice));

    lengths[5] = 0;                                                /* invalid length -> MP_ERR_LENGTHS, nothing launched */
    if (mp_forward_offline(h, imu, lengths, B, T, pose, joints, vel, contact, tran, NULL, NULL, NULL) != MP_ERR_LENGTHS) {
        fprintf(stderr, "expected MP_ERR_LENGTHS\n"); return 1; }
    lengths[5] = T;
    for (int rep = 0; rep < 3; ++rep) {                            /* first call captures the graph, later ones replay it */
        CHECK_MP(mp_reset_state(h, 1));
        CHECK_MP(mp_forward_offline(h, imu, lengths, B, T, pose, joints, vel, contact, tran, NULL, NULL, NULL));
    }
    CHECK_HIP(hipDeviceSynchronize());
    int derr = -1;
    CHECK_MP(mp_device_error(h, &derr));
    if (derr != 0) { fprintf(stderr, "device error %d\n", derr); return 1; }

    float* pose_h = (float*)malloc(n_pose * 4);
    float* tran_h = (float*)malloc((size_t)B * T * 3 * 4);
    CHECK_HIP(hipMemcpy(pose_h, pose, n_pose * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(tran_h, tran, (size_t)B * T * 3 * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < lengths[b]; ++t)
            for (int j = 0; j < 24; ++j) {
                const float* R = pose_h + (((size_t)b * T + t) * 24 + j) * 9;
                for (int a = 0; a < 3; ++a)
                    for (int c = 0; c < 3; ++c) {                  /* R^T R = I */
                        double s = 0.0;
                        for (int k = 0; k < 3; ++k) s += (double)R[3 * k + a] * R[3 * k + c];
                        const double e = fabs(s - (a == c ? 1.0 : 0.0));
                        if (e > worst) worst = e;
                    }
                if (j == 7 || j == 8 || j == 10 || j == 11 || j >= 20)   /* joint_set.ignored: identity */
                    for (int k = 0; k < 9; ++k)
                        if (R[k] != (k % 4 == 0 ? 1.0f : 0.0f)) { fprintf(stderr, "ignored joint %d not identity\n", j); return 1; }
            }
    printf("orthonormality of %d x %d x 24 local rotations: max |R^T R - I| = %.2e\n", B, T, worst);
    if (worst > 1e-3) return 1;                                   /* (random weights: some 6D rows are short, fp32 Gram-Schmidt) */
    for (int t = lengths[3]; t < T; ++t)                           /* padded frames keep the last translation */
        for (int k = 0; k < 3; ++k)
            if (tran_h[((size_t)3 * T + t) * 3 + k] != tran_h[((size_t)3 * T + lengths[3] - 1) * 3 + k]) {
                fprintf(stderr, "translation changed after the end of sequence 3\n"); return 1; }
    /* ParametricModel.forward_kinematics (articulate/model.py:208-232) and inverse_kinematics_R (:146-164) through the ABI:
     * IK(FK(local pose)) gives the local pose back */
    {
        const long N = (long)B * T;
        float *rglob = NULL, *jglob = NULL, *back = NULL;
        CHECK_HIP(hipMalloc((void**)&rglob, n_pose * 4));
        CHECK_HIP(hipMalloc((void**)&jglob, (size_t)N * 72 * 4));
        CHECK_HIP(hipMalloc((void**)&back, n_pose * 4));
        CHECK_MP(mp_fk(h, pose, NULL, N, rglob, jglob, NULL));
        CHECK_MP(mp_inverse_kinematics_r(h, rglob, N, back, NULL));
        if (mp_inverse_kinematics_r(h, rglob, N, rglob, NULL) != MP_ERR_INVALID) { fprintf(stderr, "in-place IK accepted\n"); return 1; }
        CHECK_HIP(hipDeviceSynchronize());
        float* back_h = (float*)malloc(n_pose * 4);
        CHECK_HIP(hipMemcpy(back_h, back, n_pose * 4, hipMemcpyDeviceToHost));
        double werr = 0.0;
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < lengths[b]; ++t)
                for (int k = 0; k < 216; ++k) {
                    const size_t i = ((size_t)b * T + t) * 216 + k;
                    const double e = fabs((double)back_h[i] - (double)pose_h[i]);
                    if (e > werr) werr = e;
                }
        printf("IK(FK(pose)) - pose: max %.2e\n", werr);
        if (werr > 1e-3) return 1;                                 /* (as above: rotations orthonormal to 1e-3 only) */
        free(back_h);
        (void)hipFree(rglob); (void)hipFree(jglob); (void)hipFree(back);
    }
    mp_destroy(h);
    printf("cabi_smoke: ok\n");
    return 0;
}
