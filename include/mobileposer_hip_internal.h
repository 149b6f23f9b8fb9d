/*
 * mobileposer_hip_internal.h -- test and debug hooks of libmobileposer_hip.so.  NOT part of the drop-in boundary
 * (include/mobileposer_hip.h): tests/ and tools/ bind these, the product facade does not need them.
 */
#ifndef MOBILEPOSER_HIP_INTERNAL_H
#define MOBILEPOSER_HIP_INTERNAL_H

#include "mobileposer_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook for the hidden-state exchange of the persistent kernels: 0 = pick the transport per producer from
 * its real XCC id (default), 1 = always use the any-placement write-through (sc1) transport. */
int mp_set_transport(mp_handle* h, int force_remote);
/* Test hook: make a kernel store `code` into the error word exactly as a timed-out persistent kernel would. */
int mp_debug_poke_error(mp_handle* h, int code);
/* Debug (env MP_PERSIST_PROF=1 at mp_create): per-workgroup cycle sums [grid][6] of the phases of the last
 * persistent-kernel launch: wait, sweep, mfma, reduce, cell+publish, steps. */
int mp_debug_read_prof(mp_handle* h, long long* out, int n_words);

/* Test hook: after `skip` further fused-LSTM layer launches, in each of the next `launches` ones, workgroup `block` exits at once -- exactly what its
 * cluster sees when a workgroup of the grid never becomes resident (a GPU shared with another process): the peers' waits
 * run into their time bound, poison the slab and raise the error word.  Deterministic stand-in for real starvation. */
int mp_debug_drop_workgroup(mp_handle* h, int block, int skip, int launches);

/* Measurement hook (round 6): the shader clock (MHz) a one-wave probe kernel on the handle's main stream ran at -- ticks of
 * s_memtime per tick of the constant 100 MHz s_memrealtime -- and how long the probe took.  Synchronises.  What
 * `bench.py --workload stream --cadence-hz` samples around the first ticks after an idle stretch. */
int mp_debug_clock_probe(mp_handle* h, double* shader_mhz, double* probe_us);
/* The same two clocks under load (round 6): n_cu workgroups x 4 waves each issue 4 * iters independent v_mfma_f32_32x32x2_f32
 * between their looks at the clocks.  mean / minimum over the waves of the shader MHz, mean / maximum of the time a wave took
 * (us; the work is fixed, so this is the speed of the matrix pipes whatever the counters say).  Synchronises. */
int mp_debug_clock_probe_loaded(mp_handle* h, int iters, double* mhz_mean, double* mhz_min, double* us_mean, double* us_max);
/* Test hook (round 6): the workspace plans of the handle -- how many exist, how many were ever allocated, their capacity in rows
 * (B * T) together.  Plans are kept by capacity class (mp_plans.hip get_plan): a caller that walks through sequence lengths must
 * not allocate per length. */
int mp_debug_plan_stats(mp_handle* h, int* n_plans, int* n_allocs, long long* cap_rows);

#ifdef __cplusplus
}
#endif
#endif /* MOBILEPOSER_HIP_INTERNAL_H */
