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

#ifdef __cplusplus
}
#endif
#endif /* MOBILEPOSER_HIP_INTERNAL_H */
