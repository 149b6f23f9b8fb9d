// Workspaces ("plans") of a handle by capacity class, the lengths upload, and the event-timed segments behind mp_timing_*.
#include "mp_host.h"

namespace mph {

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

static int alloc_plan_workspaces(mp_handle* h, Plan* p) {
    const size_t M = p->capRows, CB = (size_t)p->capB;
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
        // exchange areas: one per (direction, slab) -- or per (direction, sequence) where a few sequences run on the one-sequence kernels
        const size_t units = CB <= (size_t)kSeqClusterMax ? CB : (CB + 15) / 16;
        w.hx_bytes = (size_t)2 * units * ((size_t)4 * 16 * m.H + 16) * sizeof(unsigned long long);
        if (int rc = dev_alloc(h, (void**)&w.hx, w.hx_bytes, &p->allocs)) return rc;
        if (m.H == 256)
            if (int rc = dev_alloc(h, (void**)&w.hx2, w.hx_bytes, &p->allocs)) return rc;
    }
    if (int rc = dev_alloc(h, (void**)&p->r6d, M * 96 * sizeof(float), &p->allocs)) return rc;
    if (int rc = dev_alloc(h, (void**)&p->lengths_dev, CB * sizeof(int), &p->allocs)) return rc;
    HIPCHK(h, hipHostMalloc((void**)&p->lengths_pin, CB * sizeof(int), hipHostMallocDefault));
    return MP_OK;
}

// `keep`: a plan the caller is still using (mp_stream_replay holds two): never the victim of this call's eviction
int get_plan(mp_handle* h, int B, int T, Plan** out, const Plan* keep) {
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
    // (a plan enters the cache only when ALL of its workspaces exist: one that ran out of memory half-way would be handed to
    //  the next call of its class with null buffers)
    if (int rc = alloc_plan_workspaces(h, p)) {
        for (void* q : p->allocs) (void)hipFree(q);
        if (p->lengths_pin) (void)hipHostFree(p->lengths_pin);
        delete p;
        (void)hipGetLastError();                   // (hipMalloc's failure is not the next launch check's business)
        return rc;
    }
    h->plans.push_back(p);
    ++h->plan_allocs;
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

}  // namespace mph

// ================================================================================================ C ABI
extern "C" {

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

int mp_debug_plan_stats(mp_handle* h, int* n_plans, int* n_allocs, long long* cap_rows) {
    if (!h) return MP_ERR_INVALID;
    if (n_plans) *n_plans = (int)h->plans.size();
    if (n_allocs) *n_allocs = h->plan_allocs;
    if (cap_rows) { long long r = 0; for (const Plan* q : h->plans) r += (long long)q->capRows; *cap_rows = r; }
    return MP_OK;
}


}  // extern "C"
