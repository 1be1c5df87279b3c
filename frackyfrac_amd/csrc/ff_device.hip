// ff_device.hip -- the C ABI of the device path (include/frackyfrac_amd.h, sections 1 and "stage A on the device"):
// plans (create / set shard / run / destroy), unifracDists as one call and as a lazy ordered sequence, device buffers
// shared between the processes of one node, and the C++ helpers of the frcfrc command.  One of the three translation
// units of the device path (ff_plan.hpp); no kernel is launched from here.
#include <chrono>

#include "ff_plan.hpp"

#include <functional>

using namespace ff::dev;

namespace ff {
namespace dev {

int set_shard_geometry(ff_plan *pl, int32_t rank, int32_t world, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    int rc = ff_shard_rows(inf.n_samples, rank, world, &inf.row_begin, &inf.row_end);
    if (rc) return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", rank, world);
    inf.slot_begin = inf.row_begin > 0 ? inf.row_begin * (inf.row_begin - 1) / 2 : 0;
    inf.slot_end = inf.row_end > 0 ? inf.row_end * (inf.row_end - 1) / 2 : 0;
    return FF_OK;
}

}  // namespace dev
}  // namespace ff

namespace ff {
namespace dev {

int validate_problem(const ff_problem *p, char *err, size_t errlen, bool unsorted_ok)
{
    if (!p) return ff::fail(FF_ERR_ARG, err, errlen, "null problem");
    if (p->n_samples < 0 || p->n_branches < 0 || p->n_branches > (int64_t)INT32_MAX - 64)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad problem size N=%lld B=%lld",
                        (long long)p->n_samples, (long long)p->n_branches);
    if (p->n_samples > 0 && !p->indptr) return ff::fail(FF_ERR_ARG, err, errlen, "null indptr");
    if (p->n_branches > 0 && !p->branch_len) return ff::fail(FF_ERR_ARG, err, errlen, "null branch_len");
    if (p->n_samples > (int64_t)1 << 22)
        return ff::fail(FF_ERR_ARG, err, errlen, "too many samples (%lld)", (long long)p->n_samples);
    if (p->n_samples == 0) return FF_OK;
    if (p->indptr[0] != 0) return ff::fail(FF_ERR_ARG, err, errlen, "indptr[0] != 0");
    for (int64_t s = 0; s < p->n_samples; ++s)
        if (p->indptr[s + 1] < p->indptr[s])
            return ff::fail(FF_ERR_ARG, err, errlen, "indptr not monotone at sample %lld", (long long)s);
    // the flat nodes themselves: 21 M of them at C3 -- on a few host threads (this pass and the weights below were
    // 40 of the 50 ms of a one-shot ff_unifrac_dists there); the first offending sample in sample order is reported
    const unsigned nt = host_threads(p->indptr[p->n_samples]);
    std::vector<int64_t> bad_sample(nt, -1);
    std::vector<int> bad_kind(nt, 0);
    std::vector<int32_t> bad_id(nt, 0);
    ff::parallel_for(p->n_samples, nt, [&](unsigned th, int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1 && bad_sample[th] < 0; ++s) {
            const int64_t b = p->indptr[s], e = p->indptr[s + 1];
            for (int64_t t = b; t < e; ++t) {
                const int32_t id = p->branch_id[t];
                int kind = 0;
                if (id < 0 || id >= p->n_branches) kind = 1;
                else if (!unsorted_ok && t > b && id <= p->branch_id[t - 1]) kind = 2;
                else if (!(p->abnd[t] > 0) || !std::isfinite(p->abnd[t])) kind = 3;
                if (kind) {
                    bad_sample[th] = s;
                    bad_kind[th] = kind;
                    bad_id[th] = id;
                    break;
                }
            }
        }
    });
    for (unsigned th = 0; th < nt; ++th) {  // (threads hold ascending sample ranges)
        if (bad_sample[th] < 0) continue;
        const long long s = (long long)bad_sample[th];
        if (bad_kind[th] == 1) return ff::fail(FF_ERR_ARG, err, errlen, "sample %lld: branch id %d out of range", s, bad_id[th]);
        if (bad_kind[th] == 2) return ff::fail(FF_ERR_ARG, err, errlen, "sample %lld: branch ids not strictly ascending", s);
        return ff::fail(FF_ERR_ARG, err, errlen, "sample %lld: abundance must be finite and > 0", s);
    }
    return FF_OK;
}

}  // namespace dev
}  // namespace ff

namespace {

void plan_free_device(ff_plan *pl)
{
    if (!pl) return;
    (void)hipFree(pl->d_QT);
    (void)hipFree(pl->d_W);
    (void)hipFree(pl->d_num);
    (void)hipFree(pl->d_items);
    (void)hipFree(pl->d_item_ptr);
    (void)hipFree(pl->d_stamps);
    (void)hipFree(pl->d_arows);
    (void)hipFree(pl->d_aptr16);
    (void)hipFree(pl->d_cs16);
    (void)hipFree(pl->d_low_ptr);
    (void)hipFree(pl->d_low_ent);
    (void)hipFree(pl->d_low_bits);
    (void)hipFree(pl->d_Wl);
    (void)hipFree(pl->d_mlow);
    (void)hipFree(pl->d_low_tiles);
    (void)hipFree(pl->d_indptr);
    (void)hipFree(pl->d_ids);
    (void)hipFree(pl->d_abnd);
    (void)hipFree(pl->d_refine_list);
    (void)hipFree(pl->d_refine_count);
    (void)hipFree(pl->d_risk_list);
    (void)hipFree(pl->d_n_nodes);
    (void)hipFree(pl->d_wex);
    (void)hipFree(pl->d_audit_slots);
    (void)hipFree(pl->d_audit_exact);
    (void)hipFree(pl->d_Pbits);
    (void)hipFree(pl->d_Kd);
    (void)hipFree(pl->d_Kt);
    (void)hipFree(pl->d_mitems);
    (void)hipFree(pl->d_mitem_ptr);
    (void)hipFree(pl->d_partial);
    (void)hipFree(pl->d_ptiles);
    (void)hipFree(pl->d_ptile_ptr);
    (void)hipFree(pl->d_DT);
    (void)hipFree(pl->d_len);
    (void)hipFree(pl->d_len_rows);
    (void)hipFree(pl->d_host_out);
    (void)hipFree(pl->d_xtiles);
    (void)hipFree(pl->d_Xbits);
    (void)hipFree(pl->d_xutiles);
    for (auto &e : pl->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
        (void)hipEventDestroy(e.mid);
    }
}

int plan_begin(const ff_options *o, int64_t N, int64_t B, ff_plan *pl, hipDeviceProp_t *prop, char *err,
               size_t errlen)
{
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0)
        return ff::fail(FF_ERR_DEVICE, err, errlen,
                        "no HIP device available (%s); this engine has no CPU path",
                        he != hipSuccess ? hipGetErrorString(he) : "device count is 0");
    if (o->device >= 0) {
        if (o->device >= ndev) return ff::fail(FF_ERR_DEVICE, err, errlen, "device %d of %d", o->device, ndev);
        FF_HIP(hipSetDevice(o->device));
    }
    FF_HIP(hipGetDevice(&pl->device));
    (void)hipGetLastError();  // a stale error of the caller's (or of a failed plan) is not ours
    FF_HIP(hipGetDeviceProperties(prop, pl->device));
    if (strncmp(prop->gcnArchName, "gfx950", 6) != 0)
        return ff::fail(FF_ERR_DEVICE, err, errlen, "device %d is %s; this engine is built for gfx950 only",
                        pl->device, prop->gcnArchName);
    pl->weighted = o->weighted != 0;
    ff_plan_info &inf = pl->info;
    inf.audit_min_headroom = INFINITY;
    inf.active_fraction = 1.0;
    inf.n_samples = N;
    inf.n_branches = B;
    inf.n_compute_units = prop->multiProcessorCount;
    pl->shard_rank = o->rank;
    pl->shard_world = o->world;
    return set_shard_geometry(pl, o->rank, o->world, err, errlen);
}

int plan_create_impl(const ff_problem *p, const ff_options *o, ff_plan *pl, char *err, size_t errlen)
{
    hipDeviceProp_t prop;
    int rc = plan_begin(o, p->n_samples, p->n_branches, pl, &prop, err, errlen);
    if (rc) return rc;
    DeviceCsr c;
    rc = csr_from_host(p, &c, err, errlen);
    if (rc == FF_OK) rc = plan_build(o, &c, prop, pl, err, errlen);
    c.release();
    return rc;
}

// The audit's verdict on the run that has just completed, into an ff_plan_info that is handed back to a caller.
void fill_audit_info(ff_plan *pl, ff_plan_info *info)
{
    info->audit_checked = info->audit_failed = 0;
    info->audit_worst_rel_err = 0.0;
    info->audit_min_headroom = INFINITY;
    int64_t uniform = 0, found = 0, chk = 0;
    (void)ff_plan_audit(pl, &info->audit_checked, &info->audit_failed, &info->audit_worst_rel_err);
    (void)ff_plan_audit_detail(pl, &uniform, &found, &chk, &info->audit_min_headroom);
}


}  // namespace

extern "C" {

int ff_plan_create(const ff_problem *p, const ff_options *o, ff_plan **plan, char *err, size_t errlen)
{
    if (!plan) return ff::fail(FF_ERR_ARG, err, errlen, "null plan pointer");
    *plan = nullptr;
    ff_options dflt;
    ff_options_default(&dflt);
    if (!o) o = &dflt;
    if (o->world < 1 || o->rank < 0 || o->rank >= o->world)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", o->rank, o->world);
    if (o->precision < FF_PRECISION_AUTO || o->precision > FF_PRECISION_EXACT64)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad precision %d", o->precision);
    if ((o->flags & ~FF_FLAG_UNSORTED_WALK) != 0) return ff::fail(FF_ERR_ARG, err, errlen, "unknown flags %d", o->flags);
    int rc = validate_problem(p, err, errlen, (o->flags & FF_FLAG_UNSORTED_WALK) != 0);
    if (rc) return rc;
    auto *pl = new ff_plan();
    rc = plan_create_impl(p, o, pl, err, errlen);
    if (rc) {
        plan_free_device(pl);
        delete pl;
        return rc;
    }
    *plan = pl;
    return FF_OK;
}

void ff_plan_destroy(ff_plan *pl)
{
    if (!pl) return;
    plan_free_device(pl);
    delete pl;
}

int ff_plan_info_get(const ff_plan *pl, ff_plan_info *info)
{
    if (!pl || !info) return FF_ERR_ARG;
    *info = pl->info;
    return FF_OK;
}

int ff_plan_set_shard(ff_plan *pl, int32_t rank, int32_t world, char *err, size_t errlen)
{
    if (!pl) return ff::fail(FF_ERR_ARG, err, errlen, "null plan");
    if (world < 1 || rank < 0 || rank >= world) return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", rank, world);
    int cur = -1;
    FF_HIP(hipGetDevice(&cur));
    if (cur != pl->device) FF_HIP(hipSetDevice(pl->device));
    FF_HIP(hipDeviceSynchronize());  // runs of the old shard may still read the schedule
    int rc = set_shard_geometry(pl, rank, world, err, errlen);
    if (rc == FF_OK) {
        pl->shard_rank = rank;
        pl->shard_world = world;
        rc = schedule_for_shard(pl, err, errlen);
    }
    if (rc == FF_OK) FF_HIP(hipDeviceSynchronize());
    if (cur != pl->device) (void)hipSetDevice(cur);
    return rc;
}

int ff_plan_run(ff_plan *pl, void *stream, double *d_out, char *err, size_t errlen)
{
    if (!pl) return ff::fail(FF_ERR_ARG, err, errlen, "null plan");
    return plan_run_impl(pl, (hipStream_t)stream, d_out, false, err, errlen);
}

int ff_plan_run_host(ff_plan *pl, double *out, char *err, size_t errlen)
{
    if (!pl) return ff::fail(FF_ERR_ARG, err, errlen, "null plan");
    const int64_t n_slots = pl->info.slot_end - pl->info.slot_begin;
    if (n_slots <= 0) return FF_OK;
    if (!out) return ff::fail(FF_ERR_ARG, err, errlen, "null output pointer");
    int cur = -1;
    FF_HIP(hipGetDevice(&cur));
    if (cur != pl->device) FF_HIP(hipSetDevice(pl->device));
    if (n_slots > pl->host_out_cap) {
        (void)hipFree(pl->d_host_out);
        pl->d_host_out = nullptr;
        pl->host_out_cap = 0;
        FF_ALLOC(pl->d_host_out, sizeof(double) * (size_t)n_slots, "the results");
        pl->host_out_cap = n_slots;
    }
    int rc = plan_run_impl(pl, nullptr, pl->d_host_out, false, err, errlen);
    if (rc == FF_OK) {
        FF_HIP(hipMemcpy(out, pl->d_host_out, sizeof(double) * (size_t)n_slots, hipMemcpyDeviceToHost));
        bool ok = true;
        std::string why;
        if (plan_fixed32_verdict(pl, &ok, &why) == FF_OK && !ok)
            rc = ff::fail(FF_ERR_PRECISION, err, errlen, "%s: stage this problem with FF_PRECISION_EXACT64", why.c_str());
    }
    if (cur != pl->device) (void)hipSetDevice(cur);
    return rc;
}

int ff_plan_run_timed(ff_plan *pl, void *stream, double *d_out, char *err, size_t errlen)
{
    if (!pl) return ff::fail(FF_ERR_ARG, err, errlen, "null plan");
    return plan_run_impl(pl, (hipStream_t)stream, d_out, true, err, errlen);
}

// ---- device buffers shared between processes (include/frackyfrac_amd.h, "Device buffers ...") ----

static_assert(sizeof(hipIpcMemHandle_t) == sizeof(ff_ipc_handle), "ff_ipc_handle must hold a hipIpcMemHandle_t");

int ff_device_alloc(int32_t device, size_t bytes, void **dptr, char *err, size_t errlen)
{
    if (!dptr) return ff::fail(FF_ERR_ARG, err, errlen, "null pointer argument");
    *dptr = nullptr;
    int cur = -1;
    FF_HIP(hipGetDevice(&cur));
    if (device >= 0 && device != cur) FF_HIP(hipSetDevice(device));
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 1));
    if (e == hipSuccess) e = hipMemset(p, 0, std::max<size_t>(bytes, 1));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (device >= 0 && device != cur) (void)hipSetDevice(cur);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot allocate %.2f GB of device memory: %s", (double)bytes / 1e9,
                        hipGetErrorString(e));
    }
    *dptr = p;
    return FF_OK;
}

int ff_device_free(void *dptr, char *err, size_t errlen)
{
    if (dptr) FF_HIP(hipFree(dptr));
    return FF_OK;
}

int ff_ipc_export(void *dptr, ff_ipc_handle *handle, char *err, size_t errlen)
{
    if (!dptr || !handle) return ff::fail(FF_ERR_ARG, err, errlen, "null pointer argument");
    hipIpcMemHandle_t h;
    FF_HIP(hipIpcGetMemHandle(&h, dptr));
    memcpy(handle->bytes, &h, sizeof h);
    return FF_OK;
}

int ff_ipc_open(const ff_ipc_handle *handle, int32_t device, void **dptr, char *err, size_t errlen)
{
    if (!dptr || !handle) return ff::fail(FF_ERR_ARG, err, errlen, "null pointer argument");
    *dptr = nullptr;
    int cur = -1;
    FF_HIP(hipGetDevice(&cur));
    if (device >= 0 && device != cur) FF_HIP(hipSetDevice(device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle->bytes, sizeof h);
    void *p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (device >= 0 && device != cur) (void)hipSetDevice(cur);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
    }
    *dptr = p;
    return FF_OK;
}

int ff_ipc_close(void *dptr, char *err, size_t errlen)
{
    if (dptr) FF_HIP(hipIpcCloseMemHandle(dptr));
    return FF_OK;
}

int ff_device_copy_async(void *dst, const void *src, size_t bytes, void *stream, char *err, size_t errlen)
{
    if (bytes == 0) return FF_OK;
    if (!dst || !src) return ff::fail(FF_ERR_ARG, err, errlen, "null pointer argument");
    FF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FF_OK;
}

// Diagnostics, not part of the public header: copies the per-wave start/end stamps
// of the last pair_sad_kernel launch (FF_STAMPS=1) into out[4 * n_wave_slots]: 2 per wave on the 100 MHz
// wall clock, then 2 per wave on the shader clock.
int ff_debug_read_stamps(ff_plan *pl, unsigned long long *out)
{
    if (!pl || !pl->d_stamps || !out) return FF_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return FF_ERR_DEVICE;
    if (hipMemcpy(out, pl->d_stamps, sizeof(unsigned long long) * 4 * (size_t)pl->info.n_wave_slots,
                  hipMemcpyDeviceToHost) != hipSuccess)
        return FF_ERR_DEVICE;
    return FF_OK;
}

int ff_plan_timing_collect_parts(ff_plan *pl, double *total_ms, double *rare_ms, int32_t *launches)
{
    if (!pl || !total_ms || !rare_ms || !launches) return FF_ERR_ARG;
    double sum = 0, rare = 0;
    int32_t n = 0;
    const bool parts = pl->split && pl->n_low_tiles > 0;  // (the runs since the last collect are all of this shard's schedule)
    for (size_t k = 0; k < pl->events_used; ++k) {
        float ms = 0.f, low = 0.f;
        if (hipEventSynchronize(pl->events[k].second) != hipSuccess ||
            hipEventElapsedTime(&ms, pl->events[k].first, pl->events[k].second) != hipSuccess)
            return FF_ERR_DEVICE;
        if (parts && hipEventElapsedTime(&low, pl->events[k].mid, pl->events[k].second) != hipSuccess) return FF_ERR_DEVICE;
        sum += (double)ms;
        rare += (double)low;
        ++n;
    }
    pl->events_used = 0;
    *total_ms = sum;
    *rare_ms = rare;
    *launches = n;
    return FF_OK;
}

int ff_plan_timing_collect(ff_plan *pl, double *total_ms, int32_t *launches)
{
    double rare = 0;
    return ff_plan_timing_collect_parts(pl, total_ms, &rare, launches);
}

int ff_unifrac_dists(const ff_problem *p, const ff_options *o, double *out, char *err, size_t errlen)
{
    return ff::unifrac_dists_info(p, o, out, nullptr, err, errlen);
}

// ---- flat-argument forms (hosts whose FFI may not pass a struct of managed pointers: cgo) ----

static ff_problem problem_of(int64_t n_samples, int64_t n_branches, const double *branch_len, const int64_t *indptr,
                             const int32_t *branch_id, const double *abnd)
{
    ff_problem p;
    p.n_samples = n_samples;
    p.n_branches = n_branches;
    p.branch_len = branch_len;
    p.indptr = indptr;
    p.branch_id = branch_id;
    p.abnd = abnd;
    return p;
}

int ff_plan_create_csr(int64_t n_samples, int64_t n_branches, const double *branch_len, const int64_t *indptr,
                       const int32_t *branch_id, const double *abnd, const ff_options *o, ff_plan **plan, char *err,
                       size_t errlen)
{
    const ff_problem p = problem_of(n_samples, n_branches, branch_len, indptr, branch_id, abnd);
    return ff_plan_create(&p, o, plan, err, errlen);
}

int ff_unifrac_dists_csr(int64_t n_samples, int64_t n_branches, const double *branch_len, const int64_t *indptr,
                         const int32_t *branch_id, const double *abnd, const ff_options *o, double *out, char *err,
                         size_t errlen)
{
    const ff_problem p = problem_of(n_samples, n_branches, branch_len, indptr, branch_id, abnd);
    return ff_unifrac_dists(&p, o, out, err, errlen);
}

int ff_unifrac_dists_stream_csr(int64_t n_samples, int64_t n_branches, const double *branch_len, const int64_t *indptr,
                                const int32_t *branch_id, const double *abnd, const ff_options *o,
                                int64_t max_pairs_per_chunk, ff_dists_fn fn, void *user, char *err, size_t errlen)
{
    const ff_problem p = problem_of(n_samples, n_branches, branch_len, indptr, branch_id, abnd);
    return ff_unifrac_dists_stream(&p, o, max_pairs_per_chunk, fn, user, err, errlen);
}

// ---- unifracDists as a lazy ordered sequence (frcfrc/unifrac.go:209-228) ----

namespace {

// Two device result buffers and their host twins (pinned when the host lets us), one stream: sub-shard k + 1
// is reduced and copied out while the consumer is handed sub-shard k.
struct StreamBuffers {
    double *d[2] = {nullptr, nullptr}, *h[2] = {nullptr, nullptr};
    bool pinned[2] = {false, false};
    int64_t cap[2] = {0, 0};
    hipStream_t st = nullptr;
    ~StreamBuffers()
    {
        for (int b = 0; b < 2; ++b) {
            (void)hipFree(d[b]);
            if (pinned[b]) (void)hipHostFree(h[b]);
            else free(h[b]);
        }
        if (st) (void)hipStreamDestroy(st);
    }
    int reserve(int b, int64_t n, char *err, size_t errlen)
    {
        if (n <= cap[b]) return FF_OK;
        (void)hipFree(d[b]);
        d[b] = nullptr;
        if (pinned[b]) (void)hipHostFree(h[b]);
        else free(h[b]);
        h[b] = nullptr;
        cap[b] = 0;
        const size_t bytes = sizeof(double) * (size_t)n;
        if (hipMalloc(&d[b], bytes) != hipSuccess) {
            (void)hipGetLastError();
            return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot allocate %.2f GB for a sub-shard's results (a smaller "
                            "max_pairs_per_chunk makes it smaller)", (double)bytes / 1e9);
        }
        pinned[b] = hipHostMalloc(reinterpret_cast<void **>(&h[b]), bytes, hipHostMallocDefault) == hipSuccess;
        if (!pinned[b]) {
            (void)hipGetLastError();
            h[b] = static_cast<double *>(malloc(bytes));
            if (!h[b]) return ff::fail(FF_ERR_INTERNAL, err, errlen, "out of host memory for %.2f GB of results", (double)bytes / 1e9);
        }
        cap[b] = n;
        return FF_OK;
    }
};

}  // namespace

int ff_unifrac_dists_stream(const ff_problem *p, const ff_options *o, int64_t max_pairs, ff_dists_fn fn, void *user,
                            char *err, size_t errlen)
{
    if (!fn) return ff::fail(FF_ERR_ARG, err, errlen, "null callback");
    ff_options base;
    ff_options_default(&base);
    if (o) base = *o;
    if (base.world < 1 || base.rank < 0 || base.rank >= base.world)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", base.rank, base.world);
    if (max_pairs <= 0) max_pairs = (int64_t)1 << 25;
    int rc = validate_problem(p, err, errlen, (base.flags & FF_FLAG_UNSORTED_WALK) != 0);
    if (rc) return rc;
    int64_t rb = 0, re = 0;
    if (ff_shard_rows(p->n_samples, base.rank, base.world, &rb, &re) != FF_OK)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", base.rank, base.world);
    const int64_t shard_pairs = (re > 0 ? re * (re - 1) / 2 : 0) - (rb > 0 ? rb * (rb - 1) / 2 : 0);
    if (shard_pairs <= 0) return FF_OK;  // lazily: nothing to deliver, nothing staged
    // Sub-shard k of c of shard (rank, world) is shard rank * c + k of world * c: ff_shard_rows' boundaries
    // N sqrt(k / world) nest, so the sub-shards tile this shard's contiguous slot range exactly.  Shards are
    // whole 32-row blocks; a sub-shard that comes out above max_pairs is delivered in pieces.
    int64_t c = (shard_pairs + max_pairs - 1) / max_pairs;
    c = std::min<int64_t>(c, std::max<int64_t>(1, (re - rb + 31) / 32));
    c = std::min<int64_t>(c, (int64_t)INT32_MAX / base.world);
    ff_options o2 = base;
    o2.rank = (int32_t)(base.rank * c);
    o2.world = (int32_t)(base.world * c);
    ff_plan *pl = nullptr;
    rc = ff_plan_create(p, &o2, &pl, err, errlen);
    if (rc) return rc;
    struct PlanGuard {
        ff_plan *&pl;
        ~PlanGuard() { ff_plan_destroy(pl); }
    } guard{pl};
    StreamBuffers buf;
    if (hipStreamCreateWithFlags(&buf.st, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot create a stream");
    }
    auto deliver = [&](const double *h, int64_t slot0, int64_t n) -> bool {
        for (int64_t off = 0; off < n; off += max_pairs)
            if (!fn(user, slot0 + off, h + off, std::min(max_pairs, n - off))) return false;
        return true;
    };
    int prev = -1;  // buffer holding a finished sub-shard that has not been delivered yet
    int64_t prev_slot0 = 0, prev_n = 0;
    int cur = 0;
    for (int64_t k = 0; k < c; ++k) {
        if (k > 0) {
            o2.rank = (int32_t)(base.rank * c + k);
            rc = ff_plan_set_shard(pl, o2.rank, o2.world, err, errlen);
            if (rc) return rc;
        }
        const int64_t slot0 = pl->info.slot_begin, n = pl->info.slot_end - pl->info.slot_begin;
        if (n <= 0) continue;
        rc = buf.reserve(cur, n, err, errlen);
        if (rc) return rc;
        rc = plan_run_impl(pl, buf.st, buf.d[cur], false, err, errlen);
        if (rc) return rc;
        FF_HIP(hipMemcpyAsync(buf.h[cur], buf.d[cur], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, buf.st));
        bool go_on = true;
        if (prev >= 0) go_on = deliver(buf.h[prev], prev_slot0, prev_n);  // (the device works on sub-shard k meanwhile)
        prev = -1;
        FF_HIP(hipStreamSynchronize(buf.st));
        if (!go_on) return FF_OK;  // the consumer stopped: sub-shard k is dropped, the rest never computed
        bool ok = true;
        if (plan_fixed32_verdict(pl, &ok, nullptr) == FF_OK && !ok) {
            // replicates, or a failed audit (ff_plan_audit): this sub-shard again, and all later ones, in binary64
            ff_plan_destroy(pl);
            pl = nullptr;
            o2.precision = FF_PRECISION_EXACT64;
            rc = ff_plan_create(p, &o2, &pl, err, errlen);
            if (rc == FF_OK) rc = plan_run_impl(pl, buf.st, buf.d[cur], false, err, errlen);
            if (rc) return rc;
            FF_HIP(hipMemcpyAsync(buf.h[cur], buf.d[cur], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, buf.st));
            FF_HIP(hipStreamSynchronize(buf.st));
        }
        prev = cur;
        prev_slot0 = slot0;
        prev_n = n;
        cur ^= 1;
    }
    if (prev >= 0) (void)deliver(buf.h[prev], prev_slot0, prev_n);
    return FF_OK;
}

int ff_flatten_device(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr, const int64_t *leaf_idx,
                      const double *leaf_val, int leave_unnormalized, ff_flat **flat, char *err, size_t errlen)
{
    if (!tree || !leaf_ptr || !flat || n_samples < 0 || (leaf_ptr[n_samples] > 0 && (!leaf_idx || !leaf_val)))
        return ff::fail(FF_ERR_ARG, err, errlen, "ff_flatten_device: bad argument");
    if (leave_unnormalized == FF_L_REFERENCE)  // (the recursion's order is made on the host)
        return ff_flatten_leaf_csr(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized, flat, err, errlen);
    ff_options o;
    ff_options_default(&o);
    ff_plan tmp;  // only to run the device checks of plan_begin
    hipDeviceProp_t prop;
    int rc = plan_begin(&o, n_samples, (int64_t)tree->size.size(), &tmp, &prop, err, errlen);
    if (rc) return rc;
    DeviceCsr c;
    bool too_deep = false;
    rc = csr_from_leaves(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, !leave_unnormalized, &c, &too_deep, err,
                         errlen);
    if (too_deep) {
        c.release();
        return ff_flatten_leaf_csr(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized, flat, err,
                                   errlen);
    }
    if (rc) {
        c.release();
        return rc;
    }
    auto *f = new ff_flat();
    f->n_samples = c.N;
    f->n_branches = c.B;
    f->branch_len = c.h_len;
    f->indptr = c.h_indptr;
    f->branch_id.resize((size_t)c.nnz);
    f->abnd.resize((size_t)c.nnz);
    hipError_t he = hipSuccess;
    if (c.nnz > 0) {
        he = hipMemcpy(f->branch_id.data(), c.d_ids, sizeof(int32_t) * (size_t)c.nnz, hipMemcpyDeviceToHost);
        if (he == hipSuccess)
            he = hipMemcpy(f->abnd.data(), c.d_abnd, sizeof(double) * (size_t)c.nnz, hipMemcpyDeviceToHost);
    }
    c.release();
    if (he != hipSuccess) {
        delete f;
        return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: download of flat nodes failed: %s", hipGetErrorString(he));
    }
    *flat = f;
    return FF_OK;
}

int ff_plan_create_from_leaves(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr,
                               const int64_t *leaf_idx, const double *leaf_val, int leave_unnormalized,
                               const ff_options *o, ff_plan **plan, char *err, size_t errlen)
{
    if (!plan) return ff::fail(FF_ERR_ARG, err, errlen, "null plan pointer");
    *plan = nullptr;
    if (!tree || !leaf_ptr || n_samples < 0 || (leaf_ptr[n_samples] > 0 && (!leaf_idx || !leaf_val)))
        return ff::fail(FF_ERR_ARG, err, errlen, "ff_plan_create_from_leaves: bad argument");
    ff_options dflt;
    ff_options_default(&dflt);
    if (!o) o = &dflt;
    if (o->world < 1 || o->rank < 0 || o->rank >= o->world)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", o->rank, o->world);
    if (o->precision < FF_PRECISION_AUTO || o->precision > FF_PRECISION_EXACT64)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad precision %d", o->precision);
    const int64_t B = (int64_t)tree->size.size();
    if (B > (int64_t)INT32_MAX - 64 || n_samples > (int64_t)1 << 22)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad problem size N=%lld B=%lld", (long long)n_samples, (long long)B);
    if (leave_unnormalized < 0 || leave_unnormalized > FF_L_REFERENCE)
        return ff::fail(FF_ERR_ARG, err, errlen, "leave_unnormalized must be 0, 1 or FF_L_REFERENCE");
    // FF_L_REFERENCE: the lists as the reference's -l leaves them (unsorted) and the literal walk over them
    ff_options o_walk = *o;
    const bool reference_l = leave_unnormalized == FF_L_REFERENCE;
    if (reference_l) {
        o_walk.flags |= FF_FLAG_UNSORTED_WALK;
        o = &o_walk;
    }
    auto *pl = new ff_plan();
    hipDeviceProp_t prop;
    int rc = plan_begin(o, n_samples, B, pl, &prop, err, errlen);
    DeviceCsr c;
    if (rc == FF_OK) {
        bool too_deep = reference_l;  // (the recursion's order is made on the host)
        if (!reference_l)
            rc = csr_from_leaves(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, !leave_unnormalized, &c, &too_deep, err,
                                 errlen);
        if (too_deep) {  // a caterpillar: flatten on the host instead
            c.release();
            ff_flat *flat = nullptr;
            rc = ff_flatten_leaf_csr(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized, &flat, err,
                                     errlen);
            if (rc == FF_OK) {
                ff_problem p;
                ff_flat_problem(flat, &p);
                rc = csr_from_host(&p, &c, err, errlen);
                ff_flat_free(flat);
            }
        }
    }
    if (rc == FF_OK) rc = plan_build(o, &c, prop, pl, err, errlen);
    c.release();
    if (rc) {
        plan_free_device(pl);
        delete pl;
        return rc;
    }
    *plan = pl;
    return FF_OK;
}

}  // extern "C"

// Runs a freshly created plan, copies the shard's slots into out (host) and destroys the
// plan.  If FIXED32's refinement queue overflowed, `recreate` builds an EXACT64 plan and
// the shard is repeated with it.
int ff::run_plan_to_host(ff_plan *pl, const std::function<int(ff_plan **)> &recreate_exact64, double *out,
                         ff_plan_info *info_out, char *err, size_t errlen, bool shard_local)
{
    int rc = FF_OK;
    const int64_t n_slots = pl->info.slot_end - pl->info.slot_begin;
    if (n_slots > 0) {
        double *d_out = nullptr;
        hipError_t he = hipMalloc(&d_out, sizeof(double) * (size_t)n_slots);
        if (he != hipSuccess) {
            ff_plan_destroy(pl);
            return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: hipMalloc(out) failed: %s", hipGetErrorString(he));
        }
        rc = ff_plan_run(pl, nullptr, d_out, err, errlen);
        bool ok = true;
        if (rc == FF_OK && plan_fixed32_verdict(pl, &ok, nullptr) == FF_OK && !ok) {
            // more nearly-equal pairs than the refinement queue holds (the data set is mostly
            // replicates), or the audit sample missed its bar: run the whole shard in binary64
            ff_plan_destroy(pl);
            pl = nullptr;
            rc = recreate_exact64(&pl);
            if (rc == FF_OK) rc = ff_plan_run(pl, nullptr, d_out, err, errlen);
        }
        if (rc == FF_OK) {
            he = hipMemcpy(out + (shard_local ? 0 : pl->info.slot_begin), d_out, sizeof(double) * (size_t)n_slots,
                           hipMemcpyDeviceToHost);
            if (he != hipSuccess)
                rc = ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: copy of results failed: %s", hipGetErrorString(he));
        }
        (void)hipFree(d_out);
    }
    if (info_out && pl) {
        *info_out = pl->info;
        if (rc == FF_OK) fill_audit_info(pl, info_out);
    }
    ff_plan_destroy(pl);
    return rc;
}

ff::ShardRunner::ShardRunner(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr, const int64_t *leaf_idx,
                             const double *leaf_val, int leave_unnormalized, const ff_options &opt)
    : tree_(tree), n_(n_samples), lp_(leaf_ptr), li_(leaf_idx), lv_(leaf_val), unnorm_(leave_unnormalized), opt_(opt)
{
}

ff::ShardRunner::~ShardRunner()
{
    if (pl_) {
        int cur = -1;
        const bool sw = hipGetDevice(&cur) == hipSuccess && cur != pl_->device && hipSetDevice(pl_->device) == hipSuccess;
        (void)hipFree(d_out_);
        ff_plan_destroy(pl_);
        if (sw) (void)hipSetDevice(cur);
    }
}

int ff::ShardRunner::create(int32_t rank, int32_t world, int precision, char *err, size_t errlen)
{
    if (pl_) ff_plan_destroy(pl_);
    pl_ = nullptr;
    ff_options o = opt_;
    o.rank = rank;
    o.world = world;
    o.precision = precision;
    return ff_plan_create_from_leaves(tree_, n_, lp_, li_, lv_, unnorm_, &o, &pl_, err, errlen);
}

int ff::ShardRunner::device() const { return pl_ ? pl_->device : opt_.device; }

int ff::ShardRunner::prepare(int32_t rank, int32_t world, char *err, size_t errlen)
{
    if (pl_) return FF_OK;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = create(rank, world, opt_.precision, err, errlen);
    t_create += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int ff::ShardRunner::run_device(int32_t rank, int32_t world, const double **d_out, int64_t *n, ff_plan_info *info, char *err,
                                size_t errlen)
{
    using clk = std::chrono::steady_clock;
    auto since = [](clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); };
    *d_out = nullptr;
    *n = 0;
    auto t0 = clk::now();
    int rc = FF_OK;
    if (!pl_) {
        rc = create(rank, world, opt_.precision, err, errlen);
        t_create += since(t0);
    } else if (pl_->shard_rank != rank || pl_->shard_world != world) {
        rc = ff_plan_set_shard(pl_, rank, world, err, errlen);
        t_retarget += since(t0);
    }
    if (rc) return rc;
    // (a pass runs on a thread of its own: the buffers below belong on the plan's device)
    if (hipSetDevice(pl_->device) != hipSuccess) return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot select device %d", pl_->device);
    const int64_t n_slots = pl_->info.slot_end - pl_->info.slot_begin;
    if (n_slots > 0) {
        if (n_slots > d_out_cap_) {
            (void)hipFree(d_out_);
            d_out_ = nullptr;
            d_out_cap_ = 0;
            hipError_t he = hipMalloc(&d_out_, sizeof(double) * (size_t)n_slots);
            if (he != hipSuccess) {
                (void)hipGetLastError();
                return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: %s: cannot allocate %.2f GB for the results of shard %d of %d",
                                hipGetErrorString(he), 8e-9 * (double)n_slots, (int)rank, (int)world);
            }
            d_out_cap_ = n_slots;
        }
        t0 = clk::now();
        rc = ff_plan_run(pl_, nullptr, d_out_, err, errlen);
        bool ok = true;
        if (rc == FF_OK && plan_fixed32_verdict(pl_, &ok, nullptr) == FF_OK && !ok) {
            // mostly replicates, or a failed audit: binary64 from here on (run_plan_to_host does the same for one shard)
            opt_.precision = FF_PRECISION_EXACT64;
            rc = create(rank, world, FF_PRECISION_EXACT64, err, errlen);
            if (rc == FF_OK) rc = ff_plan_run(pl_, nullptr, d_out_, err, errlen);
        }
        if (rc) return rc;
        if (hipStreamSynchronize(nullptr) != hipSuccess) (void)hipGetLastError();  // (an error surfaces in the caller's next call)
        t_kernels += since(t0);
    }
    if (info) {
        *info = pl_->info;
        fill_audit_info(pl_, info);
    }
    *d_out = d_out_;
    *n = std::max<int64_t>(n_slots, 0);
    return FF_OK;
}

int ff::ShardRunner::run(int32_t rank, int32_t world, double *out, ff_plan_info *info, char *err, size_t errlen)
{
    const double *d = nullptr;
    int64_t n = 0;
    const int rc = run_device(rank, world, &d, &n, info, err, errlen);
    if (rc) return rc;
    if (n > 0) {
        if (!out) return ff::fail(FF_ERR_ARG, err, errlen, "null output pointer");
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t he = hipMemcpy(out, d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost);
        t_copy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (he != hipSuccess) {
            (void)hipGetLastError();
            return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: copy of results failed: %s", hipGetErrorString(he));
        }
    }
    return FF_OK;
}

int ff::device_count()
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

size_t ff::device_free_bytes(int device)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return 0;
    size_t free_b = 0, total_b = 0;
    if (device >= 0 && device != cur && hipSetDevice(device) != hipSuccess) return 0;
    const bool ok = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    if (device >= 0 && device != cur) (void)hipSetDevice(cur);
    return ok ? free_b : 0;
}

void ff::device_warmup(int want)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return;
    n = std::min(n, std::max(want, 1));
    for (int d = 0; d < n; ++d) {
        if (hipSetDevice(d) != hipSuccess) continue;
        // the context, and with one fill + one copy the device's queue and the runtime's own kernels: the first such
        // call of a process pays for them (0.05-0.15 s), and this thread pays it while the inputs are read
        void *p = nullptr;
        unsigned word = 0;
        if (hipMalloc(&p, 256) == hipSuccess) {
            (void)hipMemset(p, 0, 256);
            (void)hipMemcpy(&word, p, sizeof word, hipMemcpyDeviceToHost);
            (void)hipFree(p);
        }
        (void)hipGetLastError();
    }
    if (n > 0) (void)hipSetDevice(0);
}

// ff_unifrac_dists that also reports what the staging decided (used by the CLI's -stats).
int ff::unifrac_dists_info(const ff_problem *p, const ff_options *o, double *out, ff_plan_info *info_out,
                           char *err, size_t errlen)
{
    if (!out && p && p->n_samples > 1) return ff::fail(FF_ERR_ARG, err, errlen, "null output pointer");
    ff_plan *pl = nullptr;
    int rc = ff_plan_create(p, o, &pl, err, errlen);
    if (rc) return rc;
    auto again = [&](ff_plan **np) {
        ff_options o2;
        if (o) o2 = *o; else ff_options_default(&o2);
        o2.precision = FF_PRECISION_EXACT64;
        return ff_plan_create(p, &o2, np, err, errlen);
    };
    return ff::run_plan_to_host(pl, again, out, info_out, err, errlen);
}

// unifrac() with stage A on the device (used by ff_unifrac and the CLI).
int ff::unifrac_leaves_info(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr,
                            const int64_t *leaf_idx, const double *leaf_val, int leave_unnormalized,
                            const ff_options *o, double *out, ff_plan_info *info_out, char *err, size_t errlen,
                            bool shard_local)
{
    if (!out && n_samples > 1 && !shard_local) return ff::fail(FF_ERR_ARG, err, errlen, "null output pointer");
    ff_plan *pl = nullptr;
    int rc = ff_plan_create_from_leaves(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized, o, &pl,
                                        err, errlen);
    if (rc) return rc;
    auto again = [&](ff_plan **np) {
        ff_options o2;
        if (o) o2 = *o; else ff_options_default(&o2);
        o2.precision = FF_PRECISION_EXACT64;
        return ff_plan_create_from_leaves(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized, &o2, np,
                                          err, errlen);
    };
    return ff::run_plan_to_host(pl, again, out, info_out, err, errlen, shard_local);
}

