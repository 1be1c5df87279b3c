// ff_dev_text.hip -- the output side of the device path: distances -> the reference's text on the device
// (ff_kernels_fmt.hpp; replaces the formatting loop of frcfrc/frcfrc.go:58-62) and the pipeline that carries the text
// to the output file while the next pass is reduced (ff::TextPipeline, used by the frcfrc command).
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "ff_fmt_core.hpp"
#include "ff_plan.hpp"

namespace {

#include "ff_kernels_fmt.hpp"

int64_t fmt_blocks(int64_t n) { return (n + FMT_BLOCK_VALUES - 1) / FMT_BLOCK_VALUES; }

// The three launches on `st`.  d_block_bytes: fmt_blocks(n) words; d_block_off: fmt_blocks(n) + 1.
int launch_format(const double *d_vals, int64_t n, char *d_text, uint32_t *d_block_bytes, unsigned long long *d_block_off,
                  hipStream_t st, char *err, size_t errlen)
{
    const int64_t nb = fmt_blocks(n);
    if (nb <= 0 || nb > 0x7FFFFFFF) return ff::fail(FF_ERR_ARG, err, errlen, "cannot format %lld values in one call", (long long)n);
    fmt_len_kernel<<<dim3((unsigned)nb), dim3(FMT_THREADS), 0, st>>>(d_vals, n, d_block_bytes);
    fmt_scan_kernel<<<dim3(1), dim3(FMT_SCAN_THREADS), 0, st>>>(d_block_bytes, nb, d_block_off);
    fmt_write_kernel<<<dim3((unsigned)nb), dim3(FMT_THREADS), 0, st>>>(d_vals, n, d_block_off, d_text);
    FF_HIP(hipGetLastError());
    return FF_OK;
}

}  // namespace

extern "C" {

size_t ff_text_bound(int64_t n) { return n > 0 ? (size_t)n * (size_t)FMT_LINE_MAX : 0; }

int ff_format_distances_device(const double *d_values, int64_t n, char *d_text, size_t *n_bytes, void *stream, char *err,
                               size_t errlen)
{
    if (n_bytes) *n_bytes = 0;
    if (n < 0 || (n > 0 && (!d_values || !d_text)) || !n_bytes) return ff::fail(FF_ERR_ARG, err, errlen, "ff_format_distances_device: bad argument");
    if (n == 0) return FF_OK;
    const int64_t nb = fmt_blocks(n);
    ff::dev::Scratch<uint32_t> bb;
    ff::dev::Scratch<unsigned long long> bo;
    FF_HIP(bb.alloc((size_t)nb));
    FF_HIP(bo.alloc((size_t)nb + 1));
    const hipStream_t st = (hipStream_t)stream;
    const int rc = launch_format(d_values, n, d_text, bb.p, bo.p, st, err, errlen);
    if (rc) return rc;
    unsigned long long total = 0;
    FF_HIP(hipMemcpyAsync(&total, bo.p + nb, sizeof total, hipMemcpyDeviceToHost, st));
    FF_HIP(hipStreamSynchronize(st));
    *n_bytes = (size_t)total;
    return FF_OK;
}

// ---- unifracDists + the printing loop as ONE lazy sequence of text (frcfrc/unifrac.go:209-228 + frcfrc.go:58-62) ----

namespace {

// Two device text buffers (sub-shard k + 1 is reduced and formatted while sub-shard k's text is handed over) and two
// pinned host slots the text passes through on its way to the callback.
struct TextStreamBuffers {
    double *d_out = nullptr;
    int64_t out_cap = 0;
    char *d_text[2] = {nullptr, nullptr};
    int64_t text_cap[2] = {0, 0};
    uint32_t *d_block_bytes = nullptr;
    unsigned long long *d_block_off = nullptr;
    int64_t blocks_cap = 0;
    char *slot[2] = {nullptr, nullptr};
    bool pinned = false;
    int64_t slot_bytes = 0;
    hipStream_t copy = nullptr;
    ~TextStreamBuffers()
    {
        (void)hipFree(d_out);
        (void)hipFree(d_text[0]);
        (void)hipFree(d_text[1]);
        (void)hipFree(d_block_bytes);
        (void)hipFree(d_block_off);
        if (slot[0]) {
            if (pinned) (void)hipHostFree(slot[0]);
            else free(slot[0]);
        }
        if (copy) (void)hipStreamDestroy(copy);
    }
    int reserve(int buf, int64_t n, char *err, size_t errlen)
    {
        const int64_t nb = fmt_blocks(n), need = (int64_t)ff_text_bound(n);
        if (n > out_cap) {
            (void)hipFree(d_out);
            d_out = nullptr;
            out_cap = 0;
            if (hipMalloc(&d_out, sizeof(double) * (size_t)n) != hipSuccess) {
                (void)hipGetLastError();
                return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot allocate %.2f GB for a sub-shard's results (a smaller "
                                "max_pairs_per_chunk makes it smaller)", 8e-9 * (double)n);
            }
            out_cap = n;
        }
        if (need > text_cap[buf]) {
            (void)hipFree(d_text[buf]);
            d_text[buf] = nullptr;
            text_cap[buf] = 0;
            if (hipMalloc(&d_text[buf], (size_t)need) != hipSuccess) {
                (void)hipGetLastError();
                return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot allocate %.2f GB for a sub-shard's text (a smaller "
                                "max_pairs_per_chunk makes it smaller)", 1e-9 * (double)need);
            }
            text_cap[buf] = need;
        }
        if (nb > blocks_cap) {
            (void)hipFree(d_block_bytes);
            (void)hipFree(d_block_off);
            d_block_bytes = nullptr;
            d_block_off = nullptr;
            blocks_cap = 0;
            FF_HIP(hipMalloc(&d_block_bytes, sizeof(uint32_t) * (size_t)nb));
            FF_HIP(hipMalloc(&d_block_off, sizeof(unsigned long long) * (size_t)(nb + 1)));
            blocks_cap = nb;
        }
        return FF_OK;
    }
};

}  // namespace

int ff_unifrac_text_stream(const ff_problem *p, const ff_options *o, int64_t max_pairs, ff_text_fn fn, void *user, char *err,
                           size_t errlen)
{
    using namespace ff::dev;
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
    // sub-shards as in ff_unifrac_dists_stream: shard rank * c + k of world * c tile this shard's slots in order
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
    DeviceScope scope;
    FF_HIP(scope.enter(pl->device));
    TextStreamBuffers buf;
    FF_HIP(hipStreamCreateWithFlags(&buf.copy, hipStreamNonBlocking));
    {
        // two slots of up to 32 MB (a tenth of a small output: nothing is pinned that will not be used)
        int64_t want = std::min<int64_t>((int64_t)32 << 20, std::max<int64_t>((int64_t)1 << 16, shard_pairs * 20 / 4));
        want = std::max<int64_t>(want, FMT_BLOCK_BYTES_MAX);
        want = (want + 4095) / 4096 * 4096;
        void *base_ptr = nullptr;
        buf.pinned = hipHostMalloc(&base_ptr, (size_t)want * 2, hipHostMallocDefault) == hipSuccess;
        if (!buf.pinned) {
            (void)hipGetLastError();
            base_ptr = malloc((size_t)want * 2);
            if (!base_ptr) return ff::fail(FF_ERR_INTERNAL, err, errlen, "out of host memory for the text slots");
        }
        buf.slot[0] = static_cast<char *>(base_ptr);
        buf.slot[1] = buf.slot[0] + want;
        buf.slot_bytes = want;
    }
    // one sub-shard: kernels + format into text buffer `tb`, its block offsets to the host (the only wait)
    std::vector<unsigned long long> off[2];
    auto produce = [&](int tb) -> int {
        const int64_t n = pl->info.slot_end - pl->info.slot_begin;
        off[tb].assign(1, 0ull);
        if (n <= 0) return FF_OK;
        int r = buf.reserve(tb, n, err, errlen);
        if (r) return r;
        r = plan_run_impl(pl, nullptr, buf.d_out, false, err, errlen);
        if (r) return r;
        bool ok = true;
        if (plan_fixed32_verdict(pl, &ok, nullptr) == FF_OK && !ok) {
            // replicates, or a failed audit: this sub-shard again, and all later ones, in binary64
            const int32_t rk = pl->shard_rank, wd = pl->shard_world;
            ff_plan_destroy(pl);
            pl = nullptr;
            o2.precision = FF_PRECISION_EXACT64;
            o2.rank = rk;
            o2.world = wd;
            r = ff_plan_create(p, &o2, &pl, err, errlen);
            if (r == FF_OK) r = plan_run_impl(pl, nullptr, buf.d_out, false, err, errlen);
            if (r) return r;
        }
        r = launch_format(buf.d_out, n, buf.d_text[tb], buf.d_block_bytes, buf.d_block_off, nullptr, err, errlen);
        if (r) return r;
        const int64_t nb = fmt_blocks(n);
        off[tb].resize((size_t)nb + 1);
        FF_HIP(hipMemcpy(off[tb].data(), buf.d_block_off, sizeof(unsigned long long) * (size_t)(nb + 1), hipMemcpyDeviceToHost));
        return FF_OK;
    };
    // hands the text of buffer `tb` to the callback in pieces of whole blocks (whole lines) through the two slots,
    // the copy of piece q + 1 under the callback of piece q; false: the consumer stopped
    auto deliver = [&](int tb, bool *go_on) -> int {
        *go_on = true;
        const std::vector<unsigned long long> &bo = off[tb];
        const int64_t nb = (int64_t)bo.size() - 1;
        auto piece_end = [&](int64_t b) {
            const unsigned long long limit = bo[(size_t)b] + (unsigned long long)buf.slot_bytes;
            const auto it = std::upper_bound(bo.begin() + b + 1, bo.end(), limit);
            return std::max<int64_t>(b + 1, (int64_t)(it - bo.begin()) - 1);
        };
        int64_t b = 0;
        int s = 0;
        int64_t e = nb > 0 ? piece_end(0) : 0;
        if (nb > 0) FF_HIP(hipMemcpyAsync(buf.slot[s], buf.d_text[tb] + bo[0], (size_t)(bo[(size_t)e] - bo[0]), hipMemcpyDeviceToHost, buf.copy));
        while (b < nb) {
            FF_HIP(hipStreamSynchronize(buf.copy));
            const size_t len = (size_t)(bo[(size_t)e] - bo[(size_t)b]);
            const int64_t b2 = e, e2 = b2 < nb ? piece_end(b2) : b2;
            if (b2 < nb)
                FF_HIP(hipMemcpyAsync(buf.slot[s ^ 1], buf.d_text[tb] + bo[(size_t)b2], (size_t)(bo[(size_t)e2] - bo[(size_t)b2]),
                                      hipMemcpyDeviceToHost, buf.copy));
            if (len > 0 && !fn(user, buf.slot[s], len)) {
                *go_on = false;
                FF_HIP(hipStreamSynchronize(buf.copy));  // (the copy in flight targets a slot that dies with this call)
                return FF_OK;
            }
            b = b2;
            e = e2;
            s ^= 1;
        }
        return FF_OK;
    };
    rc = produce(0);
    if (rc) return rc;
    for (int64_t k = 0; k < c; ++k) {
        const int cur = (int)(k & 1);
        if (k + 1 < c) {
            // sub-shard k + 1 is reduced and formatted while sub-shard k's text is handed over: launched here, waited
            // for inside produce() only at its last step
            rc = ff_plan_set_shard(pl, (int32_t)(base.rank * c + k + 1), (int32_t)(base.world * c), err, errlen);
            if (rc) return rc;
        }
        // (set_shard synchronises the device: sub-shard k's text is complete.)  Sub-shard k + 1's pair kernels are
        // enqueued BEFORE the hand-over -- plan_run_impl is asynchronous -- and waited for after it.
        int rc_next = FF_OK;
        bool go_on = true;
        if (k + 1 < c) {
            const int64_t n = pl->info.slot_end - pl->info.slot_begin;
            if (n > 0) {
                rc_next = buf.reserve(cur ^ 1, n, err, errlen);
                if (rc_next == FF_OK) rc_next = plan_run_impl(pl, nullptr, buf.d_out, false, err, errlen);
            }
            if (rc_next) return rc_next;
        }
        rc = deliver(cur, &go_on);
        if (rc) return rc;
        if (!go_on) {
            FF_HIP(hipDeviceSynchronize());  // the sub-shard in flight is dropped, the rest never computed
            return FF_OK;
        }
        if (k + 1 < c) {
            // finish sub-shard k + 1: verdict, format, offsets (its pair kernels have been running meanwhile)
            const int64_t n = pl->info.slot_end - pl->info.slot_begin;
            off[cur ^ 1].assign(1, 0ull);
            if (n > 0) {
                bool ok = true;
                if (plan_fixed32_verdict(pl, &ok, nullptr) == FF_OK && !ok) {
                    const int32_t rk = pl->shard_rank, wd = pl->shard_world;
                    ff_plan_destroy(pl);
                    pl = nullptr;
                    o2.precision = FF_PRECISION_EXACT64;
                    o2.rank = rk;
                    o2.world = wd;
                    rc = ff_plan_create(p, &o2, &pl, err, errlen);
                    if (rc == FF_OK) rc = plan_run_impl(pl, nullptr, buf.d_out, false, err, errlen);
                    if (rc) return rc;
                }
                rc = launch_format(buf.d_out, n, buf.d_text[cur ^ 1], buf.d_block_bytes, buf.d_block_off, nullptr, err, errlen);
                if (rc) return rc;
                const int64_t nb = fmt_blocks(n);
                off[cur ^ 1].resize((size_t)nb + 1);
                FF_HIP(hipMemcpy(off[cur ^ 1].data(), buf.d_block_off, sizeof(unsigned long long) * (size_t)(nb + 1), hipMemcpyDeviceToHost));
            }
        }
    }
    return FF_OK;
}

int ff_unifrac_text_stream_csr(int64_t n_samples, int64_t n_branches, const double *branch_len, const int64_t *indptr,
                               const int32_t *branch_id, const double *abnd, const ff_options *o, int64_t max_pairs_per_chunk,
                               ff_text_fn fn, void *user, char *err, size_t errlen)
{
    ff_problem p;
    p.n_samples = n_samples;
    p.n_branches = n_branches;
    p.branch_len = branch_len;
    p.indptr = indptr;
    p.branch_id = branch_id;
    p.abnd = abnd;
    return ff_unifrac_text_stream(&p, o, max_pairs_per_chunk, fn, user, err, errlen);
}

}  // extern "C"

// ---- the pipeline -----------------------------------------------------------------------------------

namespace ff {

namespace {

constexpr int RING_SLOTS = 3;
constexpr int64_t SLOT_BYTES_MAX = (int64_t)64 << 20, SLOT_BYTES_MIN = (int64_t)1 << 16;

// What a device holds for the pipeline: up to TEXT_BUFS text buffers (the copier empties one while later passes fill
// the others: the device is not held up by the file system until it is four passes ahead of it), the per-block work
// space, a stream for the copies that the legacy stream's kernels do not wait for.
constexpr int TEXT_BUFS = 4;
struct DeviceSide {
    int device = 0;
    char *d_text[TEXT_BUFS] = {nullptr, nullptr, nullptr, nullptr};
    int64_t cap[TEXT_BUFS] = {0, 0, 0, 0};
    bool busy[TEXT_BUFS] = {false, false, false, false};
    uint32_t *d_block_bytes = nullptr;
    unsigned long long *d_block_off = nullptr;
    int64_t blocks_cap = 0;
    hipStream_t copy_stream = nullptr;
    int64_t submits = 0;
};

struct Job {  // one submit: the text of a pass on a device
    DeviceSide *side = nullptr;
    int buf = 0;
    std::vector<unsigned long long> block_off;  // [blocks + 1]
};

struct Chunk {  // a filled slot on its way to the file
    int slot = -1;
    size_t n = 0;
};

}  // namespace

struct TextPipeline::Impl {
    DistWriter *writer = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    // ring
    char *slot[RING_SLOTS] = {nullptr, nullptr, nullptr};
    bool slot_pinned = false, slot_free[RING_SLOTS] = {true, true, true};
    int64_t slot_bytes = 0;
    bool ring_ready = false;
    std::deque<Job> jobs;
    std::deque<Chunk> chunks;
    bool closing = false, copier_done = false;
    int jobs_in_flight = 0;  // submitted and not yet fully handed to the writer
    int rc = FF_OK;
    std::string error;
    std::deque<DeviceSide> sides;  // (stable addresses)
    std::thread copier, writer_thread;
    bool threads_started = false;
    double t_copy = 0, t_write = 0;  // seconds the copier spent copying, the writer writing (each on its own thread)
    bool abandoned = false;

    void set_error(int code, const char *msg)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (rc == FF_OK) {
            rc = code;
            error = msg;
        }
    }

    int alloc_ring(int64_t expected_bytes, char *err, size_t errlen)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (ring_ready) return FF_OK;
        // a third of the output per slot up to 64 MB: small outputs pin (and pay for) next to nothing
        int64_t want = std::min(SLOT_BYTES_MAX, std::max(SLOT_BYTES_MIN, expected_bytes / RING_SLOTS + 1));
        want = std::max<int64_t>(want, FMT_BLOCK_BYTES_MAX);  // (a slot holds at least one block's text)
        want = (want + 4095) / 4096 * 4096;
        void *base = nullptr;
        slot_pinned = hipHostMalloc(&base, (size_t)want * RING_SLOTS, hipHostMallocDefault) == hipSuccess;
        if (!slot_pinned) {
            (void)hipGetLastError();
            base = malloc((size_t)want * RING_SLOTS);
            if (!base) return fail(FF_ERR_INTERNAL, err, errlen, "out of host memory for the output ring (%lld bytes)", (long long)want * RING_SLOTS);
        }
        for (int k = 0; k < RING_SLOTS; ++k) slot[k] = static_cast<char *>(base) + (size_t)want * (size_t)k;
        slot_bytes = want;
        ring_ready = true;
        return FF_OK;
    }

    void copier_loop()
    {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !jobs.empty() || closing; });
                if (jobs.empty()) break;
                job = std::move(jobs.front());
                jobs.pop_front();
            }
            DeviceSide &sd = *job.side;
            bool ok = hipSetDevice(sd.device) == hipSuccess;
            if (!ok) set_error(FF_ERR_DEVICE, "HIP: cannot select the device for the output copies");
            const int64_t nb = (int64_t)job.block_off.size() - 1;
            int64_t b = 0;
            while (ok && b < nb) {
                // as many whole blocks as fit a slot (a block's text is at most FMT_BLOCK_BYTES_MAX <= slot_bytes)
                const unsigned long long byte0 = job.block_off[(size_t)b];
                int64_t e = b + 1;
                {
                    // (offsets ascend: the last block whose end is within the slot)
                    const unsigned long long limit = byte0 + (unsigned long long)slot_bytes;
                    const auto it = std::upper_bound(job.block_off.begin() + b + 1, job.block_off.end(), limit);
                    e = std::max<int64_t>(b + 1, (int64_t)(it - job.block_off.begin()) - 1);
                }
                const size_t len = (size_t)(job.block_off[(size_t)e] - byte0);
                int s = -1;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] {
                        for (int k = 0; k < RING_SLOTS; ++k)
                            if (slot_free[k]) return true;
                        return false;
                    });
                    for (int k = 0; k < RING_SLOTS && s < 0; ++k)
                        if (slot_free[k]) s = k;
                    slot_free[s] = false;
                }
                const auto c0 = std::chrono::steady_clock::now();
                hipError_t he = len > 0 ? hipMemcpyAsync(slot[s], sd.d_text[job.buf] + byte0, len, hipMemcpyDeviceToHost, sd.copy_stream) : hipSuccess;
                if (he == hipSuccess) he = hipStreamSynchronize(sd.copy_stream);
                t_copy += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
                if (he != hipSuccess) {
                    (void)hipGetLastError();
                    set_error(FF_ERR_DEVICE, (std::string("HIP: copy of the output text failed: ") + hipGetErrorString(he)).c_str());
                    ok = false;
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (ok) chunks.push_back({s, len});
                    else slot_free[s] = true;
                }
                cv.notify_all();
                b = e;
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                sd.busy[job.buf] = false;
                --jobs_in_flight;
            }
            cv.notify_all();
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            copier_done = true;
        }
        cv.notify_all();
    }

    void writer_loop()
    {
        for (;;) {
            Chunk c;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !chunks.empty() || copier_done; });
                if (chunks.empty()) break;
                c = chunks.front();
            }
            bool failed;
            {
                std::lock_guard<std::mutex> lk(mu);
                failed = rc != FF_OK;
            }
            if (!failed && c.n > 0) {
                char e[1024] = {0};
                const auto w0 = std::chrono::steady_clock::now();
                const int r = writer->write_text(slot[c.slot], c.n, e, sizeof e);
                t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
                if (r != FF_OK) set_error(r, e);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                chunks.pop_front();  // (popped only now: drain() waits for an empty queue)
                slot_free[c.slot] = true;
            }
            cv.notify_all();
        }
    }
};

TextPipeline::TextPipeline(DistWriter *writer) : impl_(new Impl())
{
    impl_->writer = writer;
}

TextPipeline::~TextPipeline()
{
    Impl &m = *impl_;
    {
        std::lock_guard<std::mutex> lk(m.mu);
        m.closing = true;
    }
    m.cv.notify_all();
    if (m.copier.joinable()) m.copier.join();
    {
        std::lock_guard<std::mutex> lk(m.mu);
        m.copier_done = true;
    }
    m.cv.notify_all();
    if (m.writer_thread.joinable()) m.writer_thread.join();
    if (m.abandoned) {
        delete impl_;
        return;
    }
    int cur = -1;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (DeviceSide &sd : m.sides) {
        if (hipSetDevice(sd.device) != hipSuccess) continue;
        for (int k = 0; k < TEXT_BUFS; ++k) (void)hipFree(sd.d_text[k]);
        (void)hipFree(sd.d_block_bytes);
        (void)hipFree(sd.d_block_off);
        if (sd.copy_stream) (void)hipStreamDestroy(sd.copy_stream);
    }
    if (have) (void)hipSetDevice(cur);
    if (m.slot[0]) {
        if (m.slot_pinned) (void)hipHostFree(m.slot[0]);
        else free(m.slot[0]);
    }
    delete impl_;
}

void TextPipeline::abandon() { impl_->abandoned = true; }

int TextPipeline::prepare(int64_t expected_bytes, char *err, size_t errlen)
{
    return impl_->alloc_ring(expected_bytes, err, errlen);
}

int TextPipeline::submit(int device, const double *d_vals, int64_t n, char *err, size_t errlen)
{
    if (n <= 0) return FF_OK;
    Impl &m = *impl_;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = m.alloc_ring((int64_t)ff_text_bound(n), err, errlen);
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(m.mu);
        if (m.rc != FF_OK) return fail(m.rc, err, errlen, "%s", m.error.c_str());
        if (!m.threads_started) {
            m.threads_started = true;
            m.copier = std::thread([&m] { m.copier_loop(); });
            m.writer_thread = std::thread([&m] { m.writer_loop(); });
        }
    }
    ff::dev::DeviceScope scope;
    FF_HIP(scope.enter(device));
    DeviceSide *sd = nullptr;
    {
        std::lock_guard<std::mutex> lk(m.mu);
        for (DeviceSide &s : m.sides)
            if (s.device == device) sd = &s;
        if (!sd) {
            m.sides.emplace_back();
            sd = &m.sides.back();
            sd->device = device;
        }
    }
    if (!sd->copy_stream) FF_HIP(hipStreamCreateWithFlags(&sd->copy_stream, hipStreamNonBlocking));
    int buf = -1;
    {
        // a buffer whose text is out (an allocated one first); all four on their way: wait for one
        std::unique_lock<std::mutex> lk(m.mu);
        auto pick = [&] {
            buf = -1;
            for (int k = 0; k < TEXT_BUFS && buf < 0; ++k)
                if (!sd->busy[k] && sd->d_text[k]) buf = k;
            for (int k = 0; k < TEXT_BUFS && buf < 0; ++k)
                if (!sd->busy[k]) buf = k;
            return buf >= 0;
        };
        m.cv.wait(lk, pick);
    }
    const int64_t nb = fmt_blocks(n), need = (int64_t)ff_text_bound(n);
    if (need > sd->cap[buf]) {
        (void)hipFree(sd->d_text[buf]);
        sd->d_text[buf] = nullptr;
        sd->cap[buf] = 0;
        if (hipMalloc(&sd->d_text[buf], (size_t)need) != hipSuccess) {
            (void)hipGetLastError();
            return fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot allocate %.2f GB for the text of a pass", (double)need / 1e9);
        }
        sd->cap[buf] = need;
    }
    if (nb > sd->blocks_cap) {
        (void)hipFree(sd->d_block_bytes);
        (void)hipFree(sd->d_block_off);
        sd->d_block_bytes = nullptr;
        sd->d_block_off = nullptr;
        sd->blocks_cap = 0;
        FF_HIP(hipMalloc(&sd->d_block_bytes, sizeof(uint32_t) * (size_t)nb));
        FF_HIP(hipMalloc(&sd->d_block_off, sizeof(unsigned long long) * (size_t)(nb + 1)));
        sd->blocks_cap = nb;
    }
    rc = launch_format(d_vals, n, sd->d_text[buf], sd->d_block_bytes, sd->d_block_off, nullptr, err, errlen);
    if (rc) return rc;
    Job job;
    job.side = sd;
    job.buf = buf;
    job.block_off.resize((size_t)nb + 1);
    FF_HIP(hipMemcpy(job.block_off.data(), sd->d_block_off, sizeof(unsigned long long) * (size_t)(nb + 1), hipMemcpyDeviceToHost));
    bytes += (int64_t)job.block_off.back();
    {
        std::lock_guard<std::mutex> lk(m.mu);
        sd->busy[buf] = true;
        ++sd->submits;
        ++m.jobs_in_flight;
        m.jobs.push_back(std::move(job));
    }
    m.cv.notify_all();
    t_submit += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return FF_OK;
}

int TextPipeline::drain(char *err, size_t errlen)
{
    Impl &m = *impl_;
    std::unique_lock<std::mutex> lk(m.mu);
    m.cv.wait(lk, [&] { return (m.jobs_in_flight == 0 && m.chunks.empty()) || !m.threads_started; });
    t_copy = m.t_copy;
    t_write = m.t_write;
    if (m.rc != FF_OK) return fail(m.rc, err, errlen, "%s", m.error.c_str());
    return FF_OK;
}

}  // namespace ff
