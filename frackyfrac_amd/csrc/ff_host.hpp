// ff_host.hpp -- internal declarations shared by the host-side sources of
// libfrackyfrac_amd (the C ABI is include/frackyfrac_amd.h).
#pragma once

#include <cerrno>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

#include "frackyfrac_amd.h"

namespace ff {

// Writes a printf-style message into the caller's err buffer; returns `code`.
int fail(int code, char *err, size_t errlen, const char *fmt, ...)
    __attribute__((format(printf, 4, 5)));

// Go's %q for a string, Go's %v for a float64, Go's %f for a float64.
std::string go_quote(const std::string &s);
std::string go_v(double f);
std::string go_f(double f);

// strconv.ParseFloat(tok, 64): returns false and fills `why` ("invalid syntax" /
// "value out of range") on failure.
bool go_parse_float(const char *b, const char *e, double *out, const char **why);

// Reads a whole file (path; gunzipped when it ends in ".gz") or stdin (path == nullptr).
int read_all(const char *path, std::string *out, char *err, size_t errlen);
// The same for a text that may be large (an abundance table): a plain regular file is read by `threads` threads
// straight into a buffer that nobody zero-fills first (a 155-MB table took 0.27 s through read_all's growing string,
// a third of its parse on 16 threads); everything else goes through read_all.
struct Text {
    const char *data = nullptr;
    size_t size = 0;
    std::unique_ptr<char[]> buf;
    std::string str;
};
int read_text(const char *path, unsigned threads, Text *out, char *err, size_t errlen);

// std::allocator whose resize() leaves trivially constructible elements uninitialised: no zero fill, no page touched
// by the thread that resizes (the threads that fill the array touch it, each its own part).
template <typename T> struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind {
        using other = NoInitAlloc<U>;
    };
    template <typename U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};

unsigned clamp_threads(int requested);
// CPUs this process may use: the smallest of the hardware's count, the scheduler affinity mask and the cgroup's
// CPU quota (cpu.max / cfs_quota_us, rounded up) -- what a container is really given, not what the machine has.
unsigned cpu_quota();

// A tuning switch (the FF_* names of INTEGRATION.md): the value given through ff_tune, else the
// environment variable of that name, else nothing.  A copy: a concurrent ff_tune cannot invalidate it.
// Switches are read when a plan is created or re-targeted, never on the launch path.
std::optional<std::string> tuning(const char *name);

// Runs fn(t, begin, end) over [0, n) split into contiguous chunks on `threads` threads.
void parallel_for(int64_t n, unsigned threads,
                  const std::function<void(unsigned, int64_t, int64_t)> &fn);

// ff_unifrac_dists + the plan's info (info may be null).
int unifrac_dists_info(const ff_problem *p, const ff_options *o, double *out, ff_plan_info *info,
                       char *err, size_t errlen);
// The same from leaf values: stage A runs on the device.
// shard_local: out[0] is the shard's first slot (out has slot_end - slot_begin entries)
// instead of slot 0 of the whole triangle.
int unifrac_leaves_info(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr,
                        const int64_t *leaf_idx, const double *leaf_val, int leave_unnormalized,
                        const ff_options *o, double *out, ff_plan_info *info, char *err, size_t errlen,
                        bool shard_local = false);
int run_plan_to_host(ff_plan *pl, const std::function<int(ff_plan **)> &recreate_exact64, double *out,
                     ff_plan_info *info, char *err, size_t errlen, bool shard_local = false);
// One device's worker for a pair space that is computed shard by shard (the CLI's passes):
// stage A and the staging happen once, every further shard only re-targets the plan
// (ff_plan_set_shard).  If a shard's refinement queue overflows (a data set of replicates) the
// runner switches to EXACT64 for good.
class ShardRunner {
public:
    ShardRunner(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr, const int64_t *leaf_idx,
                const double *leaf_val, int leave_unnormalized, const ff_options &opt);
    ~ShardRunner();
    ShardRunner(const ShardRunner &) = delete;
    ShardRunner &operator=(const ShardRunner &) = delete;
    // distances of shard `rank` of `world` into out[0 .. slot_end - slot_begin) (host memory)
    int run(int32_t rank, int32_t world, double *out, ff_plan_info *info, char *err, size_t errlen);
    // the same, the distances left on the device: *d_out (valid until the next call) holds *n of them
    int run_device(int32_t rank, int32_t world, const double **d_out, int64_t *n, ff_plan_info *info, char *err, size_t errlen);
    // builds the plan for the first shard ahead of its run (stage A, staging): what the reference's "Converting
    // abundances" phase is (unifrac.go:101-116); the first run()/run_device() of that shard then only launches
    int prepare(int32_t rank, int32_t world, char *err, size_t errlen);
    int device() const;
    // seconds the calls of run() spent: building the plan (stage A, staging), re-targeting it, in the kernels
    // (launch to the verdict's counters), copying the results out
    double t_create = 0, t_retarget = 0, t_kernels = 0, t_copy = 0;

private:
    int create(int32_t rank, int32_t world, int precision, char *err, size_t errlen);
    const ff_tree *tree_;
    int64_t n_;
    const int64_t *lp_, *li_;
    const double *lv_;
    int unnorm_;
    ff_options opt_;
    ff_plan *pl_ = nullptr;
    double *d_out_ = nullptr;
    int64_t d_out_cap_ = 0;
};

// Number of HIP devices visible (0 when there is none).
int device_count();
// Brings the HIP context of the first `want` devices up (errors are left for the first real call).
void device_warmup(int want);
// Free memory of a device in bytes (0 when it cannot be asked).
size_t device_free_bytes(int device);

// The output file of frcfrc (frcfrc.go:58-62,102): values appended in calls, one per line.
class DistWriter {
public:
    ~DistWriter();
    int open(const char *path /* null: stdout */, int threads, char *err, size_t errlen);
    int write(const double *d, int64_t n, char *err, size_t errlen);
    // n bytes of finished text (whole lines: TextPipeline, formatted on the device)
    int write_text(const char *text, size_t n, char *err, size_t errlen);
    int close(char *err, size_t errlen);

private:
    int fd_ = -1;
    bool own_ = false, gz_ = false, zst_ = false, seekable_ = false;
    unsigned nt_ = 1;
    int64_t off_ = 0;
    std::string name_;
    std::vector<std::string> bufs_, zbufs_;
};

// Large arrays that parallel loops fill: resize() does not zero them (NoInitAlloc, above).
using I64Vec = std::vector<int64_t, NoInitAlloc<int64_t>>;
using F64Vec = std::vector<double, NoInitAlloc<double>>;

// The output path of frcfrc with the formatter on the device (ff_kernels_fmt.hpp): the distances of a pass are
// turned into text in HBM, the text is copied out through a ring of pinned host slots on a stream of its own and
// appended to the file, all of it behind the main thread's back -- pass k + 1 is reduced while pass k is on its
// way out.  One copier thread (device -> slot) and one writer thread (slot -> file, DistWriter::write_text).
class TextPipeline {
public:
    explicit TextPipeline(DistWriter *writer);
    ~TextPipeline();
    TextPipeline(const TextPipeline &) = delete;
    TextPipeline &operator=(const TextPipeline &) = delete;
    // Allocates the ring for an output of about expected_bytes (any thread; optional: submit does it otherwise).
    int prepare(int64_t expected_bytes, char *err, size_t errlen);
    // d_vals[0 .. n) on `device`: formats them there (legacy stream: ordered behind the pass that produced them) and
    // queues the text for the file.  Returns when the format kernels are done -- d_vals may be overwritten then.
    int submit(int device, const double *d_vals, int64_t n, char *err, size_t errlen);
    // Waits until everything submitted is in the file; the first error of the background threads.
    int drain(char *err, size_t errlen);
    // The process is about to end: the destructor joins its threads but gives no memory back (unpinning the ring alone
    // takes 30 ms that nobody is waiting for).
    void abandon();
    double t_submit = 0;  // seconds the callers of submit() spent in it
    int64_t bytes = 0;    // text bytes produced so far
    double t_copy = 0, t_write = 0;  // after drain(): seconds the copier spent in device-to-host copies, the writer in the file

private:
    struct Impl;
    Impl *impl_;
};

// abnd[tree.Name] for every leaf (unifrac.go:38-43) as CSR over node ids.
void table_leaf_csr(const ff_table &tb, const ff_tree &tr, std::vector<int64_t> *ptr, I64Vec *idx, F64Vec *val,
                    int threads = 1);


// Index of a vector of distinct names: open addressing over (hash tag, id + 1).  A lookup
// costs one hash of the token and usually one compare -- no std::string is built per token.
struct NameIndex {
    std::vector<uint64_t> slot;  // 0 = empty
    size_t mask = 0;
    static uint64_t hash(const char *b, const char *e)
    {
        uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a, then a finalising mix
        for (const char *p = b; p < e; ++p) h = (h ^ (unsigned char)*p) * 0x100000001b3ull;
        h ^= h >> 29;
        h *= 0xbf58476d1ce4e5b9ull;
        h ^= h >> 32;
        return h;
    }
    void grow(const std::vector<std::string> &names)
    {
        const size_t n = slot.empty() ? 1024 : slot.size() * 2;
        std::vector<uint64_t> old;
        old.swap(slot);
        slot.assign(n, 0);
        mask = n - 1;
        for (uint64_t s : old)
            if (s) {
                const std::string &nm = names[(size_t)(uint32_t)s - 1];
                size_t at = (size_t)hash(nm.data(), nm.data() + nm.size()) & mask;
                while (slot[at]) at = (at + 1) & mask;
                slot[at] = s;
            }
    }
    // id of [b, e) in *names, appended when new
    int32_t intern(std::vector<std::string> *names, const char *b, const char *e)
    {
        if ((names->size() + 1) * 2 > slot.size()) grow(*names);
        const uint64_t h = hash(b, e), tag = h & 0xffffffff00000000ull;
        const size_t len = (size_t)(e - b);
        for (size_t at = (size_t)h & mask;; at = (at + 1) & mask) {
            const uint64_t s = slot[at];
            if (!s) {
                names->emplace_back(b, e);
                slot[at] = tag | (uint64_t)names->size();
                return (int32_t)names->size() - 1;
            }
            if ((s & 0xffffffff00000000ull) == tag) {
                const std::string &nm = (*names)[(size_t)(uint32_t)s - 1];
                if (nm.size() == len && memcmp(nm.data(), b, len) == 0) return (int32_t)(uint32_t)s - 1;
            }
        }
    }
};

}  // namespace ff

// The tree in enumerateNodes' numbering (frcfrc/unifrac.go:127-133).
struct ff_tree {
    std::vector<std::string> name;
    std::vector<double> dist;     // treeDists (unifrac.go:117-120)
    std::vector<int64_t> parent;  // -1 for the root; parent[id] < id
    std::vector<int64_t> size;    // nodes in the subtree rooted at id (1 = leaf)
    // name -> ids of the LEAVES carrying it (abundance goes to leaves only:
    // flatNodeOptimization, unifrac.go:18,38-43); built lazily.
    std::unordered_map<std::string, std::vector<int64_t>> leaf_ids;
    // every node name, internal ones and "" included (treeNames, unifrac.go:70-76)
    std::unordered_map<std::string, int> all_names;
    void index_names();
};

// []map[string]float64 (parser/parser.go): per sample, (species index, value) in
// insertion order; a re-assigned key keeps its slot and takes the last value.
struct ff_table {
    std::vector<std::string> species;
    ff::NameIndex index;
    std::vector<int64_t> ptr;  // [n_samples + 1]
    std::vector<int32_t, ff::NoInitAlloc<int32_t>> key;  // species index
    std::vector<double, ff::NoInitAlloc<double>> val;
    int32_t intern(const char *b, const char *e);
};

struct ff_flat {
    int64_t n_samples = 0, n_branches = 0;
    std::vector<double> branch_len;
    std::vector<int64_t> indptr;
    std::vector<int32_t> branch_id;
    std::vector<double> abnd;
};
