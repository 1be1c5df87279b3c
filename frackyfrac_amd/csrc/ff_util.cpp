// ff_util.cpp -- error plumbing, Go-compatible number formatting/parsing, threads.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <functional>
#include <thread>

#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <dlfcn.h>
#include <zlib.h>

#include <mutex>
#include <unordered_map>

#include "ff_fmt_core.hpp"
#include "ff_host.hpp"

namespace ff {

int fail(int code, char *err, size_t errlen, const char *fmt, ...)
{
    if (err && errlen) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, errlen, fmt, ap);
        va_end(ap);
    }
    return code;
}

std::string go_quote(const std::string &s)
{
    std::string o = "\"";
    char tmp[8];
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\n': o += "\\n"; break;
        case '\t': o += "\\t"; break;
        case '\r': o += "\\r"; break;
        case '\a': o += "\\a"; break;  // strconv.Quote's mnemonic escapes
        case '\b': o += "\\b"; break;
        case '\f': o += "\\f"; break;
        case '\v': o += "\\v"; break;
        default:
            if (c < 0x20 || c == 0x7f) {
                snprintf(tmp, sizeof tmp, "\\x%02x", c);
                o += tmp;
            } else {
                o += (char)c;  // UTF-8 passes through as Go prints valid runes
            }
        }
    }
    o += '"';
    return o;
}

std::string go_v(double f)
{
    char b[40];
    int n = ff_format_float(f, b);
    return std::string(b, (size_t)n);
}

std::string go_f(double f)
{
    if (std::isnan(f)) return "NaN";
    if (std::isinf(f)) return f > 0 ? "+Inf" : "-Inf";
    char b[400];
    snprintf(b, sizeof b, "%f", f);
    return b;
}

static bool ieq(const char *b, const char *e, const char *lit)
{
    size_t n = strlen(lit);
    if ((size_t)(e - b) != n) return false;
    for (size_t i = 0; i < n; ++i) {
        char c = b[i];
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
        if (c != lit[i]) return false;
    }
    return true;
}

// strconv.ParseFloat accepts: optional sign, decimal or 0x-hex mantissa with
// optional exponent, and the words inf / infinity / nan (any case).  Out-of-range
// magnitudes are an error (ErrRange), which parser.go turns into "value #k: ...".
bool go_parse_float(const char *b, const char *e, double *out, const char **why)
{
    *why = "invalid syntax";
    if (b == e) return false;
    {
        // Fast path for the tokens abundance tables are made of: DIGITS or DIGITS.DIGITS with
        // at most 15 significant digits.  The integer below and the power of ten are then both
        // exact in binary64, so one division (or none) is correctly rounded (Clinger) -- the
        // same value the general conversion gives.
        static const double P10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                     1e12, 1e13, 1e14, 1e15};
        uint64_t m = 0;
        int nd = 0, nfrac = -1;
        const char *q = b;
        for (; q < e; ++q) {
            const unsigned c = (unsigned)(*q - '0');
            if (c <= 9) {
                m = m * 10 + c;
                ++nd;
                if (nfrac >= 0) ++nfrac;
            } else if (*q == '.' && nfrac < 0 && nd > 0) {
                nfrac = 0;
            } else {
                break;
            }
        }
        if (q == e && nd > 0 && nd <= 15 && nfrac != 0) {
            *out = nfrac > 0 ? (double)m / P10[nfrac] : (double)m;
            return true;
        }
    }
    const char *p = b;
    bool neg = false;
    if (*p == '+' || *p == '-') {
        neg = (*p == '-');
        ++p;
        if (p == e) return false;
    }
    if (ieq(p, e, "inf") || ieq(p, e, "infinity")) {
        *out = neg ? -INFINITY : INFINITY;
        return true;
    }
    if (ieq(p, e, "nan")) {
        *out = NAN;
        return true;
    }
    double v = 0;
    std::from_chars_result r{};
    if (e - p > 2 && p[0] == '0' && (p[1] == 'x' || p[1] == 'X')) {
        // Go requires a p-exponent on hex floats.
        bool has_p = false;
        for (const char *q = p + 2; q < e; ++q)
            if (*q == 'p' || *q == 'P') has_p = true;
        if (!has_p) return false;
        r = std::from_chars(p + 2, e, v, std::chars_format::hex);
    } else {
        if (!((*p >= '0' && *p <= '9') || *p == '.')) return false;
        r = std::from_chars(p, e, v, std::chars_format::general);
    }
    if (r.ec == std::errc::result_out_of_range) {
        // from_chars leaves v unmodified; underflow is not an error in Go (gives 0
        // or a denormal), overflow is.
        std::string tmp(p, e);
        double s = strtod(tmp.c_str(), nullptr);
        if (std::isinf(s)) {
            *why = "value out of range";
            return false;
        }
        *out = neg ? -s : s;
        return true;
    }
    if (r.ec != std::errc() || r.ptr != e) return false;
    *out = neg ? -v : v;
    return true;
}

static bool has_suffix(const char *path, const char *suf)
{
    const size_t n = strlen(path), m = strlen(suf);
    return n >= m && strcmp(path + n - m, suf) == 0;
}

// ---- zstd and bzip2 by suffix, like gzip --------------------------------------------------------
// gostuff/aio (go.mod:7; its klauspost/compress dependency is go.mod:12) picks the codec by the file's suffix.
// This image ships libzstd.so.1 and libbz2.so.1.0 without their headers, so the few entry points used are
// declared here and bound with dlopen when a ".zst" / ".bz2" path first comes by; a host without the library
// gets an error that says so.  (Parity unpinned: aio's source is not in the reference tree; ".zst" is read and
// written, ".bz2" read only -- Go has no bzip2 writer.)
namespace {
struct ZstdBuf {  // ZSTD_inBuffer / ZSTD_outBuffer (same layout: pointer, size, position)
    void *ptr;
    size_t size, pos;
};
struct ZstdApi {
    void *(*createDStream)();
    size_t (*freeDStream)(void *);
    size_t (*decompressStream)(void *, ZstdBuf *out, ZstdBuf *in);
    unsigned (*isError)(size_t);
    const char *(*getErrorName)(size_t);
    size_t (*compressBound)(size_t);
    size_t (*compress)(void *dst, size_t cap, const void *src, size_t n, int level);
    bool ok = false;
};
const ZstdApi &zstd_api()
{
    static const ZstdApi api = [] {
        ZstdApi a{};
        void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.createDStream = (void *(*)())dlsym(h, "ZSTD_createDStream");
        a.freeDStream = (size_t(*)(void *))dlsym(h, "ZSTD_freeDStream");
        a.decompressStream = (size_t(*)(void *, ZstdBuf *, ZstdBuf *))dlsym(h, "ZSTD_decompressStream");
        a.isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        a.getErrorName = (const char *(*)(size_t))dlsym(h, "ZSTD_getErrorName");
        a.compressBound = (size_t(*)(size_t))dlsym(h, "ZSTD_compressBound");
        a.compress = (size_t(*)(void *, size_t, const void *, size_t, int))dlsym(h, "ZSTD_compress");
        a.ok = a.createDStream && a.freeDStream && a.decompressStream && a.isError && a.getErrorName && a.compressBound &&
               a.compress;
        return a;
    }();
    return api;
}
struct BzStream {  // bz_stream of bzlib.h
    char *next_in;
    unsigned avail_in, total_in_lo32, total_in_hi32;
    char *next_out;
    unsigned avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int);
    void (*bzfree)(void *, void *);
    void *opaque;
};
struct Bz2Api {
    int (*init)(BzStream *, int verbosity, int small);
    int (*run)(BzStream *);
    int (*end)(BzStream *);
    bool ok = false;
};
const Bz2Api &bz2_api()
{
    static const Bz2Api api = [] {
        Bz2Api a{};
        void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.init = (int (*)(BzStream *, int, int))dlsym(h, "BZ2_bzDecompressInit");
        a.run = (int (*)(BzStream *))dlsym(h, "BZ2_bzDecompress");
        a.end = (int (*)(BzStream *))dlsym(h, "BZ2_bzDecompressEnd");
        a.ok = a.init && a.run && a.end;
        return a;
    }();
    return api;
}

// raw: the file's bytes; out: the decompressed text (concatenated frames / streams are one text)
int unzstd(const std::string &raw, const char *path, std::string *out, char *err, size_t errlen)
{
    const ZstdApi &z = zstd_api();
    if (!z.ok) return ff::fail(FF_ERR_IO, err, errlen, "read %s: zstd support needs libzstd.so.1, which could not be loaded", path);
    void *ds = z.createDStream();
    if (!ds) return ff::fail(FF_ERR_IO, err, errlen, "read %s: zstd: out of memory", path);
    std::vector<char> buf(1 << 17);
    ZstdBuf in{const_cast<char *>(raw.data()), raw.size(), 0};
    size_t last = 0;
    while (in.pos < in.size) {
        ZstdBuf o{buf.data(), buf.size(), 0};
        last = z.decompressStream(ds, &o, &in);
        if (z.isError(last)) {
            const std::string why = z.getErrorName(last);
            z.freeDStream(ds);
            return ff::fail(FF_ERR_IO, err, errlen, "read %s: zstd: %s", path, why.c_str());
        }
        out->append(buf.data(), o.pos);
        if (o.pos == 0 && in.pos == in.size) break;
    }
    for (; last != 0;) {  // input consumed, the decoder may still hold output
        ZstdBuf o{buf.data(), buf.size(), 0};
        const size_t before = in.pos;
        last = z.decompressStream(ds, &o, &in);
        if (z.isError(last) || (o.pos == 0 && in.pos == before)) {
            z.freeDStream(ds);
            return ff::fail(FF_ERR_IO, err, errlen, "read %s: zstd: %s", path, z.isError(last) ? z.getErrorName(last) : "unexpected end of file");
        }
        out->append(buf.data(), o.pos);
    }
    z.freeDStream(ds);
    return FF_OK;
}

int unbz2(const std::string &raw, const char *path, std::string *out, char *err, size_t errlen)
{
    const Bz2Api &b = bz2_api();
    if (!b.ok) return ff::fail(FF_ERR_IO, err, errlen, "read %s: bzip2 support needs libbz2.so.1.0, which could not be loaded", path);
    std::vector<char> buf(1 << 17);
    size_t at = 0;
    while (at < raw.size()) {  // a file may hold several streams back to back
        BzStream st;
        memset(&st, 0, sizeof st);
        if (b.init(&st, 0, 0) != 0) return ff::fail(FF_ERR_IO, err, errlen, "read %s: bzip2: out of memory", path);
        st.next_in = const_cast<char *>(raw.data()) + at;
        st.avail_in = (unsigned)std::min<size_t>(raw.size() - at, 1u << 30);
        int rc = 0;
        for (;;) {
            st.next_out = buf.data();
            st.avail_out = (unsigned)buf.size();
            const unsigned in_before = st.avail_in;
            rc = b.run(&st);
            out->append(buf.data(), buf.size() - st.avail_out);
            if (rc == 4 /* BZ_STREAM_END */) break;
            if (rc != 0 /* BZ_OK */ || (st.avail_in == 0 && in_before == 0 && st.avail_out == buf.size())) {
                b.end(&st);
                return ff::fail(FF_ERR_IO, err, errlen, "read %s: bzip2: %s", path, rc == 0 ? "unexpected end of file" : "data error");
            }
            if (st.avail_in == 0) {  // next slice of a very large file
                const size_t used = (size_t)(st.next_in - raw.data());
                st.avail_in = (unsigned)std::min<size_t>(raw.size() - used, 1u << 30);
            }
        }
        at = (size_t)(st.next_in - raw.data());
        b.end(&st);
    }
    return FF_OK;
}

// One buffer as a complete zstd frame (a file may be a sequence of frames, like gzip members).
bool zstd_frame(const std::string &in, std::string *out)
{
    const ZstdApi &z = zstd_api();
    if (!z.ok) return false;
    out->resize(z.compressBound(in.size()));
    const size_t n = z.compress(&(*out)[0], out->size(), in.data(), in.size(), 1);
    if (z.isError(n)) return false;
    out->resize(n);
    return true;
}
}  // namespace

// aio.Open (frcfrc.go:93): a path ending in ".gz" (".zst", ".bz2") is decompressed on the fly (the
// reference's gostuff/aio picks the codec by suffix); anything else, and stdin, is read as is.
int read_all(const char *path, std::string *out, char *err, size_t errlen)
{
    out->clear();
    char buf[1 << 16];
    if (path && (has_suffix(path, ".zst") || has_suffix(path, ".bz2"))) {
        std::string raw;
        FILE *f = fopen(path, "rb");
        if (!f) return fail(FF_ERR_IO, err, errlen, "open %s: %s", path, strerror(errno));
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) raw.append(buf, n);
        const bool bad = ferror(f);
        fclose(f);
        if (bad) return fail(FF_ERR_IO, err, errlen, "read %s: %s", path, strerror(errno));
        return has_suffix(path, ".zst") ? unzstd(raw, path, out, err, errlen) : unbz2(raw, path, out, err, errlen);
    }
    if (path && has_suffix(path, ".gz")) {
        gzFile g = gzopen(path, "rb");
        if (!g) return fail(FF_ERR_IO, err, errlen, "open %s: %s", path, strerror(errno));
        int n;
        while ((n = gzread(g, buf, sizeof buf)) > 0) out->append(buf, (size_t)n);
        int zerr = 0;
        const char *msg = gzerror(g, &zerr);
        const bool bad = n < 0 || (zerr != Z_OK && zerr != Z_STREAM_END);
        std::string why = bad ? std::string(msg ? msg : "gzip error") : std::string();
        gzclose(g);
        if (bad) return fail(FF_ERR_IO, err, errlen, "read %s: %s", path, why.c_str());
        return FF_OK;
    }
    FILE *f = path ? fopen(path, "rb") : stdin;
    if (!f) return fail(FF_ERR_IO, err, errlen, "open %s: %s", path, strerror(errno));
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
    bool bad = ferror(f);
    if (path) fclose(f);
    if (bad) return fail(FF_ERR_IO, err, errlen, "read %s: %s", path ? path : "stdin", strerror(errno));
    return FF_OK;
}

int read_text(const char *path, unsigned threads, Text *out, char *err, size_t errlen)
{
    out->buf.reset();
    out->str.clear();
    out->data = nullptr;
    out->size = 0;
    const bool packed = path && (has_suffix(path, ".gz") || has_suffix(path, ".zst") || has_suffix(path, ".bz2"));
    if (path && !packed) {
        const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) return fail(FF_ERR_IO, err, errlen, "open %s: %s", path, strerror(errno));
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            const size_t n = (size_t)st.st_size;
            out->buf.reset(new (std::nothrow) char[n]);  // (default-initialised: not zero-filled)
            if (!out->buf) {
                ::close(fd);
                return fail(FF_ERR_IO, err, errlen, "read %s: out of memory for %zu bytes", path, n);
            }
            const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(threads, 8u), n >> 24));  // 16 MB or more each
            std::vector<int> errs(nt, 0);
            std::vector<size_t> got(nt, 0);
            parallel_for((int64_t)n, nt, [&](unsigned t, int64_t b, int64_t e) {
                int64_t at = b;
                while (at < e) {
                    const ssize_t r = pread(fd, out->buf.get() + at, (size_t)(e - at), (off_t)at);
                    if (r < 0) {
                        if (errno == EINTR) continue;
                        errs[t] = errno;
                        break;
                    }
                    if (r == 0) break;  // (the file shrank under us: what was read is the text)
                    at += r;
                }
                got[t] = (size_t)(at - b);
            });
            ::close(fd);
            size_t total = 0;
            bool whole = true;
            for (unsigned t = 0; t < nt; ++t) {
                if (errs[t]) return fail(FF_ERR_IO, err, errlen, "read %s: %s", path, strerror(errs[t]));
                if (whole) total += got[t];
                if (got[t] < (size_t)(n * (t + 1) / nt - n * t / nt)) whole = false;  // (a short part ends the text)
            }
            out->data = out->buf.get();
            out->size = total;
            return FF_OK;
        }
        ::close(fd);  // (a pipe, a device, an empty file: the general way)
    }
    const int rc = read_all(path, &out->str, err, errlen);
    out->data = out->str.data();
    out->size = out->str.size();
    return rc;
}

unsigned clamp_threads(int requested)
{
    if (requested < 1) requested = 1;
    if (requested > 256) requested = 256;
    return (unsigned)requested;
}

// The CPUs this process may really use.  A container on a large host sees every core of the machine in
// hardware_concurrency() while its cgroup grants a fraction of them (the GPU boxes of this project: 256 visible, 16
// granted): threads beyond the grant only take turns.
unsigned cpu_quota()
{
    static const unsigned q = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int c = CPU_COUNT(&set);
            if (c > 0) n = std::min(n, (unsigned)c);
        }
        auto read_file = [](const char *path, char *buf, size_t len) {
            FILE *f = fopen(path, "r");
            if (!f) return false;
            const size_t r = fread(buf, 1, len - 1, f);
            fclose(f);
            buf[r] = 0;
            return r > 0;
        };
        char b[128];
        double quota = 0;
        if (read_file("/sys/fs/cgroup/cpu.max", b, sizeof b)) {  // cgroup v2: "<quota|max> <period>"
            long long qv = 0, pv = 0;
            if (sscanf(b, "%lld %lld", &qv, &pv) == 2 && qv > 0 && pv > 0) quota = (double)qv / (double)pv;
        } else if (read_file("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", b, sizeof b)) {  // v1
            const long long qv = atoll(b);
            if (qv > 0 && read_file("/sys/fs/cgroup/cpu/cpu.cfs_period_us", b, sizeof b) && atoll(b) > 0)
                quota = (double)qv / (double)atoll(b);
        }
        if (quota > 0) n = std::min(n, (unsigned)std::max(1.0, std::ceil(quota)));
        return std::min(n, 256u);
    }();
    return q;
}

void parallel_for(int64_t n, unsigned threads,
                  const std::function<void(unsigned, int64_t, int64_t)> &fn)
{
    if (threads <= 1 || n <= 1) {
        fn(0, 0, n);
        return;
    }
    if ((int64_t)threads > n) threads = (unsigned)n;
    std::vector<std::thread> th;
    th.reserve(threads);
    for (unsigned t = 0; t < threads; ++t) {
        int64_t b = n * t / threads, e = n * (t + 1) / threads;
        th.emplace_back([&fn, t, b, e] { fn(t, b, e); });
    }
    for (auto &x : th) x.join();
}


// Compresses one buffer into a complete gzip member (RFC 1952 allows a file to be a
// sequence of members; gostuff/aio's reader, gzip(1) and zlib's gzread all read them as
// one stream), so the threads of a round can deflate their parts independently.
static bool gzip_member(const std::string &in, std::string *out)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, 1, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out->resize(deflateBound(&z, (uLong)in.size()) + 32);
    z.next_in = (Bytef *)in.data();
    z.avail_in = (uInt)in.size();
    z.next_out = (Bytef *)&(*out)[0];
    z.avail_out = (uInt)out->size();
    const int r = deflate(&z, Z_FINISH);
    out->resize(out->size() - z.avail_out);
    deflateEnd(&z);
    return r == Z_STREAM_END;
}

static bool write_fully(int fd, const char *p, size_t n, int64_t off /* -1: append at the file position */)
{
    while (n > 0) {
        const ssize_t w = off >= 0 ? pwrite(fd, p, n, (off_t)off) : write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += w;
        n -= (size_t)w;
        if (off >= 0) off += w;
    }
    return true;
}


DistWriter::~DistWriter()
{
    if (own_ && fd_ >= 0) ::close(fd_);
}

int DistWriter::open(const char *path, int threads, char *err, size_t errlen)
{
    // aio.Create (frcfrc.go:102): ".gz" / ".zst" output is compressed, by suffix
    zst_ = path && has_suffix(path, ".zst");
    if (zst_ && !zstd_api().ok)
        return fail(FF_ERR_IO, err, errlen, "open %s: zstd support needs libzstd.so.1, which could not be loaded", path);
    if (path && has_suffix(path, ".bz2"))
        return fail(FF_ERR_IO, err, errlen, "open %s: bzip2 output is not supported (the reference's aio cannot write it either)", path);
    gz_ = (path && has_suffix(path, ".gz")) || zst_;  // (gz_: the parts are compressed, each a member / frame of its own)
    name_ = path ? path : "stdout";
    fd_ = 1;
    own_ = false;
    if (path) {
        fd_ = ::open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
        if (fd_ < 0) return fail(FF_ERR_IO, err, errlen, "open %s: %s", path, strerror(errno));
        own_ = true;
    } else {
        fflush(stdout);
    }
    struct stat st;
    seekable_ = path && fstat(fd_, &st) == 0 && S_ISREG(st.st_mode);
    nt_ = clamp_threads(threads);
    bufs_.assign(nt_, std::string());
    zbufs_.assign(gz_ ? nt_ : 0, std::string());
    off_ = 0;
    return FF_OK;
}

int DistWriter::write(const double *d, int64_t n, char *err, size_t errlen)
{
    const int64_t PART = 1 << 18;  // values per thread per round (<= 25 B each)
    const unsigned nt = nt_;
    std::vector<int> ok(nt, 1);
    auto io_fail = [&] { return fail(FF_ERR_IO, err, errlen, "write %s: %s", name_.c_str(), strerror(errno)); };
    for (int64_t base = 0; base < n; base += PART * nt) {
        const int64_t m = std::min<int64_t>(PART * nt, n - base);
        for (auto &s : bufs_) s.clear();
        for (auto &s : zbufs_) s.clear();
        parallel_for(m, nt, [&](unsigned t, int64_t b, int64_t e) {
            std::string &s = bufs_[t];
            s.resize((size_t)(e - b) * 26);
            char *o = &s[0];
            for (int64_t i = b; i < e; ++i) {
                o += ff_format_float(d[base + i], o);
                *o++ = '\n';
            }
            s.resize((size_t)(o - s.data()));
            if (gz_) ok[t] = (zst_ ? zstd_frame(s, &zbufs_[t]) : gzip_member(s, &zbufs_[t])) ? 1 : 0;
        });
        std::vector<std::string> &outb = gz_ ? zbufs_ : bufs_;
        if (gz_)
            for (unsigned t = 0; t < nt; ++t)
                if (!ok[t]) return fail(FF_ERR_IO, err, errlen, "write %s: %s error", name_.c_str(), zst_ ? "zstd" : "gzip");
        if (seekable_ && nt > 1) {
            std::vector<int64_t> off(nt + 1, off_);
            for (unsigned t = 0; t < nt; ++t) off[t + 1] = off[t] + (int64_t)outb[t].size();
            int first_errno = 0;
            parallel_for(nt, nt, [&](unsigned, int64_t b, int64_t e) {
                for (int64_t t = b; t < e; ++t)
                    if (!outb[(size_t)t].empty() &&
                        !write_fully(fd_, outb[(size_t)t].data(), outb[(size_t)t].size(), off[(size_t)t])) {
                        ok[(size_t)t] = 0;
                        first_errno = errno;
                    }
            });
            for (unsigned t = 0; t < nt; ++t)
                if (!ok[t]) {
                    errno = first_errno;
                    return io_fail();
                }
            off_ = off[nt];
        } else {
            for (unsigned t = 0; t < nt; ++t) {
                if (outb[t].empty()) continue;
                // (a regular file written by one thread still goes by offset: the parts of later
                // calls must land behind these)
                if (!write_fully(fd_, outb[t].data(), outb[t].size(), seekable_ ? off_ : -1)) return io_fail();
                off_ += (int64_t)outb[t].size();
            }
        }
    }
    return FF_OK;
}

// Finished text (whole lines, formatted on the device): the parts of a call go to a regular file in parallel at
// their byte offsets; compressed output makes every part a gzip member / zstd frame of its own, as write() does.
int DistWriter::write_text(const char *text, size_t n, char *err, size_t errlen)
{
    if (n == 0) return FF_OK;
    auto io_fail = [&] { return fail(FF_ERR_IO, err, errlen, "write %s: %s", name_.c_str(), strerror(errno)); };
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt_, n >> 20));  // a part is a megabyte or more
    if (gz_) {
        std::vector<int> ok(nt, 1);
        for (auto &s : zbufs_) s.clear();
        std::vector<std::string> parts(nt);
        parallel_for((int64_t)n, nt, [&](unsigned t, int64_t b, int64_t e) {
            parts[t].assign(text + b, text + e);
            ok[t] = (zst_ ? zstd_frame(parts[t], &zbufs_[t]) : gzip_member(parts[t], &zbufs_[t])) ? 1 : 0;
        });
        for (unsigned t = 0; t < nt; ++t) {
            if (!ok[t]) return fail(FF_ERR_IO, err, errlen, "write %s: %s error", name_.c_str(), zst_ ? "zstd" : "gzip");
            if (zbufs_[t].empty()) continue;
            if (!write_fully(fd_, zbufs_[t].data(), zbufs_[t].size(), seekable_ ? off_ : -1)) return io_fail();
            off_ += (int64_t)zbufs_[t].size();
        }
        return FF_OK;
    }
    // (Buffered writes to ONE file serialise on its inode lock: 2.6 GB take 0.28 s with 1 thread or with 16.  Copying
    // the parts through a shared mapping of the file instead -- page faults do not take that lock -- was measured on the
    // GPU box and is five times slower, 1.34 s: profiles/r05_cli_c4_mmap.txt.)
    if (seekable_ && nt > 1) {
        std::vector<int> ok(nt, 1);
        int first_errno = 0;
        parallel_for((int64_t)n, nt, [&](unsigned t, int64_t b, int64_t e) {
            if (!write_fully(fd_, text + b, (size_t)(e - b), off_ + b)) {
                ok[t] = 0;
                first_errno = errno;
            }
        });
        for (unsigned t = 0; t < nt; ++t)
            if (!ok[t]) {
                errno = first_errno;
                return io_fail();
            }
    } else if (!write_fully(fd_, text, n, seekable_ ? off_ : -1)) {
        return io_fail();
    }
    off_ += (int64_t)n;
    return FF_OK;
}

int DistWriter::close(char *err, size_t errlen)
{
    int rc = FF_OK;
    if (own_ && fd_ >= 0 && ::close(fd_) != 0) rc = fail(FF_ERR_IO, err, errlen, "close %s: %s", name_.c_str(), strerror(errno));
    fd_ = -1;
    own_ = false;
    return rc;
}

}  // namespace ff

extern "C" {

const char *ff_version(void) { return "frackyfrac_amd " FF_VERSION_STRING " (gfx950)"; }

void ff_options_default(ff_options *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->weighted = 0;
    o->precision = FF_PRECISION_AUTO;
    o->device = -1;
    o->rank = 0;
    o->world = 1;
}

int ff_cpu_quota(void) { return (int)ff::cpu_quota(); }

int64_t ff_num_pairs(int64_t n) { return n < 2 ? 0 : n * (n - 1) / 2; }

// Equal-pairs contiguous row shards: rows [0, r) hold r(r-1)/2 pairs, so shard
// boundaries sit at N*sqrt(k/world), rounded to the 32-row tile height.
int ff_shard_rows(int64_t n, int32_t rank, int32_t world, int64_t *rb, int64_t *re)
{
    if (n < 0 || world < 1 || rank < 0 || rank >= world || !rb || !re) return FF_ERR_ARG;
    auto bound = [&](int32_t k) -> int64_t {
        if (k <= 0) return 0;
        if (k >= world) return n;
        double x = (double)n * std::sqrt((double)k / (double)world);
        int64_t r = (int64_t)std::llround(x / 32.0) * 32;
        if (r < 0) r = 0;
        if (r > n) r = n;
        return r;
    };
    *rb = bound(rank);
    *re = bound(rank + 1);
    if (*re < *rb) *re = *rb;
    return FF_OK;
}

// fmt.Fprintln(w, f) without the newline: strconv.FormatFloat(f, 'g', -1, 64).  The digits and the layout are
// ff_fmt_core.hpp's -- the code the device formatter runs (ff_kernels_fmt.hpp), so host and device print the same text.
int ff_format_float(double f, char *buf)
{
    uint64_t bits;
    memcpy(&bits, &f, sizeof bits);
    return ff::fmt::format_bits(bits, buf);
}

// frcfrc.go:58-62: one fmt.Fprintln(fout, f) per value.  Rounds of `threads` parts: every
// thread formats its part of the round (and, for ".gz", deflates it into a gzip member of its
// own); a regular file then takes the parts in parallel at their byte offsets (pwrite), a pipe
// or stdout in order.
int ff_write_distances(const char *path, const double *d, int64_t n, int threads,
                       char *err, size_t errlen)
{
    ff::DistWriter w;
    int rc = w.open(path, threads, err, errlen);
    if (rc == FF_OK) rc = w.write(d, n, err, errlen);
    const int rc2 = w.close(rc == FF_OK ? err : nullptr, rc == FF_OK ? errlen : 0);
    return rc != FF_OK ? rc : rc2;
}

}  // extern "C"

// ---- tuning switches: ff_tune overrides, then the environment -----------------------------

namespace {
std::mutex g_tune_mu;
std::unordered_map<std::string, std::string> g_tune;
}  // namespace

std::optional<std::string> ff::tuning(const char *name)
{
    std::lock_guard<std::mutex> lk(g_tune_mu);  // (also serialises the getenv calls of this library)
    auto it = g_tune.find(name);
    if (it != g_tune.end()) return it->second;
    const char *e = getenv(name);
    if (e) return std::string(e);
    return std::nullopt;
}

extern "C" int ff_tune(const char *name, const char *value)
{
    if (!name || strncmp(name, "FF_", 3) != 0) return FF_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (value) g_tune[name] = value;
    else g_tune.erase(name);
    return FF_OK;
}

