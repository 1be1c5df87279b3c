// ff_table.cpp -- dense and sparse abundance-table loaders.
//
// Replaces parser.ParseAbundance (parser/parser.go:21-57) and
// parser.ParseSparseAbundance (parser/parser.go:85-100), with the validation
// rules and error texts of parseRow (:59-81), parseSparseRow (:102-127),
// splitSparse (:129-140) and iterRows (:142-155; 32 MiB maximum line).
// Instead of one Go map per sample the table keeps one species dictionary and
// CSR rows of (species index, value); a key assigned twice keeps its first
// position and its last value, which is what a map assignment leaves behind.
#include <algorithm>
#include <chrono>
#include <cmath>

#include "ff_host.hpp"

int32_t ff_table::intern(const char *b, const char *e) { return index.intern(&species, b, e); }

namespace {

const size_t MAX_LINE = (size_t)1 << 25;  // sc.Buffer(nil, 1<<25), parser.go:145

inline bool is_space(char c)
{
    // regexp \S+ (parser.go:17) splits on the RE2 \s class: \t \n \f \r and space -- NOT \v,
    // which stays inside a token (and then fails ParseFloat or the species lookup)
    return c == ' ' || c == '\t' || c == '\n' || c == '\f' || c == '\r';
}

// bufio.ScanLines: lines end at '\n', one trailing '\r' is dropped, a final
// unterminated non-empty line counts.
struct Lines {
    const char *p, *e;
    bool next(const char **b, const char **le)
    {
        if (p >= e) return false;
        const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
        const char *end = q ? q : e;
        *b = p;
        *le = (end > p && end[-1] == '\r') ? end - 1 : end;
        p = q ? q + 1 : e;
        return true;
    }
};

struct Row {  // one sample being assembled, with last-assignment-wins keys
    std::vector<int32_t> key;
    std::vector<double> val;
    // where[k] is key k's position in this row iff stamp[k] == cur (no per-row clearing)
    std::vector<uint32_t> stamp, where;
    uint32_t cur = 0;
    void clear()
    {
        key.clear();
        val.clear();
        if (++cur == 0) {  // the stamp wrapped: forget everything once
            std::fill(stamp.begin(), stamp.end(), 0u);
            cur = 1;
        }
    }
    void set(int32_t k, double v)
    {
        if ((size_t)k >= stamp.size()) {
            const size_t n = std::max<size_t>((size_t)k + 1, stamp.size() * 2);
            stamp.resize(n, 0u);
            where.resize(n, 0u);
        }
        if (stamp[(size_t)k] == cur) {
            val[where[(size_t)k]] = v;
            return;
        }
        stamp[(size_t)k] = cur;
        where[(size_t)k] = (uint32_t)key.size();
        key.push_back(k);
        val.push_back(v);
    }
};

int bad_value(int i, double f, char *err, size_t errlen)
{
    return ff::fail(FF_ERR_PARSE, err, errlen, "value #%d: bad value: %s", i, ff::go_f(f).c_str());
}

int bad_float(int i, const char *b, const char *e, const char *why, char *err, size_t errlen)
{
    std::string tok(b, e);
    return ff::fail(FF_ERR_PARSE, err, errlen, "value #%d: strconv.ParseFloat: parsing %s: %s", i,
                    ff::go_quote(tok).c_str(), why);
}



// ---- row parsers (one line each); keys are species ids of `names` (dense) or of the
// ---- caller's interner (sparse)

struct LineRef {
    const char *b, *e;
};

int parse_dense_row(const char *b, const char *e, const std::vector<int32_t> &names, Row *row, char *err,
                    size_t errlen)
{
    // parseRow, parser.go:59-81: the count is checked before any value is parsed
    size_t nparts = 0;
    for (const char *p = b; p < e;) {
        while (p < e && is_space(*p)) ++p;
        const char *q = p;
        while (q < e && !is_space(*q)) ++q;
        if (q > p) ++nparts;
        p = q;
    }
    if (nparts != names.size())
        return ff::fail(FF_ERR_PARSE, err, errlen, "has %zu values, expected %zu", nparts, names.size());
    row->clear();
    size_t i = 0;
    for (const char *p = b; p < e;) {
        while (p < e && is_space(*p)) ++p;
        const char *q = p;
        while (q < e && !is_space(*q)) ++q;
        if (q > p) {
            double f;
            const char *why;
            if (!ff::go_parse_float(p, q, &f, &why)) return bad_float((int)i + 1, p, q, why, err, errlen);
            if (std::isnan(f) || std::isinf(f) || f < 0) return bad_value((int)i + 1, f, err, errlen);
            if (f != 0) row->set(names[i], f);
            ++i;
        }
        p = q;
    }
    return FF_OK;
}

template <class Intern>
int parse_sparse_row(const char *b, const char *e, Intern &&intern, Row *row, char *err, size_t errlen)
{
    row->clear();  // a blank line is an empty sample (parser_test.go:29-35)
    int i = 0;
    for (const char *p = b; p < e;) {
        while (p < e && is_space(*p)) ++p;
        const char *q = p;
        while (q < e && !is_space(*q)) ++q;
        if (q > p) {
            ++i;
            const char *colon = nullptr;  // splitSparse: the LAST colon (parser.go:129-140)
            for (const char *c = q; c > p;)
                if (*--c == ':') {
                    colon = c;
                    break;
                }
            if (!colon) {
                std::string tok(p, q);
                return ff::fail(FF_ERR_PARSE, err, errlen, "value #%d: no colon in %s", i, ff::go_quote(tok).c_str());
            }
            if (colon == p) return ff::fail(FF_ERR_PARSE, err, errlen, "value #%d: empty species name", i);
            double f;
            const char *why;
            if (!ff::go_parse_float(colon + 1, q, &f, &why)) return bad_float(i, colon + 1, q, why, err, errlen);
            if (std::isnan(f) || std::isinf(f) || f < 0) return bad_value(i, f, err, errlen);
            if (f == 0)
                return ff::fail(FF_ERR_PARSE, err, errlen, "value #%d: zeros are not allowed in sparse format", i);
            row->set(intern(p, colon), f);
        }
        p = q;
    }
    return FF_OK;
}

// The rows of [first, last) parsed by one thread into a private partial table.
struct Partial {
    std::vector<std::string> names;    // sparse only: thread-local dictionary
    ff::NameIndex dict;
    std::vector<int64_t> cnt;          // entries per row
    std::vector<int32_t> key;
    std::vector<double> val;
    int64_t err_line = -1;             // first failing line of this thread
    int err_code = FF_OK;
    char err[512] = {0};
};

int parse_table(const char *text, size_t len, bool sparse, int nthreads, ff_table **out, char *err, size_t errlen)
{
    if (!text || !out) return ff::fail(FF_ERR_ARG, err, errlen, "ff_table_parse: null argument");
    std::vector<LineRef> lines;
    {
        Lines ls{text, text + len};
        const char *b, *e;
        while (ls.next(&b, &e)) {
            if ((size_t)(e - b) > MAX_LINE) return ff::fail(FF_ERR_PARSE, err, errlen, "bufio.Scanner: token too long");
            lines.push_back({b, e});
        }
    }
    auto *t = new ff_table();
    t->ptr.push_back(0);
    std::vector<int32_t> names;
    size_t first = 0;
    if (!sparse) {  // header, parser.go:32-40
        if (lines.empty()) {
            *out = t;
            return FF_OK;
        }
        for (const char *p = lines[0].b; p < lines[0].e;) {
            while (p < lines[0].e && is_space(*p)) ++p;
            const char *q = p;
            while (q < lines[0].e && !is_space(*q)) ++q;
            if (q > p) names.push_back(t->intern(p, q));
            p = q;
        }
        if (names.empty()) {
            delete t;
            return ff::fail(FF_ERR_PARSE, err, errlen, "row #1 has 0 values");
        }
        first = 1;
    }
    const int64_t nrows = (int64_t)lines.size() - (int64_t)first;
    unsigned nt = ff::clamp_threads(nthreads);
    if ((int64_t)nt > std::max<int64_t>(nrows, 1)) nt = (unsigned)std::max<int64_t>(nrows, 1);
    std::vector<Partial> parts(nt);
    ff::parallel_for(nrows, nt, [&](unsigned tid, int64_t b, int64_t e) {
        Partial &pt = parts[tid];
        Row row;
        auto intern = [&pt](const char *p, const char *q) -> int32_t { return pt.dict.intern(&pt.names, p, q); };
        for (int64_t r = b; r < e; ++r) {
            const LineRef &ln = lines[first + (size_t)r];
            const int rc = sparse ? parse_sparse_row(ln.b, ln.e, intern, &row, pt.err, sizeof pt.err)
                                  : parse_dense_row(ln.b, ln.e, names, &row, pt.err, sizeof pt.err);
            if (rc) {
                pt.err_line = r;
                pt.err_code = rc;
                return;
            }
            pt.cnt.push_back((int64_t)row.key.size());
            pt.key.insert(pt.key.end(), row.key.begin(), row.key.end());
            pt.val.insert(pt.val.end(), row.val.begin(), row.val.end());
        }
    });
    // the first failing row in row order wins, as with the reference's ordered pipeline
    for (const Partial &pt : parts)
        if (pt.err_code) {
            delete t;
            return ff::fail(pt.err_code, err, errlen, "%s", pt.err);
        }
    // merge: thread dictionaries into the table's (first appearance in row order keeps the
    // lower id), then every partial copies its rows to its place in parallel
    std::vector<size_t> key_at(parts.size() + 1, 0), row_at(parts.size() + 1, 0);
    std::vector<std::vector<int32_t>> remap(parts.size());
    for (size_t q = 0; q < parts.size(); ++q) {
        const Partial &pt = parts[q];
        key_at[q + 1] = key_at[q] + pt.key.size();
        row_at[q + 1] = row_at[q] + pt.cnt.size();
        if (sparse) {
            remap[q].resize(pt.names.size());
            for (size_t k = 0; k < pt.names.size(); ++k)
                remap[q][k] = t->intern(pt.names[k].data(), pt.names[k].data() + pt.names[k].size());
        }
    }
    t->key.resize(key_at.back());
    t->val.resize(key_at.back());
    t->ptr.resize(row_at.back() + 1);
    ff::parallel_for((int64_t)parts.size(), (unsigned)parts.size(), [&](unsigned, int64_t qb, int64_t qe) {
        for (int64_t q = qb; q < qe; ++q) {
            const Partial &pt = parts[(size_t)q];
            int32_t *kd = t->key.data() + key_at[(size_t)q];
            if (sparse)
                for (size_t k = 0; k < pt.key.size(); ++k) kd[k] = remap[(size_t)q][(size_t)pt.key[k]];
            else if (!pt.key.empty())
                memcpy(kd, pt.key.data(), sizeof(int32_t) * pt.key.size());
            if (!pt.val.empty()) memcpy(t->val.data() + key_at[(size_t)q], pt.val.data(), sizeof(double) * pt.val.size());
            int64_t at = (int64_t)key_at[(size_t)q];
            for (size_t r = 0; r < pt.cnt.size(); ++r) {
                at += pt.cnt[r];
                t->ptr[row_at[(size_t)q] + r + 1] = at;
            }
        }
    });
    *out = t;
    return FF_OK;
}

}  // namespace

extern "C" {

int ff_table_parse_dense(const char *text, size_t len, ff_table **out, char *err, size_t errlen)
{
    return parse_table(text, len, false, 1, out, err, errlen);
}

int ff_table_parse_sparse(const char *text, size_t len, ff_table **out, char *err, size_t errlen)
{
    return parse_table(text, len, true, 1, out, err, errlen);
}

int ff_table_parse_mt(const char *text, size_t len, int sparse, int threads, ff_table **out, char *err,
                      size_t errlen)
{
    return parse_table(text, len, sparse != 0, threads, out, err, errlen);
}

int ff_table_read_file_mt(const char *path, int sparse, int threads, ff_table **table, char *err, size_t errlen)
{
    ff::Text text;
    int rc = ff::read_text(path, ff::clamp_threads(threads), &text, err, errlen);
    if (rc) return rc;
    return parse_table(text.data, text.size, sparse != 0, threads, table, err, errlen);
}

int ff_table_read_file(const char *path, int sparse, ff_table **table, char *err, size_t errlen)
{
    return ff_table_read_file_mt(path, sparse, 1, table, err, errlen);
}

void ff_table_free(ff_table *t) { delete t; }
int64_t ff_table_num_samples(const ff_table *t) { return t ? (int64_t)t->ptr.size() - 1 : 0; }
int64_t ff_table_sample_size(const ff_table *t, int64_t s)
{
    if (!t || s < 0 || s + 1 >= (int64_t)t->ptr.size()) return -1;
    return t->ptr[(size_t)s + 1] - t->ptr[(size_t)s];
}
int ff_table_sample_entry(const ff_table *t, int64_t s, int64_t k, const char **name, double *value)
{
    int64_t n = ff_table_sample_size(t, s);
    if (n < 0 || k < 0 || k >= n) return FF_ERR_ARG;
    size_t at = (size_t)(t->ptr[(size_t)s] + k);
    if (name) *name = t->species[(size_t)t->key[at]].c_str();
    if (value) *value = t->val[at];
    return FF_OK;
}

}  // extern "C"
