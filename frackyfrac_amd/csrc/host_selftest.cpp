// host_selftest.cpp -- exercises the host-side C ABI (tree, tables, validation, stage A,
// formatting, writer, flag parsing) without touching the device; built with
// -fsanitize=address,undefined by `make asan` and run by tests/test_sanitizers.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>
#include <zlib.h>

#include "ff_host.hpp"
#include "frackyfrac_amd.h"

static int fails = 0;
#define CHECK(c)                                                     \
    do {                                                             \
        if (!(c)) {                                                  \
            fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); \
            ++fails;                                                 \
        }                                                            \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main()
{
    char err[256];
    // trees: valid, odd, broken, and random byte soup (must fail cleanly, never crash)
    const char *trees[] = {"(s2:3,s1:1,s3:5);", "((s1:1,s2:3):2,(s3:2,s4:5):1);", "a;", " ( 'it''s':1e-1,[c]b:2,(,c:3)x:4 )r:.5;",
                           "", "(", "(a,b", "(a,b));", "(a:x);", "((((((;", "'unterminated", "(a:1,b:2)c:3", "[[[", "(a,b)[x];"};
    for (const char *t : trees) {
        ff_tree *tr = nullptr;
        int rc = ff_tree_parse(t, strlen(t), &tr, err, sizeof err);
        if (rc == 0) {
            CHECK(ff_tree_num_nodes(tr) >= 1);
            for (int64_t k = 0; k < ff_tree_num_nodes(tr); ++k) CHECK(ff_tree_name(tr, k) != nullptr);
            CHECK(ff_tree_name(tr, -1) == nullptr);
            ff_tree_free(tr);
        } else {
            CHECK(tr == nullptr && strlen(err) > 0);
        }
    }
    for (int it = 0; it < 3000; ++it) {
        std::string s;
        const char alphabet[] = "(),:;'[] ab1.e-\n";
        size_t n = rnd() % 40;
        for (size_t k = 0; k < n; ++k) s += alphabet[rnd() % (sizeof alphabet - 1)];
        ff_tree *tr = nullptr;
        if (ff_tree_parse(s.data(), s.size(), &tr, err, sizeof err) == 0) ff_tree_free(tr);
    }
    // tables
    const char *dense[] = {"   aa  bbbb    \n1\t2\n 3  \t  4 \t\n", "a a b\n1 2 3\n", "", "\n", "a\nx\n", "a b\n1\n", "a\n-1\n", "a\n1e999\n", "a\r\n1\r\n"};
    for (const char *t : dense)
        for (int nt : {1, 3}) {
            ff_table *tb = nullptr;
            if (ff_table_parse_mt(t, strlen(t), 0, nt, &tb, err, sizeof err) == 0) {
                for (int64_t s = 0; s < ff_table_num_samples(tb); ++s)
                    for (int64_t k = 0; k < ff_table_sample_size(tb, s); ++k) {
                        const char *nm;
                        double v;
                        CHECK(ff_table_sample_entry(tb, s, k, &nm, &v) == 0 && v > 0);
                    }
                CHECK(ff_table_sample_entry(tb, 99, 0, nullptr, nullptr) != 0);
                ff_table_free(tb);
            }
        }
    const char *sparse[] = {"a:11 b:222  \n  b:32 c:7\n\nd:1\tc:4\ta:10\n", "c:d:e::5\n", "a:1 b\n", ":1\n", "a:0\n", "a:nan\n", "a:", "\n\n"};
    for (const char *t : sparse)
        for (int nt : {1, 4}) {
            ff_table *tb = nullptr;
            if (ff_table_parse_mt(t, strlen(t), 1, nt, &tb, err, sizeof err) == 0) ff_table_free(tb);
        }
    for (int it = 0; it < 3000; ++it) {
        std::string s;
        const char alphabet[] = "ab:1.e-+ \t\n0xnI";
        size_t n = rnd() % 60;
        for (size_t k = 0; k < n; ++k) s += alphabet[rnd() % (sizeof alphabet - 1)];
        ff_table *tb = nullptr;
        if (ff_table_parse_mt(s.data(), s.size(), (int)(rnd() & 1), 1 + (int)(rnd() % 3), &tb, err, sizeof err) == 0) ff_table_free(tb);
    }
    // validation + stage A on the reference's weighted example
    {
        ff_tree *tr = nullptr;
        ff_table *tb = nullptr;
        const char *t = "((s1:1,s2:3):2,(s3:2,s4:5):1);", *a = "s1 s2 s3 s4\n4 1 0 0\n0 2 3 0\n";
        CHECK(ff_tree_parse(t, strlen(t), &tr, err, sizeof err) == 0);
        CHECK(ff_table_parse_dense(a, strlen(a), &tb, err, sizeof err) == 0);
        CHECK(ff_validate_species(tb, tr, err, sizeof err) == 0);
        for (int leave = 0; leave < 2; ++leave) {
            ff_flat *fl = nullptr;
            CHECK(ff_flatten(tb, tr, leave, &fl, err, sizeof err) == 0);
            ff_problem p;
            ff_flat_problem(fl, &p);
            CHECK(p.n_samples == 2 && p.n_branches == 7 && p.indptr[2] == 9);
            double sum = 0;
            for (int64_t k = p.indptr[0]; k < p.indptr[1]; ++k) sum += p.abnd[k];
            CHECK(leave ? sum == 15.0 : std::fabs(sum - 1.0) < 1e-15);
            ff_flat_free(fl);
        }
        ff_table *bad = nullptr;
        CHECK(ff_table_parse_sparse("zz:1\n", 5, &bad, err, sizeof err) == 0);
        CHECK(ff_validate_species(bad, tr, err, sizeof err) == FF_ERR_SPECIES);
        CHECK(strcmp(err, "sample #1 has value 1 for species \"zz\" which is not in the tree") == 0);
        ff_table_free(bad);
        ff_table_free(tb);
        ff_tree_free(tr);
    }
    // formatting: random doubles round-trip; writer
    {
        char buf[40];
        std::vector<double> v;
        for (int it = 0; it < 20000; ++it) {
            uint64_t bits = rnd();
            double d;
            memcpy(&d, &bits, 8);
            int n = ff_format_float(d, buf);
            CHECK(n > 0 && n < 32);
            buf[n] = 0;
            if (std::isfinite(d)) CHECK(strtod(buf, nullptr) == d);
            v.push_back((double)(rnd() >> 11) * 0x1p-53);
        }
        CHECK(ff_write_distances("/dev/null", v.data(), (int64_t)v.size(), 3, err, sizeof err) == 0);
    }
    // the large-text reader (parallel preads into an unzeroed buffer) and the writer's text path (device-formatted
    // text: plain parts by offset, gzip members, a non-seekable sink)
    {
        char path[] = "/tmp/ff_selftest_XXXXXX";
        const int fd = mkstemp(path);
        CHECK(fd >= 0);
        std::string big;
        while (big.size() < (size_t)40 << 20) {
            char line[64];
            big.append(line, (size_t)snprintf(line, sizeof line, "%.17g\n", (double)(rnd() >> 11) * 0x1p-53));
        }
        CHECK(write(fd, big.data(), big.size()) == (ssize_t)big.size());
        close(fd);
        for (unsigned nt : {1u, 2u, 3u}) {
            ff::Text t;
            CHECK(ff::read_text(path, nt, &t, err, sizeof err) == 0);
            CHECK(t.size == big.size() && memcmp(t.data, big.data(), big.size()) == 0);
        }
        {
            ff::Text t;
            CHECK(ff::read_text("/nonexistent/file", 2, &t, err, sizeof err) == FF_ERR_IO && strlen(err) > 0);
        }
        for (int threads : {1, 4}) {
            ff::DistWriter w;
            CHECK(w.open(path, threads, err, sizeof err) == 0);
            const size_t cut = big.size() / 3;
            CHECK(w.write_text(big.data(), cut, err, sizeof err) == 0);
            CHECK(w.write_text(big.data() + cut, big.size() - cut, err, sizeof err) == 0);
            CHECK(w.write_text(big.data(), 0, err, sizeof err) == 0);
            CHECK(w.close(err, sizeof err) == 0);
            ff::Text t;
            CHECK(ff::read_text(path, 2, &t, err, sizeof err) == 0);
            CHECK(t.size == big.size() && memcmp(t.data, big.data(), big.size()) == 0);
        }
        {
            std::string gz = std::string(path) + ".gz";
            ff::DistWriter w;
            CHECK(w.open(gz.c_str(), 3, err, sizeof err) == 0);
            CHECK(w.write_text(big.data(), big.size() / 2, err, sizeof err) == 0);
            CHECK(w.write_text(big.data() + big.size() / 2, big.size() - big.size() / 2, err, sizeof err) == 0);
            CHECK(w.close(err, sizeof err) == 0);
            std::string back;
            CHECK(ff::read_all(gz.c_str(), &back, err, sizeof err) == 0);
            CHECK(back == big);
            unlink(gz.c_str());
        }
        {
            ff::DistWriter w;
            CHECK(w.open("/dev/null", 4, err, sizeof err) == 0);
            CHECK(w.write_text(big.data(), big.size(), err, sizeof err) == 0);
            CHECK(w.close(err, sizeof err) == 0);
        }
        unlink(path);
    }
    // shards and flag parsing paths that end before the device
    {
        int64_t rb, re, prev = 0;
        for (int r = 0; r < 5; ++r) {
            CHECK(ff_shard_rows(1000, r, 5, &rb, &re) == 0 && rb == prev);
            prev = re;
        }
        CHECK(prev == 1000 && ff_shard_rows(10, 5, 5, &rb, &re) != 0);
        const char *argv1[] = {"frcfrc", "-w"};
        CHECK(ff_frcfrc_main(2, (char **)argv1) == 2);
        const char *argv2[] = {"frcfrc", "-t", "x", "-l"};
        CHECK(ff_frcfrc_main(4, (char **)argv2) == 2);
        const char *argv3[] = {"frcfrc", "-t", "/nonexistent.tree"};
        CHECK(ff_frcfrc_main(3, (char **)argv3) == 2);
    }
    if (fails) {
        fprintf(stderr, "%d checks failed\n", fails);
        return 1;
    }
    puts("host selftest ok");
    return 0;
}
