// ff_cli.cpp -- the `frcfrc` command (frcfrc/frcfrc.go:29-125) on top of the C ABI.
//
// Same flags as the reference (frcfrc.go:18-27): -i -o -t -w -s -p -l, parsed with
// the conventions of Go's flag package (-x, --x, -x=v, "-x v" for non-booleans,
// parsing stops at the first non-flag or "--"), the same validation messages
// (frcfrc.go:78-86), the same stderr phase lines and the same output: one
// distance per line in IterPairs order, formatted like fmt.Fprintln (frcfrc.go:58-62).
// Errors print "ERROR: <msg>" and exit 2 (common/common.go:13-18).
// Extensions (not in the reference): -precision auto|fixed32|exact64, -stats, -gpus N.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <thread>

#include "ff_host.hpp"

namespace {

const char *USAGE =
    "FrackyFrac calculates UniFrac on the given abundance table.\n"
    "Outputs one distance per line in the order (1,2),(1,3),(2,3)...(1,n)...(n-1,n).\n"
    "\n"
    "Params:\n";

void usage()
{
    fputs(USAGE, stderr);
    // flag.PrintDefaults(): flags in lexical order
    fputs("  -gpus int\n    \tNumber of GPUs to shard the pair space over (default 1)\n", stderr);
    fputs("  -i string\n    \tPath to input file (default stdin)\n", stderr);
    fputs("  -l\tLeave abundance values unnormalized (default normalize each sample to sum up to 1)\n", stderr);
    fputs("  -l-sorted\n    \tWith -l: merge SORTED lists (what -l evidently means; the reference leaves them unsorted, and so does -l alone)\n", stderr);
    fputs("  -o string\n    \tPath to output file (default stdout)\n", stderr);
    fputs("  -p int\n    \tNumber of threads (default 1: here, the CPUs the process may use)\n", stderr);
    fputs("  -precision string\n    \tDevice arithmetic: auto, fixed32 or exact64 (default \"auto\")\n", stderr);
    fputs("  -s\tInput is in sparse format\n", stderr);
    fputs("  -stats\n    \tPrint device statistics to stderr\n", stderr);
    fputs("  -t string\n    \tPath to tree file, required\n", stderr);
    fputs("  -w\tUse weighted UniFrac (default unweighted)\n", stderr);
}

int die(const char *msg)
{
    fprintf(stderr, "ERROR: %s\n", msg);  // common.ExitIfError
    return 2;
}

struct Flags {
    std::string in, out, tree, precision = "auto";
    bool weighted = false, sparse = false, nnorm = false, stats = false, lcompat = false, lsorted = false;
    long nt = 1, gpus = 1;
    bool nt_given = false;
};

bool parse_bool(const std::string &v, bool *out)
{
    static const char *T[] = {"1", "t", "T", "true", "TRUE", "True"};
    static const char *F[] = {"0", "f", "F", "false", "FALSE", "False"};
    for (auto s : T)
        if (v == s) {
            *out = true;
            return true;
        }
    for (auto s : F)
        if (v == s) {
            *out = false;
            return true;
        }
    return false;
}

// Returns -1 to continue, otherwise an exit code.
int parse_flags(int argc, char **argv, Flags *f)
{
    for (int a = 1; a < argc; ++a) {
        std::string s = argv[a];
        if (s.size() < 2 || s[0] != '-') break;  // first non-flag ends parsing
        size_t dashes = 1;
        if (s[1] == '-') {
            dashes = 2;
            if (s.size() == 2) break;  // "--"
        }
        std::string name = s.substr(dashes), value;
        bool has_value = false;
        size_t eq = name.find('=');
        if (eq != std::string::npos) {
            value = name.substr(eq + 1);
            name = name.substr(0, eq);
            has_value = true;
        }
        if (name.empty() || name[0] == '-' || name[0] == '=') {
            fprintf(stderr, "bad flag syntax: %s\n", s.c_str());
            usage();
            return 2;
        }
        bool *bp = name == "w" ? &f->weighted : name == "s" ? &f->sparse : name == "l" ? &f->nnorm
                   : name == "stats" ? &f->stats : name == "l-compat" ? &f->lcompat : name == "l-sorted" ? &f->lsorted : nullptr;
        if (bp) {
            if (has_value) {
                if (!parse_bool(value, bp)) {
                    fprintf(stderr, "invalid boolean value %s for -%s: parse error\n", ff::go_quote(value).c_str(), name.c_str());
                    usage();
                    return 2;
                }
            } else {
                *bp = true;
            }
            continue;
        }
        if (name == "h" || name == "help") {
            usage();
            return 0;
        }
        std::string *sp = name == "i" ? &f->in : name == "o" ? &f->out : name == "t" ? &f->tree
                          : name == "precision" ? &f->precision : nullptr;
        if (!sp && name != "p" && name != "gpus") {
            fprintf(stderr, "flag provided but not defined: -%s\n", name.c_str());
            usage();
            return 2;
        }
        if (!has_value) {
            if (a + 1 >= argc) {
                fprintf(stderr, "flag needs an argument: -%s\n", name.c_str());
                usage();
                return 2;
            }
            value = argv[++a];
        }
        if (sp) {
            *sp = value;
        } else {
            char *end = nullptr;
            errno = 0;
            long v = strtol(value.c_str(), &end, 0);
            if (value.empty() || *end || errno) {
                fprintf(stderr, "invalid value %s for flag -%s: parse error\n", ff::go_quote(value).c_str(), name.c_str());
                usage();
                return 2;
            }
            (name == "p" ? f->nt : f->gpus) = v;
            if (name == "p") f->nt_given = true;
        }
    }
    return -1;
}

std::string go_duration(double sec)
{
    char b[64];
    if (sec < 1e-6) snprintf(b, sizeof b, "%.0fns", sec * 1e9);
    else if (sec < 1e-3) snprintf(b, sizeof b, "%.3fµs", sec * 1e6);
    else if (sec < 1) snprintf(b, sizeof b, "%.6fms", sec * 1e3);
    else if (sec < 60) snprintf(b, sizeof b, "%.9fs", sec);
    else {
        long m = (long)(sec / 60);
        double s = sec - 60.0 * (double)m;
        if (m >= 60) snprintf(b, sizeof b, "%ldh%ldm%.9fs", m / 60, m % 60, s);
        else snprintf(b, sizeof b, "%ldm%.9fs", m, s);
    }
    return b;
}

}  // namespace

extern "C" int ff_frcfrc_main(int argc, char **argv)
{
    if (argc <= 1) {  // frcfrc.go:71-74
        usage();
        return 0;
    }
    Flags f;
    int rc = parse_flags(argc, argv, &f);
    if (rc >= 0) return rc;
    if (f.tree.empty()) return die("please provide a tree file with -t");  // frcfrc.go:78-80
    if (f.nt < 1) {                                                         // :81-83
        char m[64];
        snprintf(m, sizeof m, "bad number of threads: %ld", f.nt);
        return die(m);
    }
    if (f.nnorm && !f.weighted) return die("-l can only be used with weighted unifrac");  // :84-86
    // The reference's -l skips normalizeFlatNodes and with it the SORT of the lists (unifrac.go:57-59,108-110), so its
    // merge walk mis-pairs branches.  A drop-in prints what the reference prints: -l gives the reference's own
    // values, bit for bit (FF_L_REFERENCE: unsorted lists, the literal walk -- what the cgo shim does under -l too);
    // -l -l-sorted the evidently intended ones (sorted lists, raw abundances; every kernel of the engine applies).
    // (-l-compat, round 4's name for what is now the default, is still accepted.)
    if (f.lcompat && !f.nnorm) return die("-l-compat can only be used with -l");
    if (f.lsorted && !f.nnorm) return die("-l-sorted can only be used with -l");
    if (f.lsorted && f.lcompat) return die("-l-sorted and -l-compat exclude each other");
    if (f.gpus < 1 || f.gpus > 64) {
        char m[64];
        snprintf(m, sizeof m, "bad number of GPUs: %ld", f.gpus);
        return die(m);
    }
    ff_options opt;
    ff_options_default(&opt);
    opt.weighted = f.weighted;
    if (f.precision == "auto") opt.precision = FF_PRECISION_AUTO;
    else if (f.precision == "fixed32") opt.precision = FF_PRECISION_FIXED32;
    else if (f.precision == "exact64") opt.precision = FF_PRECISION_EXACT64;
    else return die("bad -precision: want auto, fixed32 or exact64");

    // -p: the reference's default is ONE thread, for a program whose every phase runs on the CPU.  Here the pair space
    // is the device's, and what is left for the host -- parsing the table, writing the file -- takes the CPUs the
    // process is allowed (its cgroup quota / affinity mask) unless -p says fewer or more.
    const int nt = f.nt_given ? (int)std::min<long>(f.nt, 256) : (int)ff::cpu_quota();
    char err[1024];
    auto t0 = std::chrono::steady_clock::now();
    auto last = t0;
    double phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // tree, load, validate, open, convert, distances, write, close
    auto lap = [&](int k) {
        auto now = std::chrono::steady_clock::now();
        phase[k] = std::chrono::duration<double>(now - last).count();
        last = now;
    };
    // the HIP context takes ~0.15 s to come up: do that while the inputs are read
    std::thread warm([&f] { ff::device_warmup((int)f.gpus); });
    struct Joiner {
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{warm};
    fputs("Reading tree\n", stderr);
    ff_tree *tree = nullptr;
    if (ff_tree_read_file(f.tree.c_str(), &tree, err, sizeof err)) return die(err);
    lap(0);

    fputs("Loading abundances\n", stderr);
    ff_table *table = nullptr;
    if (ff_table_read_file_mt(f.in.empty() ? nullptr : f.in.c_str(), f.sparse, nt, &table, err, sizeof err)) {
        ff_tree_free(tree);
        return die(err);
    }
    lap(1);

    fputs("Validating\n", stderr);
    if (ff_validate_species(table, tree, err, sizeof err)) {
        ff_table_free(table);
        ff_tree_free(tree);
        return die(err);
    }
    lap(2);

    // frcfrc.go:55: the output is opened once the input is known to be good
    ff::DistWriter writer;
    if (writer.open(f.out.empty() ? nullptr : f.out.c_str(), nt, err, sizeof err)) {
        ff_table_free(table);
        ff_tree_free(tree);
        return die(err);
    }
    lap(3);

    fputs("Converting abundances\n", stderr);  // unifrac.go:101
    std::vector<int64_t> leaf_ptr;
    ff::I64Vec leaf_idx;
    ff::F64Vec leaf_val;
    ff::table_leaf_csr(*table, *tree, &leaf_ptr, &leaf_idx, &leaf_val, nt);
    const int64_t n = ff_table_num_samples(table);
    ff_table_free(table);
    warm.join();
    // The pair space is cut into equal-pair row shards (ff_shard_rows).  -gpus G runs G of them
    // at a time, one host thread per shard, shard g of a pass on device g modulo the devices
    // present.  (Across processes the same shards are gathered: frackyfrac_amd/distributed.py.)
    // Shards hold at most 2^25 pairs (fewer when the device is short of memory): larger
    // problems run in several passes.  The distances of a pass never come to the host as numbers: they are
    // formatted where they are (ff_kernels_fmt.hpp) and the text goes to the file through a ring of pinned
    // slots while the next pass is reduced (ff::TextPipeline).  Like the reference, which streams pair by
    // pair (unifrac.go:209-228), memory does not grow with the square of the samples.
    const int ndev = ff::device_count();
    if (ndev <= 0) {
        ff_tree_free(tree);
        return die("no HIP device available; this engine has no CPU path");
    }
    const int64_t P = ff_num_pairs(n);
    int64_t budget = (int64_t)1 << 25;  // pairs per shard: 256 MB of results + up to 0.8 GB of their text, up to four times
    {
        const size_t free_b = ff::device_free_bytes(0);
        // 4 B accumulator + 8 B result + 4 x 25 B of text per pair, the staged matrix and the rest in the other 40 %
        if (free_b > 0) budget = std::min<int64_t>(budget, (int64_t)((double)free_b * 0.6 / 112.0));
        if (const char *e = getenv("FF_CLI_MAX_PAIRS"))  // (tests)
            if (atoll(e) > 0) budget = atoll(e);
        budget = std::max<int64_t>(budget, 1);
    }
    const int64_t G = f.gpus;
    int64_t passes = std::max<int64_t>(1, (P + budget * G - 1) / (budget * G));
    passes = std::min<int64_t>(passes, std::max<int64_t>(1, (n + 31) / 32));  // shards are whole 32-row blocks
    const int64_t world = G * passes;
    if (world > INT32_MAX) {
        ff_tree_free(tree);
        return die("too many samples");
    }
    // the ring's pinned memory is made ready while the plan is built (about 20 bytes of text per pair)
    ff::TextPipeline pipe(&writer);
    char ring_err[1024] = {0};
    int ring_rc = 0;
    std::thread ring([&] { ring_rc = pipe.prepare(std::min<int64_t>(P, budget * G) * 20, ring_err, sizeof ring_err); });
    struct Joiner2 {
        std::thread &t;
        ~Joiner2() { if (t.joinable()) t.join(); }
    } joiner2{ring};
    ff_plan_info info{};
    // one runner per device slot: stage A and the staging happen once -- here, under "Converting abundances", where the
    // reference does the same work (abundanceToFlatNodes + normalizeFlatNodes, unifrac.go:101-116); later passes
    // re-target the plan
    std::vector<std::unique_ptr<ff::ShardRunner>> runners;
    rc = 0;
    {
        std::vector<int> rcs((size_t)G, 0);
        std::vector<std::string> errs((size_t)G);
        for (int64_t g = 0; g < G; ++g) {
            ff_options o = opt;
            o.device = (int32_t)(g % ndev);
            runners.emplace_back(new ff::ShardRunner(tree, n, leaf_ptr.data(), leaf_idx.data(), leaf_val.data(),
                                                     f.nnorm ? (f.lsorted ? 1 : FF_L_REFERENCE) : 0, o));
        }
        auto prep = [&](int64_t g) {
            char e[1024] = {0};
            rcs[(size_t)g] = P > 0 ? runners[(size_t)g]->prepare((int32_t)g, (int32_t)world, e, sizeof e) : 0;
            errs[(size_t)g] = e;
        };
        if (G == 1) {
            prep(0);
        } else {
            std::vector<std::thread> th;
            for (int64_t g = 0; g < G; ++g) th.emplace_back(prep, g);
            for (auto &t : th) t.join();
        }
        for (int64_t g = 0; g < G && rc == 0; ++g)
            if (rcs[(size_t)g]) {
                rc = rcs[(size_t)g];
                snprintf(err, sizeof err, "%s", errs[(size_t)g].c_str());
            }
    }
    ring.join();
    if (rc == 0 && ring_rc != 0) {
        rc = ring_rc;
        snprintf(err, sizeof err, "%s", ring_err);
    }
    if (rc) {
        runners.clear();
        ff_tree_free(tree);
        return die(err);
    }
    lap(4);

    fputs("Calculating distances\n", stderr);  // unifrac.go:122
    for (int64_t pass = 0; pass < passes && rc == 0; ++pass) {
        std::vector<int> rcs((size_t)G, 0);
        std::vector<std::string> errs((size_t)G);
        std::vector<ff_plan_info> infos((size_t)G);
        std::vector<const double *> d_out((size_t)G, nullptr);
        std::vector<int64_t> n_out((size_t)G, 0);
        auto one = [&](int64_t g) {
            char e[1024] = {0};
            rcs[(size_t)g] = runners[(size_t)g]->run_device((int32_t)(pass * G + g), (int32_t)world, &d_out[(size_t)g], &n_out[(size_t)g],
                                                            &infos[(size_t)g], e, sizeof e);
            errs[(size_t)g] = e;
        };
        if (G == 1) {
            one(0);
        } else {
            std::vector<std::thread> th;
            for (int64_t g = 0; g < G; ++g) th.emplace_back(one, g);
            for (auto &t : th) t.join();
        }
        for (int64_t g = 0; g < G && rc == 0; ++g)
            if (rcs[(size_t)g]) {
                rc = rcs[(size_t)g];
                snprintf(err, sizeof err, "%s", errs[(size_t)g].c_str());
            }
        if (pass == 0) info = infos[0];
        for (int64_t g = 0; g < G && rc == 0; ++g) {  // the audit's verdict over every shard of every pass
            if (pass > 0 || g > 0) {
                info.audit_checked += infos[(size_t)g].audit_checked;
                info.audit_failed += infos[(size_t)g].audit_failed;
                info.audit_worst_rel_err = std::max(info.audit_worst_rel_err, infos[(size_t)g].audit_worst_rel_err);
                info.audit_min_headroom = std::min(info.audit_min_headroom, infos[(size_t)g].audit_min_headroom);
            }
        }
        // in IterPairs order: shard g of the pass behind shard g - 1
        for (int64_t g = 0; g < G && rc == 0; ++g)
            rc = pipe.submit(runners[(size_t)g]->device(), d_out[(size_t)g], n_out[(size_t)g], err, sizeof err);
    }
    lap(5);
    if (rc == 0) rc = pipe.drain(err, sizeof err);
    lap(6);
    const double det[5] = {runners[0]->t_create, runners[0]->t_retarget, runners[0]->t_kernels, runners[0]->t_copy, pipe.t_submit};
    const long long text_bytes = (long long)pipe.bytes;
    const double pipe_copy = pipe.t_copy, pipe_write = pipe.t_write;
    runners.clear();
    ff_tree_free(tree);
    if (rc == 0) rc = writer.close(err, sizeof err);
    if (rc) return die(err);
    if (getenv("FF_CLI_FAST_EXIT")) pipe.abandon();  // (set by the frcfrc executable, which ends right after this returns)
    lap(7);
    // Unweighted in fixed point is the reference bit for bit only when every branch length is a
    // multiple of 2^-scale; say so when it was not (the values are then within 1e-6, like weighted)
    if (!f.weighted && info.precision == FF_PRECISION_FIXED32 && !info.lengths_exact)
        fputs("Note: branch lengths are not multiples of a power of two that fits 31 bits: unweighted distances are "
              "within 1e-6 (relative) of the reference's, not bit-identical; -precision exact64 gives the reference's bits\n",
              stderr);
    if (f.stats)
        // seconds: the reference's own phases (its stderr lines); convert holds stage A and the staging on the device, as
        // the reference's "Converting abundances" holds abundanceToFlatNodes; write is what was left of the output once
        // the last pass was reduced (the rest went out under the passes).  detail: where the device side spent its time.
        fprintf(stderr,
                "{\"precision\": \"%s\", \"scale_log2\": %d, \"lengths_exact\": %d, \"bit_exact\": %s, \"tiles\": %lld, "
                "\"items\": %lld, \"wave_slots\": %lld, \"staged_bytes\": %.0f, \"passes\": %lld, \"threads\": %d, \"audit\": {\"checked\": %lld, "
                "\"failed\": %lld, \"worst_rel_err\": %.3g, \"min_headroom\": %s}, \"seconds\": {\"tree\": %.3f, "
                "\"load\": %.3f, \"validate\": %.3f, \"open\": %.3f, \"convert\": %.3f, \"distances\": %.3f, \"write\": %.3f, \"close\": %.3f}, "
                "\"detail\": {\"plan\": %.3f, \"retarget\": %.3f, \"kernels\": %.3f, \"format\": %.3f, \"text_copy\": %.3f, \"text_write\": %.3f, \"text_bytes\": %lld}}\n",
                info.precision == FF_PRECISION_FIXED32 ? "fixed32" : "exact64", info.scale_log2,
                info.lengths_exact,
                info.precision == FF_PRECISION_EXACT64 || (!f.weighted && info.lengths_exact) ? "true" : "false",
                (long long)info.n_tiles, (long long)info.n_items,
                (long long)info.n_wave_slots, info.staged_bytes, (long long)passes, nt, (long long)info.audit_checked,
                (long long)info.audit_failed, info.audit_worst_rel_err,
                std::isfinite(info.audit_min_headroom) ? std::to_string(info.audit_min_headroom).c_str() : "null", phase[0],
                phase[1], phase[2], phase[3], phase[4], phase[5], phase[6], phase[7], det[0], det[1], det[2], det[4], pipe_copy, pipe_write, text_bytes);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "Took %s\n", go_duration(sec).c_str());
    fputs("Done\n", stderr);
    return 0;
}

// ---- sprspr: dense table -> sparse table (sprspr/sprspr.go:19-44) ---------------------------

namespace {
const char *SPRSPR_USAGE =
    "SparseySparse converts dense format abundance tables to sparse format.\n"
    "\n"
    "Usage:\n"
    "sprspr < INPUT_FILE > OUTPUT_FILE\n"
    "\n"
    "Reading standard input...\n";
}

// One line per sample: its non-zero entries as name:value, tab-separated, the value as Go's %g
// prints a float64 (= %v: shortest digits that round-trip).  The reference walks a Go map, so the
// order of the entries within a line is random there; here it is the order of the table's header.
extern "C" int ff_table_write_sparse(const ff_table *table, const char *path, char *err, size_t errlen)
{
    if (!table) return ff::fail(FF_ERR_ARG, err, errlen, "null table");
    std::string out;
    char buf[40];
    const int64_t n = ff_table_num_samples(table);
    for (int64_t s = 0; s < n; ++s) {
        for (int64_t k = table->ptr[(size_t)s]; k < table->ptr[(size_t)s + 1]; ++k) {
            if (k > table->ptr[(size_t)s]) out += '\t';
            out += table->species[(size_t)table->key[(size_t)k]];
            out += ':';
            out.append(buf, (size_t)ff_format_float(table->val[(size_t)k], buf));
        }
        out += '\n';
    }
    FILE *fp = path ? fopen(path, "wb") : stdout;
    if (!fp) return ff::fail(FF_ERR_IO, err, errlen, "open %s: %s", path, strerror(errno));
    const bool ok = fwrite(out.data(), 1, out.size(), fp) == out.size() && fflush(fp) == 0;
    if (path) fclose(fp);
    return ok ? FF_OK : ff::fail(FF_ERR_IO, err, errlen, "write %s: %s", path ? path : "stdout", strerror(errno));
}

// The whole `sprspr` command (sprspr/sprspr.go:14-17): usage to stderr, stdin -> stdout.
extern "C" int ff_sprspr_main(int argc, char **argv)
{
    (void)argc;
    (void)argv;
    fputs(SPRSPR_USAGE, stderr);  // fmt.Fprintln(os.Stderr, usageMessage)
    char err[1024];
    ff_table *table = nullptr;
    int rc = ff_table_read_file_mt(nullptr, 0, 2, &table, err, sizeof err);  // parser.ParseAbundance(r, 2, ...)
    if (rc == FF_OK) rc = ff_table_write_sparse(table, nullptr, err, sizeof err);
    const int64_t n = table ? ff_table_num_samples(table) : 0;
    ff_table_free(table);
    if (rc) return die(err);
    fprintf(stderr, "%lld samples\n", (long long)n);  // (ptimer's closing line; its wording is unpinned)
    return 0;
}

