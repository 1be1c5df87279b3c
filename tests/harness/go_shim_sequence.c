/* go_shim_sequence.c -- the calls of bindings/go/unifrac_gpu.go, in C, call for call:
 *
 *   stream (what the shim does):
 *     ff_options_default -> ff_unifrac_dists_stream_csr(n, B, treeDists, indptr, ids, abnd, &o, chunk,
 *                                                       deliver, &handle, err, errlen)
 *     with a callback that consumes the distances one by one and returns 0 when its consumer stops.
 *   text (the shim's unifracTextGPU: the distances as the lines the reference prints, formatted on the device):
 *     ff_options_default -> ff_unifrac_text_stream_csr(n, B, treeDists, indptr, ids, abnd, &o, chunk,
 *                                                      write_text, &handle, err, errlen)
 *     with a callback that writes every piece to stdout and returns 0 once its writer has taken `stop after` pieces.
 *   plan (a host that keeps the staged plan):
 *     ff_options_default -> ff_plan_create_csr -> { ff_plan_set_shard -> ff_plan_info_get ->
 *     ff_plan_run_host [-> on FF_ERR_PRECISION: destroy, create EXACT64, set_shard, run_host] }
 *     for every shard -> ff_plan_destroy,
 *
 * fed the way the shim is fed ([][]flatNode as CSR + treeDists, every array a separate argument: no struct
 * of pointers crosses the boundary).  gcc-compiled against the public header only; the flat nodes come
 * from ff_flatten on a tree and a table file so that the test can run it on the reference's golden files.
 * Prints the distances like fmt.Fprintln(w, f); with a stop count, the consumer "breaks" after that many.
 *
 *   go_shim_sequence <stream|text|plan> <tree> <table> <dense|sparse> <weighted 0|1> [chunk pairs | shards] [stop after]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "frackyfrac_amd.h"

#define DIE(...) do { fprintf(stderr, "ERROR: " __VA_ARGS__); fprintf(stderr, "\n"); exit(2); } while (0)

typedef struct consumer { /* the shim's gpuSeq: yield is "print one line", false after stop_after lines */
    int64_t next_slot, printed, stop_after, calls, calls_after_stop;
    int stopped;
} consumer;

static int deliver(void *user, int64_t slot_begin, const double *dists, int64_t n) /* ffDeliver */
{
    consumer *c = *(consumer **)user; /* the shim passes &handle and looks the handle up */
    ++c->calls;
    if (c->stopped) {
        ++c->calls_after_stop;
        return 0;
    }
    if (slot_begin != c->next_slot) DIE("piece starts at slot %lld, expected %lld", (long long)slot_begin, (long long)c->next_slot);
    for (int64_t k = 0; k < n; ++k) {
        if (c->stop_after >= 0 && c->printed >= c->stop_after) { /* yield returned false */
            c->stopped = 1;
            return 0;
        }
        char buf[40];
        const int len = ff_format_float(dists[k], buf);
        printf("%.*s\n", len, buf);
        ++c->printed;
    }
    c->next_slot += n;
    return 1;
}

typedef struct text_sink { /* the shim's gpuText: an io.Writer that fails after stop_after writes */
    int64_t pieces, bytes, lines, stop_after, calls_after_stop;
    int stopped;
} text_sink;

static int write_text(void *user, const char *text, size_t n) /* ffWriteText */
{
    text_sink *t = *(text_sink **)user;
    if (t->stopped) {
        ++t->calls_after_stop;
        return 0;
    }
    if (t->stop_after >= 0 && t->pieces >= t->stop_after) { /* w.Write returned an error */
        t->stopped = 1;
        return 0;
    }
    if (n == 0 || text[n - 1] != '\n') DIE("a piece of %zu bytes does not end with a newline", n);
    if (fwrite(text, 1, n, stdout) != n) DIE("write to stdout failed");
    for (size_t k = 0; k < n; ++k) t->lines += text[k] == '\n';
    ++t->pieces;
    t->bytes += (int64_t)n;
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 6) DIE("usage: go_shim_sequence <stream|text|plan> <tree> <table> <dense|sparse> <weighted> [chunk|shards] [stop after]");
    const int stream = strcmp(argv[1], "stream") == 0;
    char err[1024];
    ff_tree *tree = NULL;
    ff_table *table = NULL;
    ff_flat *flat = NULL;
    if (ff_tree_read_file(argv[2], &tree, err, sizeof err)) DIE("%s", err);
    if (ff_table_read_file(argv[3], strcmp(argv[4], "sparse") == 0, &table, err, sizeof err)) DIE("%s", err);
    if (ff_validate_species(table, tree, err, sizeof err)) DIE("%s", err);
    if (ff_flatten(table, tree, 0, &flat, err, sizeof err)) DIE("%s", err);
    ff_problem p; /* only to get at the arrays: the shim has them as Go slices */
    ff_flat_problem(flat, &p);

    ff_options o;
    ff_options_default(&o);
    o.weighted = atoi(argv[5]);
    if (getenv("FF_SHIM_PRECISION")) o.precision = atoi(getenv("FF_SHIM_PRECISION"));
    const int64_t total = ff_num_pairs(p.n_samples);
    int64_t printed = 0;
    if (strcmp(argv[1], "text") == 0) {
        text_sink t = {0, 0, 0, argc > 7 ? atoll(argv[7]) : -1, 0, 0};
        text_sink *handle = &t;
        const int rc = ff_unifrac_text_stream_csr(p.n_samples, p.n_branches, p.branch_len, p.indptr, p.branch_id, p.abnd, &o,
                                                  argc > 6 ? atoll(argv[6]) : 0, write_text, &handle, err, sizeof err);
        if (rc) DIE("%s", err);
        if (t.calls_after_stop) DIE("%lld calls after the writer stopped", (long long)t.calls_after_stop);
        printed = t.lines;
        if (t.stopped) {
            fprintf(stderr, "stopped after %lld pieces, %lld lines of %lld\n", (long long)t.pieces, (long long)t.lines, (long long)total);
            printed = total;
        }
    } else if (stream) {
        consumer c = {0, 0, argc > 7 ? atoll(argv[7]) : -1, 0, 0, 0};
        consumer *handle = &c;
        const int rc = ff_unifrac_dists_stream_csr(p.n_samples, p.n_branches, p.branch_len, p.indptr, p.branch_id, p.abnd, &o,
                                                   argc > 6 ? atoll(argv[6]) : 0, deliver, &handle, err, sizeof err);
        if (rc) DIE("%s", err);
        if (c.calls_after_stop) DIE("%lld calls after the consumer stopped", (long long)c.calls_after_stop);
        printed = c.printed;
        if (c.stop_after >= 0 && c.stop_after < total) {
            if (printed != c.stop_after) DIE("%lld distances before the stop, expected %lld", (long long)printed, (long long)c.stop_after);
            fprintf(stderr, "stopped after %lld of %lld distances, %lld pieces delivered\n", (long long)printed, (long long)total,
                    (long long)c.calls);
            printed = total;
        }
    } else {
        ff_plan *plan = NULL;
        if (ff_plan_create_csr(p.n_samples, p.n_branches, p.branch_len, p.indptr, p.branch_id, p.abnd, &o, &plan, err, sizeof err))
            DIE("%s", err);
        int32_t shards = argc > 6 ? atoi(argv[6]) : (int32_t)(total / (1 << 25) + 1);
        for (int32_t r = 0; r < shards; ++r) {
            if (ff_plan_set_shard(plan, r, shards, err, sizeof err)) DIE("%s", err);
            ff_plan_info info;
            ff_plan_info_get(plan, &info);
            const int64_t m = info.slot_end - info.slot_begin;
            if (info.slot_begin != printed) DIE("shard %d starts at slot %lld, expected %lld", r, (long long)info.slot_begin, (long long)printed);
            if (m == 0) continue;
            double *part = malloc(sizeof(double) * (size_t)m);
            int rc = ff_plan_run_host(plan, part, err, sizeof err);
            if (rc == FF_ERR_PRECISION) {
                ff_plan_destroy(plan);
                plan = NULL;
                o.precision = FF_PRECISION_EXACT64;
                if (ff_plan_create_csr(p.n_samples, p.n_branches, p.branch_len, p.indptr, p.branch_id, p.abnd, &o, &plan, err, sizeof err) ||
                    ff_plan_set_shard(plan, r, shards, err, sizeof err))
                    DIE("%s", err);
                rc = ff_plan_run_host(plan, part, err, sizeof err);
            }
            if (rc) DIE("%s", err);
            for (int64_t k = 0; k < m; ++k) {
                char buf[40];
                const int len = ff_format_float(part[k], buf);
                printf("%.*s\n", len, buf);
            }
            printed += m;
            free(part);
        }
        ff_plan_destroy(plan);
    }
    if (printed != total) DIE("%lld of %lld distances", (long long)printed, (long long)total);
    ff_flat_free(flat);
    ff_table_free(table);
    ff_tree_free(tree);
    return 0;
}
