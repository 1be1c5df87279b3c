/* go_shim_sequence.c -- the call sequence of bindings/go/unifrac_gpu.go, in C, call for call:
 *   ff_options_default -> ff_plan_create -> { ff_plan_set_shard -> ff_plan_info_get ->
 *   ff_plan_run_host [-> on FF_ERR_PRECISION: destroy, create EXACT64, set_shard, run_host] }
 *   for every shard -> ff_plan_destroy,
 * fed the way the shim is fed ([][]flatNode as CSR + treeDists).  gcc-compiled against the public
 * header only; the flat nodes come from ff_flatten on a tree and a table file so that the test can
 * run it on the reference's golden files.  Prints the distances like fmt.Fprintln(w, f).
 *
 *   go_shim_sequence <tree> <table> <dense|sparse> <weighted 0|1> [shards]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "frackyfrac_amd.h"

#define DIE(...) do { fprintf(stderr, "ERROR: " __VA_ARGS__); fprintf(stderr, "\n"); exit(2); } while (0)

int main(int argc, char **argv)
{
    if (argc < 5) DIE("usage: go_shim_sequence <tree> <table> <dense|sparse> <weighted> [shards]");
    char err[1024];
    ff_tree *tree = NULL;
    ff_table *table = NULL;
    ff_flat *flat = NULL;
    if (ff_tree_read_file(argv[1], &tree, err, sizeof err)) DIE("%s", err);
    if (ff_table_read_file(argv[2], strcmp(argv[3], "sparse") == 0, &table, err, sizeof err)) DIE("%s", err);
    if (ff_validate_species(table, tree, err, sizeof err)) DIE("%s", err);
    if (ff_flatten(table, tree, 0, &flat, err, sizeof err)) DIE("%s", err);
    ff_problem p; /* what the shim builds from [][]flatNode and treeDists */
    ff_flat_problem(flat, &p);

    ff_options o;
    ff_options_default(&o);
    o.weighted = atoi(argv[4]);
    if (getenv("FF_SHIM_PRECISION")) o.precision = atoi(getenv("FF_SHIM_PRECISION"));
    ff_plan *plan = NULL;
    if (ff_plan_create(&p, &o, &plan, err, sizeof err)) DIE("%s", err);
    const int64_t total = ff_num_pairs(p.n_samples);
    int32_t shards = argc > 5 ? atoi(argv[5]) : (int32_t)(total / (1 << 25) + 1);
    int64_t printed = 0;
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
            if (ff_plan_create(&p, &o, &plan, err, sizeof err) || ff_plan_set_shard(plan, r, shards, err, sizeof err)) DIE("%s", err);
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
    if (printed != total) DIE("%lld of %lld distances", (long long)printed, (long long)total);
    ff_plan_destroy(plan);
    ff_flat_free(flat);
    ff_table_free(table);
    ff_tree_free(tree);
    return 0;
}
