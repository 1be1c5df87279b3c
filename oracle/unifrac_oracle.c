/*
 * oracle/unifrac_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement, in plain C, of the UniFrac hot path of fluhus/frackyfrac
 * (reference snapshot 2025-01-03).  It exists so that the HIP path can be
 * checked against the reference's algorithm on machines where the reference
 * itself (Go 1.23 + un-vendored modules) cannot be built.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (frackyfrac_amd/csrc) never links or calls anything here.
 *
 * Pinning: the restatement reproduces every golden vector the reference holds
 * for this path (testdata/{uwtd1,uwtd2,wtd}.want, frcfrc/unifrac_test.go:22,44,65,
 * common/common_test.go:9-10); see tests/test_oracle_golden.py.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference root).  All arithmetic is IEEE binary64 in the reference's order;
 * build with -ffp-contract=off so no FMA contraction changes a rounding.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* frcfrc/unifrac.go:137-140  type flatNode struct { id int; abnd float64 } */
typedef struct {
    int64_t id;
    double abnd;
} orc_flatnode;

/*
 * frcfrc/unifrac.go:32-53  abundanceToFlatNodes.
 *
 * The tree is given in the numbering enumerateNodes assigns (unifrac.go:127-133:
 * pre-order, root = 0), as subtree sizes: the children of `id` are id+1,
 * id+1+size[id+1], ... while < id+size[id].  leaf_abnd[id] is abnd[tree.Name]
 * for that node (0 when the map has no such key); like the reference with
 * flatNodeOptimization = true (unifrac.go:18,38-43) it is consulted for leaves
 * only.  The recursion is unrolled onto an explicit stack but keeps the
 * reference's order of float additions: children left to right, then the
 * node's own value (:34-43), and appends {id,sum} iff sum > 0 (:49-51), which
 * yields the list in post-order.
 *
 * out must have room for n nodes.  Returns the number of flat nodes written.
 */
int64_t orc_abundance_to_flat_nodes(int64_t n, const int64_t *size,
                                    const double *leaf_abnd, orc_flatnode *out)
{
    if (n <= 0) return 0;
    int64_t *stk_id = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t *stk_next = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    double *stk_sum = (double *)malloc(sizeof(double) * (size_t)n);
    int64_t sp = 0, nout = 0;
    stk_id[0] = 0;
    stk_next[0] = 1;
    stk_sum[0] = 0.0;
    while (sp >= 0) {
        int64_t id = stk_id[sp];
        int64_t end = id + size[id];
        if (stk_next[sp] < end) { /* for _, c := range tree.Children (:35) */
            int64_t c = stk_next[sp];
            stk_next[sp] = c + size[c];
            ++sp;
            stk_id[sp] = c;
            stk_next[sp] = c + 1;
            stk_sum[sp] = 0.0;
            continue;
        }
        double sum = stk_sum[sp];
        if (size[id] == 1) { /* len(tree.Children) == 0 (:39) */
            double a = leaf_abnd[id];
            if (a > 0) sum += a; /* :40-42 */
        }
        if (sum > 0) { /* :49-51 */
            out[nout].id = id;
            out[nout].abnd = sum;
            ++nout;
        }
        --sp;
        if (sp >= 0) stk_sum[sp] += sum; /* sum += abundanceToFlatNodes(...) (:36) */
    }
    free(stk_id);
    free(stk_next);
    free(stk_sum);
    return nout;
}

static int cmp_flatnode_id(const void *a, const void *b)
{
    int64_t x = ((const orc_flatnode *)a)->id, y = ((const orc_flatnode *)b)->id;
    return (x > y) - (x < y);
}

/*
 * frcfrc/unifrac.go:56-67  normalizeFlatNodes: sort by id (:57-59), sum the
 * abundances of ALL flat nodes -- leaves, internal nodes and root -- in
 * ascending id order (:60-63), divide each by that sum (:64-66).  ids are
 * unique, so the unstable sort.Slice has one possible result.
 */
void orc_normalize_flat_nodes(orc_flatnode *nodes, int64_t n)
{
    qsort(nodes, (size_t)n, sizeof(orc_flatnode), cmp_flatnode_id);
    double sum = 0.0;
    for (int64_t i = 0; i < n; ++i) sum += nodes[i].abnd;
    for (int64_t i = 0; i < n; ++i) nodes[i].abnd /= sum;
}

/* Sort only: the evidently intended behaviour of -l;
 * the reference itself skips the sort together with the division
 * (unifrac.go:108-110), see orc_unifrac_dists' callers in oracle.py. */
void orc_sort_flat_nodes(orc_flatnode *nodes, int64_t n)
{
    qsort(nodes, (size_t)n, sizeof(orc_flatnode), cmp_flatnode_id);
}

/* frcfrc/unifrac.go:144-171  unifracDistUnweighted. */
double orc_dist_unweighted(const orc_flatnode *a, int64_t na, const orc_flatnode *b,
                           int64_t nb, const double *tree_dists)
{
    double result = 0.0, common = 0.0;
    int64_t i = 0, j = 0;
    while (i < na && j < nb) {
        if (a[i].id < b[j].id) {
            result += tree_dists[a[i].id];
            ++i;
            continue;
        }
        if (a[i].id > b[j].id) {
            result += tree_dists[b[j].id];
            ++j;
            continue;
        }
        common += tree_dists[a[i].id];
        ++i;
        ++j;
    }
    for (; i < na; ++i) result += tree_dists[a[i].id];
    for (; j < nb; ++j) result += tree_dists[b[j].id];
    result /= (result + common);
    return result;
}

/* frcfrc/unifrac.go:174-205  unifracDistWeighted. */
double orc_dist_weighted(const orc_flatnode *a, int64_t na, const orc_flatnode *b,
                         int64_t nb, const double *tree_dists)
{
    double numer = 0.0, denom = 0.0;
    int64_t i = 0, j = 0;
    while (i < na && j < nb) {
        if (a[i].id < b[j].id) {
            numer += tree_dists[a[i].id] * a[i].abnd;
            denom += tree_dists[a[i].id] * a[i].abnd;
            ++i;
            continue;
        }
        if (a[i].id > b[j].id) {
            numer += tree_dists[b[j].id] * b[j].abnd;
            denom += tree_dists[b[j].id] * b[j].abnd;
            ++j;
            continue;
        }
        numer += tree_dists[a[i].id] * fabs(a[i].abnd - b[j].abnd);
        denom += tree_dists[a[i].id] * (a[i].abnd + b[j].abnd);
        ++i;
        ++j;
    }
    for (; i < na; ++i) {
        numer += tree_dists[a[i].id] * a[i].abnd;
        denom += tree_dists[a[i].id] * a[i].abnd;
    }
    for (; j < nb; ++j) {
        numer += tree_dists[b[j].id] * b[j].abnd;
        denom += tree_dists[b[j].id] * b[j].abnd;
    }
    return numer / denom;
}

/*
 * frcfrc/unifrac.go:209-228  unifracDists + common/common.go:21-31 IterPairs.
 *
 * Samples are CSR: sample s owns nodes[indptr[s] .. indptr[s+1]).  Output slot
 * k = i*(i-1)/2 + j holds the distance of the pair IterPairs yields k-th, i.e.
 * {s[i], s[j]} for i = 0..N-1, j = 0..i-1 (element 0 = the higher index, so
 * the distance functions get (a = sample i, b = sample j)).  Only slots
 * [pair_begin, pair_end) are computed (bench.py times a bounded prefix); the
 * reference's ppln.Serial worker pool (unifrac.go:212) is restated as
 * nthreads threads over contiguous slot ranges -- per-pair arithmetic is
 * unchanged and the ordered delivery is the slot index.
 */
typedef struct {
    const int64_t *indptr;
    const orc_flatnode *nodes;
    const double *tree_dists;
    int weighted;
    int64_t begin, end;
    double *out;
} orc_job;

static void pair_of_slot(int64_t k, int64_t *pi, int64_t *pj)
{
    /* invert k = i(i-1)/2 + j, 0 <= j < i */
    int64_t i = (int64_t)((1.0 + sqrt(1.0 + 8.0 * (double)k)) / 2.0);
    while (i * (i - 1) / 2 > k) --i;
    while ((i + 1) * i / 2 <= k) ++i;
    *pi = i;
    *pj = k - i * (i - 1) / 2;
}

static void *orc_worker(void *arg)
{
    orc_job *jb = (orc_job *)arg;
    if (jb->begin >= jb->end) return NULL;
    int64_t i, j;
    pair_of_slot(jb->begin, &i, &j);
    for (int64_t k = jb->begin; k < jb->end; ++k) {
        const orc_flatnode *a = jb->nodes + jb->indptr[i];
        const orc_flatnode *b = jb->nodes + jb->indptr[j];
        int64_t na = jb->indptr[i + 1] - jb->indptr[i];
        int64_t nb = jb->indptr[j + 1] - jb->indptr[j];
        jb->out[k] = jb->weighted ? orc_dist_weighted(a, na, b, nb, jb->tree_dists)
                                  : orc_dist_unweighted(a, na, b, nb, jb->tree_dists);
        if (++j == i) { /* for j := range i (common.go:24) */
            ++i;
            j = 0;
        }
    }
    return NULL;
}

int orc_unifrac_dists(int64_t n_samples, const int64_t *indptr, const orc_flatnode *nodes,
                      const double *tree_dists, int weighted, int nthreads,
                      int64_t pair_begin, int64_t pair_end, double *out)
{
    int64_t npairs = n_samples * (n_samples - 1) / 2;
    if (pair_begin < 0) pair_begin = 0;
    if (pair_end > npairs) pair_end = npairs;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    orc_job jobs[256];
    pthread_t th[256];
    int64_t total = pair_end - pair_begin;
    if (total <= 0) return 0;
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].indptr = indptr;
        jobs[t].nodes = nodes;
        jobs[t].tree_dists = tree_dists;
        jobs[t].weighted = weighted;
        jobs[t].begin = pair_begin + total * t / nthreads;
        jobs[t].end = pair_begin + total * (t + 1) / nthreads;
        jobs[t].out = out;
    }
    if (nthreads == 1) {
        orc_worker(&jobs[0]);
        return 0;
    }
    for (int t = 0; t < nthreads; ++t)
        if (pthread_create(&th[t], NULL, orc_worker, &jobs[t]) != 0) return -1;
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    return 0;
}

/*
 * Stage A for many samples at once (frcfrc/unifrac.go:102-116): sample s has
 * leaf values leaf_idx/leaf_val[leaf_ptr[s] .. leaf_ptr[s+1]) (node ids of the
 * leaves that carry abundance; a duplicated leaf name appears once per node).
 * Writes CSR flat nodes; out_nodes needs room for n_samples * n entries in the
 * worst case, so callers size it from a first pass with out_nodes == NULL,
 * which only fills out_indptr.  mode: 0 = normalise (default path, :109),
 * 1 = leave post-order and un-normalised (the reference under -l, :108-110),
 * 2 = sort only (intended -l semantics).
 */
int orc_flatten_samples(int64_t n, const int64_t *size, int64_t n_samples,
                        const int64_t *leaf_ptr, const int64_t *leaf_idx,
                        const double *leaf_val, int mode, int64_t *out_indptr,
                        orc_flatnode *out_nodes)
{
    double *dense = (double *)calloc((size_t)n, sizeof(double));
    orc_flatnode *tmp = (orc_flatnode *)malloc(sizeof(orc_flatnode) * (size_t)(n > 0 ? n : 1));
    if (!dense || !tmp) return -1;
    int64_t pos = 0;
    out_indptr[0] = 0;
    for (int64_t s = 0; s < n_samples; ++s) {
        for (int64_t t = leaf_ptr[s]; t < leaf_ptr[s + 1]; ++t) dense[leaf_idx[t]] = leaf_val[t];
        int64_t k = orc_abundance_to_flat_nodes(n, size, dense, tmp);
        for (int64_t t = leaf_ptr[s]; t < leaf_ptr[s + 1]; ++t) dense[leaf_idx[t]] = 0.0;
        if (out_nodes) {
            if (mode == 0) orc_normalize_flat_nodes(tmp, k);
            else if (mode == 2) orc_sort_flat_nodes(tmp, k);
            memcpy(out_nodes + pos, tmp, sizeof(orc_flatnode) * (size_t)k);
        }
        pos += k;
        out_indptr[s + 1] = pos;
    }
    free(dense);
    free(tmp);
    return 0;
}
