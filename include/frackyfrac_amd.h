/*
 * frackyfrac_amd.h -- C ABI of the MI355X-native UniFrac distance engine.
 *
 * This is the drop-in boundary for ONE path of fluhus/frackyfrac: the all-pairs
 * UniFrac reduction of the `frcfrc` tool (frcfrc/unifrac.go), plus the thin host
 * surface either side of it (Newick tree, dense/sparse abundance tables, the
 * lower-triangle text output).  The reference has no FFI; its seams are Go
 * function calls inside package main.  Each entry point below names the
 * reference function it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - Every function that can fail returns 0 on success and a non-zero ff_status
 *     otherwise, and writes a NUL-terminated message into err (if err != NULL and
 *     errlen > 0).  The CLI prints it as "ERROR: <msg>" and exits 2, as
 *     common.ExitIfError does (common/common.go:13-18).
 *   - Plain pointers and sizes only; all host buffers are caller-owned and are
 *     not retained after the call returns (cgo pointer-passing rule).
 *   - The compute path is HIP on gfx950 only.  There is no CPU fallback: without
 *     a usable GPU the compute entry points fail with FF_ERR_DEVICE.
 *   - Pair order everywhere is common.IterPairs (common/common.go:21-31): slot
 *     k = i*(i-1)/2 + j for sample indices i > j >= 0, i.e. numpy.tril_indices(N,-1).
 *   - Threads: the library keeps no state outside the objects it hands out (ff_tune's
 *     switches apart, which a lock guards).  Different plans, trees, tables and one-call
 *     entry points may be used from different host threads at the same time -- a Go
 *     host's goroutines --; ONE object is used by one thread at a time.  A call leaves
 *     the calling thread's current HIP device as it found it.
 */
#ifndef FRACKYFRAC_AMD_H
#define FRACKYFRAC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FF_VERSION_STRING "0.1.0"

typedef enum ff_status {
    FF_OK = 0,
    FF_ERR_ARG = 1,     /* bad argument / malformed problem                        */
    FF_ERR_PARSE = 2,   /* malformed tree or table text (message follows parser.go) */
    FF_ERR_SPECIES = 3, /* validateSpecies failure                                  */
    FF_ERR_DEVICE = 4,  /* no GPU, HIP error, out of device memory                  */
    FF_ERR_IO = 5,      /* file open/read/write                                     */
    FF_ERR_INTERNAL = 6,
    FF_ERR_PRECISION = 7 /* ff_plan_run_host: more nearly identical pairs than the FIXED32 plan can
                            re-compute exactly (a data set of replicates), or its run-time audit
                            failed (ff_plan_audit); stage an EXACT64 plan */
} ff_status;

/* ------------------------------------------------------------------------- *
 * 1. The hot path: frcfrc/unifrac.go:209  unifracDists(nodes, treeDists, weighted)
 * ------------------------------------------------------------------------- */

/*
 * Inputs of unifracDists: N sorted sparse vectors of flat nodes
 * (frcfrc/unifrac.go:137-140 `flatNode{id, abnd}`) in CSR form, and
 * treeDists[B] (unifrac.go:117-120: the branch length of every node in
 * enumerateNodes' pre-order numbering, root included).
 */
typedef struct ff_problem {
    int64_t n_samples;          /* N                                                   */
    int64_t n_branches;         /* B = number of tree nodes; ids are 0..B-1            */
    const double *branch_len;   /* [B]   treeDists                                     */
    const int64_t *indptr;      /* [N+1] sample s owns entries indptr[s]..indptr[s+1]  */
    const int32_t *branch_id;   /* [nnz] flatNode.id, strictly ascending within a sample */
    const double *abnd;         /* [nnz] flatNode.abnd, > 0                            */
} ff_problem;

/* How the pairwise sums are carried on the device (DESIGN.md "Arithmetic"). */
typedef enum ff_precision {
    FF_PRECISION_AUTO = 0,    /* EXACT64 when pairs*branches <= 2^32 (about a millisecond), when
                                 FIXED32 is not applicable, and for UNWEIGHTED whenever a branch
                                 length is off the binary grid FIXED32 would need (any real
                                 phylogeny): unweighted results are then the reference's bits;
                                 else FIXED32                                                  */
    FF_PRECISION_FIXED32 = 1, /* 32-bit fixed point, integer sums: order-independent, exact
                                 for unweighted whenever all branch lengths are k * 2^-e,
                                 else within 1e-6 relative (per-branch shared rounding offset,
                                 binary64 denominators, exact re-computation of near-equal
                                 pairs, run-time audit);
                                 weighted runs on the vector ALU (v_sad_u32), unweighted on
                                 the int8 matrix cores (same integers, same results)          */
    FF_PRECISION_EXACT64 = 2  /* binary64 in the reference's own summation order: bit-for-bit
                                 the reference for every input it accepts (negative, infinite
                                 and NaN branch lengths included: a tree with a non-finite
                                 length is reduced by the literal merge walk, which like the
                                 reference never touches a branch neither sample of a pair
                                 has); weighted about 4x slower
                                 than FIXED32, unweighted about 2x slower than the vector-ALU
                                 FIXED32 path (40x slower than the matrix cores)              */
} ff_precision;

typedef struct ff_options {
    int32_t weighted;     /* the `weighted bool` argument / the -w flag (frcfrc.go:22)        */
    int32_t precision;    /* ff_precision                                                     */
    int32_t device;       /* HIP device ordinal; -1 = current device                          */
    int32_t rank;         /* pair-space shard: this process computes the rows of shard `rank` */
    int32_t world;        /* of `world` equal-work shards (1 = everything); see ff_shard_rows */
    int32_t flags;        /* FF_FLAG_*                                                        */
    int32_t reserved[2];  /* must be 0                                                        */
} ff_options;

/* ff_options.flags.
 * FF_FLAG_UNSORTED_WALK: the lists of the problem are NOT ascending in branch id and the distance of a
 * pair is whatever the reference's two-pointer merge (unifrac.go:148-167, 178-203) makes of them as they
 * stand.  That is what frcfrc -l computes in the reference: with -l it skips normalizeFlatNodes and, with
 * it, the sort (unifrac.go:57-59,108-110), so the lists stay in the recursion's post-order and the merge
 * mis-pairs branches.  This engine's -l sorts; with this flag (leave_unnormalized =
 * FF_L_REFERENCE in the entry points that flatten, `frcfrc -l -l-compat`) it reproduces the reference
 * bit for bit instead: one thread per pair walks the two lists literally, in binary64. */
#define FF_FLAG_UNSORTED_WALK 1

/* Fills *o with the defaults: unweighted, AUTO, current device, rank 0 of 1. */
void ff_options_default(ff_options *o);

/* Number of output slots for n samples: n*(n-1)/2. */
int64_t ff_num_pairs(int64_t n_samples);

/*
 * The row range [*row_begin, *row_end) of sample indices i whose pairs (i, j<i)
 * shard `rank` of `world` owns.  Shards are contiguous in IterPairs order, so a
 * shard's results are the contiguous output slots
 * [row_begin*(row_begin-1)/2, row_end*(row_end-1)/2); boundaries are chosen so
 * every shard holds the same number of pairs to within one 32-row block.
 */
int ff_shard_rows(int64_t n_samples, int32_t rank, int32_t world,
                  int64_t *row_begin, int64_t *row_end);

/*
 * Replaces unifracDists (frcfrc/unifrac.go:209-228): computes the distances of
 * this shard's pairs into out (host memory).  out has ff_num_pairs(N) slots and
 * is indexed by the global slot number; slots outside the shard are untouched.
 * Blocking.  Ordered delivery (ppln.Serial, unifrac.go:212) is the slot index;
 * the reference's early-stop (yield == false, unifrac.go:222) has no
 * counterpart HERE: the whole shard is produced in one device pass.  ff_unifrac_dists_stream
 * below is the entry point that keeps it.
 */
int ff_unifrac_dists(const ff_problem *p, const ff_options *o, double *out,
                     char *err, size_t errlen);

/*
 * unifracDists as what it is in the reference: a LAZY, ORDERED sequence that stops computing when
 * its consumer stops (iter.Seq[float64], frcfrc/unifrac.go:209-228; early stop at :221-226).
 * The pair space of this shard (o->rank of o->world) is walked in sub-shards of at most
 * max_pairs_per_chunk distances (<= 0: 2^25 = 256 MB); each finished sub-shard is handed to
 * `fn` as dists[0..n) = the distances of the consecutive global slots slot_begin .. slot_begin+n-1
 * (common.IterPairs order), sub-shards in ascending slot order, every slot exactly once.  The
 * buffer belongs to the library and is valid only during the call of fn.
 *   fn returns non-zero: continue.  fn returns 0: stop -- no further sub-shard is delivered, the
 *   one in flight on the device is drained and dropped, and the call returns FF_OK.
 * fn is called on the CALLING thread, never concurrently with itself (a cgo callback may
 * therefore call the `yield` of the iterator it sits in), while the device already reduces the
 * next sub-shard.  The inputs are staged once; a sub-shard that fails FIXED32's guarantee
 * (FF_ERR_PRECISION conditions, see ff_plan_audit) is repeated in EXACT64 before it is delivered,
 * and so is everything after it.  Nothing is staged or computed if the shard is empty.
 */
typedef int (*ff_dists_fn)(void *user, int64_t slot_begin, const double *dists, int64_t n);
int ff_unifrac_dists_stream(const ff_problem *p, const ff_options *o, int64_t max_pairs_per_chunk,
                            ff_dists_fn fn, void *user, char *err, size_t errlen);

/*
 * The same entry points with the fields of ff_problem as separate arguments, for hosts whose FFI
 * forbids passing a struct that holds pointers into managed memory.  cgo: "Go code may pass a Go
 * pointer to C provided the Go memory to which it points does not contain any Go pointers" -- a
 * Go-allocated C.ff_problem filled with unsafe.SliceData(...) of Go slices breaks that rule
 * (cgocheck panics), four slice pointers as top-level arguments do not.  Nothing is retained
 * after the call returns.
 */
int ff_unifrac_dists_csr(int64_t n_samples, int64_t n_branches, const double *branch_len,
                         const int64_t *indptr, const int32_t *branch_id, const double *abnd,
                         const ff_options *o, double *out, char *err, size_t errlen);
int ff_unifrac_dists_stream_csr(int64_t n_samples, int64_t n_branches, const double *branch_len,
                                const int64_t *indptr, const int32_t *branch_id, const double *abnd,
                                const ff_options *o, int64_t max_pairs_per_chunk, ff_dists_fn fn,
                                void *user, char *err, size_t errlen);

/* -- The same path with the staged inputs resident in HBM (benchmarks, pipelines) -- */

typedef struct ff_plan ff_plan; /* staged matrix + tile schedule on one device */
typedef struct ff_tree ff_tree; /* section 2 */

/* What the staging decided; read back with ff_plan_info. */
typedef struct ff_plan_info {
    int32_t precision;        /* FF_PRECISION_FIXED32 or FF_PRECISION_EXACT64 actually used   */
    int32_t scale_log2;       /* FIXED32: values are floor(x * 2^scale_log2 + u_branch)       */
    int32_t lengths_exact;    /* FIXED32 unweighted: 1 if every branch length is an exact
                                 multiple of 2^-scale_log2 (results then bit-exact)           */
    int32_t n_compute_units;  /* CUs of the device                                            */
    int64_t n_samples, n_branches;
    int64_t ld;               /* staged matrix leading dimension (samples, padded)            */
    int64_t rows_padded;      /* staged matrix rows (branches, padded)                        */
    int64_t row_begin, row_end; /* shard                                                      */
    int64_t slot_begin, slot_end; /* output slots this plan writes                            */
    int64_t n_tiles;          /* pair tiles of the shard                                      */
    int64_t n_items;          /* (tile, branch-range) work items after balancing              */
    int64_t n_wave_slots;     /* persistent waves the main kernel runs                        */
    double staged_bytes;      /* bytes of the staged matrix in HBM                            */
    double elements;          /* sum over items of tile pairs * branches = |a-b| terms issued */
    int32_t kernel;           /* ff_kernel: which kernel does the pair reduction              */
    int32_t n_digits;         /* FF_KERNEL_MFMA_I8: base-128 digits of the staged rows' integer
                                 lengths (graded rows: of the longest row, at most 4)         */
    int64_t n_rows;           /* branches staged: n_branches, or only those some sample has a
                                 flat node on when that drops a tenth of them (compaction)    */
    int32_t n_sweeps;         /* FF_KERNEL_MFMA_I8: passes over the branch range per pair tile */
    int32_t planes_per_sweep; /* FF_KERNEL_MFMA_I8: digit planes multiplied per pass: 1 or 2
                                 base-128 digits, or 3 signed digits (graded rows, below)      */
    int64_t rows_three_planes;/* FF_KERNEL_MFMA_I8, graded rows (lengths of more than two base-128
                                 digits: rows sorted by length, longest first, a branch longer than
                                 three signed digits hold as several rows): the leading rows of
                                 rows_padded whose blocks take three planes; the rest take two  */
    /* FIXED32's run-time audit of the LAST COMPLETED RUN (ff_plan_audit, ff_plan_audit_detail).  Filled by
       the blocking entry points that hand an ff_plan_info back after running (ff_unifrac and the frcfrc
       command: their -stats); ff_plan_info_get leaves them 0 / infinity -- ask ff_plan_audit.          */
    int64_t audit_checked;     /* pairs held against their binary64 value: the uniform sample + the pairs
                                  just above the refinement rule's bound                                  */
    int64_t audit_failed;      /* of those, further than 0.5e-6 (relative) from it                        */
    double audit_worst_rel_err;
    double audit_min_headroom; /* smallest (U * 1e-6 - 2) / (5 sqrt(k)) over the run's pairs that were not
                                  re-computed exactly (the rule queues everything under 1); infinity: none */
    double active_fraction;    /* FIXED32 weighted: share of the shard's (32-sample block, staged row) cells in
                                  which some sample has the branch -- what pair_sad_sparse_kernel walks (it skips
                                  the rest) and what decides between it and the dense kernel; 1 where not counted */
    int64_t rare_rows;         /* FIXED32 weighted on a sparse table: staged rows kept OUT of the matrix (few samples
                                  reach them) and reduced by pair_low_kernel over the pairs that both have them; 0: none */
    double rare_updates;       /* pair_low_kernel's own work per pass: the sum over the rare rows of n_r (n_r - 1) / 2, n_r =
                                  the samples with a flat node on the row (an update = one min + one LDS add).
                                  A row shard's plan: the whole triangle's count times the shard's share of the pairs */
} ff_plan_info;

typedef enum ff_kernel {
    FF_KERNEL_SAD_U32 = 0,   /* pair_sad_kernel: v_sad_u32 pair tiles (FIXED32)                    */
    FF_KERNEL_EXACT_F64 = 1, /* pair_exact64_kernel (EXACT64)                                      */
    FF_KERNEL_MFMA_I8 = 2,   /* pair_common_mfma_kernel: FIXED32 unweighted, int8 matrix cores     */
    FF_KERNEL_SAD_U32_SPARSE = 3, /* pair_sad_sparse_kernel: as SAD_U32, skipping branch rows on which
                                     none of a tile's 32 samples has a flat node (sparse tables)   */
    FF_KERNEL_MFMA_I8_SMALL = 4,  /* pair_common_small_kernel: as MFMA_I8 for a shard smaller than one round
                                     of it (few hundred samples): 32 x 32 tiles, one launch per pass     */
    FF_KERNEL_EXACT_F64_UNW = 5,  /* pair_exact_unw_kernel: EXACT64 unweighted from presence bits -- the
                                     reference's two chains of binary64 additions per pair and no other
                                     arithmetic (unifrac.go:144-171)                                      */
    FF_KERNEL_WALK_F64 = 6,       /* pair_walk_kernel: the reference's merge walk itself, a thread per pair,
                                     over lists as they stand (FF_FLAG_UNSORTED_WALK)                     */
    FF_KERNEL_EXACT_F64_SKIP = 7  /* pair_exact64_skip_kernel: EXACT64 weighted; a branch a row has not adds
                                     l * y twice (unifrac.go:186-187) instead of the six operations of
                                     :191-192 on a zero -- same bits as FF_KERNEL_EXACT_F64               */
} ff_kernel;

/* Stage: quantise / densify the flat nodes into the branch-major matrix in HBM
 * (DESIGN.md "Data layout"), build the tile schedule.  Synchronous. */
int ff_plan_create(const ff_problem *p, const ff_options *o, ff_plan **plan,
                   char *err, size_t errlen);
/* ff_plan_create with the fields of ff_problem as separate arguments (see ff_unifrac_dists_csr). */
int ff_plan_create_csr(int64_t n_samples, int64_t n_branches, const double *branch_len,
                       const int64_t *indptr, const int32_t *branch_id, const double *abnd,
                       const ff_options *o, ff_plan **plan, char *err, size_t errlen);
/* The same from leaf values: stage A runs on the device and the flat nodes never leave
 * it (this is what ff_unifrac and the frcfrc command use). */
int ff_plan_create_from_leaves(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr,
                               const int64_t *leaf_idx, const double *leaf_val,
                               int leave_unnormalized, const ff_options *o, ff_plan **plan,
                               char *err, size_t errlen);
void ff_plan_destroy(ff_plan *plan);
int ff_plan_info_get(const ff_plan *plan, ff_plan_info *info);

/* Re-targets a staged plan at shard `rank` of `world`: the staged matrix stays where it is, the
 * work schedule and the accumulators are rebuilt (milliseconds).  Lets one staging serve every
 * shard of a pair space too large for one pass (the lazy iter.Seq of unifrac.go:209-228 becomes
 * a loop over shards), or a rank take over another rank's rows.  Synchronises the device. */
int ff_plan_set_shard(ff_plan *plan, int32_t rank, int32_t world, char *err, size_t errlen);

/*
 * One pass of the hot path over the staged inputs: launches the pair kernels on
 * `stream` (a hipStream_t; NULL = the null stream) and returns without
 * synchronising.  d_out is DEVICE memory with room for slot_end - slot_begin
 * doubles: d_out[k - slot_begin] receives the distance of global slot k.
 */
int ff_plan_run(ff_plan *plan, void *stream, double *d_out, char *err, size_t errlen);

/* ff_plan_run for callers without device memory of their own (a cgo host): runs the current
 * shard on the null stream, waits, and copies its slot_end - slot_begin distances to `out`
 * (HOST memory).  Returns FF_ERR_PRECISION instead of results that could miss the tolerance. */
int ff_plan_run_host(ff_plan *plan, double *out, char *err, size_t errlen);

/*
 * Like ff_plan_run but brackets the dominant kernel (the pair-tile reduction) with
 * a fresh pair of HIP events on `stream`.  ff_plan_timing_collect synchronises on
 * every pair recorded since the previous collect and returns the summed duration of
 * that kernel (ms) and the number of launches, so a benchmark can time K steps
 * without a host sync inside its timed region.
 */
int ff_plan_run_timed(ff_plan *plan, void *stream, double *d_out, char *err, size_t errlen);
int ff_plan_timing_collect(ff_plan *plan, double *total_ms, int32_t *launches);
/* The same with the share of the rare rows' kernel (pair_low_kernel, ff_plan_info.rare_rows > 0: the rest of total_ms is
 * the matrix rows' kernel; 0 where the plan is not split) -- so that a benchmark can price each kernel on its own rows. */
int ff_plan_timing_collect_parts(ff_plan *plan, double *total_ms, double *rare_ms, int32_t *launches);

/*
 * FIXED32 re-computes the few pairs whose integer sum is too small for the 1e-6
 * relative bar (nearly identical samples) with the reference's binary64 merge walk
 * on the device.  After a run has completed: how many pairs the last run queued and
 * the queue's capacity; queued > capacity means the surplus kept its fixed-point
 * value (ff_unifrac_dists then repeats the shard in EXACT64 by itself).
 */
int ff_plan_refined_pairs(ff_plan *plan, int64_t *queued, int64_t *capacity);

/*
 * FIXED32's run-time audit, in two parts.  (1) When a shard is scheduled, a fixed pseudo-random
 * sample of its pairs (4096 per 2^23 pairs of the shard, at most 65536) is computed in binary64 on
 * the device; every run compares what it delivered for them.  (2) Every run also collects the
 * pairs that stand just above the refinement rule's bound -- headroom
 * (U * 1e-6 - 2) / (5 sqrt(k)) in [1, 1.25): the ones the statistical argument protects least --
 * and computes up to 4096 of them in binary64 on the spot.  After a run has completed: the pairs
 * checked (both parts), how many of them were further than 0.5e-6 (relative) from their binary64
 * value, and the largest relative error seen.
 * failed > 0 means the run must not be trusted to the 1e-6 bar: ff_plan_run_host returns
 * FF_ERR_PRECISION, ff_unifrac_dists / ff_unifrac / the CLI repeat the shard in EXACT64.
 * checked == 0 when the plan's integers are exact (EXACT64; unweighted on the binary grid).
 */
int ff_plan_audit(ff_plan *plan, int64_t *checked, int64_t *failed, double *max_rel_err);
/* The parts: the uniform sample's size, how many pairs of the last run had a headroom under 1.25,
 * how many of those were computed in binary64 (the first 4096), and the smallest headroom over all
 * pairs of the run that were not re-computed exactly (infinity: none, or no audit). */
int ff_plan_audit_detail(ff_plan *plan, int64_t *uniform_checked, int64_t *risk_found,
                         int64_t *risk_checked, double *min_headroom);

/* -- Device buffers shared between the processes of one node (the multi-GPU gather) --
 *
 * With one process per GPU, every rank's distances end in ONE array on the root's device: the
 * root allocates it (ff_device_alloc), exports it (ff_ipc_export), every other rank maps it
 * (ff_ipc_open) and copies its finished slice of the IterPairs-ordered output straight into it
 * over its own xGMI link (ff_device_copy_async on a side stream, i.e. by the copy engines, while
 * its compute units already reduce the next batch).  Plain HIP IPC underneath
 * (hipIpcGetMemHandle / hipIpcOpenMemHandle); the 64 handle bytes travel by whatever channel
 * the host has (a Go host: a pipe or a socket; frackyfrac_amd/distributed.py: the process
 * group's object broadcast).  A mapping stays valid until ff_ipc_close; the owner must outlive
 * every mapping.  This pool's driver only supports dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0
 * must be set in the environment of every process before its first HIP call. */
typedef struct ff_ipc_handle {
    unsigned char bytes[64];
} ff_ipc_handle;
/* `bytes` of device memory on HIP device `device` (-1: the current one), zero-filled. */
int ff_device_alloc(int32_t device, size_t bytes, void **dptr, char *err, size_t errlen);
int ff_device_free(void *dptr, char *err, size_t errlen);
/* dptr must be the start of an allocation made by ff_device_alloc in THIS process. */
int ff_ipc_export(void *dptr, ff_ipc_handle *handle, char *err, size_t errlen);
/* Maps another process's allocation into this process; the pointer is usable on `device`
 * (-1: the current one) -- peer access to the owner's device is enabled on first use. */
int ff_ipc_open(const ff_ipc_handle *handle, int32_t device, void **dptr, char *err, size_t errlen);
int ff_ipc_close(void *dptr, char *err, size_t errlen);
/* Device-to-device copy (local or into a mapped peer buffer) on `stream` (hipStream_t; NULL =
 * the null stream); returns without synchronising. */
int ff_device_copy_async(void *dst, const void *src, size_t bytes, void *stream, char *err, size_t errlen);

/* ------------------------------------------------------------------------- *
 * 2. Host surface either side of the hot path
 * ------------------------------------------------------------------------- */

/* newick.Node tree as the path uses it (Name, Distance, Children), flattened in
 * the numbering of enumerateNodes (frcfrc/unifrac.go:127-133). */
/* (typedef struct ff_tree ff_tree; is declared above, next to ff_plan) */

/* Replaces readTree (frcfrc/frcfrc.go:109-114): first tree of a Newick text.
 * "no tree in the given file" when the text holds none. */
int ff_tree_parse(const char *text, size_t len, ff_tree **tree, char *err, size_t errlen);
int ff_tree_read_file(const char *path, ff_tree **tree, char *err, size_t errlen);
void ff_tree_free(ff_tree *tree);
int64_t ff_tree_num_nodes(const ff_tree *tree);
/* Borrowed views, valid until ff_tree_free: */
const double *ff_tree_branch_len(const ff_tree *tree);   /* [B] treeDists             */
const int64_t *ff_tree_parent(const ff_tree *tree);      /* [B] parent id, -1 = root  */
const int64_t *ff_tree_subtree_size(const ff_tree *tree);/* [B] nodes under id, incl. */
const char *ff_tree_name(const ff_tree *tree, int64_t id);

/* []map[string]float64 as the loaders produce it: one species->value map per sample. */
typedef struct ff_table ff_table;

/* Replace parser.ParseAbundance (parser/parser.go:21) and
 * parser.ParseSparseAbundance (parser/parser.go:85). */
int ff_table_parse_dense(const char *text, size_t len, ff_table **table, char *err, size_t errlen);
int ff_table_parse_sparse(const char *text, size_t len, ff_table **table, char *err, size_t errlen);
int ff_table_read_file(const char *path /* NULL = stdin */, int sparse, ff_table **table,
                       char *err, size_t errlen);
/* The same with the rows parsed on `threads` host threads (the ngoroutines argument of
 * the reference's loaders, parser.go:21,85); the first failing row in row order is the
 * one reported, as with the reference's ordered pipeline. */
int ff_table_parse_mt(const char *text, size_t len, int sparse, int threads, ff_table **table,
                      char *err, size_t errlen);
int ff_table_read_file_mt(const char *path /* NULL = stdin */, int sparse, int threads,
                          ff_table **table, char *err, size_t errlen);
void ff_table_free(ff_table *table);
int64_t ff_table_num_samples(const ff_table *table);
int64_t ff_table_sample_size(const ff_table *table, int64_t sample);
/* k-th entry of a sample in insertion order (a duplicated key keeps its first
 * position and its last value, as a Go map assignment would). */
int ff_table_sample_entry(const ff_table *table, int64_t sample, int64_t k,
                          const char **name, double *value);

/* The table in the reference's sparse format: one line per sample, its non-zero entries as
 * name:value (Go's %g), tab-separated -- what the `sprspr` tool writes (sprspr/sprspr.go:19-44;
 * entries in the order of the table's header: the reference's order is a Go map's, i.e. random).
 * path NULL = stdout. */
int ff_table_write_sparse(const ff_table *table, const char *path, char *err, size_t errlen);
/* Whole `sprspr` command (sprspr/sprspr.go:14-17): dense table on stdin -> sparse table on stdout. */
int ff_sprspr_main(int argc, char **argv);

/* Replaces validateSpecies (frcfrc/unifrac.go:80-93). */
int ff_validate_species(const ff_table *table, const ff_tree *tree, char *err, size_t errlen);

/* Stage A (frcfrc/unifrac.go:32-67,99-116) on the host: abundanceToFlatNodes +
 * normalizeFlatNodes for every sample, in the reference's order of float
 * additions.  leave_unnormalized = the -l flag (frcfrc.go:25): 1 = lists are sorted
 * but not divided (the evidently intended semantics); FF_L_REFERENCE = what the
 * reference really does under -l: neither divided NOR SORTED (it skips the sort with
 * the division, unifrac.go:57-59,108-110), i.e. every list in the
 * order the recursion appends it, a node after its subtree.  Lists in that order
 * go with FF_FLAG_UNSORTED_WALK; ff_unifrac and ff_plan_create_from_leaves set it
 * themselves when given FF_L_REFERENCE. */
#define FF_L_REFERENCE 2
typedef struct ff_flat ff_flat; /* owns the CSR arrays an ff_problem points into */
int ff_flatten(const ff_table *table, const ff_tree *tree, int leave_unnormalized,
               ff_flat **flat, char *err, size_t errlen);
/* The same from leaf values given by node id (no names): sample s has values
 * leaf_val[leaf_ptr[s]..leaf_ptr[s+1]) at leaf node ids leaf_idx[...]. */
int ff_flatten_leaf_csr(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr,
                        const int64_t *leaf_idx, const double *leaf_val,
                        int leave_unnormalized, ff_flat **flat, char *err, size_t errlen);
/* Stage A on the DEVICE: the same flat nodes, bit for bit (a node
 * adds its children in ascending order, as the reference's recursion does), computed
 * level by level over a dense branch-major matrix in HBM.  Trees deeper than 4096
 * levels are flattened on the host instead. */
int ff_flatten_device(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr,
                      const int64_t *leaf_idx, const double *leaf_val,
                      int leave_unnormalized, ff_flat **flat, char *err, size_t errlen);
void ff_flat_free(ff_flat *flat);
void ff_flat_problem(const ff_flat *flat, ff_problem *p); /* borrowed views */

/* Replaces unifrac (frcfrc/unifrac.go:97-124): stage A on the device + distances into
 * out (host, ff_num_pairs(N) slots).  o->weighted selects the metric. */
int ff_unifrac(const ff_table *table, const ff_tree *tree, const ff_options *o,
               int leave_unnormalized, double *out, char *err, size_t errlen);

/* fmt.Fprintln(w, f) for a float64 (frcfrc/frcfrc.go:59): Go's %v -- shortest
 * round-trip digits, %e form when the decimal exponent is < -4 or >= 6
 * (strconv's shortest-%g rule), "NaN", "+Inf".  Writes the
 * text WITHOUT the newline, returns its length (buf must hold 32 bytes). */
int ff_format_float(double f, char *buf);
/* Writes `n` distances, one per line, to path (NULL = stdout), formatting row
 * blocks in parallel on `threads` host threads. */
int ff_write_distances(const char *path, const double *d, int64_t n, int threads,
                       char *err, size_t errlen);

/* CPUs this process may use: the smallest of the machine's count, the scheduler affinity mask and the cgroup CPU
 * quota (rounded up).  What the frcfrc command gives its loaders and its writer when -p is not given. */
int ff_cpu_quota(void);

/* The same formatter ON THE DEVICE (the loop of frcfrc/frcfrc.go:58-62 for distances that are still in HBM): the
 * n values at d_values (device memory) become the lines the reference prints, in order, at d_text (device memory,
 * at least ff_text_bound(n) bytes = 25 per value); *n_bytes = the length of the text.  Same digits and layout as
 * ff_format_float, byte for byte (one implementation, csrc/ff_fmt_core.hpp).  The launches go to `stream` (NULL: the
 * legacy stream); the call returns once the text is complete.  The frcfrc command uses this path: a pass's distances
 * never reach the host as numbers. */
size_t ff_text_bound(int64_t n);
int ff_format_distances_device(const double *d_values, int64_t n, char *d_text, size_t *n_bytes, void *stream,
                               char *err, size_t errlen);

/* unifracDists AND the loop that prints it (frcfrc/unifrac.go:209-228 + frcfrc/frcfrc.go:58-62) as one lazy ordered
 * sequence of TEXT: the lines `for f := range dists { fmt.Fprintln(w, f) }` would write, formatted on the device,
 * handed to `fn` in pieces of whole lines (<= 32 MB each), in order, on the calling thread; fn returns 0 to stop --
 * nothing further is computed.  What a host whose own formatter is the bottleneck calls instead of
 * ff_unifrac_dists_stream (the reference's Fprintln loop is 5 s of a C4-sized run on one goroutine; this is 0.3):
 * the callback's body is `w.Write(text[:n])`.  Same sub-shards, same FIXED32 -> EXACT64 repeat, same laziness as
 * ff_unifrac_dists_stream; max_pairs_per_chunk bounds device memory (33 bytes per pair of a sub-shard), not the
 * pieces.  The _csr form takes the problem's fields as separate arguments (cgo's pointer rule). */
typedef int (*ff_text_fn)(void *user, const char *text, size_t n_bytes);
int ff_unifrac_text_stream(const ff_problem *problem, const ff_options *options, int64_t max_pairs_per_chunk,
                           ff_text_fn fn, void *user, char *err, size_t errlen);
int ff_unifrac_text_stream_csr(int64_t n_samples, int64_t n_branches, const double *branch_len, const int64_t *indptr,
                               const int32_t *branch_id, const double *abnd, const ff_options *options,
                               int64_t max_pairs_per_chunk, ff_text_fn fn, void *user, char *err, size_t errlen);

/* Whole `frcfrc` command (frcfrc/frcfrc.go:29-67): argv as the reference's flags
 * -i -o -t -w -s -p -l.  Returns the process exit code (0, or 2 after printing
 * "ERROR: ..." to stderr). */
int ff_frcfrc_main(int argc, char **argv);

/* The synthetic abundance tables of the benchmark (recipe in frackyfrac_amd/synth.py
 * and csrc/ff_synth.cpp: splitmix64 -> xoshiro256**, one stream per sample, so any range of samples
 * can be generated on its own).  Samples [sample_begin, sample_end) of a table over n_leaves leaves:
 * first the number of leaves each holds, then -- with ptr = their exclusive prefix sums -- the leaf
 * ordinals (0 .. n_leaves-1, ascending within a sample) and integer counts.  Host code, no GPU. */
int ff_synth_counts(int64_t n_leaves, double density, uint64_t seed, int64_t sample_begin, int64_t sample_end,
                    int threads, int64_t *counts);
int ff_synth_fill(int64_t n_leaves, double density, uint64_t seed, int64_t sample_begin, int64_t sample_end,
                  int threads, const int64_t *ptr, int64_t *leaf_ordinal, double *value);

/* The tuning switches INTEGRATION.md lists as FF_* environment variables (FF_WAVES_PER_WG,
 * FF_XCD_SLICES, FF_SPARSE_MIN, FF_AUDIT, ...), for a host that cannot or should not change its
 * environment: a value set here wins over the environment variable of the same name; value NULL
 * removes the override.  Process-wide; switches are read when a plan is created or re-targeted
 * (never per launch), so set them before ff_plan_create.  Defaults are the measured best. */
int ff_tune(const char *name, const char *value);

const char *ff_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FRACKYFRAC_AMD_H */
