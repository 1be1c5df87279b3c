// ff_dev_stage.hip -- inputs -> HBM (one of the three translation units of the device path: ff_plan.hpp).
//
// Stage A on the device (abundanceToFlatNodes + normalizeFlatNodes, frcfrc/unifrac.go:32-67,99-116) or the flat nodes
// from the host; the choice of arithmetic (FIXED32 / EXACT64: DESIGN.md "Arithmetic"); branch compaction; and the staged
// operands of every pair kernel -- the branch-major u32 matrix of the v_sad_u32 kernels, presence words and digit
// planes of the matrix-core kernels, the binary64 matrix and the presence bits of the EXACT64 kernels.
#include "ff_plan.hpp"

namespace {

using namespace ff::sched;

#include "ff_kernels_stage.hpp"
#include "ff_kernels_stage_a.hpp"

}  // namespace

namespace ff {
namespace dev {

using namespace ff::sched;

// Host flat nodes -> device (the inner-seam entry: ff_plan_create / ff_unifrac_dists).
int csr_from_host(const ff_problem *p, DeviceCsr *c, char *err, size_t errlen)
{
    const int64_t N = p->n_samples, B = p->n_branches;
    c->N = N;
    c->B = B;
    c->nnz = N > 0 ? p->indptr[N] : 0;
    c->h_indptr.assign((size_t)N + 1, 0);
    if (N > 0) memcpy(c->h_indptr.data(), p->indptr, sizeof(int64_t) * (size_t)(N + 1));
    c->h_len.assign(p->branch_len, p->branch_len + B);
    c->h_weight.assign((size_t)N, 0.0);
    ff::parallel_for(N, host_threads(c->nnz), [&](unsigned, int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; ++s) {
            double w = 0;
            for (int64_t t = p->indptr[s]; t < p->indptr[s + 1]; ++t) w += p->branch_len[p->branch_id[t]] * p->abnd[t];
            c->h_weight[(size_t)s] = w;
        }
    });
    FF_HIP(hipMalloc(&c->d_indptr, sizeof(int64_t) * (size_t)(N + 1)));
    FF_HIP(hipMalloc(&c->d_ids, sizeof(int32_t) * (size_t)std::max<int64_t>(c->nnz, 1)));
    FF_HIP(hipMalloc(&c->d_abnd, sizeof(double) * (size_t)std::max<int64_t>(c->nnz, 1)));
    FF_HIP(hipMalloc(&c->d_len, sizeof(double) * (size_t)std::max<int64_t>(B, 1)));
    FF_HIP(hipMemcpy(c->d_indptr, c->h_indptr.data(), sizeof(int64_t) * (size_t)(N + 1), hipMemcpyHostToDevice));
    if (c->nnz > 0) {
        FF_HIP(hipMemcpy(c->d_ids, p->branch_id, sizeof(int32_t) * (size_t)c->nnz, hipMemcpyHostToDevice));
        FF_HIP(hipMemcpy(c->d_abnd, p->abnd, sizeof(double) * (size_t)c->nnz, hipMemcpyHostToDevice));
    }
    if (B > 0) FF_HIP(hipMemcpy(c->d_len, p->branch_len, sizeof(double) * (size_t)B, hipMemcpyHostToDevice));
    return FF_OK;
}

// Stage A on the device: leaf values -> flat nodes.  Returns
// FF_ERR_INTERNAL + *too_deep when the tree has more levels than it is worth launching
// kernels for (a caterpillar); the caller then flattens on the host.
int csr_from_leaves(const ff_tree *t, int64_t N, const int64_t *leaf_ptr, const int64_t *leaf_idx,
                    const double *leaf_val, bool normalize, DeviceCsr *c, bool *too_deep, char *err,
                    size_t errlen)
{
    *too_deep = false;
    const int64_t B = (int64_t)t->size.size();
    c->N = N;
    c->B = B;
    c->h_len = t->dist;
    const int64_t n_leafvals = leaf_ptr[N];
    for (int64_t k = 0; k < n_leafvals; ++k)
        if (leaf_idx[k] < 0 || leaf_idx[k] >= B)
            return ff::fail(FF_ERR_ARG, err, errlen, "leaf index %lld out of range", (long long)leaf_idx[k]);
    // The tree the sums run over: the whole tree, or -- when the samples touch less than 90 % of
    // it -- the tree induced by the leaves that carry abundance and their ancestors (W nodes,
    // pre-order kept; node_of[w] = original id).  An absent child adds +0.0 to its parent's sum,
    // which changes no bit, so the flat nodes are the same; the dense S matrix is W x N
    // instead of B x N.
    std::vector<int32_t> node_of;   // empty: identity
    std::vector<int64_t> w_parent, w_size, w_lidx;
    const int64_t *parentP = t->parent.data(), *sizeP = t->size.data(), *lidxP = leaf_idx;
    int64_t W = B;
    if (env_int("FF_COMPACT", 1) != 0 && B > 1 && n_leafvals > 0) {
        std::vector<unsigned char> used((size_t)B, 0);
        for (int64_t k = 0; k < n_leafvals; ++k)
            if (t->size[(size_t)leaf_idx[k]] == 1 && leaf_val[k] > 0) used[(size_t)leaf_idx[k]] = 1;
        int64_t cnt = 0;
        for (int64_t id = B - 1; id >= 1; --id)  // parent[id] < id
            if (used[(size_t)id]) {
                used[(size_t)t->parent[(size_t)id]] = 1;
                ++cnt;
            }
        cnt += used[0];
        if (cnt > 1 && cnt * 10 <= B * 9) {
            W = cnt;
            std::vector<int32_t> row_of((size_t)B, 0);
            node_of.reserve((size_t)W);
            for (int64_t id = 0; id < B; ++id)
                if (used[(size_t)id]) {
                    row_of[(size_t)id] = (int32_t)node_of.size();
                    node_of.push_back((int32_t)id);
                }
            w_parent.assign((size_t)W, -1);
            w_size.assign((size_t)W, 1);
            for (int64_t w = 1; w < W; ++w) w_parent[(size_t)w] = row_of[(size_t)t->parent[(size_t)node_of[(size_t)w]]];
            for (int64_t w = W - 1; w >= 1; --w) w_size[(size_t)w_parent[(size_t)w]] += w_size[(size_t)w];
            // an entry that is not a leaf with abundance goes to the root, which is internal here
            w_lidx.resize((size_t)n_leafvals);
            for (int64_t k = 0; k < n_leafvals; ++k)
                w_lidx[(size_t)k] = used[(size_t)leaf_idx[k]] && t->size[(size_t)leaf_idx[k]] == 1
                                        ? row_of[(size_t)leaf_idx[k]] : 0;
            parentP = w_parent.data();
            sizeP = w_size.data();
            lidxP = w_lidx.data();
        }
    }
    // levels: depth of every node; internal nodes grouped by level, deepest first
    std::vector<int32_t> depth((size_t)W, 0);
    int32_t max_depth = 0;
    for (int64_t id = 1; id < W; ++id) {
        depth[(size_t)id] = depth[(size_t)parentP[(size_t)id]] + 1;
        max_depth = std::max(max_depth, depth[(size_t)id]);
    }
    if (max_depth > 4096) {
        *too_deep = true;
        return FF_ERR_INTERNAL;
    }
    std::vector<int64_t> child_ptr((size_t)W + 1, 0);
    std::vector<int32_t> child_idx;
    child_idx.reserve((size_t)W);
    std::vector<std::vector<int32_t>> by_level((size_t)max_depth + 1);
    for (int64_t id = 0; id < W; ++id) {
        const int64_t end = id + sizeP[(size_t)id];
        for (int64_t ch = id + 1; ch < end; ch += sizeP[(size_t)ch]) child_idx.push_back((int32_t)ch);  // ascending
        child_ptr[(size_t)id + 1] = (int64_t)child_idx.size();
        if (sizeP[(size_t)id] > 1) by_level[(size_t)depth[(size_t)id]].push_back((int32_t)id);
    }
    std::vector<int32_t> order;
    std::vector<int> level_ptr{0};
    for (int32_t L = max_depth; L >= 0; --L) {
        order.insert(order.end(), by_level[(size_t)L].begin(), by_level[(size_t)L].end());
        level_ptr.push_back((int)order.size());
    }
    const int64_t ld = round_up(std::max<int64_t>(N, 1), 64);
    double *d_S = nullptr, *d_lval = nullptr, *d_div = nullptr, *d_weight = nullptr;
    int64_t *d_lptr = nullptr, *d_lidx = nullptr, *d_size = nullptr, *d_cptr = nullptr, *d_count = nullptr;
    int32_t *d_cidx = nullptr, *d_order = nullptr, *d_node_of = nullptr;
    auto cleanup = [&] {
        (void)hipFree(d_node_of);
        (void)hipFree(d_S); (void)hipFree(d_lval); (void)hipFree(d_div); (void)hipFree(d_weight);
        (void)hipFree(d_lptr); (void)hipFree(d_lidx); (void)hipFree(d_size); (void)hipFree(d_cptr);
        (void)hipFree(d_count); (void)hipFree(d_cidx); (void)hipFree(d_order);
    };
#define FF_HIP_C(call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (void)hipGetLastError();                                                            \
            cleanup();                                                                          \
            return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: %s failed: %s", #call,             \
                            hipGetErrorString(e_));                                             \
        }                                                                                       \
    } while (0)
    const size_t s_bytes = sizeof(double) * (size_t)std::max<int64_t>(W, 1) * (size_t)ld;
    FF_HIP_C(hipMalloc(&d_S, s_bytes));
    FF_HIP_C(hipMemset(d_S, 0, s_bytes));
    FF_HIP_C(hipMalloc(&d_lptr, sizeof(int64_t) * (size_t)(N + 1)));
    FF_HIP_C(hipMalloc(&d_lidx, sizeof(int64_t) * (size_t)std::max<int64_t>(n_leafvals, 1)));
    FF_HIP_C(hipMalloc(&d_lval, sizeof(double) * (size_t)std::max<int64_t>(n_leafvals, 1)));
    FF_HIP_C(hipMalloc(&d_size, sizeof(int64_t) * (size_t)std::max<int64_t>(W, 1)));
    FF_HIP_C(hipMalloc(&d_cptr, sizeof(int64_t) * (size_t)(W + 1)));
    if (!node_of.empty()) {
        FF_HIP_C(hipMalloc(&d_node_of, sizeof(int32_t) * (size_t)W));
        FF_HIP_C(hipMemcpy(d_node_of, node_of.data(), sizeof(int32_t) * (size_t)W, hipMemcpyHostToDevice));
    }
    FF_HIP_C(hipMalloc(&d_cidx, sizeof(int32_t) * std::max<size_t>(child_idx.size(), 1)));
    FF_HIP_C(hipMalloc(&d_order, sizeof(int32_t) * std::max<size_t>(order.size(), 1)));
    FF_HIP_C(hipMalloc(&d_count, sizeof(int64_t) * (size_t)std::max<int64_t>(N, 1)));
    FF_HIP_C(hipMalloc(&d_div, sizeof(double) * (size_t)std::max<int64_t>(N, 1)));
    FF_HIP_C(hipMalloc(&d_weight, sizeof(double) * (size_t)std::max<int64_t>(N, 1)));
    FF_HIP_C(hipMalloc(&c->d_len, sizeof(double) * (size_t)std::max<int64_t>(B, 1)));
    FF_HIP_C(hipMemcpy(d_lptr, leaf_ptr, sizeof(int64_t) * (size_t)(N + 1), hipMemcpyHostToDevice));
    if (n_leafvals > 0) {
        FF_HIP_C(hipMemcpy(d_lidx, lidxP, sizeof(int64_t) * (size_t)n_leafvals, hipMemcpyHostToDevice));
        FF_HIP_C(hipMemcpy(d_lval, leaf_val, sizeof(double) * (size_t)n_leafvals, hipMemcpyHostToDevice));
    }
    if (B > 0) {
        FF_HIP_C(hipMemcpy(d_size, sizeP, sizeof(int64_t) * (size_t)W, hipMemcpyHostToDevice));
        FF_HIP_C(hipMemcpy(c->d_len, t->dist.data(), sizeof(double) * (size_t)B, hipMemcpyHostToDevice));
    }
    FF_HIP_C(hipMemcpy(d_cptr, child_ptr.data(), sizeof(int64_t) * (size_t)(W + 1), hipMemcpyHostToDevice));
    if (!child_idx.empty())
        FF_HIP_C(hipMemcpy(d_cidx, child_idx.data(), sizeof(int32_t) * child_idx.size(), hipMemcpyHostToDevice));
    if (!order.empty())
        FF_HIP_C(hipMemcpy(d_order, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice));
    if (N > 0 && B > 0) {
        if (n_leafvals > 0)
            stage_a_scatter_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_lptr, d_lidx, d_lval, d_size, d_S, ld);
        const unsigned sblocks = (unsigned)((N + 255) / 256);
        for (size_t L = 0; L + 1 < level_ptr.size(); ++L) {
            int b0 = level_ptr[L], b1 = level_ptr[L + 1];
            while (b0 < b1) {  // grid.y is limited to 65535
                const int chunk = std::min(b1 - b0, 65535);
                stage_a_level_kernel<<<dim3(sblocks, (unsigned)chunk), dim3(256)>>>(d_order, b0, b0 + chunk, d_cptr,
                                                                                    d_cidx, d_S, ld, N);
                b0 += chunk;
            }
        }
        stage_a_count_kernel<<<dim3((unsigned)((N + 63) / 64)), dim3(64)>>>(d_S, ld, W, N, d_count, d_div);
    }
    FF_HIP_C(hipGetLastError());
    std::vector<int64_t> cnt((size_t)N, 0);
    if (N > 0 && B > 0) FF_HIP_C(hipMemcpy(cnt.data(), d_count, sizeof(int64_t) * (size_t)N, hipMemcpyDeviceToHost));
    c->h_indptr.assign((size_t)N + 1, 0);
    for (int64_t s2 = 0; s2 < N; ++s2) c->h_indptr[(size_t)s2 + 1] = c->h_indptr[(size_t)s2] + cnt[(size_t)s2];
    c->nnz = c->h_indptr[(size_t)N];
    FF_HIP_C(hipMalloc(&c->d_indptr, sizeof(int64_t) * (size_t)(N + 1)));
    FF_HIP_C(hipMalloc(&c->d_ids, sizeof(int32_t) * (size_t)std::max<int64_t>(c->nnz, 1)));
    FF_HIP_C(hipMalloc(&c->d_abnd, sizeof(double) * (size_t)std::max<int64_t>(c->nnz, 1)));
    FF_HIP_C(hipMemcpy(c->d_indptr, c->h_indptr.data(), sizeof(int64_t) * (size_t)(N + 1), hipMemcpyHostToDevice));
    c->h_weight.assign((size_t)N, 0.0);
    if (N > 0 && B > 0) {
        stage_a_fill_kernel<<<dim3((unsigned)((N + 63) / 64)), dim3(64)>>>(d_S, ld, W, N, c->d_indptr, d_div,
                                                                           normalize ? 1 : 0, c->d_len, d_node_of,
                                                                           c->d_ids, c->d_abnd, d_weight);
        FF_HIP_C(hipGetLastError());
        FF_HIP_C(hipMemcpy(c->h_weight.data(), d_weight, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost));
    }
#undef FF_HIP_C
    cleanup();
    return FF_OK;
}

namespace {

// (d_indptr, d_ids, d_len: the flat nodes and treeDists on the device -- the plan owns them by now)
Quant choose_quant(const DeviceCsr &c, bool weighted, const int64_t *d_indptr, const int32_t *d_ids, const double *d_abnd,
                   const double *d_len)
{
    Quant q;
    const int64_t B = c.B, N = c.N;
    for (int64_t b = 0; b < B; ++b)
        if (!std::isfinite(c.h_len[(size_t)b]) || c.h_len[(size_t)b] < 0) {
            q.why_not = "negative or non-finite branch length";
            return q;
        }
    const double LIMIT = 2147483647.0;  // every W_s must stay below 2^31 so that U < 2^32
    if (weighted) {
        double wmax = 0, wmin_pos = INFINITY;
        int64_t nnz_max = 0;
        for (int64_t s = 0; s < N; ++s) {
            const double w = c.h_weight[(size_t)s];
            if (!std::isfinite(w)) {
                q.why_not = "non-finite sample weight";
                return q;
            }
            wmax = std::max(wmax, w);
            if (w > 0) wmin_pos = std::min(wmin_pos, w);
            nnz_max = std::max(nnz_max, c.h_indptr[(size_t)s + 1] - c.h_indptr[(size_t)s]);
        }
        if (wmax == 0) {  // every distance is 0/0
            q.fixed_ok = true;
            q.e = 0;
            return q;
        }
        // a sample far lighter than the heaviest one would keep too few bits
        if (wmin_pos < wmax * 0x1p-10) {
            q.why_not = "sample weights span more than 2^10";
            return q;
        }
        int ex;
        std::frexp((LIMIT - (double)nnz_max - 2.0) / wmax, &ex);  // 2^(ex-1) <= ratio < 2^ex
        q.e = ex - 1;
        q.fixed_ok = true;
        return q;
    }
    // unweighted: smallest e making every length an integer, if every sample's sum still fits
    double lmax = 0;
    int e_exact = -2000;
    for (int64_t b = 0; b < B; ++b) {
        const double l = c.h_len[(size_t)b];
        lmax = std::max(lmax, l);
        if (l == 0) continue;
        int ex;
        const double m = std::frexp(l, &ex);  // l = m * 2^ex, 0.5 <= m < 1
        const uint64_t mi = (uint64_t)std::ldexp(m, 53);
        const int tz = __builtin_ctzll(mi);
        const int lowbit = ex - 53 + tz;  // l is a multiple of 2^lowbit
        e_exact = std::max(e_exact, -lowbit);
    }
    q.klen.assign((size_t)B, 0);
    // What has to stay below 2^31 is a SAMPLE's sum of integer lengths (U = W_i + W_j - 2 common), not the
    // tree's: a sample reaches a fraction of the tree, and the bits this leaves go to the resolution.
    // (Scaled by the tree's total, C3's shape with inexact lengths kept so few bits per pair that most
    // pairs failed the refinement rule and went to the binary64 walk: 55 ms a pass instead of 0.5.)
    double wl = 0;  // max over samples of sum_b l_b over the sample's flat nodes
    int64_t nnz_max = 0;
    if (N > 0 && c.nnz > 0) {
        double *d_w = nullptr;
        std::vector<double> hw((size_t)N);
        bool ok = hipMalloc(&d_w, sizeof(double) * (size_t)N) == hipSuccess;
        if (ok) {
            exact_weight_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, d_abnd, d_len, 0, d_w);
            ok = hipGetLastError() == hipSuccess &&
                 hipMemcpy(hw.data(), d_w, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost) == hipSuccess;
        }
        (void)hipFree(d_w);
        if (!ok) {
            q.why_not = "device error while summing the samples' branch lengths";
            return q;
        }
        for (int64_t s = 0; s < N; ++s) {
            wl = std::max(wl, hw[(size_t)s]);
            nnz_max = std::max(nnz_max, c.h_indptr[(size_t)s + 1] - c.h_indptr[(size_t)s]);
        }
    }
    if (wl == 0) {  // no sample has a branch of positive length: every distance is 0/0
        q.fixed_ok = true;
        q.e = 0;
        q.lengths_exact = 1;
        return q;
    }
    if (e_exact > -2000 && e_exact < 1000 && std::ldexp(std::max(wl, lmax), e_exact) <= LIMIT - 2.0) {
        q.e = e_exact;
        q.lengths_exact = 1;
    } else {
        // every sample's sum (each length rounded up by less than 1) below 2^31.  (Until round 3 every length was
        // also kept below 2^28 -- four base-128 digits, two sweeps of the matrix-core kernel -- which cost a tree
        // with a few very long branches most of its resolution: log-normal lengths of sigma 2.5 sent four pairs in
        // five to the binary64 walk.  Graded staging, stage_for_mfma, multiplies a long branch as several rows.)
        int ex;
        std::frexp((LIMIT - (double)nnz_max - 2.0) / wl, &ex);
        q.e = ex - 1;
        q.lengths_exact = 0;
    }
    // the branch's shared rounding offset (ff_dither.hpp); an exact length is its own integer.  (A branch no
    // sample has a flat node on may be longer than any sample's sum: its integer is never used, only kept in range.)
    for (int64_t b = 0; b < B; ++b)
        q.klen[(size_t)b] = (uint32_t)(int64_t)std::min(LIMIT, std::floor(std::ldexp(c.h_len[(size_t)b], q.e) + ff::branch_dither(b)));
    q.fixed_ok = true;
    return q;
}

// What the staging steps of a plan share.
struct StageCtx {
    const ff_options *o;
    DeviceCsr *c;
    const hipDeviceProp_t *prop;
    ff_plan *pl;
    int64_t R = 0;                       // staged rows: B, or the branches in use (compaction)
    Scratch<int32_t> row_of;             // branch id -> staged row (null: identity)
    std::vector<int32_t> branch_of_row;  // staged row -> branch id (empty: identity)
    std::vector<unsigned char> branch_used;  // [B] 1: some sample has a flat node on the branch (empty: not known)
    Quant q;
};

// The names the staging code is written in.
#define FF_STAGE_NAMES                                                                      \
    const ff_options *o = x.o;                                                              \
    DeviceCsr *c = x.c;                                                                     \
    const hipDeviceProp_t &prop = *x.prop;                                                  \
    ff_plan *pl = x.pl;                                                                     \
    const int64_t N = c->N, B = c->B, nnz = c->nnz, R = x.R;                                \
    const bool weighted = pl->weighted != 0;                                                \
    ff_plan_info &inf = pl->info;                                                           \
    const int64_t n_slots = inf.slot_end - inf.slot_begin;                                  \
    int64_t *d_indptr = pl->d_indptr;                                                       \
    int32_t *d_ids = pl->d_ids;                                                             \
    double *d_abnd = pl->d_abnd, *d_len = pl->d_len;                                        \
    Scratch<int32_t> &row_of = x.row_of;                                                    \
    std::vector<int32_t> &branch_of_row = x.branch_of_row;                                  \
    Quant &q = x.q;                                                                         \
    (void)o; (void)prop; (void)N; (void)B; (void)nnz; (void)R; (void)weighted; (void)n_slots; \
    (void)d_indptr; (void)d_ids; (void)d_abnd; (void)d_len; (void)row_of; (void)branch_of_row; (void)q


int compact_branches(StageCtx &x, char *err, size_t errlen)
{
    x.R = x.c->B;
    FF_STAGE_NAMES;
    // Branch compaction.  A branch no sample has a flat node on is a zero row of the staged
    // matrix and adds |0 - 0| (or +0.0) to every pair: with a reference phylogeny much larger
    // than what the samples cover, most rows are like that.  Rows are renumbered over the
    // branches in use (ascending, so EXACT64 keeps the reference's order) when that drops
    // at least a tenth of them.  R = staged rows.
    std::vector<int32_t> h_row_of;
    if (env_int("FF_COMPACT", 1) != 0 && B > 0 && nnz > 0) {
        Scratch<unsigned char> mark;
        FF_HIP(mark.alloc((size_t)B));
        FF_HIP(hipMemset(mark.p, 0, (size_t)B));
        mark_branches_kernel<<<dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 1 << 20)), dim3(256)>>>(d_ids, nnz, mark.p);
        FF_HIP(hipGetLastError());
        std::vector<unsigned char> hm((size_t)B);
        FF_HIP(hipMemcpy(hm.data(), mark.p, (size_t)B, hipMemcpyDeviceToHost));
        int64_t used = 0;
        for (unsigned char m : hm) used += m;
        x.branch_used = hm;
        if (used * 10 <= B * 9) {
            h_row_of.assign((size_t)B, 0);
            branch_of_row.reserve((size_t)used);
            for (int64_t b = 0; b < B; ++b)
                if (hm[(size_t)b]) {
                    h_row_of[(size_t)b] = (int32_t)branch_of_row.size();
                    branch_of_row.push_back((int32_t)b);
                }
            x.R = used;
            FF_HIP(row_of.alloc((size_t)B));
            FF_HIP(hipMemcpy(row_of.p, h_row_of.data(), sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice));
        }
    }
    return FF_OK;
}

// FIXED32 unweighted on the matrix cores: presence / digit planes, sample-major, and the MFMA schedule.
int stage_for_mfma(StageCtx &x, char *err, size_t errlen)
{
    FF_STAGE_NAMES;
    // presence bits (64-bit words, pairs of slabs major) and per-row digits, zero padded to whole tiles and quads of slabs
    pl->mfma = true;
    inf.kernel = FF_KERNEL_MFMA_I8;
    inf.lengths_exact = q.lengths_exact;
    inf.scale_log2 = q.e;
    // A branch no sample has a flat node on multiplies presence bits that are all zero: its integer length is never
    // used (choose_quant only keeps it in range, up to 2^31), so it must not decide the digits of the sweep or be cut
    // into hundreds of all-zero rows when fewer than a tenth of the branches are like that and the rows stay as they are.
    if (!x.branch_used.empty())
        for (int64_t b = 0; b < B; ++b)
            if (!x.branch_used[(size_t)b]) q.klen[(size_t)b] = 0;
    uint32_t kmax = 0;
    for (uint32_t k : q.klen) kmax = std::max(kmax, k);
    auto digits_of = [](uint32_t k) {
        int d = 1;
        while (d < 5 && (k >> (7 * d)) != 0) ++d;
        return d;
    };
    int digits = digits_of(kmax);
    // The staged rows: (branch, integer length of the row).  Up to two base-128 digits -- short binary fractions,
    // C3's generator -- a row is a branch in use, in ascending order, and a sweep multiplies both digit planes.
    // Longer lengths (any real phylogeny: the integers then take the 31-bit budget of a sample's sum) are staged
    // GRADED: three signed digits d0 + 128 d1 + 32768 d2 cover a length up to TRI_KMAX in one sweep of three
    // MFMAs per block (pair_common_mfma_kernel<.., GRADED>), two of them one up to DUO_KMAX; common(i, j) is
    // linear in the lengths, so a longer branch becomes several rows with the same presence bits whose lengths
    // add up to its own, and the order of the rows is free, so they are sorted by length, longest first: the
    // sweep multiplies three planes up to the first slab without a third digit and two from there on.  With
    // lengths spread over orders of magnitude most rows are of the second kind.  FF_MFMA_GRADED=0: base-128
    // digits in branch order, two planes per sweep, as many sweeps as it takes.
    struct StagedRow {
        int32_t branch;
        uint32_t k;
    };
    std::vector<StagedRow> rows;
    bool graded = false;
    if (digits > 2 && env_int("FF_MFMA_GRADED", 1) != 0) {
        int64_t pieces = 0;
        for (int64_t r = 0; r < R; ++r) {
            const int64_t k = q.klen[(size_t)(branch_of_row.empty() ? r : branch_of_row[(size_t)r])];
            pieces += std::max<int64_t>(1, (k + TRI_KMAX - 1) / TRI_KMAX);
        }
        // (a few long branches, not a tree of them: three planes over x times the rows against two sweeps of two
        // planes, or three sweeps from five base-128 digits)
        if (pieces <= (digits < 5 ? R + R / 4 : R + R * 4 / 5) + 1024 && pieces < ((int64_t)1 << 30)) {
            graded = true;
            rows.reserve((size_t)pieces);
            for (int64_t r = 0; r < R; ++r) {
                const int32_t b = (int32_t)(branch_of_row.empty() ? r : branch_of_row[(size_t)r]);
                const int64_t k = q.klen[(size_t)b], n = std::max<int64_t>(1, (k + TRI_KMAX - 1) / TRI_KMAX);
                for (int64_t j = 0; j < n; ++j) rows.push_back({b, (uint32_t)(k / n + (j < k % n ? 1 : 0))});
            }
            std::stable_sort(rows.begin(), rows.end(), [](const StagedRow &u, const StagedRow &v) { return u.k > v.k; });
            digits = digits_of(rows.empty() ? 0u : rows[0].k);  // (of the rows: what the small-shard kernel multiplies)
        }
    }
    if (!graded) {
        rows.reserve((size_t)R);
        for (int64_t r = 0; r < R; ++r) {
            const int32_t b = (int32_t)(branch_of_row.empty() ? r : branch_of_row[(size_t)r]);
            rows.push_back({b, q.klen[(size_t)b]});
        }
    }
    const int64_t Rs = (int64_t)rows.size();  // staged rows
    pl->m_digits = digits;
    pl->m_graded = graded;
    inf.n_digits = digits;
    const int64_t n8 = round_up(N, M_TILE_I);
    const int64_t n_slabs = mfma_staged_slabs(Rs);  // whole quads of slabs
    const int64_t ldb = n_slabs * M_KSLAB;
    pl->m_ldb = ldb;
    pl->m_n8 = n8;
    inf.ld = n8;
    inf.rows_padded = ldb;
    // (+ M_PAD_SLABS slabs of zeros behind the arrays: the kernel's prefetches run past an item's end)
    const size_t bits_bytes = sizeof(unsigned long long) * (size_t)mfma_alloc_slabs(Rs) * (size_t)n8;
    const size_t plane_alloc = (size_t)(mfma_alloc_slabs(Rs) * M_KSLAB);
    const size_t digit_bytes = plane_alloc * (size_t)(digits + 1);  // (+1: a single-digit item reads its plane twice)
    inf.staged_bytes = (double)bits_bytes + (double)digit_bytes + (graded ? 3.0 * (double)plane_alloc : 0.0);
    FF_ALLOC(pl->d_Pbits, bits_bytes, "the presence bits");
    FF_HIP(hipMalloc(&pl->d_Kd, digit_bytes));
    FF_HIP(hipMemset(pl->d_Pbits, 0, bits_bytes));
    FF_HIP(hipMalloc(&pl->d_W, sizeof(unsigned long long) * (size_t)n8));
    FF_HIP(hipMemset(pl->d_W, 0, sizeof(unsigned long long) * (size_t)n8));
    {
        // digits in the kernel's order of the 64 rows of a slab: chunk C, dword kk, byte q holds
        // row 32 * (C >> 1) + 8 * q + 4 * (C & 1) + kk (ff_kernels_mfma.hpp)
        auto pos_of = [](int64_t r) {
            const int64_t slab = r / M_KSLAB, w = r % M_KSLAB;  // w = 32 * h + 8 * q + 4 * c1 + kk
            const int64_t h = w >> 5, qq = (w >> 3) & 3, c1 = (w >> 2) & 1, kk = w & 3;
            return (size_t)(slab * M_KSLAB + (2 * h + c1) * 16 + kk * 4 + qq);
        };
        std::vector<int8_t> kd(digit_bytes, 0);
        for (int64_t r = 0; r < Rs; ++r) {
            const uint32_t k = rows[(size_t)r].k;
            const size_t pos = pos_of(r);
            for (int d = 0; d < digits; ++d) kd[(size_t)d * (size_t)ldb + pos] = (int8_t)((k >> (7 * d)) & 127u);
        }
        FF_HIP(hipMemcpy(pl->d_Kd, kd.data(), digit_bytes, hipMemcpyHostToDevice));
        pl->m_duo_from_slab = 0;
        if (graded) {
            std::vector<int8_t> kt(plane_alloc * 3, 0);
            int64_t first_duo = 0;  // the first row whose length (and every later one's) needs no third digit
            for (int64_t r = 0; r < Rs; ++r) {
                int8_t d[3];
                tri_digits((int64_t)rows[(size_t)r].k, d);
                const size_t pos = pos_of(r);
                kt[pos] = d[0];
                kt[(size_t)ldb + pos] = d[1];
                kt[2 * (size_t)ldb + pos] = d[2];
                if ((int64_t)rows[(size_t)r].k > DUO_KMAX) first_duo = r + 1;
            }
            pl->m_duo_from_slab = (int)((std::min(first_duo, Rs) + M_KSLAB - 1) / M_KSLAB);
            FF_HIP(hipMalloc(&pl->d_Kt, kt.size()));
            FF_HIP(hipMemcpy(pl->d_Kt, kt.data(), kt.size(), hipMemcpyHostToDevice));
        }
    }
    Scratch<uint32_t> klen;
    Scratch<int32_t> row_ptr, row_list;  // graded: the rows of a branch (it may have several)
    FF_HIP(klen.alloc((size_t)B));
    FF_HIP(hipMemcpy(klen.p, q.klen.data(), sizeof(uint32_t) * (size_t)B, hipMemcpyHostToDevice));
    if (graded) {
        std::vector<int32_t> ptr((size_t)B + 1, 0), list((size_t)std::max<int64_t>(Rs, 1));
        for (const StagedRow &sr : rows) ++ptr[(size_t)sr.branch + 1];
        for (int64_t b = 0; b < B; ++b) ptr[(size_t)b + 1] += ptr[(size_t)b];
        std::vector<int32_t> at(ptr.begin(), ptr.end() - 1);
        for (int64_t r = 0; r < Rs; ++r) list[(size_t)at[(size_t)rows[(size_t)r].branch]++] = (int32_t)r;
        FF_HIP(row_ptr.alloc(ptr.size()));
        FF_HIP(row_list.alloc(list.size()));
        FF_HIP(hipMemcpy(row_ptr.p, ptr.data(), sizeof(int32_t) * ptr.size(), hipMemcpyHostToDevice));
        FF_HIP(hipMemcpy(row_list.p, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice));
    }
    if (nnz > 0)
        stage_mfma_bits_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, klen.p, graded ? nullptr : row_of.p,
                                                                  row_ptr.p, row_list.p, pl->d_Pbits, n8, n_slabs, pl->d_W);
    FF_HIP(hipGetLastError());
    FF_HIP(hipDeviceSynchronize());
    klen.release();
    return schedule_mfma(pl, err, errlen);
}

// FIXED32 on the vector ALU: the branch-major u32 matrix, column sums, the sparse decision, the wave schedule.
int stage_for_sad(StageCtx &x, char *err, size_t errlen)
{
    FF_STAGE_NAMES;
    const int64_t ld = round_up(std::max<int64_t>(N, 1), TILE_J);
    // Sparse tables -- and dense ones with many samples: the rows that few samples reach leave the matrix
    // (ff_kernels_low.hpp).  A matrix row costs the dense kernel one lane-op per pair of the triangle:
    // P / 33e12 s (measured: 0.97 us a row at 8,192 samples, 0.26 at 4,096, 3.9 at 16,384).  A rare row r with n_r flat
    // nodes costs pair_low_kernel n_r^2 / 2 updates of an LDS accumulator whatever the number of pairs: priced below per
    // candidate side of its blocks of pairs, in steps of a wave at 1.3e-11 s chip-wide, the last, part-filled round of
    // the blocks on 2 workgroups per CU counted.  A row is rare when that is cheaper than its matrix row, up to
    // N / LOW_SHARE_DIV samples, and the split is taken when the whole estimate saves 7 % or more (ms per pass without ->
    // with: profiles/r05_sparse_split.txt).  FF_SPARSE_SPLIT = 1 / 0 forces / forbids it.
    Scratch<int32_t> qt_row_of, low_of;   // branch id -> row of the matrix / rare row (-1: not there)
    int64_t Rq = R, Rl = 0;
    {
        const int force = env_int("FF_SPARSE_SPLIT", -1);
        if (weighted && force != 0 && N >= 2 * LOW_TILE_MAX && nnz > 0 && R > 32) {
            Scratch<int32_t> cnt;
            FF_HIP(cnt.alloc((size_t)R));
            FF_HIP(hipMemset(cnt.p, 0, sizeof(int32_t) * (size_t)R));
            row_counts_kernel<<<dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 1 << 16)), dim3(256)>>>(d_ids, nnz, row_of.p, cnt.p);
            FF_HIP(hipGetLastError());
            std::vector<int32_t> h_cnt((size_t)R);
            FF_HIP(hipMemcpy(h_cnt.data(), cnt.p, sizeof(int32_t) * (size_t)R, hipMemcpyDeviceToHost));
            // (priced for THIS plan's shard: its pairs, and its blocks of pairs -- a row shard of a multi-GPU run holds a
            // G-th of the blocks, and how they fill their rounds weighs on the block side: profiles/r05_shard_balance.txt)
            const double Ps = std::max(1.0, (double)n_slots);
            constexpr double ROW_RATE = 33e12, STEP_COST = 0.13e-10, ROUND_STEPS = 4, WORD_STEPS = 8, ZERO_STEPS = 5;
            const double t_row = Ps / ROW_RATE;
            const double slots = 2.0 * (double)std::max(1, prop.multiProcessorCount);  // blocks in flight
            auto kernel_rows = [](double active_rows, double rows) { return active_rows <= 0.72 * rows ? active_rows / 0.77 : rows; };
            // (the sparse-aware kernel is taken from 28 % inactive cells and walks a row at 0.77 of the dense rate)
            double now_active = 0;
            for (int64_t r = 0; r < R; ++r)
                now_active += 1.0 - std::pow(1.0 - std::min(1.0, (double)h_cnt[(size_t)r] / (double)N), (double)TILE_I);  // P(a 32-sample block has it)
            const double t_now = kernel_rows(now_active, (double)R) * t_row;
            const int forced_tile = env_int("FF_LOW_TILE", 0);
            const int64_t cap = N / LOW_SHARE_DIV;
            double t_split = 0, rare_max = 0;
            int tile = 0;
            int64_t n_low = 0;
            std::vector<double> price((size_t)cap + 1);
            for (int cand : LOW_TILES) {
                bool known = false;
                for (int t : LOW_TILES) known = known || t == forced_tile;
                if (known && cand != forced_tile) continue;
                double tc = 0;  // the shard's blocks: sample blocks bi that hold one of its rows, times bj <= bi
                for (int64_t bi = inf.row_begin / cand; bi * cand < std::min<int64_t>(inf.row_end, N); ++bi) tc += (double)(bi + 1);
                tc = std::max(tc, 1.0);
                const double fill = std::ceil(tc / slots) * slots / tc;  // the last round's idle slots
                const double nblk = (double)((N + cand - 1) / cand);
                // price[n]: a rare row with n flat nodes, in seconds of the chip.  n / nblk entries per sample block, a block
                // has any with p = 1 - exp(-that), so the row is met in tc p^2 of the shard's blocks of pairs with a = n /
                // (nblk p) entries on each side.  There its B entries take a / 64 of a round of the wave, and the round lasts
                // as long as the longest A list among its 64 / a rows -- a + k sqrt(a), k the expected maximum of that many
                // normals -- + 1.5 for the last quad's zero adds + ROUND_STEPS for the search and the loads' latency.  Per
                // bitmap word and block of pairs: WORD_STEPS where the word has any row in both sample blocks (the index
                // look-ups, the scan), ZERO_STEPS where it has none; a row carries a 64th of its word.
                // One STEP_COST reproduces pair_low_kernel within 12 % (5 % rms) at every block side at C3, C4, C5 and
                // C5's tree at 1 % and 0.2 % leaf density (tools/experiments/low_tile_sweep.sh,
                // profiles/r05_low_tile_sweep.txt), and picks the side that measures best at each.
                // Rare: every row up to the first n that a matrix row would do cheaper (or, forced, up to the cap -- the
                // tests' small problems).
                static const double KMAX[7] = {0, 0.56, 1.03, 1.42, 1.77, 2.07, 2.33};  // E max of 1, 2, 4 .. 64 normals
                int64_t rmax = 0;
                for (int64_t n = 1; n <= cap; ++n) {
                    const double na = (double)n / nblk, p = 1.0 - std::exp(-na), a = na / p, both = p * p;
                    const double lg = std::min(6.0, std::max(0.0, std::log2(64.0 / a)));
                    const int l0 = std::min(5, (int)lg);
                    const double k = KMAX[l0] + (KMAX[l0 + 1] - KMAX[l0]) * (lg - l0);
                    const double row_steps = both * a / 64.0 * (a + k * std::sqrt(a) + 1.5 + ROUND_STEPS);
                    const double word_steps = ((1.0 - std::pow(1.0 - both, 64.0)) * WORD_STEPS + ZERO_STEPS) / 64.0;
                    price[(size_t)n] = tc * (row_steps + word_steps) * STEP_COST * fill;
                    if (force <= 0 && price[(size_t)n] >= t_row) break;
                    rmax = n;
                }
                double high_active = 0, t_low = 0;
                int64_t nl = 0;
                for (int64_t r = 0; r < R; ++r) {
                    const int64_t n = h_cnt[(size_t)r];
                    if (n <= rmax) {
                        ++nl;
                        t_low += price[(size_t)n];
                    } else {
                        high_active += 1.0 - std::pow(1.0 - std::min(1.0, (double)n / (double)N), (double)TILE_I);
                    }
                }
                const double t = kernel_rows(high_active, (double)(R - nl)) * t_row + t_low;
                if (tile == 0 || t < t_split) {
                    tile = cand;
                    t_split = t;
                    rare_max = (double)rmax;
                    n_low = nl;
                }
            }
            // (the index is (rare rows + 1) x sample blocks words, twice while it is built: not beyond 4 GB; and it counts
            // the rare rows' entries -- their flat nodes -- in 32 bits)
            double rare_entries = 0;
            for (int64_t r = 0; r < R; ++r)
                if ((double)h_cnt[(size_t)r] <= rare_max) rare_entries += (double)h_cnt[(size_t)r];
            const bool fits = 8.0 * (double)(n_low + 1) * (double)((N + std::max(tile, 1) - 1) / std::max(tile, 1)) <= 4e9 &&
                              rare_entries < 4.0e9;
            const bool take = R - n_low >= 16 && n_low > 0 && fits && (force > 0 || t_split <= 0.93 * t_now);
            if (take) {
                std::vector<int32_t> h_qt((size_t)B, -1), h_low((size_t)B, -1);
                // The matrix rows keep the staged order.  The rare rows are numbered by DESCENDING sample count: the 64
                // rows of a bitmap word are worked together by a wave of pair_low_kernel, a round of it lasts as long as the
                // longest list among the rows it holds, and rows of like weight side by side keep those alike (in branch
                // order the round's longest list was 13 times its average one).  Integer sums: any numbering gives the same M.
                Rq = 0;
                std::vector<int64_t> rare;
                for (int64_t r = 0; r < R; ++r) {
                    const int64_t b = branch_of_row.empty() ? r : branch_of_row[(size_t)r];
                    if ((double)h_cnt[(size_t)r] <= rare_max) rare.push_back(r);  // (n_r = 0 cannot be: a staged row has a flat node)
                    else h_qt[(size_t)b] = (int32_t)Rq++;
                }
                std::stable_sort(rare.begin(), rare.end(), [&](int64_t u, int64_t v) { return h_cnt[(size_t)u] > h_cnt[(size_t)v]; });
                for (int64_t r : rare) h_low[(size_t)(branch_of_row.empty() ? r : branch_of_row[(size_t)r])] = (int32_t)Rl++;
                FF_HIP(qt_row_of.alloc((size_t)B));
                FF_HIP(low_of.alloc((size_t)B));
                FF_HIP(hipMemcpy(qt_row_of.p, h_qt.data(), sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice));
                FF_HIP(hipMemcpy(low_of.p, h_low.data(), sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice));
                pl->split = true;
                pl->low_tile = tile;
                pl->low_rows = Rl;
                inf.rare_rows = Rl;
                double upd = 0;  // (the kernel's own work: bench.py's roofline.parts prices it)
                for (int64_t r : rare) upd += 0.5 * (double)h_cnt[(size_t)r] * (double)(h_cnt[(size_t)r] - 1);
                inf.rare_updates = upd * Ps / std::max(1.0, 0.5 * (double)N * (double)(N - 1));
            }
        }
    }
    const int32_t *stage_row_of = pl->split ? qt_row_of.p : row_of.p;
    const int64_t rows = sad_staged_rows(Rq);
    inf.ld = ld;
    inf.rows_padded = rows;
    inf.lengths_exact = weighted ? 0 : q.lengths_exact;
    const size_t qt_bytes = sizeof(uint32_t) * (size_t)sad_alloc_rows(Rq) * (size_t)ld;
    inf.staged_bytes = (double)qt_bytes;
    FF_ALLOC(pl->d_QT, qt_bytes, "the staged branch x sample matrix");
    FF_HIP(hipMalloc(&pl->d_W, sizeof(unsigned long long) * (size_t)ld));
    Scratch<uint32_t> klen;
    if (!weighted) {
        FF_HIP(klen.alloc((size_t)B));
        if (B > 0) FF_HIP(hipMemcpy(klen.p, q.klen.data(), sizeof(uint32_t) * (size_t)B, hipMemcpyHostToDevice));
    }
    std::vector<unsigned long long> hW((size_t)ld);
    const int nb = pl->split ? (int)((N + pl->low_tile - 1) / pl->low_tile) : 0;
    if (pl->split) {
        FF_HIP(hipMalloc(&pl->d_Wl, sizeof(uint32_t) * (size_t)ld));
        FF_HIP(hipMemset(pl->d_Wl, 0, sizeof(uint32_t) * (size_t)ld));
    }
    int e = q.e;
    for (int attempt = 0;; ++attempt) {
        FF_HIP(hipMemset(pl->d_QT, 0, qt_bytes));
        FF_HIP(hipMemset(pl->d_W, 0, sizeof(unsigned long long) * (size_t)ld));
        if (N > 0 && nnz > 0)
            stage_fixed32_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, d_abnd, d_len, klen.p,
                                                                    weighted ? 1 : 0, e, stage_row_of, pl->d_QT, ld);
        if (rows > 0) {
            const int64_t rpb = std::max<int64_t>(64, round_up(rows, 256) / 256);
            dim3 grid((unsigned)(ld / 64), (unsigned)((rows + rpb - 1) / rpb));
            colsum_kernel<<<grid, dim3(64)>>>(pl->d_QT, ld, rows, rpb, pl->d_W);
        }
        if (pl->split)  // the rare rows' share of every column sum: into Wl, and into W (every W_s must stay below 2^31)
            low_sums_counts_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, d_abnd, d_len, e, low_of.p, Rl + 1, pl->low_tile, nullptr,
                                                                      pl->d_Wl, pl->d_W);
        FF_HIP(hipGetLastError());
        FF_HIP(hipMemcpy(hW.data(), pl->d_W, sizeof(unsigned long long) * (size_t)ld, hipMemcpyDeviceToHost));
        unsigned long long wmax = 0;
        for (auto w : hW) wmax = std::max(wmax, w);
        if (wmax <= 2147483647ull) break;
        if (!weighted || attempt >= 3)
            return ff::fail(FF_ERR_INTERNAL, err, errlen, "FIXED32 staging overflow (max column sum %llu)", wmax);
        --e;  // rounding pushed a column over the bound: drop one bit
    }
    klen.release();
    inf.scale_log2 = e;
    if (pl->split) {
        // the rare rows' entries grouped by sample block: counts -> positions -> (sample, value) pairs -> row bitmaps
        pl->low_blocks = nb;
        pl->low_words = (Rl + 63) / 64;
        const int64_t rows1 = Rl + 1;
        const size_t cells = (size_t)rows1 * (size_t)nb;
        FF_ALLOC(pl->d_low_ptr, sizeof(uint32_t) * cells, "the rare rows' index");
        FF_HIP(hipMemset(pl->d_low_ptr, 0, sizeof(uint32_t) * cells));
        low_sums_counts_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, d_abnd, d_len, e, low_of.p, rows1, pl->low_tile, pl->d_low_ptr,
                                                                  nullptr, nullptr);
        Scratch<uint32_t> totals, cursor;
        FF_HIP(totals.alloc((size_t)nb + 1));
        FF_HIP(cursor.alloc(cells));
        FF_HIP(hipMemset(cursor.p, 0, sizeof(uint32_t) * cells));
        // per block: exclusive prefix over its rows1 positions (the last one becomes the block's total), then the
        // blocks' bases
        scan_segments_kernel<<<dim3((unsigned)nb), dim3(1024)>>>(pl->d_low_ptr, rows1, rows1, totals.p);
        FF_HIP(hipGetLastError());
        std::vector<uint32_t> h_tot((size_t)nb), h_base((size_t)nb + 1, 0u);
        FF_HIP(hipMemcpy(h_tot.data(), totals.p, sizeof(uint32_t) * (size_t)nb, hipMemcpyDeviceToHost));
        uint64_t run = 0;
        for (int k = 0; k < nb; ++k) {
            h_base[(size_t)k] = (uint32_t)run;
            run += h_tot[(size_t)k];
        }
        if (run >= 0xFFFFFFF0ull) return ff::fail(FF_ERR_INTERNAL, err, errlen, "too many rare-row entries");
        const uint32_t n_entries = (uint32_t)run;
        FF_HIP(hipMemcpy(totals.p, h_base.data(), sizeof(uint32_t) * (size_t)nb, hipMemcpyHostToDevice));
        low_add_base_kernel<<<dim3((unsigned)((cells + 255) / 256)), dim3(256)>>>(pl->d_low_ptr, rows1, nb, totals.p);
        FF_ALLOC(pl->d_low_ent, sizeof(uint2) * ((size_t)n_entries + 4), "the rare rows' entries");
        FF_HIP(hipMemset(pl->d_low_ent + n_entries, 0, sizeof(uint2) * 4));  // (spare: pair_low_kernel loads four at a time, and adds the zeros)
        low_fill_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, d_abnd, d_len, e, low_of.p, rows1, pl->low_tile, pl->d_low_ptr, cursor.p,
                                                           pl->d_low_ent);
        FF_ALLOC(pl->d_low_bits, sizeof(unsigned long long) * (size_t)nb * (size_t)pl->low_words, "the rare rows' bitmaps");
        low_bits_kernel<<<dim3((unsigned)pl->low_words, (unsigned)nb), dim3(64)>>>(pl->d_low_ptr, Rl, rows1, pl->low_words, pl->d_low_bits);
        FF_HIP(hipGetLastError());
        FF_HIP(hipDeviceSynchronize());  // (the scratch arrays go out of scope)
        inf.staged_bytes += 4.0 * (double)cells + 8.0 * (double)n_entries + 8.0 * (double)nb * (double)pl->low_words;
    }
    pl->n_workgroups = prop.multiProcessorCount;  // persistent: one workgroup per CU
    pl->lds_bytes = 96 * 1024;  // unused dynamic LDS sized so that exactly one workgroup fits a CU
    // activity of every (i-block, branch row): decides between the dense and the
    // sparse-aware kernel
    if (env_int("FF_SPARSE", 1) != 0 && rows > 0 && N > 0) {
        const int64_t n_iblocks = ld / TILE_I, words = (rows + SLACK_ROWS + 63) / 64;
        Scratch<unsigned long long> act64;
        FF_HIP(act64.alloc((size_t)(n_iblocks * words)));
        build_activity_kernel<<<dim3((unsigned)words, (unsigned)n_iblocks), dim3(64)>>>(pl->d_QT, ld, rows, words,
                                                                                        act64.p);
        FF_HIP(hipGetLastError());
        std::vector<unsigned long long> a64((size_t)(n_iblocks * words));
        FF_HIP(hipMemcpy(a64.data(), act64.p, sizeof(unsigned long long) * a64.size(), hipMemcpyDeviceToHost));
        act64.release();
        // only the i-blocks this shard's tiles use count for the decision
        const int64_t ib0 = inf.row_begin / TILE_I, ib1 = (inf.row_end + TILE_I - 1) / TILE_I;
        int64_t active = 0;
        for (int64_t ib = ib0; ib < ib1; ++ib)
            for (int64_t w = 0; w < words; ++w) active += __builtin_popcountll(a64[(size_t)(ib * words + w)]);
        const double total = (double)std::max<int64_t>(1, (ib1 - ib0) * rows);
        const double inactive = 1.0 - (double)active / total;
        inf.active_fraction = (double)active / total;
        const auto thr = ff::tuning("FF_SPARSE_MIN");
        // the list walk runs at about 0.77 of the dense loop's rate per row (shallower
        // prefetch, per-row address arithmetic), so it pays from about a quarter upwards
        if (inactive >= (thr && !thr->empty() ? atof(thr->c_str()) : 0.28)) {
            // per i-block: the list of active rows and, every 16 rows, where the list stands
            const int64_t marks = rows / (2 * KSTEP) + 1;
            pl->aptr_stride = marks;
            std::vector<uint32_t> arows, aptr((size_t)(n_iblocks * marks), 0u);
            arows.reserve((size_t)active + SPARSE_LIST_PAD);
            for (int64_t ib = 0; ib < n_iblocks; ++ib)
                for (int64_t r = 0; r <= rows; ++r) {
                    if (r % (2 * KSTEP) == 0) aptr[(size_t)(ib * marks + r / (2 * KSTEP))] = (uint32_t)arows.size();
                    if (r < rows && ((a64[(size_t)(ib * words + r / 64)] >> (r % 64)) & 1ull)) arows.push_back((uint32_t)r);
                }
            if (arows.size() >= 0xFFFFFFF0ull)
                return ff::fail(FF_ERR_INTERNAL, err, errlen, "active-row list too long");
            arows.resize(arows.size() + SPARSE_LIST_PAD, (uint32_t)rows);  // (spare entries: the batch prefetch, ff_schedule.hpp)
            pl->zero_row = (int32_t)rows;  // first slack row: zero in every column
            FF_HIP(hipMalloc(&pl->d_arows, sizeof(uint32_t) * arows.size()));
            FF_HIP(hipMemcpy(pl->d_arows, arows.data(), sizeof(uint32_t) * arows.size(), hipMemcpyHostToDevice));
            FF_HIP(hipMalloc(&pl->d_aptr16, sizeof(uint32_t) * aptr.size()));
            FF_HIP(hipMemcpy(pl->d_aptr16, aptr.data(), sizeof(uint32_t) * aptr.size(), hipMemcpyHostToDevice));
            FF_HIP(hipMalloc(&pl->d_cs16, sizeof(uint32_t) * (size_t)(marks * ld)));
            {
                static_assert(PFX_CHUNK_ROWS % (2 * KSTEP) == 0, "prefix16: a chunk holds whole marks");
                const int64_t n_chunks = (rows + PFX_CHUNK_ROWS - 1) / PFX_CHUNK_ROWS;
                Scratch<uint32_t> chunk_sums;
                FF_HIP(chunk_sums.alloc((size_t)(n_chunks * ld)));
                const dim3 grid((unsigned)((ld + 63) / 64), (unsigned)n_chunks);
                prefix16_sums_kernel<<<grid, dim3(64)>>>(pl->d_QT, ld, rows, chunk_sums.p);
                prefix16_scan_kernel<<<dim3((unsigned)((ld + 63) / 64)), dim3(64)>>>(chunk_sums.p, ld, n_chunks);
                prefix16_fill_kernel<<<grid, dim3(64)>>>(pl->d_QT, ld, rows, chunk_sums.p, pl->d_cs16);
                FF_HIP(hipGetLastError());
                FF_HIP(hipDeviceSynchronize());  // (chunk_sums goes out of scope)
            }
            pl->sparse = true;
            inf.kernel = FF_KERNEL_SAD_U32_SPARSE;
        }
    }
    return schedule_sad(pl, err, errlen);
}

// EXACT64 unweighted: presence bits (a word per 32 staged rows and sample), the lengths by staged row, the tiles.
int stage_for_exact_unw(StageCtx &x, char *err, size_t errlen)
{
    FF_STAGE_NAMES;
    pl->xu = true;
    inf.kernel = FF_KERNEL_EXACT_F64_UNW;
    const int64_t ldx = xu_ld(N);
    pl->xu_ldx = ldx;
    pl->xu_slabs = (int)xu_slabs(R);
    inf.ld = ldx;
    inf.rows_padded = xu_slabs(R) * XU_SLAB;
    const size_t bits_bytes = sizeof(uint32_t) * (size_t)xu_alloc_slabs(R) * (size_t)ldx;
    const size_t len_count = (size_t)xu_alloc_lengths(R);
    inf.staged_bytes = (double)bits_bytes + 8.0 * (double)len_count;
    FF_ALLOC(pl->d_Xbits, bits_bytes, "the presence bits");
    FF_HIP(hipMemset(pl->d_Xbits, 0, bits_bytes));
    if (N > 0 && nnz > 0)
        stage_xbits_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, row_of.p, pl->d_Xbits, ldx);
    FF_HIP(hipGetLastError());
    std::vector<double> lr(len_count, 0.0);  // treeDists by staged row, zeros behind
    for (int64_t r = 0; r < R; ++r) lr[(size_t)r] = c->h_len[(size_t)(branch_of_row.empty() ? r : branch_of_row[(size_t)r])];
    FF_HIP(hipMalloc(&pl->d_len_rows, sizeof(double) * len_count));
    FF_HIP(hipMemcpy(pl->d_len_rows, lr.data(), sizeof(double) * len_count, hipMemcpyHostToDevice));
    return schedule_exact_unw(pl, err, errlen);
}

// EXACT64: the branch-major binary64 matrix and its tiles.
int stage_for_exact64(StageCtx &x, char *err, size_t errlen)
{
    FF_STAGE_NAMES;
    if (!weighted && N > 0 && B > 0 && env_int("FF_EXACT_UNW", 1) != 0) return stage_for_exact_unw(x, err, errlen);
    const int64_t ld = round_up(std::max<int64_t>(N, 1), X_TILE_J);
    inf.ld = ld;
    inf.rows_padded = R;
    // (+ X_VALUES_PAD values: a tile whose height does not divide 64 reads up to H - 1 operands past the last row's end)
    const size_t dt_bytes = sizeof(double) * ((size_t)std::max<int64_t>(R, 1) * (size_t)ld + X_VALUES_PAD);
    inf.staged_bytes = (double)dt_bytes;
    FF_ALLOC(pl->d_DT, dt_bytes, "the staged binary64 matrix");
    FF_HIP(hipMemset(pl->d_DT, 0, dt_bytes));
    if (N > 0 && nnz > 0)
        stage_exact64_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, d_abnd, weighted ? 1 : 0,
                                                                row_of.p, pl->d_DT, ld);
    FF_HIP(hipGetLastError());
    if (row_of.p) {  // the walk reads treeDists by staged row
        std::vector<double> lr((size_t)R);
        for (int64_t r = 0; r < R; ++r) lr[(size_t)r] = c->h_len[(size_t)branch_of_row[(size_t)r]];
        FF_HIP(hipMalloc(&pl->d_len_rows, sizeof(double) * (size_t)R));
        FF_HIP(hipMemcpy(pl->d_len_rows, lr.data(), sizeof(double) * (size_t)R, hipMemcpyHostToDevice));
    }
    return schedule_exact64(pl, err, errlen);
}

}  // namespace

// Stages the device-resident flat nodes and builds the schedule.  Takes ownership of *c.
int plan_build(const ff_options *o, DeviceCsr *c, const hipDeviceProp_t &prop, ff_plan *pl, char *err,
               size_t errlen)
{
    const int64_t N = c->N, B = c->B;
    const bool weighted = pl->weighted != 0;
    ff_plan_info &inf = pl->info;
    pl->d_len = c->d_len;  // the plan owns the device arrays from here on
    pl->d_indptr = c->d_indptr;
    pl->d_ids = c->d_ids;
    pl->d_abnd = c->d_abnd;
    c->d_len = nullptr;
    c->d_indptr = nullptr;
    c->d_ids = nullptr;
    c->d_abnd = nullptr;

    // A branch length that is not a finite number (the Newick reader takes "inf" and "nan" as strconv.ParseFloat does):
    // the reference's walk never touches a branch NEITHER sample of a pair has (unifrac.go:178-203), so such a pair keeps
    // a finite distance, while every dense reformulation multiplies the length by the pair's absent / absent zeros --
    // Inf * 0 = NaN.  pair_exact_unw_kernel forms its operands by masking the length's bits and is safe; the weighted
    // EXACT64 kernels (and the unweighted one on the weighted kernel's arithmetic, FF_EXACT_UNW=0) are not: such a tree
    // goes to the literal walk, the reference's own operations on the pairs' own lists.  (FIXED32 refuses it: choose_quant.)
    bool nonfinite_len = false;
    for (int64_t b = 0; b < B; ++b) nonfinite_len = nonfinite_len || !std::isfinite(c->h_len[(size_t)b]);
    const bool dense_unsafe = nonfinite_len && o->precision != FF_PRECISION_FIXED32 && N > 0 &&
                              (weighted || env_int("FF_EXACT_UNW", 1) == 0);
    if ((o->flags & FF_FLAG_UNSORTED_WALK) || dense_unsafe) {
        // nothing to stage: the walk reads the flat nodes as they stand, and every reformulation above (dense rows,
        // integer sums, presence bits) assumes lists a merge pairs up correctly
        pl->walk = true;
        inf.n_rows = B;
        inf.rows_padded = B;
        inf.precision = FF_PRECISION_EXACT64;
        inf.kernel = FF_KERNEL_WALK_F64;
        inf.staged_bytes = 12.0 * (double)c->nnz;
        inf.n_tiles = inf.n_items = 0;
        inf.n_wave_slots = (int64_t)inf.n_compute_units * 8 * 4;
        inf.elements = 0;
        FF_HIP(hipDeviceSynchronize());
        return FF_OK;
    }
    StageCtx x;
    x.o = o;
    x.c = c;
    x.prop = &prop;
    x.pl = pl;
    int rc = compact_branches(x, err, errlen);
    if (rc) return rc;
    const int64_t R = x.R;
    Quant &q = x.q;
    inf.n_rows = R;

    int prec = o->precision;
    const bool is_auto = prec == FF_PRECISION_AUTO;
    // AUTO: problems small enough that the binary64 walk costs about a millisecond
    // get the reference's exact roundings (this covers all of the reference's own
    // test data); everything larger takes the fixed-point path -- except UNWEIGHTED with
    // a branch length off the binary grid (below).
    if (is_auto && (double)ff_num_pairs(N) * (double)R <= 4294967296.0)
        prec = FF_PRECISION_EXACT64;
    if (prec != FF_PRECISION_EXACT64) {
        q = choose_quant(*c, weighted, pl->d_indptr, pl->d_ids, pl->d_abnd, pl->d_len);
        if (!q.fixed_ok) {
            if (prec == FF_PRECISION_FIXED32)
                return ff::fail(FF_ERR_ARG, err, errlen, "FIXED32 not applicable: %s", q.why_not.c_str());
            prec = FF_PRECISION_EXACT64;
        } else if (is_auto && !weighted && !q.lengths_exact) {
            // The reference's unweighted value is what its two chains of additions round to (unifrac.go:144-171), and
            // the bar for unweighted is its bits, not a tolerance: integer lengths that carry a rounding (any real
            // phylogeny) cannot give them, pair_exact_unw_kernel does (C3's shape: 10 ms a pass against 0.3 on the
            // matrix cores -- a thirtieth of what the command spends reading the table and writing the distances).
            // FIXED32 on such lengths stays available on request: within 1e-6, with refinement and audit.
            prec = FF_PRECISION_EXACT64;
        } else {
            prec = FF_PRECISION_FIXED32;
        }
    }
    inf.precision = prec;
    inf.kernel = prec == FF_PRECISION_EXACT64 ? FF_KERNEL_EXACT_F64 : FF_KERNEL_SAD_U32;

    const bool use_mfma = prec == FF_PRECISION_FIXED32 && !weighted && env_int("FF_UNWEIGHTED_MFMA", 1) != 0 && N > 0 && B > 0;
    if (use_mfma) rc = stage_for_mfma(x, err, errlen);
    else if (prec == FF_PRECISION_FIXED32) rc = stage_for_sad(x, err, errlen);
    else rc = stage_for_exact64(x, err, errlen);
    if (rc) return rc;
    FF_HIP(hipDeviceSynchronize());
    // FIXED32 whose integers carry a rounding (weighted; unweighted with lengths off the binary
    // grid) divides by binary64 weights, so that only the numerator's rounding reaches a distance
    if (prec == FF_PRECISION_FIXED32 && (weighted || !inf.lengths_exact) && N > 0) {
        FF_HIP(hipMalloc(&pl->d_wex, sizeof(double) * (size_t)N));
        exact_weight_kernel<<<dim3((unsigned)N), dim3(256)>>>(pl->d_indptr, pl->d_ids, pl->d_abnd, pl->d_len,
                                                               weighted ? 1 : 0, pl->d_wex);
        FF_HIP(hipGetLastError());
        FF_HIP(hipDeviceSynchronize());
    }
    // FIXED32 keeps the flat nodes resident for refine_exact_kernel unless the integer
    // sums are exact already (unweighted with lengths on the binary grid)
    if (prec == FF_PRECISION_FIXED32 && (weighted || !inf.lengths_exact) && env_int("FF_REFINE", 1)) {
        pl->refine = true;
        rc = alloc_refine_queue(pl, err, errlen);
        if (rc) return rc;
    } else {
        (void)hipFree(pl->d_indptr);
        (void)hipFree(pl->d_ids);
        (void)hipFree(pl->d_abnd);
        pl->d_indptr = nullptr;
        pl->d_ids = nullptr;
        pl->d_abnd = nullptr;
    }
    return FF_OK;
}

}  // namespace dev
}  // namespace ff
