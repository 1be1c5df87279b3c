// ff_flatten.cpp -- species validation and stage A of the UniFrac path on the host.
//
// Replaces validateSpecies/treeNames (frcfrc/unifrac.go:70-93) and the
// per-sample conversion of unifrac() (unifrac.go:99-116): abundanceToFlatNodes
// (:32-53) followed by normalizeFlatNodes (:56-67).  This is the "post-order
// accumulation over the tree once on the host" of the design; the pairwise
// stage B runs on the GPU (ff_dev_run.hip).
//
// The recursion of abundanceToFlatNodes is replaced by one descending-id sweep
// over the pre-order arrays (children have larger ids than their parent), but
// the ORDER of the float additions is the reference's: a node's sum adds its
// children left to right, then its own leaf value.  Flat nodes come out in
// ascending id, i.e. already in the order normalizeFlatNodes sorts them into;
// the normaliser is the reference's: the sum, in that order, of ALL flat-node
// abundances (leaves, internal nodes and root -- SURVEY.md Q1), not the sample
// total.
#include <cmath>
#include <thread>

#include "ff_host.hpp"

namespace {

// One sample: leaf values already scattered into val[] (zero elsewhere).
// Appends (id, abundance) with abundance > 0 in ascending id.  `sum` is scratch [B].
// post_order (may be null): emit in THAT order instead -- what the reference's lists look like under -l, where
// normalizeFlatNodes is skipped together with its sort (unifrac.go:57-59,108-110) and the recursion's own order,
// children before their parent, stays.
void flatten_one(const ff_tree &t, const double *val, double *sum, bool normalize,
                 std::vector<int32_t> *ids, std::vector<double> *ab, const std::vector<int32_t> *post_order = nullptr)
{
    const int64_t B = (int64_t)t.size.size();
    for (int64_t id = B - 1; id >= 0; --id) {
        double s = 0.0;
        const int64_t end = id + t.size[(size_t)id];
        for (int64_t c = id + 1; c < end; c += t.size[(size_t)c]) s += sum[c];  // :35-37
        if (t.size[(size_t)id] == 1) {                                          // :39
            double a = val[id];
            if (a > 0) s += a;                                                   // :40-42
        }
        sum[id] = s;
    }
    size_t first = ids->size();
    double total = 0.0;
    if (post_order) {
        for (int32_t id : *post_order)
            if (sum[id] > 0) {  // :49-51, appended when the recursion returns from the node
                ids->push_back(id);
                ab->push_back(sum[id]);
            }
        return;
    }
    for (int64_t id = 0; id < B; ++id)
        if (sum[id] > 0) {  // :49-51
            ids->push_back((int32_t)id);
            ab->push_back(sum[id]);
            total += sum[id];  // normalizeFlatNodes :60-63, ascending id
        }
    if (normalize)
        for (size_t k = first; k < ab->size(); ++k) (*ab)[k] /= total;  // :64-66
}

int flatten_impl(const ff_tree &t, int64_t ns, const int64_t *leaf_ptr, const int64_t *leaf_idx,
                 const double *leaf_val, bool normalize, bool reference_l_order, unsigned threads, ff_flat **out,
                 char *err, size_t errlen)
{
    const int64_t B = (int64_t)t.size.size();
    // the order in which abundanceToFlatNodes' recursion appends (unifrac.go:35-52): a node after its subtree
    std::vector<int32_t> post_order;
    if (reference_l_order && B > 0 && B <= INT32_MAX) {
        post_order.reserve((size_t)B);
        std::vector<std::pair<int64_t, int64_t>> st;  // (node, next child to visit)
        st.push_back({0, 1});
        while (!st.empty()) {
            auto &top = st.back();
            const int64_t id = top.first, end = id + t.size[(size_t)id];
            if (top.second < end) {
                const int64_t c = top.second;
                top.second = c + t.size[(size_t)c];
                st.push_back({c, c + 1});
            } else {
                post_order.push_back((int32_t)id);
                st.pop_back();
            }
        }
    }
    if (B > INT32_MAX) return ff::fail(FF_ERR_ARG, err, errlen, "tree has too many nodes (%lld)", (long long)B);
    for (int64_t k = 0; k < leaf_ptr[ns]; ++k)
        if (leaf_idx[k] < 0 || leaf_idx[k] >= B)
            return ff::fail(FF_ERR_ARG, err, errlen, "leaf index %lld out of range", (long long)leaf_idx[k]);
    auto *f = new ff_flat();
    f->n_samples = ns;
    f->n_branches = B;
    f->branch_len = t.dist;  // treeDists, unifrac.go:117-120
    struct Part {
        std::vector<int32_t> ids;
        std::vector<double> ab;
        std::vector<int64_t> cnt;
        int64_t begin = 0;
    };
    if (threads > (unsigned)std::max<int64_t>(ns, 1)) threads = (unsigned)std::max<int64_t>(ns, 1);
    std::vector<Part> parts(threads);
    ff::parallel_for(ns, threads, [&](unsigned tid, int64_t b, int64_t e) {
        Part &p = parts[tid];
        p.begin = b;
        std::vector<double> val((size_t)B, 0.0), sum((size_t)B, 0.0);
        for (int64_t s = b; s < e; ++s) {
            for (int64_t k = leaf_ptr[s]; k < leaf_ptr[s + 1]; ++k) val[(size_t)leaf_idx[k]] = leaf_val[k];
            size_t before = p.ids.size();
            flatten_one(t, val.data(), sum.data(), normalize, &p.ids, &p.ab, reference_l_order ? &post_order : nullptr);
            p.cnt.push_back((int64_t)(p.ids.size() - before));
            for (int64_t k = leaf_ptr[s]; k < leaf_ptr[s + 1]; ++k) val[(size_t)leaf_idx[k]] = 0.0;
        }
    });
    f->indptr.assign((size_t)ns + 1, 0);
    size_t total = 0;
    for (auto &p : parts) total += p.ids.size();
    f->branch_id.reserve(total);
    f->abnd.reserve(total);
    int64_t s = 0;
    for (auto &p : parts) {
        for (int64_t c : p.cnt) {
            f->indptr[(size_t)s + 1] = f->indptr[(size_t)s] + c;
            ++s;
        }
        f->branch_id.insert(f->branch_id.end(), p.ids.begin(), p.ids.end());
        f->abnd.insert(f->abnd.end(), p.ab.begin(), p.ab.end());
    }
    *out = f;
    return FF_OK;
}

}  // namespace

namespace ff {

// abnd[tree.Name] for leaves only (unifrac.go:38-43): species -> leaf ids resolved once;
// a name carried by several leaves feeds each of them; a key naming an internal node is
// never looked up.
void table_leaf_csr(const ff_table &tb, const ff_tree &tr, std::vector<int64_t> *ptr, I64Vec *idx, F64Vec *val,
                    int threads)
{
    std::vector<const std::vector<int64_t> *> where(tb.species.size(), nullptr);
    for (size_t k = 0; k < tb.species.size(); ++k) {
        auto it = tr.leaf_ids.find(tb.species[k]);
        if (it != tr.leaf_ids.end()) where[k] = &it->second;
    }
    const int64_t ns = (int64_t)tb.ptr.size() - 1;
    const unsigned nt = clamp_threads(threads);
    // count, prefix, fill: samples are independent
    ptr->assign((size_t)ns + 1, 0);
    parallel_for(ns, nt, [&](unsigned, int64_t sb, int64_t se) {
        for (int64_t s = sb; s < se; ++s) {
            int64_t c = 0;
            for (int64_t k = tb.ptr[(size_t)s]; k < tb.ptr[(size_t)s + 1]; ++k)
                if (const auto *ids = where[(size_t)tb.key[(size_t)k]]) c += (int64_t)ids->size();
            (*ptr)[(size_t)s + 1] = c;
        }
    });
    for (int64_t s = 0; s < ns; ++s) (*ptr)[(size_t)s + 1] += (*ptr)[(size_t)s];
    idx->resize((size_t)(*ptr)[(size_t)ns]);
    val->resize((size_t)(*ptr)[(size_t)ns]);
    parallel_for(ns, nt, [&](unsigned, int64_t sb, int64_t se) {
        for (int64_t s = sb; s < se; ++s) {
            size_t at = (size_t)(*ptr)[(size_t)s];
            for (int64_t k = tb.ptr[(size_t)s]; k < tb.ptr[(size_t)s + 1]; ++k) {
                const auto *ids = where[(size_t)tb.key[(size_t)k]];
                if (!ids) continue;
                for (int64_t id : *ids) {
                    (*idx)[at] = id;
                    (*val)[at++] = tb.val[(size_t)k];
                }
            }
        }
    });
}

}  // namespace ff

extern "C" {

int ff_validate_species(const ff_table *tb, const ff_tree *tr, char *err, size_t errlen)
{
    if (!tb || !tr) return ff::fail(FF_ERR_ARG, err, errlen, "ff_validate_species: null argument");
    // species that are tree names (any node, internal and "" included: unifrac.go:70-76)
    std::vector<char> ok(tb->species.size(), 0);
    for (size_t k = 0; k < tb->species.size(); ++k) ok[k] = tr->all_names.count(tb->species[k]) ? 1 : 0;
    const int64_t ns = (int64_t)tb->ptr.size() - 1;
    for (int64_t s = 0; s < ns; ++s)
        for (int64_t k = tb->ptr[(size_t)s]; k < tb->ptr[(size_t)s + 1]; ++k)
            if (!ok[(size_t)tb->key[(size_t)k]])
                return ff::fail(FF_ERR_SPECIES, err, errlen,
                                "sample #%lld has value %s for species %s which is not in the tree",
                                (long long)s + 1, ff::go_v(tb->val[(size_t)k]).c_str(),
                                ff::go_quote(tb->species[(size_t)tb->key[(size_t)k]]).c_str());
    return FF_OK;
}

int ff_flatten_leaf_csr(const ff_tree *tree, int64_t ns, const int64_t *leaf_ptr,
                        const int64_t *leaf_idx, const double *leaf_val, int leave_unnormalized,
                        ff_flat **flat, char *err, size_t errlen)
{
    if (!tree || !leaf_ptr || !flat || ns < 0 || (leaf_ptr[ns] > 0 && (!leaf_idx || !leaf_val)))
        return ff::fail(FF_ERR_ARG, err, errlen, "ff_flatten_leaf_csr: bad argument");
    unsigned hw = std::thread::hardware_concurrency();
    if (leave_unnormalized < 0 || leave_unnormalized > FF_L_REFERENCE)
        return ff::fail(FF_ERR_ARG, err, errlen, "ff_flatten_leaf_csr: leave_unnormalized must be 0, 1 or FF_L_REFERENCE");
    return flatten_impl(*tree, ns, leaf_ptr, leaf_idx, leaf_val, !leave_unnormalized, leave_unnormalized == FF_L_REFERENCE,
                        ff::clamp_threads(hw ? (int)hw : 1), flat, err, errlen);
}

int ff_flatten(const ff_table *tb, const ff_tree *tr, int leave_unnormalized, ff_flat **flat,
               char *err, size_t errlen)
{
    if (!tb || !tr || !flat) return ff::fail(FF_ERR_ARG, err, errlen, "ff_flatten: null argument");
    std::vector<int64_t> ptr;
    ff::I64Vec idx;
    ff::F64Vec val;
    ff::table_leaf_csr(*tb, *tr, &ptr, &idx, &val);
    return ff_flatten_leaf_csr(tr, (int64_t)ptr.size() - 1, ptr.data(), idx.data(), val.data(), leave_unnormalized,
                               flat, err, errlen);
}

void ff_flat_free(ff_flat *f) { delete f; }

void ff_flat_problem(const ff_flat *f, ff_problem *p)
{
    if (!f || !p) return;
    p->n_samples = f->n_samples;
    p->n_branches = f->n_branches;
    p->branch_len = f->branch_len.data();
    p->indptr = f->indptr.data();
    p->branch_id = f->branch_id.data();
    p->abnd = f->abnd.data();
}

int ff_unifrac(const ff_table *table, const ff_tree *tree, const ff_options *o,
               int leave_unnormalized, double *out, char *err, size_t errlen)
{
    if (!table || !tree) return ff::fail(FF_ERR_ARG, err, errlen, "ff_unifrac: null argument");
    std::vector<int64_t> ptr;
    ff::I64Vec idx;
    ff::F64Vec val;
    ff::table_leaf_csr(*table, *tree, &ptr, &idx, &val);
    return ff::unifrac_leaves_info(tree, (int64_t)ptr.size() - 1, ptr.data(), idx.data(), val.data(),
                                   leave_unnormalized, o, out, nullptr, err, errlen);
}

}  // extern "C"
