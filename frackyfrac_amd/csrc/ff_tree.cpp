// ff_tree.cpp -- Newick reader -> the flat tree the UniFrac path consumes.
//
// Replaces readTree (frcfrc/frcfrc.go:109-114) and the newick.Node fields the
// path touches: Name (unifrac.go:40,73), Distance (:119), Children (:35,39) and
// PreOrder (:72,129).  The reference's reader is github.com/fluhus/biostuff
// v1.0.0 (go.mod:6), not vendored; the grammar here is standard Newick:
//   tree   := subtree ';'
//   subtree:= '(' subtree (',' subtree)* ')' label? (':' length)?  |  label? (':' length)?
//   label  := unquoted run of characters other than ( ) , : ; [ ] and blanks,
//             or a single-quoted string with '' as the escaped quote
//   [ ... ] comments and blanks between tokens are skipped.
// Nodes are numbered in order of first appearance, which is pre-order
// (enumerateNodes, unifrac.go:127-133): root = 0, parent[id] < id.  The parser
// is iterative, so a 50k-leaf caterpillar does not overflow the stack.
#include <cmath>

#include "ff_host.hpp"

void ff_tree::index_names()
{
    leaf_ids.clear();
    all_names.clear();
    for (size_t k = 0; k < name.size(); ++k) {
        all_names.emplace(name[k], 1);
        if (size[k] == 1) leaf_ids[name[k]].push_back((int64_t)k);
    }
}

namespace {

struct Lexer {
    const char *p, *e, *base;
    void skip()
    {
        for (;;) {
            while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
            if (p < e && *p == '[') {
                int depth = 0;
                while (p < e) {
                    if (*p == '[') ++depth;
                    else if (*p == ']' && --depth == 0) {
                        ++p;
                        break;
                    }
                    ++p;
                }
                continue;
            }
            return;
        }
    }
    static bool is_delim(char c)
    {
        return c == '(' || c == ')' || c == ',' || c == ':' || c == ';' || c == '[' || c == ']' ||
               c == ' ' || c == '\t' || c == '\n' || c == '\r';
    }
    std::string label()
    {
        std::string s;
        if (p < e && *p == '\'') {
            ++p;
            while (p < e) {
                if (*p == '\'') {
                    if (p + 1 < e && p[1] == '\'') {
                        s += '\'';
                        p += 2;
                        continue;
                    }
                    ++p;
                    break;
                }
                s += *p++;
            }
            return s;
        }
        const char *b = p;
        while (p < e && !is_delim(*p)) ++p;
        s.assign(b, p);
        return s;
    }
};

}  // namespace

extern "C" {

int ff_tree_parse(const char *text, size_t len, ff_tree **out, char *err, size_t errlen)
{
    if (!text || !out) return ff::fail(FF_ERR_ARG, err, errlen, "ff_tree_parse: null argument");
    Lexer lx{text, text + len, text};
    lx.skip();
    if (lx.p >= lx.e)
        return ff::fail(FF_ERR_PARSE, err, errlen, "no tree in the given file");  // frcfrc.go:113
    auto *t = new ff_tree();
    std::vector<int64_t> open;  // ids of the internal nodes whose ')' is pending
    auto new_node = [&]() -> int64_t {
        int64_t id = (int64_t)t->name.size();
        t->name.emplace_back();
        t->dist.push_back(0.0);
        t->parent.push_back(open.empty() ? -1 : open.back());
        t->size.push_back(1);
        return id;
    };
    auto bad = [&](const char *what) {
        long off = (long)(lx.p - lx.base);
        delete t;
        return ff::fail(FF_ERR_PARSE, err, errlen, "newick: %s at offset %ld", what, off);
    };
    // reads label? (':' length)? for node `id`
    auto tail = [&](int64_t id) -> bool {
        lx.skip();
        if (lx.p < lx.e && !Lexer::is_delim(*lx.p)) t->name[(size_t)id] = lx.label();
        else if (lx.p < lx.e && *lx.p == '\'') t->name[(size_t)id] = lx.label();
        lx.skip();
        if (lx.p < lx.e && *lx.p == ':') {
            ++lx.p;
            lx.skip();
            const char *b = lx.p;
            while (lx.p < lx.e && !Lexer::is_delim(*lx.p)) ++lx.p;
            double v;
            const char *why;
            if (!ff::go_parse_float(b, lx.p, &v, &why)) return false;
            t->dist[(size_t)id] = v;
        }
        return true;
    };
    int64_t cur = -1;       // node whose tail was just read (awaiting ',' ')' or ';')
    bool expect_node = true;
    bool done = false;
    while (!done) {
        lx.skip();
        if (expect_node) {
            if (lx.p < lx.e && *lx.p == '(') {
                ++lx.p;
                int64_t id = new_node();
                open.push_back(id);
                continue;  // first child follows
            }
            int64_t id = new_node();  // a leaf (possibly unnamed)
            if (!tail(id)) return bad("bad branch length");
            cur = id;
            expect_node = false;
            continue;
        }
        if (lx.p >= lx.e) return bad("unexpected end of text, expected ';'");
        char c = *lx.p;
        if (c == ',') {
            if (open.empty()) return bad("',' outside parentheses");
            ++lx.p;
            expect_node = true;
        } else if (c == ')') {
            if (open.empty()) return bad("unbalanced ')'");
            ++lx.p;
            int64_t id = open.back();
            open.pop_back();
            if (!tail(id)) return bad("bad branch length");
            cur = id;
        } else if (c == ';') {
            if (!open.empty()) return bad("unbalanced '('");
            ++lx.p;
            done = true;
        } else {
            return bad("expected ',' ')' or ';'");
        }
    }
    (void)cur;
    for (int64_t i = (int64_t)t->name.size() - 1; i > 0; --i) t->size[(size_t)t->parent[(size_t)i]] += t->size[(size_t)i];
    t->index_names();
    *out = t;
    return FF_OK;
}

int ff_tree_read_file(const char *path, ff_tree **tree, char *err, size_t errlen)
{
    std::string text;
    if (!path) return ff::fail(FF_ERR_ARG, err, errlen, "please provide a tree file with -t");
    int rc = ff::read_all(path, &text, err, errlen);
    if (rc) return rc;
    return ff_tree_parse(text.data(), text.size(), tree, err, errlen);
}

void ff_tree_free(ff_tree *t) { delete t; }
int64_t ff_tree_num_nodes(const ff_tree *t) { return t ? (int64_t)t->name.size() : 0; }
const double *ff_tree_branch_len(const ff_tree *t) { return t->dist.data(); }
const int64_t *ff_tree_parent(const ff_tree *t) { return t->parent.data(); }
const int64_t *ff_tree_subtree_size(const ff_tree *t) { return t->size.data(); }
const char *ff_tree_name(const ff_tree *t, int64_t id)
{
    if (!t || id < 0 || id >= (int64_t)t->name.size()) return nullptr;
    return t->name[(size_t)id].c_str();
}

}  // extern "C"
