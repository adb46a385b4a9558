// Host side of the C ABI (include/cafe_mi355x.h): context, schedule, per-call enqueue.
//
// What of the reference this replaces, per scorer call:
//   base_model::infer_family_likelihoods   src/base_model.cpp:53-112
//   gamma_model::infer_family_likelihoods  src/gamma_core.cpp:169-246 (+ can_infer :123)
//   matrix_cache / matrix_cache_key        src/matrix_cache.h:42-61, src/matrix_cache.cpp:99-171
//   inference_prune                        src/core.cpp:133-144
// The tree is flattened once into a schedule of leaf-gather and GEMM launches (post-order,
// Sethi-Ullman child order so that few likelihood panels are live), families are de-duplicated
// once (build_reference_list, base_model.cpp:27) and stay resident on the device; a call uploads
// only the scalars that prepare_calculation changes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "cafe_ctx.h"

using namespace cafe;

namespace cafe {

void set_err(cafe_ctx* c, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    c->err = buf;
}

}  // namespace cafe

namespace {

// Sethi-Ullman style need: panels live while evaluating node v (leaves need none).
int panel_need(const cafe_ctx* c, int v, std::vector<int>& need) {
    std::vector<int> kid;
    for (int u : c->children[v])
        if (c->leaf_taxon[u] < 0) kid.push_back(panel_need(c, u, need));
    std::sort(kid.begin(), kid.end(), std::greater<int>());
    int n = (int)kid.size() + 1;
    for (size_t i = 0; i < kid.size(); ++i) n = std::max(n, (int)i + kid[i]);
    need[v] = n;
    return n;
}

struct PanelAlloc {
    bool reuse = true;          // false: every panel gets an id of its own (grouped schedule; the arena is planned afterwards)
    std::vector<int> free_list;
    int high = 0;
    int get() {
        if (reuse && !free_list.empty()) { int p = free_list.back(); free_list.pop_back(); return p; }
        return high++;
    }
    void put(int p) { if (reuse) free_list.push_back(p); }
};

int emit_node(cafe_ctx* c, int v, const std::vector<int>& need, PanelAlloc& pa) {
    std::vector<int> inner, leaves;
    for (int u : c->children[v]) (c->leaf_taxon[u] < 0 ? inner : leaves).push_back(u);
    std::vector<int> order = inner;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return need[x] > need[y]; });
    std::map<int, int> panel_of;
    for (size_t idx = 0; idx < order.size(); ++idx) panel_of[order[idx]] = emit_node(c, order[idx], need, pa);
    const int dst = pa.get();
    bool init = false;
    // A parent with interior children folds (up to kMaxLeafPerOp of) its leaf children into the epilogue of
    // the first GEMM; a parent with leaf children only (a cherry) is a pure gather.  Extra leaves gather-multiply.
    // (one leaf, without an error model or with a 3-tap one: the specialised epilogues of prune_gemm.hip)
    // A child with fewer distinct columns than its parent (subtree-level de-duplication) first gets its factor
    // P . L over ITS columns in a scratch panel, which a combine pass spreads over the parent's columns.
    const bool first_direct = !inner.empty() && (!c->subtree_dedup || c->edge_identity[inner[0]]);
    size_t fused = (!first_direct || leaves.empty() || (c->n_dev != 0 && c->n_dev != 3)) ? 0 : 1;
    std::vector<std::pair<int, int>> factors;             // (child, scratch panel) waiting to be assembled
    auto is_direct = [&](int u) { return !c->subtree_dedup || c->edge_identity[u]; };
    if (inner.size() == 2 && leaves.empty() && is_direct(inner[0]) != is_direct(inner[1])) {
        // one child shares the parent's columns, the other has fewer: the smaller one's factor GEMM runs first over ITS
        // columns, the other's GEMM then writes the parent's panel and multiplies the gathered factor in (the product
        // of two numbers: the same bits whichever child comes first)
        const int big = is_direct(inner[0]) ? inner[0] : inner[1], small = big == inner[0] ? inner[1] : inner[0];
        const int scratch = pa.get();
        Op f{};
        f.type = 1; f.parent = v; f.src_panel = panel_of[small]; f.child = small; f.to_root = (v == c->root);
        f.dst_panel = scratch; f.mode = 0; f.to_factor = true;
        c->ops.push_back(f);
        Op g{};
        g.type = 1; g.parent = v; g.src_panel = panel_of[big]; g.child = big; g.to_root = (v == c->root);
        g.dst_panel = dst; g.mode = 0; g.has_gath = true; g.gath_child = small; g.gath_panel = scratch;
        c->ops.push_back(g);
        pa.put(scratch);
        for (int u : inner) pa.put(panel_of[u]);
        return dst;
    }
    for (size_t gi = 0; gi < inner.size(); ++gi) {   // child order of the reference (probability.cpp:205 walks _descendants in order)
        const int u = inner[gi];
        const bool direct = is_direct(u);
        Op op{};
        op.type = 1;
        op.parent = v;
        op.src_panel = panel_of[u];
        op.child = u;
        op.to_root = (v == c->root);
        if (direct) {
            op.dst_panel = dst;
            op.mode = init ? 1 : 0;
            if (gi == 0) {
                op.n_leaf = (int)fused;
                for (size_t l = 0; l < fused; ++l) op.leaf_node[l] = leaves[l];
            }
            c->ops.push_back(op);
            init = true;
        } else {
            const int scratch = pa.get();
            op.dst_panel = scratch;
            op.mode = 0;
            op.to_factor = true;
            c->ops.push_back(op);
            factors.emplace_back(u, scratch);
        }
    }
    // assemble the parent's panel: up to two factor panels and two leaf children per pass, written once
    size_t li = fused, fi = 0;
    while (li < leaves.size() || fi < factors.size()) {
        Op op{};
        op.type = 0;
        op.parent = v;
        op.dst_panel = dst;
        op.to_root = (v == c->root);
        const size_t max_leaf = factors.empty() ? (size_t)kMaxLeafPerOp : 2;     // the fast kernel takes two of each
        op.n_leaf = (int)std::min<size_t>(max_leaf, leaves.size() - li);
        for (int l = 0; l < op.n_leaf; ++l) op.leaf_node[l] = leaves[li + l];
        li += op.n_leaf;
        op.n_src = (int)std::min<size_t>(2, factors.size() - fi);
        for (int j = 0; j < op.n_src; ++j) { op.src_child[j] = factors[fi + j].first; op.src_panels[j] = factors[fi + j].second; }
        fi += op.n_src;
        op.mode = init ? 1 : 0;
        c->ops.push_back(op);
        init = true;
    }
    for (auto& fs : factors) pa.put(fs.second);
    for (int u : inner) pa.put(panel_of[u]);
    return dst;
}

// Subtree-level de-duplication (host side, once): the distinct patterns of leaf counts under every interior node, the
// column of each child for every column of its parent, and the leaf children's counts per parent column.  Columns are
// numbered by first occurrence in (distinct-)family order, so the root's columns are the distinct families themselves
// and a child with as many patterns as its parent has them in the same order (an identity map: no combine pass).
int compute_patterns(cafe_ctx* c, const cafe_problem* p, const std::vector<int64_t>& uniq) {
    const int n = c->n_nodes, T = c->n_taxa;
    const int64_t F = c->F_uniq;
    c->pat_cols.assign(n, 0);
    c->edge_identity.assign(n, 0);
    c->d_edge_map.assign(n, nullptr);
    c->d_leaf_cnt.assign(n, nullptr);
    c->leaf_rank.assign(n, 0);
    // ---- 1. every interior node's own patterns, children first.  Patterns are numbered in the order of the node's
    // HEAVY child's pattern numbers (the interior child with the most patterns; ties and cherries: first occurrence in
    // family order), so that along the heavy path a parent's columns map to non-decreasing child columns: the
    // gathers of the assemble passes and of K2's gathered-factor epilogue then read the big factor panel in order.
    std::vector<std::vector<int32_t>> pid(n);            // [interior node][distinct family] own pattern index
    std::vector<std::vector<int64_t>> rep(n);            // [interior node][own pattern] first distinct family showing it
    for (int v = 0; v < n; ++v) {
        if (c->leaf_taxon[v] >= 0) continue;
        std::vector<int> inner, leaves;
        for (int u : c->children[v]) (c->leaf_taxon[u] < 0 ? inner : leaves).push_back(u);
        const size_t kw = inner.size() + leaves.size();
        pid[v].resize(F);
        if (v == c->root) {                              // the root keeps one column per family of the context (K4 reads them
            rep[v].resize(F);                            // by family index), also when identical families were kept apart
            for (int64_t f = 0; f < F; ++f) { pid[v][f] = (int32_t)f; rep[v][f] = f; }
            continue;
        }
        std::unordered_map<std::string, int32_t> seen;
        seen.reserve((size_t)F * 2);
        std::vector<int64_t> first;                      // raw pattern (first-occurrence number) -> first family
        std::vector<int32_t> key(kw);
        for (int64_t f = 0; f < F; ++f) {
            size_t k = 0;
            for (int u : inner) key[k++] = pid[u][f];
            for (int u : leaves) key[k++] = p->counts[uniq[f] * T + c->leaf_taxon[u]];
            std::string ks(reinterpret_cast<const char*>(key.data()), sizeof(int32_t) * kw);
            auto it = seen.find(ks);
            if (it == seen.end()) {
                it = seen.emplace(std::move(ks), (int32_t)first.size()).first;
                first.push_back(f);
            }
            pid[v][f] = it->second;
        }
        int heavy = -1;
        for (int u : inner) if (heavy < 0 || rep[u].size() > rep[heavy].size()) heavy = u;
        const size_t U = first.size();
        std::vector<int32_t> order(U), renum(U);
        for (size_t i = 0; i < U; ++i) order[i] = (int32_t)i;
        // largest leaf count under v, per pattern: the primary key (columns of similar size share a 128-column tile, whose
        // all-zero rows K2 skips); within equal sizes the heavy child's numbering
        std::vector<int32_t> big(U, 0), small(U, 0x7fffffff);
        {
            std::vector<int> under;                       // taxa under v
            std::vector<int> stack(1, v);
            while (!stack.empty()) {
                const int w = stack.back(); stack.pop_back();
                if (c->leaf_taxon[w] >= 0) under.push_back(c->leaf_taxon[w]);
                for (int u : c->children[w]) stack.push_back(u);
            }
            for (size_t i = 0; i < U; ++i)
                for (int t : under) {
                    const int32_t x = p->counts[uniq[first[i]] * T + t];
                    big[i] = std::max(big[i], x);
                    small[i] = std::min(small[i], x);
                }
        }
        // (the lower end of a column's non-zero rows follows its largest count, the upper end its smallest)
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
            if (big[x] != big[y]) return big[x] < big[y];
            if (small[x] != small[y]) return small[x] < small[y];
            return heavy >= 0 && pid[heavy][first[x]] < pid[heavy][first[y]];
        });
        rep[v].resize(U);
        for (size_t i = 0; i < U; ++i) { renum[order[i]] = (int32_t)i; rep[v][i] = first[order[i]]; }
        for (int64_t f = 0; f < F; ++f) pid[v][f] = renum[pid[v][f]];
    }
    // ---- 2. column space of every interior node, parents first: its own patterns, or its parent's columns, which makes
    // the edge direct (the GEMM's epilogue writes the parent's panel).  A GEMM column costs about 0.12 us, a column of an
    // assemble pass 0.024 us (one factor + leaf) to 0.036 us (two factors) at the bench shape, so:
    //  * all interior children inherit when together they add < 15 % GEMM columns (store / multiply epilogues, one leaf
    //    sibling fused);
    //  * of two interior children (no leaf sibling) the larger one inherits alone when it adds < 12 % (an assemble pass
    //    saved is worth about that many GEMM columns): the smaller one keeps its own columns and its factor is gathered
    //    in the larger one's epilogue -- no assemble pass either.
    std::vector<int> space(n, -1);
    space[c->root] = c->root;
    for (int v = n - 1; v >= 0; --v) {
        if (c->leaf_taxon[v] >= 0) continue;
        std::vector<int> inner;
        int n_leaves = 0;
        for (int u : c->children[v]) { if (c->leaf_taxon[u] < 0) inner.push_back(u); else ++n_leaves; }
        const double Uv = (double)rep[space[v]].size();
        double extra = 0;
        for (int u : inner) extra += 1.0 - (double)rep[u].size() / Uv;
        static const double thr_all = std::getenv("CAFE_INHERIT_ALL") ? std::atof(std::getenv("CAFE_INHERIT_ALL")) : 0.15;
        static const double thr_big = std::getenv("CAFE_INHERIT_BIG") ? std::atof(std::getenv("CAFE_INHERIT_BIG")) : 0.12;
        const bool inherit = !inner.empty() && n_leaves <= 1 && extra < thr_all;
        for (int u : inner) space[u] = inherit ? space[v] : u;
        if (!inherit && inner.size() == 2 && n_leaves == 0) {
            const int big = rep[inner[0]].size() >= rep[inner[1]].size() ? inner[0] : inner[1];
            if (1.0 - (double)rep[big].size() / Uv < thr_big) space[big] = space[v];
        }
    }
    // ---- 3. tables
    for (int v = 0; v < n; ++v) {
        if (c->leaf_taxon[v] >= 0) continue;
        std::vector<int> inner, leaves;
        for (int u : c->children[v]) (c->leaf_taxon[u] < 0 ? inner : leaves).push_back(u);
        const std::vector<int64_t>& cols = rep[space[v]];                // representative family of every column of v's panel
        const int64_t U = (int64_t)cols.size(), Up = round_up64(U, kBN);
        c->pat_cols[v] = Up;
        for (size_t l = 0; l < leaves.size(); ++l) c->leaf_rank[leaves[l]] = (int)l;
        if (!leaves.empty()) {
            std::vector<int32_t> tab(leaves.size() * (size_t)Up, 0);
            for (size_t l = 0; l < leaves.size(); ++l)
                for (int64_t u2 = 0; u2 < U; ++u2) tab[l * Up + u2] = p->counts[uniq[cols[u2]] * T + c->leaf_taxon[leaves[l]]];
            HIP_TRY(c, hipMalloc(&c->d_leaf_cnt[v], tab.size() * sizeof(int32_t)));
            HIP_TRY(c, hipMemcpy(c->d_leaf_cnt[v], tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        for (int u : inner) {
            if (space[u] == space[v]) { c->edge_identity[u] = 1; continue; }      // the child's columns ARE the parent's
            std::vector<int32_t> map((size_t)Up, 0);
            bool same = (int64_t)rep[u].size() == U;     // as many own patterns as the parent has columns, in the same order?
            for (int64_t u2 = 0; u2 < U; ++u2) { map[u2] = pid[u][cols[u2]]; same = same && map[u2] == (int32_t)u2; }
            if (same) { c->edge_identity[u] = 1; continue; }
            HIP_TRY(c, hipMalloc(&c->d_edge_map[u], map.size() * sizeof(int32_t)));
            HIP_TRY(c, hipMemcpy(c->d_edge_map[u], map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    return CAFE_OK;
}

// matrix_cache_key (matrix_cache.h:42-61)
inline void quantize(double lambda, double t, long* lq, long* tq) {
    *lq = long(lambda * 1000000000);
    *tq = long(t * 1000);
}

void free_device(cafe_ctx* c) {
    if (!c->device_ready) return;            // nothing was created on a device (argument / device errors)
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    comm_release(c);
    for (auto& g : c->graphs) if (g.second.exec) hipGraphExecDestroy(g.second.exec);
    hipFree(c->d_counts); hipFree(c->d_weights); hipFree(c->pool.base); hipFree(c->kpool.base); hipFree(c->kpool.ext); hipFree(c->pool.ext); hipFree(c->d_params); hipFree(c->d_panels);
    hipFree(c->d_ext_nodes);
    for (auto ptr : c->d_colext) hipFree(ptr);
    for (auto ptr : c->d_tileext) hipFree(ptr);
    hipFree(c->d_fam_out); hipFree(c->d_fam_lik); hipFree(c->d_cat_out); hipFree(c->d_failed);
    hipFree(c->d_scratch); hipFree(c->d_result); hipFree(c->d_stamps);
    for (auto ptr : c->d_edge_map) hipFree(ptr);
    for (auto ptr : c->d_leaf_cnt) hipFree(ptr);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->h_result) hipHostFree(c->h_result);
    if (c->h_ext) hipHostFree(c->h_ext);
    auto free_desc = [](DescSet& d) { hipFree(d.d_gemm_ops); hipFree(d.d_plan_desc); hipFree(d.d_plan); d = DescSet(); };
    free_desc(c->desc);
    for (auto& g : c->graphs) free_desc(g.second.desc);
    hipFree(c->d_gather_ops); hipFree(c->d_lt); hipFree(c->d_lt_pairs);
    if (c->h_gemm_stage) hipHostFree(c->h_gemm_stage);
    if (c->h_plan_desc) hipHostFree(c->h_plan_desc);
    if (c->ev_upload) hipEventDestroy(c->ev_upload);
    for (auto& e : c->ev) if (e) hipEventDestroy(e);
    for (auto& e : c->gemm_ev) hipEventDestroy(e);
    c->graphs.clear();
    if (c->stream) hipStreamDestroy(c->stream);
}

int create_impl(cafe_ctx* c, const cafe_problem* p) {
    const bool device_counts = p && (p->flags & kFlagDeviceCounts);      // internal: the caller fills d_counts on the device
    if (!p || p->n_nodes < 3 || !p->parent || !p->branch_length || !p->leaf_taxon || (!p->counts && !device_counts)) {
        set_err(c, "cafe_create: missing tree or family arrays");
        return CAFE_ERR_ARGUMENT;
    }
    if (p->n_families < 1 || p->n_taxa < 2 || p->max_family_size < 1 || p->max_root_family_size < 1) {
        set_err(c, "cafe_create: empty family table or non-positive max sizes");
        return CAFE_ERR_ARGUMENT;
    }
    c->n_nodes = p->n_nodes; c->n_taxa = p->n_taxa; c->M = p->max_family_size; c->R = p->max_root_family_size;
    c->N = std::max(c->M, c->R) + 1;                                   // base_model.cpp:77
    c->n_lambdas = std::max(1, p->n_lambdas); c->single_lambda = p->single_lambda;
    c->Kmax = std::max(1, p->max_categories); c->n_dev = p->n_deviations; c->device = p->device;
    if (c->Kmax > CAFE_MAX_CATEGORIES) { set_err(c, "cafe_create: more than %d gamma categories", CAFE_MAX_CATEGORIES); return CAFE_ERR_ARGUMENT; }
    c->parent.assign(p->parent, p->parent + p->n_nodes);
    c->blen.assign(p->branch_length, p->branch_length + p->n_nodes);
    c->leaf_taxon.assign(p->leaf_taxon, p->leaf_taxon + p->n_nodes);
    if (p->lambda_index) c->lam_idx.assign(p->lambda_index, p->lambda_index + p->n_nodes);
    else c->lam_idx.assign(p->n_nodes, 0);
    c->children.assign(p->n_nodes, {});
    for (int v = 0; v < p->n_nodes; ++v) {
        int par = c->parent[v];
        if (par < 0) {
            if (c->root >= 0) { set_err(c, "cafe_create: more than one root"); return CAFE_ERR_ARGUMENT; }
            c->root = v;
        } else if (par >= p->n_nodes || par <= v) {
            set_err(c, "cafe_create: node %d: parent %d must come after its children", v, par);
            return CAFE_ERR_ARGUMENT;
        } else {
            c->children[par].push_back(v);
        }
        if (c->lam_idx[v] < 0 || c->lam_idx[v] >= c->n_lambdas) { set_err(c, "cafe_create: lambda index out of range at node %d", v); return CAFE_ERR_ARGUMENT; }
    }
    if (c->root < 0) { set_err(c, "cafe_create: no root"); return CAFE_ERR_ARGUMENT; }
    for (int v = 0; v < p->n_nodes; ++v) {
        bool leaf = c->children[v].empty();
        if (leaf != (c->leaf_taxon[v] >= 0) || (leaf && c->leaf_taxon[v] >= c->n_taxa)) {
            set_err(c, "cafe_create: leaf_taxon inconsistent with the tree at node %d", v);
            return CAFE_ERR_ARGUMENT;
        }
    }
    if (c->children[c->root].empty()) { set_err(c, "cafe_create: the root is a leaf"); return CAFE_ERR_ARGUMENT; }
    if (c->N > bd_matrix_max_order()) { set_err(c, "cafe_create: matrix order %d exceeds %d", c->N, bd_matrix_max_order()); return CAFE_ERR_ARGUMENT; }

    // families: range check + de-duplication (build_reference_list, base_model.cpp:27-51)
    c->F_all = p->n_families;
    const int T = c->n_taxa;
    for (int64_t i = 0; !device_counts && i < c->F_all * T; ++i)
        if (p->counts[i] < 0 || p->counts[i] > c->M) {
            set_err(c, "cafe_create: family %lld has a count outside [0, %d]", (long long)(i / T), c->M);
            return CAFE_ERR_ARGUMENT;
        }
    c->ref_of.resize(c->F_all);
    std::vector<int64_t> uniq;                 // first occurrence of each distinct row
    if ((p->flags & CAFE_FLAG_NO_DEDUP) || device_counts) {
        uniq.resize(c->F_all);
        for (int64_t f = 0; f < c->F_all; ++f) { uniq[f] = f; c->ref_of[f] = f; }
        c->weights.assign(c->F_all, 1.0);
    } else {
        std::unordered_map<std::string, int64_t> seen;
        seen.reserve((size_t)c->F_all * 2);
        for (int64_t f = 0; f < c->F_all; ++f) {
            std::string key(reinterpret_cast<const char*>(p->counts + f * T), sizeof(int32_t) * T);
            auto it = seen.find(key);
            if (it == seen.end()) {
                seen.emplace(std::move(key), (int64_t)uniq.size());
                c->ref_of[f] = (int64_t)uniq.size();
                uniq.push_back(f);
                c->weights.push_back(1.0);
            } else {
                c->ref_of[f] = it->second;
                c->weights[it->second] += 1.0;
            }
        }
    }
    c->F_uniq = (int64_t)uniq.size();
    c->Fp = round_up64(c->F_uniq, kBN);
    // Columns in order of the families' largest count: a likelihood column is exactly zero far from the observed sizes
    // (node_extent_kernel), K2 skips the all-zero rows of a 128-column tile of its B operand, and a tile of families of
    // similar size has many of them.  Internal order only: ref_of maps every family of the table to its column.
    if (!device_counts) {
        std::vector<int32_t> key(c->F_uniq);
        for (int64_t u = 0; u < c->F_uniq; ++u) key[u] = *std::max_element(p->counts + uniq[u] * T, p->counts + (uniq[u] + 1) * T);
        std::vector<int64_t> perm(c->F_uniq), inv(c->F_uniq);
        for (int64_t u = 0; u < c->F_uniq; ++u) perm[u] = u;
        std::stable_sort(perm.begin(), perm.end(), [&](int64_t x, int64_t y) { return key[x] < key[y]; });
        std::vector<int64_t> nu(c->F_uniq);
        std::vector<double> nw(c->F_uniq);
        for (int64_t i = 0; i < c->F_uniq; ++i) { nu[i] = uniq[perm[i]]; nw[i] = c->weights[perm[i]]; inv[perm[i]] = i; }
        uniq.swap(nu);
        c->weights.swap(nw);
        for (int64_t f = 0; f < c->F_all; ++f) c->ref_of[f] = inv[c->ref_of[f]];
    }

    // device
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_err(c, "cafe_create: no HIP device available (this library has no CPU path)"); return CAFE_ERR_DEVICE; }
    if (c->device < 0 || c->device >= ndev) { set_err(c, "cafe_create: device %d out of range (%d devices)", c->device, ndev); return CAFE_ERR_DEVICE; }
    HIP_TRY(c, hipSetDevice(c->device));
    c->device_ready = true;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

    // counts, taxon-major, padded families replicate an all-zero family
    {
        std::vector<int32_t> tm((size_t)T * c->Fp, 0);
        for (int64_t u = 0; !device_counts && u < c->F_uniq; ++u)
            for (int t = 0; t < T; ++t) tm[(size_t)t * c->Fp + u] = p->counts[uniq[u] * T + t];
        HIP_TRY(c, hipMalloc(&c->d_counts, tm.size() * sizeof(int32_t)));
        HIP_TRY(c, hipMemcpy(c->d_counts, tm.data(), tm.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        std::vector<double> w(c->Fp, 0.0);
        std::copy(c->weights.begin(), c->weights.end(), w.begin());
        HIP_TRY(c, hipMalloc(&c->d_weights, w.size() * sizeof(double)));
        HIP_TRY(c, hipMemcpy(c->d_weights, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
    }

    // subtree-level de-duplication tables (the schedule depends on them)
    c->subtree_dedup = !device_counts && !(p->flags & CAFE_FLAG_NO_SUBTREE_DEDUP);
    if (c->subtree_dedup) { const int rc = compute_patterns(c, p, uniq); if (rc != CAFE_OK) return rc; }
    std::vector<int> need(c->n_nodes, 0);
    panel_need(c, c->root, need);

    // matrix pools: one slot per (distinct quantized branch length, lambda index) pair and category, per layout.
    // Leaf branches use row-major matrices (K3 gathers a column), interior branches k-major ones (K2's A).
    c->pair_of.assign(c->n_nodes, -1);
    {
        std::map<std::pair<long, int>, int> seen[2], any;
        for (int v = 0; v < c->n_nodes; ++v) {
            if (v == c->root) continue;
            const int layout = c->leaf_taxon[v] >= 0 ? 0 : 1;
            long lq, tq;
            quantize(0.0, c->blen[v], &lq, &tq);
            const auto key = std::make_pair(tq, c->lam_idx[v]);
            auto it = seen[layout].find(key);
            if (it == seen[layout].end()) {
                it = seen[layout].emplace(key, (int)c->pair_tq[layout].size()).first;
                c->pair_tq[layout].push_back(tq);
                c->pair_lam[layout].push_back(c->lam_idx[v]);
            }
            c->pair_of[v] = it->second;
            any.emplace(key, 0);
        }
        c->n_pairs[0] = (int)c->pair_tq[0].size();
        c->n_pairs[1] = (int)c->pair_tq[1].size();
        c->n_distinct_pairs = (int)any.size();
    }
    c->kc = round_up(c->M + 1, kBK);
    c->pool.n = c->N;
    c->pool.ld = round_up(c->N, 16);
    c->pool.stride = (int64_t)c->N * c->pool.ld;
    c->max_slots = c->n_pairs[0] * c->Kmax;
    c->kpool.n = c->N;
    c->kpool.rows = c->kc;
    c->kpool.k_valid = c->M + 1;
    c->kpool.kmajor = 1;
    c->kpool.ld = round_up(c->N - 1, 16) + round_up(kMaxBM, 16) + 16;     // a row tile may start at any valid row
    c->kpool.stride = (int64_t)c->kc * c->kpool.ld;
    c->kpool.ext_blocks = (c->N - 1 + 15) / 16;
    c->max_kslots = c->n_pairs[1] * c->Kmax;
    c->slot_of.assign((size_t)c->n_nodes * c->Kmax, -1);
    for (int v = 0; v < c->n_nodes; ++v) {
        if (v == c->root) continue;
        const int layout = c->leaf_taxon[v] >= 0 ? 0 : 1;
        for (int k = 0; k < c->Kmax; ++k) c->slot_of[(size_t)v * c->Kmax + k] = k * c->n_pairs[layout] + c->pair_of[v];
    }
    const size_t pool_bytes = (size_t)std::max(1, c->max_slots) * c->pool.stride * sizeof(double);
    const size_t kpool_bytes = (size_t)std::max(1, c->max_kslots) * c->kpool.stride * sizeof(double);
    if (hipMalloc(&c->pool.base, pool_bytes) != hipSuccess || hipMalloc(&c->kpool.base, kpool_bytes) != hipSuccess) {
        set_err(c, "cafe_create: cannot allocate %.2f GB for %d transition matrices of order %d", (pool_bytes + kpool_bytes) / 1e9,
                c->max_slots + c->max_kslots, c->N);
        return CAFE_ERR_MEMORY;
    }
    // padding columns / rows of both layouts are never written by K1 and must read as 0
    HIP_TRY(c, hipMemset(c->pool.base, 0, pool_bytes));
    HIP_TRY(c, hipMemset(c->kpool.base, 0, kpool_bytes));
    // non-zero extents of the k-major matrices (K1 writes them, K2 skips the K tiles outside them)
    HIP_TRY(c, hipMalloc(&c->kpool.ext, sizeof(int32_t) * 2 * (size_t)std::max(1, c->max_kslots) * c->kpool.ext_blocks));
    HIP_TRY(c, hipMemset(c->kpool.ext, 0, sizeof(int32_t) * 2 * (size_t)std::max(1, c->max_kslots) * c->kpool.ext_blocks));
    c->pool.ext_blocks = c->N;               // row-major: one entry per column x of a leaf branch's matrix
    HIP_TRY(c, hipMalloc(&c->pool.ext, sizeof(int32_t) * 2 * (size_t)std::max(1, c->max_slots) * c->pool.ext_blocks));
    HIP_TRY(c, hipMemset(c->pool.ext, 0, sizeof(int32_t) * 2 * (size_t)std::max(1, c->max_slots) * c->pool.ext_blocks));
    // diagnostic CAFE_NO_KSKIP: every K tile of every launch.  Small matrices (mammals: N = 141, 9 K tiles): a row tile spans
    // most of the band anyway, and the extent kernels and lookups cost more than the few K tiles they would save (measured:
    // 0.34 -> 0.38 ms per call with them) -- no extents below N = 256 unless CAFE_FORCE_KSKIP asks for them
    if (std::getenv("CAFE_NO_KSKIP") || (c->N < 256 && !std::getenv("CAFE_FORCE_KSKIP"))) {
        (void)hipFree(c->kpool.ext); c->kpool.ext = nullptr;
        (void)hipFree(c->pool.ext); c->pool.ext = nullptr;
    }
    if (c->kpool.ext && c->N >= 256)         // (small matrices: one row tile spans most of the band anyway, and a copy per call is not free)
        HIP_TRY(c, hipHostMalloc(&c->h_ext, sizeof(int32_t) * 2 * (size_t)std::max(1, c->max_kslots) * c->kpool.ext_blocks, hipHostMallocDefault));
    c->stats.matrix_bytes = (int64_t)(pool_bytes + kpool_bytes);

    // per-call parameter block (layout: cafe_ctx.h), device + pinned mirror
    {
        size_t off = sizeof(SlotParam) * (size_t)(c->max_slots + c->max_kslots);
        off = (off + 63) / 64 * 64;
        const size_t off_prior = off; off += sizeof(double) * c->R;
        const size_t off_logprior = off; off += sizeof(double) * c->R;
        const size_t off_cat = off; off += sizeof(double) * c->Kmax;
        const size_t off_err = off; off += sizeof(double) * (size_t)(c->M + 1) * std::max(1, c->n_dev);
        c->params_bytes = off;
        HIP_TRY(c, hipMalloc(&c->d_params, c->params_bytes));
        HIP_TRY(c, hipMemset(c->d_params, 0, c->params_bytes));
        c->d_slots = reinterpret_cast<SlotParam*>(c->d_params);
        c->d_prior = reinterpret_cast<double*>(c->d_params + off_prior);
        c->d_logprior = reinterpret_cast<double*>(c->d_params + off_logprior);
        c->d_catprobs = reinterpret_cast<double*>(c->d_params + off_cat);
        c->d_err = c->n_dev > 0 ? reinterpret_cast<double*>(c->d_params + off_err) : nullptr;
        c->stage_bytes = c->params_bytes;
        HIP_TRY(c, hipHostMalloc(&c->h_stage, c->stage_bytes, hipHostMallocDefault));
        std::memset(c->h_stage, 0, c->stage_bytes);
    }
    HIP_TRY(c, hipHostMalloc(&c->h_result, 4 * sizeof(double), hipHostMallocDefault));
    c->h_poison = c->h_result + 2;                       // what a failing rank of a communicator feeds the all-reduce
    c->h_poison[0] = 0.0;
    c->h_poison[1] = std::numeric_limits<double>::quiet_NaN();
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
    for (auto& e : c->ev) HIP_TRY(c, hipEventCreate(&e));
    {
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
        c->n_cu = prop.multiProcessorCount;
        const char* sl = std::getenv("CAFE_GEMM_STAMPS_LAUNCH");       // diagnostics: read once, never on the call path
        c->stamps_launch = sl ? std::atol(sl) : -1;
        if (std::getenv("CAFE_USE_GRAPH")) c->use_graph = 1;
        // small matrices (a K2 launch is one round of tiles and lasts as long as one tile): 16-deep K tiles, half as many DMA
        // round trips per tile; otherwise 8-deep ones, four workgroups per CU
        c->kb = c->N < 256 ? 16 : 8;
        if (const char* e = std::getenv("CAFE_KB")) c->kb = std::atoi(e) == 16 ? 16 : 8;
        const char* fm = std::getenv("CAFE_FORCE_TILE");             // diagnostic, like cafe_debug_force_tile
        if (fm && std::atoi(fm) >= 2 && std::atoi(fm) <= 9) c->force_mi = std::atoi(fm);
    }

    // outputs
    HIP_TRY(c, hipMalloc(&c->d_fam_out, sizeof(double) * c->Fp));
    HIP_TRY(c, hipMalloc(&c->d_fam_lik, sizeof(double) * c->Fp));
    HIP_TRY(c, hipMalloc(&c->d_cat_out, sizeof(double) * c->Fp * c->Kmax));
    HIP_TRY(c, hipMalloc(&c->d_failed, sizeof(int32_t) * c->Fp));
    HIP_TRY(c, hipMemset(c->d_failed, 0, sizeof(int32_t) * c->Fp));
    HIP_TRY(c, hipMalloc(&c->d_scratch, sizeof(double) * (2 * c->n_scratch + 1)));      // partials + the ticket counter of the final sum
    HIP_TRY(c, hipMemset(c->d_scratch, 0, sizeof(double) * (2 * c->n_scratch + 1)));
    HIP_TRY(c, hipMalloc(&c->d_result, sizeof(double) * 2));
    if (std::getenv("CAFE_GEMM_STAMPS")) {
        c->stamps_words = (size_t)6 * 8 * ((c->Fp / kBN + 8) * 16) * c->Kmax;
        HIP_TRY(c, hipMalloc(&c->d_stamps, c->stamps_words * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemset(c->d_stamps, 0, c->stamps_words * sizeof(unsigned long long)));
    }

    // likelihood panels: rows padded so that every panel can be a GEMM B operand (kc rows) or the root (R rows)
    // a factor GEMM stores transposed, [column][16 - out_off + panel row] (prune_gemm.hip): factor_ld rows per column, and a
    // panel slot must be able to hold a factor of as many columns
    c->factor_ld = round_up(std::max(c->M + 1, c->R) + 16, 16);
    c->rows_pad = std::max(std::max(c->kc, round_up(c->R, kBK)), c->factor_ld);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
    c->workspace_limit = p->workspace_limit;
    const size_t budget = p->workspace_limit ? p->workspace_limit : (size_t)(free_b * 0.80);
    // K2 addresses a panel category through a 32-bit buffer descriptor: rows_pad * cols * 8 bytes must stay below 4 GB
    const int64_t desc_cols = (int64_t)(0xFFFFFFF0ll / ((int64_t)c->rows_pad * 8)) / kBN * kBN;

    // ---- the schedule.  Preferred (the table fits one column chunk with a place of its own for every panel): GROUPED --
    // the ops are levelled by their dependencies into steps, a step's ops of one kernel variant share a launch, and the
    // arena is planned from the panels' lifetimes.  Otherwise: one op per launch in post-order with the Sethi-Ullman slot
    // pool (few live panels), in as many column chunks as the workspace asks for.
    auto cols_of = [&](int v) -> int64_t { return c->subtree_dedup ? c->pat_cols[v] : c->Fp; };
    size_t panel_doubles = 0;
    c->grouped = !std::getenv("CAFE_NO_GROUPS");
    if (c->grouped) {
        PanelAlloc pa;
        pa.reuse = false;
        c->ops.clear();
        c->root_panel = emit_node(c, c->root, need, pa);
        c->panels.assign(pa.high, Panel());
        // what each panel is: the transposed factor of a child (its own columns) or a node's panel
        for (const Op& op : c->ops) {
            Panel& P = c->panels[op.dst_panel];
            P.factor = op.type == 1 && op.to_factor;
            P.cols = P.factor ? cols_of(op.child) : cols_of(op.parent);
            P.kstride = P.cols * (P.factor ? c->factor_ld : c->rows_pad);
        }
        // steps: an op runs one step after the last op it depends on -- the writers of what it reads (its children's panels,
        // gathered factors, and its own destination when it multiplies)
        {
            std::vector<int> last_writer(c->panels.size(), -1);
            int n_steps = 0;
            for (size_t i = 0; i < c->ops.size(); ++i) {
                Op& op = c->ops[i];
                int st = 0;
                auto dep = [&](int panel) { if (last_writer[panel] >= 0) st = std::max(st, c->ops[last_writer[panel]].step + 1); };
                if (op.type == 1) { dep(op.src_panel); if (op.has_gath) dep(op.gath_panel); }
                for (int j = 0; j < op.n_src; ++j) dep(op.src_panels[j]);
                dep(op.dst_panel);                           // (a store is the first writer: no-op; a multiply follows the store)
                op.step = st;
                last_writer[op.dst_panel] = (int)i;
                n_steps = std::max(n_steps, st + 1);
            }
            for (Panel& P : c->panels) { P.first_step = 0x7fffffff; P.last_step = -1; }
            for (const Op& op : c->ops) {
                auto use = [&](int panel) { Panel& P = c->panels[panel]; P.first_step = std::min(P.first_step, op.step); P.last_step = std::max(P.last_step, op.step); };
                use(op.dst_panel);
                if (op.type == 1) { use(op.src_panel); if (op.has_gath) use(op.gath_panel); }
                for (int j = 0; j < op.n_src; ++j) use(op.src_panels[j]);
            }
            c->panels[c->root_panel].last_step = n_steps;    // K4 and cafe_get_root_likelihoods read it after the last step
            // arena: first fit over the steps; a panel's place is free again after the step that reads it last
            std::vector<std::pair<int64_t, int64_t>> holes;  // (offset, length) sorted by offset
            int64_t top = 0;
            std::vector<std::vector<int>> born(n_steps + 1), dies(n_steps + 1);
            for (size_t i = 0; i < c->panels.size(); ++i) { born[c->panels[i].first_step].push_back((int)i); dies[c->panels[i].last_step].push_back((int)i); }
            for (int st = 0; st <= n_steps; ++st) {
                std::sort(born[st].begin(), born[st].end(), [&](int x, int y) { return c->panels[x].kstride > c->panels[y].kstride; });
                for (int id : born[st]) {
                    Panel& P = c->panels[id];
                    const int64_t len = round_up64(P.kstride * c->Kmax, 64);          // 512-byte granules
                    bool placed = false;
                    for (size_t h = 0; h < holes.size() && !placed; ++h)
                        if (holes[h].second >= len) {
                            P.offset = holes[h].first;
                            holes[h].first += len; holes[h].second -= len;
                            if (holes[h].second == 0) holes.erase(holes.begin() + h);
                            placed = true;
                        }
                    if (!placed) {
                        if (!holes.empty() && holes.back().first + holes.back().second == top) {     // grow the hole at the top
                            P.offset = holes.back().first;
                            top = P.offset + len;
                            holes.pop_back();
                        } else {
                            P.offset = top;
                            top += len;
                        }
                    }
                }
                for (int id : dies[st]) {
                    const Panel& P = c->panels[id];
                    const int64_t len = round_up64(P.kstride * c->Kmax, 64);
                    auto it = std::lower_bound(holes.begin(), holes.end(), std::make_pair(P.offset, (int64_t)0));
                    it = holes.insert(it, std::make_pair(P.offset, len));
                    if (it + 1 != holes.end() && it->first + it->second == (it + 1)->first) { it->second += (it + 1)->second; holes.erase(it + 1); }
                    if (it != holes.begin() && (it - 1)->first + (it - 1)->second == it->first) { (it - 1)->second += it->second; holes.erase(it); }
                }
            }
            panel_doubles = (size_t)top;
        }
        int64_t widest = 0;
        for (const Panel& P : c->panels) widest = std::max(widest, P.cols);
        // (a place of its own for every panel takes several times the slot pool: not when that is more than half the workspace)
        if (panel_doubles * sizeof(double) + 65536 > budget / 2 || widest > desc_cols) c->grouped = false;
    }
    if (c->grouped) {
        c->chunk_cols = c->Fp;
        c->n_panels = (int)c->panels.size();
        c->panel_kstride = (int64_t)c->rows_pad * c->Fp;     // (the root panel's, what K4 reads)
        c->panel_stride = 0;
    } else {
        PanelAlloc pa;
        c->ops.clear();
        c->root_panel = emit_node(c, c->root, need, pa);
        c->n_panels = pa.high;
        size_t per_col = (size_t)c->n_panels * c->Kmax * c->rows_pad * sizeof(double);
        int64_t cols = std::min<int64_t>((int64_t)(budget / per_col) / kBN * kBN, desc_cols);
        if (c->subtree_dedup && cols < c->Fp) {
            // several column chunks: the per-node column maps address whole panels, so this case keeps one column per
            // family in every panel (the schedule without combine passes needs no more panels than the one with them)
            c->subtree_dedup = false;
            c->ops.clear();
            PanelAlloc pb;
            c->root_panel = emit_node(c, c->root, need, pb);
            c->n_panels = pb.high;
            per_col = (size_t)c->n_panels * c->Kmax * c->rows_pad * sizeof(double);
            cols = std::min<int64_t>((int64_t)(budget / per_col) / kBN * kBN, desc_cols);
        }
        if (cols < kBN) { set_err(c, "cafe_create: %zu bytes of workspace cannot hold %d panels of one 128-family tile", budget, c->n_panels); return CAFE_ERR_MEMORY; }
        c->chunk_cols = std::min<int64_t>(cols, c->Fp);
        c->panel_kstride = (int64_t)c->rows_pad * c->chunk_cols;
        c->panel_stride = c->panel_kstride * c->Kmax;
        c->panels.assign(c->n_panels, Panel());
        for (int i = 0; i < c->n_panels; ++i) {              // slots as wide as the widest panel; a node uses a prefix with its own leading dimension
            c->panels[i].cols = c->chunk_cols;
            c->panels[i].offset = (int64_t)i * c->panel_stride;
            c->panels[i].kstride = c->panel_kstride;
        }
        for (size_t i = 0; i < c->ops.size(); ++i) c->ops[i].step = (int)i;
        panel_doubles = (size_t)c->n_panels * c->panel_stride;
    }
    // (+64 KB: the assemble pass reads whole 64-row tiles of a transposed factor, up to a tile past its last column)
    const size_t panel_bytes = panel_doubles * sizeof(double) + 65536;
    if (hipMalloc(&c->d_panels, panel_bytes) != hipSuccess) {
        set_err(c, "cafe_create: cannot allocate %.2f GB of likelihood panels", panel_bytes / 1e9);
        return CAFE_ERR_MEMORY;
    }
    // rows beyond what the first writer of a panel covers must not hold NaN bit patterns
    HIP_TRY(c, hipMemset(c->d_panels, 0, panel_bytes));
    c->stats.panel_bytes = (int64_t)panel_bytes;
    c->stats.n_unique_families = c->F_uniq;
    c->stats.n_chunks = (c->Fp + c->chunk_cols - 1) / c->chunk_cols;

    // zero extents of the panels: descriptors of the interior non-root nodes, children before parents, level by level
    c->d_colext.assign(c->n_nodes, nullptr);
    c->d_tileext.assign(c->n_nodes, nullptr);
    // (one column per family at every node -- CAFE_FLAG_NO_SUBTREE_DEDUP, device-written counts -- works the same way as long
    // as the families fit one column chunk: every edge is the identity and the counts are the family table itself)
    c->panel_extents = (c->subtree_dedup || c->stats.n_chunks == 1) && c->kpool.ext && c->pool.ext && !std::getenv("CAFE_NO_PANEL_EXTENTS");
    c->no_asm_skip = std::getenv("CAFE_NO_ASM_SKIP") != nullptr;
    if (c->panel_extents) {
        std::vector<int> level(c->n_nodes, -1);
        int max_level = -1;
        for (int v = 0; v < c->n_nodes && c->panel_extents; ++v) {
            if (c->leaf_taxon[v] >= 0 || v == c->root) continue;
            int lv = 0, n_leaf = 0, n_inner = 0;
            for (int u : c->children[v]) {
                if (c->leaf_taxon[u] >= 0) ++n_leaf; else { ++n_inner; lv = std::max(lv, level[u] + 1); }
            }
            if (n_leaf > kMaxExtChildren || n_inner > kMaxExtChildren) c->panel_extents = false;   // (a wide polytomy: no extents)
            level[v] = lv;
            max_level = std::max(max_level, lv);
        }
        if (c->panel_extents) {
            std::vector<ExtNode> nodes;
            for (int lv = 0; lv <= max_level; ++lv) {
                cafe_ctx::ExtLevel L{(int)nodes.size(), 0, 0};
                for (int v = 0; v < c->n_nodes; ++v) {
                    if (level[v] != lv) continue;
                    ExtNode nd{};
                    nd.cols = (int32_t)(c->subtree_dedup ? c->pat_cols[v] : c->Fp);
                    HIP_TRY(c, hipMalloc(&c->d_colext[v], sizeof(int32_t) * 2 * (size_t)c->Kmax * nd.cols));
                    HIP_TRY(c, hipMalloc(&c->d_tileext[v], sizeof(int32_t) * 2 * (size_t)c->Kmax * (nd.cols / kBN)));
                    nd.colext = c->d_colext[v];
                    nd.tileext = c->d_tileext[v];
                    nd.cnt = c->subtree_dedup ? c->d_leaf_cnt[v] : c->d_counts;
                    nd.cnt_ld = c->subtree_dedup ? c->pat_cols[v] : c->Fp;
                    for (int u : c->children[v]) {
                        if (c->leaf_taxon[u] >= 0) {
                            nd.leaf_pair[nd.n_leaf] = c->pair_of[u];
                            nd.leaf_row[nd.n_leaf] = c->subtree_dedup ? c->leaf_rank[u] : c->leaf_taxon[u];
                            ++nd.n_leaf;
                        } else {
                            nd.inner_pair[nd.n_inner] = c->pair_of[u];
                            nd.inner_cols[nd.n_inner] = (int32_t)(c->subtree_dedup ? c->pat_cols[u] : c->Fp);
                            nd.inner_map[nd.n_inner] = (!c->subtree_dedup || c->edge_identity[u]) ? nullptr : c->d_edge_map[u];
                            nd.inner_colext[nd.n_inner] = c->d_colext[u];
                            ++nd.n_inner;
                        }
                    }
                    L.max_col_tiles = std::max(L.max_col_tiles, nd.cols / kBN);
                    nodes.push_back(nd);
                    ++L.count;
                }
                if (L.count) c->ext_levels.push_back(L);
            }
            HIP_TRY(c, hipMalloc(&c->d_ext_nodes, sizeof(ExtNode) * std::max<size_t>(1, nodes.size())));
            HIP_TRY(c, hipMemcpy(c->d_ext_nodes, nodes.data(), sizeof(ExtNode) * nodes.size(), hipMemcpyHostToDevice));
        }
    }


    // ---- launches: the ops of a step that share a kernel variant go out together
    if (const char* e = std::getenv("CAFE_PLAN_FIXED")) c->plan_fixed = std::max(0, atoi(e));
    if (const char* e = std::getenv("CAFE_PLAN_BIAS")) c->plan_bias = std::min(50, std::max(0, atoi(e)));
    if (const char* e = std::getenv("CAFE_PLAN_BIAS4")) {
        int v[4];
        if (std::sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0)
            for (int i = 0; i < 4; ++i) c->plan_bias4[i] = v[i];
    }
    if (const char* e = std::getenv("CAFE_PLAN_BIAS3")) {
        int a = 100, b = 100, d = 100;
        if (std::sscanf(e, "%d,%d,%d", &a, &b, &d) == 3 && a > 0 && b > 0 && d > 0) { c->plan_bias3[0] = a; c->plan_bias3[1] = b; c->plan_bias3[2] = d; }
    }
    // Leaf branches whose matrix an assemble pass multiplies with a factor get a transposed copy (leaf_transpose_kernel, every
    // call): the pass then reads the leaf's column as lines, like the factor's, instead of 8 bytes per matrix row (4.0 -> 6 TB/s).
    // A copy costs 16 N^2 bytes per category and call whatever the number of columns, so a branch gets one only when the
    // passes that read it write enough columns: >= lt_min N.  Bench table: 31 branches, 139.5 -> 137.9 ms per call; its 1/8
    // shards copy 2 to 6 branches and take what they took (20.1 / 20.2 ms; with all 30 copied: +0.3 to +0.6 ms).
    std::vector<int> lt_of_pair(std::max(1, c->n_pairs[0]), -1);
    if (!std::getenv("CAFE_NO_LEAF_T")) {
        double lt_min = 6.0;
        if (const char* e = std::getenv("CAFE_LEAF_T_MIN")) lt_min = atof(e);
        std::vector<int64_t> served(lt_of_pair.size(), 0);
        auto eligible = [](const Op& op) { return op.type == 0 && op.n_src >= 1 && op.n_src <= 2 && op.n_leaf >= 1 && op.n_leaf <= 2; };
        for (const Op& op : c->ops)
            if (eligible(op))
                for (int l = 0; l < op.n_leaf; ++l) served[c->pair_of[op.leaf_node[l]]] += cols_of(op.parent) * std::max<int64_t>(1, c->stats.n_chunks);
        for (size_t pr = 0; pr < served.size(); ++pr)
            if (served[pr] > 0 && (double)served[pr] >= lt_min * c->N) { lt_of_pair[pr] = (int)c->lt_pairs.size(); c->lt_pairs.push_back((int)pr); }
        const size_t lt_bytes = sizeof(double) * ((size_t)c->lt_pairs.size() * c->Kmax * (size_t)(c->M + 1) * c->factor_ld + 2 * kBN);
        size_t free_now = 0, total_now = 0;
        HIP_TRY(c, hipMemGetInfo(&free_now, &total_now));
        const bool fits = !c->lt_pairs.empty() && (size_t)c->lt_pairs.size() * c->Kmax <= 65535u && lt_bytes <= free_now / 4 &&
                          (!c->workspace_limit || lt_bytes <= c->workspace_limit / 8);
        if (fits) {
            HIP_TRY(c, hipMalloc(&c->d_lt, lt_bytes));
            HIP_TRY(c, hipMemset(c->d_lt, 0, lt_bytes));
            HIP_TRY(c, hipMalloc(&c->d_lt_pairs, sizeof(int32_t) * c->lt_pairs.size()));
            HIP_TRY(c, hipMemcpy(c->d_lt_pairs, c->lt_pairs.data(), sizeof(int32_t) * c->lt_pairs.size(), hipMemcpyHostToDevice));
            for (Op& op : c->ops) {
                if (!eligible(op)) continue;
                op.leaf_t = true;
                for (int l = 0; l < op.n_leaf; ++l) op.leaf_t = op.leaf_t && lt_of_pair[c->pair_of[op.leaf_node[l]]] >= 0;
            }
        } else {
            c->lt_pairs.clear();
        }
    }
    {
        std::vector<size_t> idx(c->ops.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
        auto key = [&](const Op& o) -> int {                 // launch order inside a step: factor GEMMs, the other GEMMs, then K3
            if (o.type == 1) return o.to_factor ? 0 : 1 + (o.has_gath ? 2 : (o.n_leaf ? 1 : 0)) * 2 + o.mode;
            return 16 + o.n_src * 32 + o.n_leaf * 2 + o.mode + (o.leaf_t ? 1024 : 0);
        };
        std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) {
            const Op &a = c->ops[x], &b = c->ops[y];
            if (a.step != b.step) return a.step < b.step;
            if (a.to_root != b.to_root) return b.to_root;
            return key(a) < key(b);
        });
        for (size_t i : idx) {
            Op& o = c->ops[i];
            const bool fresh = c->groups.empty() || c->groups.back().step != o.step || c->groups.back().type != o.type ||
                               key(c->ops[c->groups.back().ops[0]]) != key(o) || c->groups.back().to_root != o.to_root ||
                               (int)c->groups.back().ops.size() >= (o.type == 1 ? kMaxGroupOps : 512);
            if (fresh) {
                Group g;
                g.type = o.type; g.step = o.step; g.to_root = o.to_root;
                g.first_desc = o.type == 1 ? c->n_gemm_ops : c->n_gather_ops;
                if (o.type == 1) g.variant = GemmVariant{o.mode, o.has_gath ? 2 : (o.n_leaf ? 1 : 0), o.to_factor ? 1 : 0};
                c->groups.push_back(g);
                c->n_gemm_groups += o.type == 1;
            }
            o.desc = o.type == 1 ? c->n_gemm_ops++ : c->n_gather_ops++;
            c->groups.back().ops.push_back((int)i);
        }
    }
    // static descriptors (n_row_tiles of a K2 op follows the tile height, chosen per call)
    c->h_gemm_ops.assign(std::max(1, c->n_gemm_ops), GemmOp{});
    c->h_gather_ops.assign(std::max(1, c->n_gather_ops), GatherArgs{});
    const int64_t lt_kstride = (int64_t)(c->M + 1) * c->factor_ld;
    for (const Op& op : c->ops) {
        const int32_t* cnt_base = c->subtree_dedup ? c->d_leaf_cnt[op.parent] : c->d_counts;
        const int64_t cnt_ld = c->subtree_dedup ? c->pat_cols[op.parent] : c->Fp;
        auto cnt_row = [&](int leaf) { return c->subtree_dedup ? c->leaf_rank[leaf] : c->leaf_taxon[leaf]; };
        const Panel& D = c->panels[op.dst_panel];
        if (op.type == 1) {
            GemmOp& g = c->h_gemm_ops[op.desc];
            const Panel& S = c->panels[op.src_panel];
            for (int k = 0; k < c->Kmax; ++k) g.slot[k] = c->slot_of[(size_t)op.child * c->Kmax + k];
            g.src = c->d_panels + S.offset; g.src_kstride = S.kstride;
            g.dst = c->d_panels + D.offset; g.dst_kstride = D.kstride;
            g.ld = (int32_t)cols_of(op.child);               // the GEMM runs over the child's columns (= the parent's when direct)
            g.n_col_tiles = g.ld / kBN;
            g.rows = op.to_root ? c->R : c->M;               // parent sizes 1..rows
            g.out_off = op.to_root ? 0 : 1;
            g.dst_ldt = op.to_factor ? c->factor_ld : 0;
            g.n_leaf = op.n_leaf;
            if (op.n_leaf) {
                g.taxon = cnt_row(op.leaf_node[0]);
                for (int k = 0; k < c->Kmax; ++k) g.leaf_slot[k] = c->slot_of[(size_t)op.leaf_node[0] * c->Kmax + k];
            }
            g.counts = cnt_base; g.counts_ld = cnt_ld;
            if (op.has_gath) {
                const Panel& G = c->panels[op.gath_panel];
                g.gath_src = c->d_panels + G.offset; g.gath_kstride = G.kstride;
                g.gath_ld = c->factor_ld;
                g.gath_map = c->d_edge_map[op.gath_child];
            }
            g.bext = c->panel_extents ? c->d_tileext[op.child] : nullptr;
        } else {
            GatherArgs& g = c->h_gather_ops[op.desc];
            g.n_leaf = op.n_leaf;
            for (int l = 0; l < op.n_leaf; ++l) {
                g.taxon[l] = cnt_row(op.leaf_node[l]);
                for (int k = 0; k < c->Kmax; ++k) g.slot[l][k] = c->slot_of[(size_t)op.leaf_node[l] * c->Kmax + k];
            }
            g.counts = cnt_base; g.counts_ld = cnt_ld;
            g.dst = c->d_panels + D.offset; g.panel_kstride = D.kstride; g.ld = (int32_t)cols_of(op.parent);
            g.row_off = op.to_root ? 1 : 0;
            g.rows = op.to_root ? c->R : c->M + 1;
            g.rows_store = op.to_root ? c->R : c->kc;
            g.mode = op.mode;
            g.n_src = op.n_src;
            for (int j = 0; j < op.n_src; ++j) {
                const Panel& S = c->panels[op.src_panels[j]];
                g.src[j] = c->d_panels + S.offset; g.kstride_src[j] = S.kstride;
                g.ld_src[j] = c->factor_ld;
                g.map[j] = c->d_edge_map[op.src_child[j]];
            }
            // (the root's vector is read whole by the reduction and has no extent record)
            g.tileext = c->panel_extents && !op.to_root && !c->no_asm_skip ? c->d_tileext[op.parent] : nullptr;
            g.lt_kstride = lt_kstride;
            if (op.leaf_t)
                for (int l = 0; l < op.n_leaf && l < kMaxLeafPerOp; ++l)
                    g.lt[l] = c->d_lt + (int64_t)lt_of_pair[c->pair_of[op.leaf_node[l]]] * c->Kmax * lt_kstride;
        }
    }
    HIP_TRY(c, hipMalloc(&c->d_gather_ops, sizeof(GatherArgs) * c->h_gather_ops.size()));
    HIP_TRY(c, hipMemcpy(c->d_gather_ops, c->h_gather_ops.data(), sizeof(GatherArgs) * c->h_gather_ops.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipHostMalloc(&c->h_gemm_stage, sizeof(GemmOp) * c->h_gemm_ops.size(), hipHostMallocDefault));
    // tile lists of the K2 launches: room for the tallest list any tile height can ask for
    // (the planner is one 64-lane wave per XCD, a lane per workgroup: MI355X has 32 CUs x 2 workgroups per XCD)
    {
        size_t entries = 0;
        for (const Group& g : c->groups) {
            if (g.type != 1) continue;
            size_t worst = 0;
            for (int mi = 2; mi <= 9; ++mi) {
                int64_t tiles = 0;
                for (int oi : g.ops) {
                    const Op& op = c->ops[oi];
                    const int rows = op.to_root ? c->R : c->M;
                    const int64_t gc = c->subtree_dedup ? c->pat_cols[op.child] : c->chunk_cols;
                    tiles += prune_gemm_tiles_xcd0(c->Kmax, (int)(gc / kBN), (rows + 16 * mi - 1) / (16 * mi));
                }
                worst = std::max(worst, (size_t)8 * (size_t)(tiles + kPlanLanes * (1 + kPlanSlack)));   // >= 8 * nlb * (ceil(tiles / nlb) + slack), any K <= Kmax
            }
            entries += worst;
        }
        c->plan_entries = entries;
        HIP_TRY(c, hipHostMalloc(&c->h_plan_desc, sizeof(PlanLaunch) * std::max(1, c->n_gemm_groups), hipHostMallocDefault));
    }

    if (std::getenv("CAFE_DUMP_SCHEDULE")) {             // diagnostic: the launch list with its column counts
        std::fprintf(stderr, "cafe schedule: %s, %zu ops in %zu launches, %d panels, %.2f GB\n", c->grouped ? "grouped" : "one op per launch", c->ops.size(),
                     c->groups.size(), c->n_panels, c->stats.panel_bytes / 1e9);
        for (const Group& g : c->groups) {
            std::fprintf(stderr, "cafe schedule: step %d %s x%zu\n", g.step, g.type == 1 ? "K2" : "K3", g.ops.size());
            for (int oi : g.ops) {
                const Op& op = c->ops[oi];
                const int64_t pc = c->subtree_dedup ? c->pat_cols[op.parent] : c->chunk_cols;
                if (op.type == 1)
                    std::fprintf(stderr, "cafe schedule:   gemm child %d -> parent %d cols %lld %s%s%s leaf %d\n", op.child, op.parent,
                                 (long long)(c->subtree_dedup ? c->pat_cols[op.child] : c->chunk_cols), op.to_factor ? "factor(transposed)" : (op.mode ? "multiply" : "store"),
                                 op.has_gath ? " +gathered-factor" : "", op.to_root ? " root" : "", op.n_leaf);
                else
                    std::fprintf(stderr, "cafe schedule:   %s parent %d cols %lld factors %d leaves %d %s\n", op.n_src ? "assemble" : "leaf-gather", op.parent,
                                 (long long)pc, op.n_src, op.n_leaf, op.mode ? "multiply" : "store");
            }
        }
    }
    int n_gemm = c->n_gemm_groups;
    for (auto& op : c->ops) {
        c->stats.n_gather_epilogues += op.type == 1 && op.has_gath;
        c->stats.n_assemble_passes += op.type == 0 && op.n_src > 0;
        c->stats.n_leaf_passes += op.type == 0 && op.n_src == 0;
    }
    c->gemm_ev.resize((size_t)2 * n_gemm * c->stats.n_chunks);
    for (auto& e : c->gemm_ev) HIP_TRY(c, hipEventCreate(&e));
    HIP_TRY(c, hipDeviceSynchronize());
    return CAFE_OK;
}

}  // namespace

namespace cafe {

bool lambdas_valid(const cafe_ctx* c, const double* lam) {
    if (c->single_lambda) return lam[0] > 0;                                   // lambda.h:58
    for (int i = 0; i < c->n_lambdas; ++i) if (lam[i] < 0) return false;       // lambda.cpp:59
    return true;
}

// Slot parameters of a call into the pinned stage: slot = k * n_pairs[layout] + pair, built from the de-quantized key
// like matrix_cache.cpp:148-149 (lambda.h:39, :82-88 for lambda * multiplier).
void fill_slots(cafe_ctx* c, const double* lambdas, const double* multipliers, int K) {
    SlotParam* h[2] = {reinterpret_cast<SlotParam*>(c->h_stage), reinterpret_cast<SlotParam*>(c->h_stage) + c->max_slots};
    for (int layout = 0; layout < 2; ++layout) {
        const int P = c->n_pairs[layout];
        for (int k = 0; k < K; ++k) {
            const double mult = multipliers ? multipliers[k] : 1.0;
            for (int p = 0; p < P; ++p) {
                const long lq = long(lambdas[c->pair_lam[layout][p]] * mult * 1000000000);     // matrix_cache.h:47
                const double lambda_q = double(lq) / 1000000000.0, t_q = double(c->pair_tq[layout][p]) / 1000.0;
                const double alpha = lambda_q * t_q / (1 + lambda_q * t_q);
                const double coeff = 1 - 2 * alpha;
                SlotParam sp;
                sp.alpha = alpha;
                sp.oma2 = (1 - alpha) * (1 - alpha);
                sp.zero = !(coeff > 0 && coeff != 1);       // saturated (coeff < 0) or degenerate: rows s>=1 are 0
                sp.pad = 0;
                h[layout][k * P + p] = sp;
            }
        }
    }
    c->n_slots_last = K * c->n_pairs[0];
    c->n_kslots_last = K * c->n_pairs[1];
    c->stats.n_matrices = (int64_t)K * c->n_distinct_pairs;
}

// For the callers outside the scorer path (reconstruct.hip): slot parameters uploaded on `s`, K1 launched.  Afterwards
// slot_of[node * Kmax + k] (static) names the matrix of every branch and category.
int prepare_matrices(cafe_ctx* c, const double* lambdas, const double* multipliers, int K, hipStream_t s) {
    if (c->upload_pending) { HIP_TRY(c, hipEventSynchronize(c->ev_upload)); c->upload_pending = false; }
    fill_slots(c, lambdas, multipliers, K);
    const size_t nb = sizeof(SlotParam) * (size_t)(c->max_slots + c->max_kslots);
    HIP_TRY(c, hipMemcpyAsync(c->d_params, c->h_stage, nb, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipEventRecord(c->ev_upload, s));
    c->upload_pending = true;
    HIP_TRY(c, launch_bd_matrix_build_both(c->pool, c->kpool, c->d_slots, c->d_slots + c->max_slots, c->n_slots_last, c->n_kslots_last, s));
    return CAFE_OK;
}

}  // namespace cafe

namespace {

// Host-only rejections; true => the call's value is +inf without touching the device.
bool rejected(const cafe_ctx* c, const cafe_params* pr, int K) {
    if (!lambdas_valid(c, pr->lambdas)) return true;                           // base_model.cpp:56 / gamma_core.cpp:125
    if (pr->model != CAFE_MODEL_GAMMA) return false;
    if (pr->alpha < 0) return true;                                            // gamma_core.cpp:128
    // gamma_core.cpp:131-139: longest branch x largest multiplier x largest lambda saturated?
    bool first = true;
    double longest = 0;
    for (int v = 0; v < c->n_nodes; ++v) {       // clade::get_branch_lengths: the set of t > 0, root included
        double t = c->blen[v];
        if (!(t > 0.0)) continue;
        if (first || t > longest) { longest = t; first = false; }
    }
    double lm = *std::max_element(pr->multipliers, pr->multipliers + K);
    double ll = *std::max_element(pr->lambdas, pr->lambdas + c->n_lambdas);
    double lambda = lm * ll;
    double alpha = lambda * longest / (1 + lambda * longest);                  // matrix_cache.cpp:115
    return (1 - 2 * alpha) < 0;
}

// Row-tile height of one K2 launch (a group of ops) from the (previous call's) non-zero extents of its matrices: a tile runs
// only the K tiles inside the union of its 16-row blocks' extents, so a lower tile hugs the band of a short branch more
// closely (at config 4 the launches execute 69 % of all K tiles with 144-row tiles, 64 % with 80-row ones) but is a little
// less efficient per MFMA and fills the persistent grid in different rounds.  Costs in units of one K tile of one 16-row
// block; the efficiency factors are measured (forced tile heights): even heights stage a padded A tile.
int pick_tile_height(const cafe_ctx* c, const int32_t* ext, const Group& g, int K, int64_t chunk_cols) {
    static const double eff[10] = {0, 0, 0, 0, 1.12, 1.03, 1.30, 1.06, 1.05, 1.00};
    const int nb = c->kpool.ext_blocks;
    const int kBK = c->kb;
    const int n_k = (c->M + 1 + kBK - 1) / kBK;
    const double overhead = 2.0 * 16 / kBK;  // prologue + epilogue of a tile, in K tiles
    int best = 9;
    double best_cost = 1e300;
    for (int mi = 9; mi >= 4; --mi) {
        if (mi == 6) continue;
        const int slots = prune_gemm_wg_per_cu(mi, c->kb) * c->n_cu / 8 * 8;
        double work = 0, tiles = 0;          // sum over (op, category, row tile, column tile) of (K tiles + overhead) * height; tiles
        for (int oi : g.ops) {
            const Op& op = c->ops[oi];
            const int rows = op.to_root ? c->R : c->M;
            const int n_col_tiles = (int)((c->subtree_dedup ? c->pat_cols[op.child] : chunk_cols) / kBN);
            const int row_tiles = (rows + 16 * mi - 1) / (16 * mi);
            for (int k = 0; k < K; ++k) {
                const int32_t* e = ext + (size_t)c->slot_of[(size_t)op.child * c->Kmax + k] * nb * 2;
                for (int rt = 0; rt < row_tiles; ++rt) {
                    int lo = 0x7fffffff, hi = -1;
                    for (int b = rt * mi; b < rt * mi + mi && b < nb; ++b) { lo = std::min(lo, e[2 * b]); hi = std::max(hi, e[2 * b + 1]); }
                    int nkt = n_k;
                    if (hi >= lo) nkt = std::min(hi, c->M) / kBK - lo / kBK + 1; else nkt = 1;
                    work += (nkt + overhead) * mi * n_col_tiles;
                }
            }
            tiles += (double)row_tiles * K * n_col_tiles;
        }
        const double rounds = std::ceil(tiles / slots);
        const double cost = std::max(work / slots, rounds * (work / tiles)) * eff[mi];
        if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best = mi; }
    }
    return best;
}

// Host side of a call's launches for K categories and a chunk `cols` columns wide: the tile height of every K2 group (from
// the previous call's extents when there are any), the per-op row-tile counts, the planner's descriptors.  sync: upload now
// (before a graph capture); otherwise on `s`, and only what changed since the last call.
int prepare_descriptors(cafe_ctx* c, DescSet& ds, int K, int64_t cols, const std::vector<int32_t>* prev_ext, bool sync, hipStream_t s, bool* plan_needed) {
    const size_t n_ops = c->h_gemm_ops.size();
    if (!ds.d_gemm_ops) {
        HIP_TRY(c, hipMalloc(&ds.d_gemm_ops, sizeof(GemmOp) * n_ops));
        HIP_TRY(c, hipMalloc(&ds.d_plan_desc, sizeof(PlanLaunch) * std::max(1, c->n_gemm_groups)));
        HIP_TRY(c, hipMalloc(&ds.d_plan, sizeof(int2) * std::max<size_t>(1, c->plan_entries)));
    }
    ds.group_mi.assign(c->n_gemm_groups, 0);
    ds.group_blocks.assign(c->n_gemm_groups, 0);
    ds.group_rounds.assign(c->n_gemm_groups, 0);
    ds.group_plan_off.assign(c->n_gemm_groups, 0);
    std::vector<GemmOp> ops = c->h_gemm_ops;
    std::vector<PlanLaunch> plans(c->n_gemm_groups);
    size_t used = 0;
    int gi = 0;
    for (const Group& g : c->groups) {
        if (g.type != 1) continue;
        int mi = c->force_mi;                               // 0: picked per launch
        if (!mi && prev_ext) mi = pick_tile_height(c, prev_ext->data(), g, K, cols);
        if (!mi) {                                          // no extents (yet): whole rounds x height
            int64_t tiles_by_mi[10] = {0};
            for (int h = 2; h <= 9; ++h)
                for (int oi : g.ops) {
                    const Op& op = c->ops[oi];
                    const int64_t gc = c->subtree_dedup ? c->pat_cols[op.child] : cols;
                    tiles_by_mi[h] += (int64_t)(((op.to_root ? c->R : c->M) + 16 * h - 1) / (16 * h)) * (gc / kBN) * K;
                }
            mi = prune_gemm_pick_mi(tiles_by_mi, c->n_cu, c->kb);
        }
        int64_t tiles0 = 0;
        for (int oi : g.ops) {
            const Op& op = c->ops[oi];
            GemmOp& d = ops[op.desc];
            d.n_row_tiles = (d.rows + 16 * mi - 1) / (16 * mi);
            const int64_t gc = c->subtree_dedup ? c->pat_cols[op.child] : cols;
            tiles0 += prune_gemm_tiles_xcd0(K, (int)(gc / kBN), d.n_row_tiles);
        }
        const int blocks = prune_gemm_blocks(tiles0, c->n_cu, mi, c->kb), nlb = blocks / 8;
        const int rounds = (int)((tiles0 + nlb - 1) / nlb) + kPlanSlack;
        const size_t need = (size_t)8 * nlb * rounds;
        if (used + need > c->plan_entries) { set_err(c, "internal: tile lists do not fit (%zu + %zu > %zu)", used, need, c->plan_entries); return CAFE_ERR_STATE; }
        PlanLaunch& L = plans[gi];
        L = PlanLaunch{};
        L.aext = c->kpool.ext; L.ext_blocks = c->kpool.ext_blocks;
        L.ops = ds.d_gemm_ops + g.first_desc; L.n_ops = (int)g.ops.size();
        L.uniform_ld = c->subtree_dedup || c->grouped ? 0 : (int32_t)cols;
        L.mi = mi; L.n_categories = K; L.k_valid = c->M + 1; L.kb = c->kb;
        L.blocks_per_xcd = nlb; L.rounds = rounds; L.fixed = std::max(1, c->plan_fixed * 8 / c->kb); L.bias = c->plan_bias;
        for (int i = 0; i < 4; ++i) L.bias3[i] = nlb == 128 ? c->plan_bias4[i] : (i < 3 ? c->plan_bias3[i] : 100);
        L.plan = ds.d_plan + used;
        ds.group_mi[gi] = mi; ds.group_blocks[gi] = blocks; ds.group_rounds[gi] = rounds; ds.group_plan_off[gi] = used;
        used += need;
        ++gi;
    }
    const bool ops_same = ds.gemm_ops_sent.size() == n_ops && std::memcmp(ds.gemm_ops_sent.data(), ops.data(), sizeof(GemmOp) * n_ops) == 0;
    const bool plans_same = ds.plan_desc_sent.size() == plans.size() &&
                            (plans.empty() || std::memcmp(ds.plan_desc_sent.data(), plans.data(), sizeof(PlanLaunch) * plans.size()) == 0);
    if (!ops_same) {
        if (sync) {
            HIP_TRY(c, hipMemcpy(ds.d_gemm_ops, ops.data(), sizeof(GemmOp) * n_ops, hipMemcpyHostToDevice));
        } else {
            std::memcpy(c->h_gemm_stage, ops.data(), sizeof(GemmOp) * n_ops);       // (the previous upload from here was waited for: ev_upload)
            HIP_TRY(c, hipMemcpyAsync(ds.d_gemm_ops, c->h_gemm_stage, sizeof(GemmOp) * n_ops, hipMemcpyHostToDevice, s));
        }
        ds.gemm_ops_sent = ops;
    }
    if (!plans_same && !plans.empty()) {
        if (sync) {
            HIP_TRY(c, hipMemcpy(ds.d_plan_desc, plans.data(), sizeof(PlanLaunch) * plans.size(), hipMemcpyHostToDevice));
        } else {
            std::memcpy(c->h_plan_desc, plans.data(), sizeof(PlanLaunch) * plans.size());
            HIP_TRY(c, hipMemcpyAsync(ds.d_plan_desc, c->h_plan_desc, sizeof(PlanLaunch) * plans.size(), hipMemcpyHostToDevice, s));
        }
        ds.plan_desc_sent = plans;
    }
    // with extents the lists follow this call's matrices: planned every call; without, only when something they depend on changed
    const bool static_ok = ds.plan_static_valid && ds.plan_static_K == K && ds.plan_static_cols == cols && ops_same && plans_same;
    *plan_needed = c->kpool.ext != nullptr || !static_ok;
    ds.plan_static_valid = true; ds.plan_static_K = K; ds.plan_static_cols = cols;
    return CAFE_OK;
}

// The device work of one call, enqueued on `s` (or recorded into a graph being captured on `s`): parameter upload,
// K1, the schedule (K2 / K3 launches), K4, the final sum into d_out.  Everything that changes between calls of the
// same shape travels through the parameter block; kernel arguments depend only on (reduction, K, error model).
int record_call(cafe_ctx* c, DescSet& ds, int K, bool gamma, bool rootmax, bool use_err, double* d_out, hipStream_t s, bool events, bool capturing) {
    c->stats.gemm_flops = c->stats.gemm_bytes = c->stats.gemm_flops_per_family = c->stats.gemm_flops_dense = 0;
    c->stats.gemm_launches = 0;
    c->desc_last = &ds;
    // ---- host: tile heights and descriptors (the previous call's extents are in h_ext: that call was waited for)
    const bool have_ext = c->h_ext && c->h_ext_valid && c->h_ext_K == K;       // the previous call's extents (same shape)
    std::vector<int32_t> prev_ext;
    if (have_ext) prev_ext.assign(c->h_ext, c->h_ext + (size_t)2 * c->n_kslots_last * c->kpool.ext_blocks);
    bool plan_needed = true;
    const int64_t cols0 = std::min<int64_t>(c->chunk_cols, c->Fp);
    if (!capturing) {                        // (a capture's descriptors were prepared and uploaded before it began)
        const int rc = prepare_descriptors(c, ds, K, cols0, have_ext ? &prev_ext : nullptr, false, s, &plan_needed);
        if (rc != CAFE_OK) return rc;
    }
    if (c->panels_dirty) {                   // the last call returned NaN: no stale NaN may meet a zero of a padded K step
        HIP_TRY(c, hipMemsetAsync(c->d_panels, 0, (size_t)c->stats.panel_bytes, s));
        c->panels_dirty = false;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_params, c->h_stage, c->params_bytes, hipMemcpyHostToDevice, s));
    if (events) HIP_TRY(c, hipEventRecord(c->ev[0], s));
    HIP_TRY(c, launch_bd_matrix_build_both(c->pool, c->kpool, c->d_slots, c->d_slots + c->max_slots, c->n_slots_last, c->n_kslots_last, s));
    if (events) HIP_TRY(c, hipEventRecord(c->ev[1], s));
    if (c->debug_fail_in > 0 && --c->debug_fail_in == 0) {                     // test hook: a device error in the middle of a call
        set_err(c, "injected failure (cafe_debug_fail_next)");
        return CAFE_ERR_DEVICE;
    }
    if (c->h_ext) {                          // this call's extents for the next one; lands while the K2 launches run
        HIP_TRY(c, hipMemcpyAsync(c->h_ext, c->kpool.ext, sizeof(int32_t) * 2 * (size_t)c->n_kslots_last * c->kpool.ext_blocks, hipMemcpyDeviceToHost, s));
        c->h_ext_valid = true;
        c->h_ext_K = K;
    }
    if (c->panel_extents) {                  // zero extents of every node's panel for this call's matrices (extents.hip)
        ExtArgs ea{};
        ea.nodes = c->d_ext_nodes;
        ea.leaf_ext = c->pool.ext; ea.leaf_ext_blocks = c->pool.ext_blocks; ea.n_pairs_leaf = c->n_pairs[0];
        ea.kext = c->kpool.ext; ea.kext_blocks = c->kpool.ext_blocks; ea.n_pairs_inner = c->n_pairs[1];
        ea.M = c->M;
        ea.err = use_err ? c->d_err : nullptr; ea.n_dev = use_err ? c->n_dev : 0;
        for (const auto& L : c->ext_levels) {
            ea.first = L.first; ea.count = L.count;
            HIP_TRY(c, launch_node_extents(ea, L.max_col_tiles, K, s));
        }
    }

    const bool leaf_t = c->d_lt && !use_err;
    c->lt_used_last = leaf_t;
    if (leaf_t) {                            // transposed copies of the leaf matrices that meet a factor in an assemble pass
        LeafTArgs lt{};
        lt.pool = c->pool; lt.pairs = c->d_lt_pairs; lt.pool_pairs = c->n_pairs[0]; lt.n_list = (int)c->lt_pairs.size();
        lt.n_x = c->M + 1; lt.ld_t = c->factor_ld; lt.dst = c->d_lt;
        lt.kstride = (int64_t)(c->M + 1) * c->factor_ld; lt.pair_stride = lt.kstride * c->Kmax;
        HIP_TRY(c, launch_leaf_transpose(lt, lt.n_list, K, s));
    }

    // ---- prune, chunk by chunk (one chunk unless the workspace is limited)
    c->gemm_ev_used = 0;
    c->gemm_launches_info.clear();
    int64_t planned_cols = cols0;
    for (int64_t f0 = 0; f0 < c->Fp; f0 += c->chunk_cols) {
        const int64_t cols = std::min<int64_t>(c->chunk_cols, c->Fp - f0);
        const int uniform_ld = (c->subtree_dedup || c->grouped) ? 0 : (int)cols;
        if (cols != planned_cols) {          // the last chunk is narrower: its own descriptors and lists (same stream: in order)
            if (capturing) { set_err(c, "internal: a captured call cannot re-plan for a narrower last chunk"); return CAFE_ERR_STATE; }
            HIP_TRY(c, hipStreamSynchronize(s));             // the descriptors (and their pinned stage) are still in use by the chunks before
            const int rc = prepare_descriptors(c, ds, K, cols, have_ext ? &prev_ext : nullptr, false, s, &plan_needed);
            if (rc != CAFE_OK) return rc;
            plan_needed = true;
            planned_cols = cols;
            ds.plan_static_valid = false;
        }
        if (plan_needed && c->n_gemm_groups > 0)
            HIP_TRY(c, launch_tile_plan(ds.d_plan_desc, c->n_gemm_groups, *std::max_element(ds.group_rounds.begin(), ds.group_rounds.end()), s));
        plan_needed = false;
        int gi = 0;
        for (size_t g_index = 0; g_index < c->groups.size(); ++g_index) {
            const Group& g = c->groups[g_index];
            if (g.type == 0) {
                GatherGroup gg{};
                gg.pool = c->pool;
                gg.ops = c->d_gather_ops + g.first_desc; gg.n_ops = (int)g.ops.size(); gg.n_categories = K;
                gg.max_family_size = c->M;
                gg.err = use_err ? c->d_err : nullptr; gg.n_dev = use_err ? c->n_dev : 0;
                gg.uniform_ld = uniform_ld;
                gg.f0 = c->subtree_dedup ? 0 : f0;
                gg.leaf_t = leaf_t ? 1 : 0;
                HIP_TRY(c, launch_leaf_gather_group(gg, c->h_gather_ops.data() + g.first_desc, s));
                continue;
            }
            GemmArgs a{};
            a.pool = c->kpool; a.lpool = c->pool;
            a.ops = ds.d_gemm_ops + g.first_desc; a.n_ops = (int)g.ops.size();
            a.k_valid = c->M + 1; a.kb = c->kb; a.mi = ds.group_mi[gi]; a.n_categories = K;
            a.uniform_ld = uniform_ld;
            a.f0 = c->subtree_dedup ? 0 : f0;
            a.err = use_err ? c->d_err : nullptr; a.max_family_size = c->M;
            a.stamps = (c->stamps_launch < 0 || c->stamps_launch == (long)c->stats.gemm_launches) ? c->d_stamps : nullptr;
            a.plan = ds.d_plan + ds.group_plan_off[gi]; a.plan_rounds = ds.group_rounds[gi];
            if (events && c->gemm_ev_used + 2 <= c->gemm_ev.size()) {       // start / stop events ride on the dispatch itself
                HIP_TRY(c, launch_prune_gemm(a, g.variant, ds.group_blocks[gi], s, c->gemm_ev[c->gemm_ev_used], c->gemm_ev[c->gemm_ev_used + 1]));
                c->gemm_ev_used += 2;
            } else {
                HIP_TRY(c, launch_prune_gemm(a, g.variant, ds.group_blocks[gi], s));
            }
            c->stats.gemm_launches += 1;
            c->gemm_launches_info.push_back({(int)g_index, K, a.mi, cols});
            for (int oi : g.ops) {
                const Op& op = c->ops[oi];
                const double gc = (double)(c->subtree_dedup ? c->pat_cols[op.child] : cols);
                const int rows = op.to_root ? c->R : c->M;
                c->stats.gemm_flops_dense += 2.0 * rows * (c->M + 1) * gc * K;
                c->stats.gemm_flops += 2.0 * rows * (c->M + 1) * gc * K;      // (cafe_executed_flops counts what the tiles really ran)
                c->stats.gemm_flops_per_family += 2.0 * rows * (c->M + 1) * (double)cols * K;
                c->stats.gemm_bytes += 8.0 * K * ((double)rows * (c->M + 1) + (double)(c->M + 1) * gc + (double)rows * gc);
            }
            ++gi;
        }
        if (events && f0 + c->chunk_cols >= c->Fp) HIP_TRY(c, hipEventRecord(c->ev[2], s));
        ReduceArgs r{};
        r.root = c->d_panels + c->panels[c->root_panel].offset;
        r.panel_kstride = c->panels[c->root_panel].kstride; r.ld = (int)cols; r.R = c->R; r.K = K; r.model = rootmax ? 2 : (gamma ? 1 : 0);
        r.prior = c->d_prior; r.log_prior = c->d_logprior; r.cat_probs = c->d_catprobs;
        r.f0 = f0; r.nf = std::max<int64_t>(0, std::min<int64_t>(cols, c->F_uniq - f0));
        r.fam_out = c->d_fam_out; r.fam_lik = c->d_fam_lik; r.cat_out = c->d_cat_out; r.failed = c->d_failed;
        HIP_TRY(c, launch_root_reduce(r, s));
        c->last_chunk_f0 = f0;
        c->last_chunk_nf = r.nf;
        // (several chunks with extents: the lists depend on this chunk's panel extents -- none: extents need one chunk -- and
        // on the matrices, the same for every chunk: no re-plan)
    }
    c->plan_launches_last = c->n_gemm_groups;
    // (the pair also goes straight into pinned host memory: cafe_score without a communicator reads it there after the
    // stream has drained, no device-to-host copy)
    HIP_TRY(c, launch_final_sum(c->d_fam_out, c->d_weights, c->d_failed, c->F_uniq, c->d_scratch, c->n_scratch, d_out, c->h_result, s));
    if (events) { HIP_TRY(c, hipEventRecord(c->ev[3], s)); c->events_valid = true; }
    return CAFE_OK;
}

// rootmax: the p-value path (probability.cpp:273-317, 391-444) prunes with the plain lambda, no error model and
// no prior, and keeps max_j L_root[j] per family instead of the scorer's reduction.
int enqueue(cafe_ctx* c, const cafe_params* pr, double* d_out, hipStream_t s, bool rootmax = false) {
    if (!pr || !pr->lambdas || (!rootmax && !pr->prior)) { set_err(c, "cafe_score: lambdas and prior are required"); return CAFE_ERR_ARGUMENT; }
    const bool gamma = !rootmax && pr->model == CAFE_MODEL_GAMMA;
    const bool use_err = !rootmax && c->n_dev > 0;
    const int K = gamma ? pr->n_categories : 1;
    if (gamma && (K < 1 || K > c->Kmax || !pr->multipliers || !pr->cat_probs)) {
        set_err(c, "cafe_score: gamma model needs 1..%d categories with multipliers and cat_probs", c->Kmax);
        return CAFE_ERR_ARGUMENT;
    }
    if (!rootmax && (c->n_dev > 0) != (pr->error_model != nullptr)) {
        set_err(c, "cafe_score: error model %s but the problem was created with n_deviations=%d", pr->error_model ? "given" : "missing", c->n_dev);
        return CAFE_ERR_ARGUMENT;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    c->last_stream = s;
    if (c->upload_pending) { HIP_TRY(c, hipEventSynchronize(c->ev_upload)); c->upload_pending = false; }
    c->have_results = false;
    c->events_valid = false;
    c->K_last = K;
    c->model_last = rootmax ? CAFE_MODEL_BASE : pr->model;
    c->rootmax_last = rootmax;

    c->last_rejected = rootmax ? !lambdas_valid(c, pr->lambdas) : rejected(c, pr, K);
    if (c->last_rejected) {
        double* hr = reinterpret_cast<double*>(c->h_stage);
        hr[0] = 0.0; hr[1] = 1.0;
        HIP_TRY(c, hipMemcpyAsync(d_out, hr, 2 * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipEventRecord(c->ev_upload, s));
        c->upload_pending = true;
        c->stats.n_matrices = 0;
        c->stats.gemm_flops = c->stats.gemm_bytes = c->stats.gemm_flops_per_family = 0;
        c->stats.gemm_launches = 0;
        return CAFE_OK;
    }

    // ---- the call's parameters into the pinned mirror of the device block
    fill_slots(c, pr->lambdas, gamma ? pr->multipliers : nullptr, K);
    {
        double* h_prior = reinterpret_cast<double*>(c->h_stage + ((char*)c->d_prior - c->d_params));
        double* h_logprior = reinterpret_cast<double*>(c->h_stage + ((char*)c->d_logprior - c->d_params));
        double* h_cat = reinterpret_cast<double*>(c->h_stage + ((char*)c->d_catprobs - c->d_params));
        for (int j = 0; j < c->R; ++j) {
            const double eq = rootmax ? 1.0 : (double)pr->prior[j];    // compute() returns float (root_equilibrium_distribution.h:15)
            h_prior[j] = eq;
            h_logprior[j] = std::log(eq);
        }
        for (int k = 0; k < K; ++k) h_cat[k] = gamma ? pr->cat_probs[k] : 1.0;
        if (use_err) std::memcpy(c->h_stage + ((char*)c->d_err - c->d_params), pr->error_model, sizeof(double) * (size_t)(c->M + 1) * c->n_dev);
    }

    const bool graph_ok = c->use_graph && !c->profile && !c->d_stamps && c->force_mi == 0;
    if (!graph_ok) {
        const int rc = record_call(c, c->desc, K, gamma, rootmax, use_err, d_out, s, c->profile != 0, false);
        if (rc != CAFE_OK) return rc;
    } else {
        const int key = (rootmax ? 2 : (gamma ? 1 : 0)) + 4 * K;
        cafe_ctx::CallGraph& cg = c->graphs[key];
        auto drop = [&]() { hipFree(cg.desc.d_gemm_ops); hipFree(cg.desc.d_plan_desc); hipFree(cg.desc.d_plan); c->graphs.erase(key); };
        if (!cg.exec) {
            // capture on the context's own stream (idle: calls are sequential), replay on the caller's.  The graph gets
            // descriptors and tile lists of its own, uploaded before the capture begins; its tile heights are frozen.
            if (c->stats.n_chunks > 1) { drop(); set_err(c, "cafe_set_graphs: a call in several column chunks cannot be captured"); return CAFE_ERR_STATE; }
            {
                const bool have_ext = c->h_ext && c->h_ext_valid && c->h_ext_K == K;
                std::vector<int32_t> prev_ext;
                if (have_ext) prev_ext.assign(c->h_ext, c->h_ext + (size_t)2 * c->n_kslots_last * c->kpool.ext_blocks);
                bool plan_needed = true;
                const int rp = prepare_descriptors(c, cg.desc, K, std::min<int64_t>(c->chunk_cols, c->Fp), have_ext ? &prev_ext : nullptr, true, c->stream, &plan_needed);
                if (rp != CAFE_OK) { drop(); return rp; }
            }
            hipGraph_t graph = nullptr;
            HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            const int rc = record_call(c, cg.desc, K, gamma, rootmax, use_err, c->d_result, c->stream, false, true);
            const hipError_t ee = hipStreamEndCapture(c->stream, &graph);
            if (rc != CAFE_OK) { if (graph) (void)hipGraphDestroy(graph); drop(); return rc; }
            if (ee != hipSuccess || !graph) { drop(); set_err(c, "hipStreamEndCapture failed: %s", hipGetErrorString(ee)); return CAFE_ERR_DEVICE; }
            const hipError_t ei = hipGraphInstantiate(&cg.exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ei != hipSuccess) { drop(); set_err(c, "hipGraphInstantiate failed: %s", hipGetErrorString(ei)); return CAFE_ERR_DEVICE; }
            cg.stats = c->stats;
        }
        const int64_t n_mat = c->stats.n_matrices;
        c->stats.gemm_flops = cg.stats.gemm_flops; c->stats.gemm_bytes = cg.stats.gemm_bytes;
        c->stats.gemm_flops_per_family = cg.stats.gemm_flops_per_family; c->stats.gemm_launches = cg.stats.gemm_launches;
        c->stats.gemm_flops_dense = cg.stats.gemm_flops_dense;
        c->stats.n_matrices = n_mat;
        c->last_chunk_f0 = (c->Fp - 1) / c->chunk_cols * c->chunk_cols;
        c->last_chunk_nf = std::max<int64_t>(0, std::min<int64_t>(c->chunk_cols, c->F_uniq - c->last_chunk_f0));
        HIP_TRY(c, hipGraphLaunch(cg.exec, s));
        if (d_out != c->d_result) HIP_TRY(c, hipMemcpyAsync(d_out, c->d_result, 2 * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(c, hipEventRecord(c->ev_upload, s));
    c->upload_pending = true;
    c->have_results = true;
    return CAFE_OK;
}


// Flops the K2 launches of the last (profiled) call EXECUTED: a (row tile, column tile) pair runs only the K tiles inside
// matrix extent x panel extent, so read the extents this call published and count, per launch, what its tiles ran.  Reads
// the extents back (a few synchronous copies, milliseconds of host work): for measurement, once, not per call.
// per_block: count a row block only over the K tiles inside its own extent (what the kernel issues); false: every block over
// its tile's whole K range (the count of rounds 2 and 3a, kept for comparison)
double count_executed_flops(cafe_ctx* c, std::vector<double>* per_launch = nullptr, bool per_block = true) {
    if (c->gemm_launches_info.empty()) return c->stats.gemm_flops;
    const int kBK = c->kb;
    const int nb = c->kpool.ext_blocks;
    std::vector<int32_t> ext;
    if (c->kpool.ext) {
        ext.resize((size_t)2 * c->max_kslots * nb);
        if (hipMemcpy(ext.data(), c->kpool.ext, ext.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
    }
    double executed = 0;
    std::vector<int32_t> bext;
    for (const auto& L : c->gemm_launches_info) {
        const double before = executed;
        const int mi = L.mi, bm = 16 * mi;
        for (int oi : c->groups[L.group].ops) {
            const Op& op = c->ops[oi];
            const int rows = op.to_root ? c->R : c->M;
            const int64_t cols = c->subtree_dedup ? c->pat_cols[op.child] : L.cols;
            const int n_ct = (int)(cols / kBN);
            const bool have_b = c->kpool.ext && c->panel_extents && c->d_tileext[op.child];
            if (have_b) {
                bext.resize((size_t)2 * L.K * n_ct);
                if (hipMemcpy(bext.data(), c->d_tileext[op.child], bext.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
            }
            for (int k = 0; k < L.K; ++k) {
                const int32_t* e = c->kpool.ext ? ext.data() + (size_t)c->slot_of[(size_t)op.child * c->Kmax + k] * nb * 2 : nullptr;
                for (int row0 = 0; row0 < rows; row0 += bm) {
                    int alo = 0, ahi = c->M;
                    if (e) {
                        alo = 0x7fffffff; ahi = -1;
                        for (int b = row0 / 16; b < row0 / 16 + mi && b < nb; ++b) { alo = std::min(alo, e[2 * b]); ahi = std::max(ahi, e[2 * b + 1]); }
                    }
                    for (int ct = 0; ct < (have_b ? n_ct : 1); ++ct) {
                        int lo = alo, hi = ahi;
                        if (have_b) { lo = std::max(lo, bext[((size_t)k * n_ct + ct) * 2]); hi = std::min(hi, bext[((size_t)k * n_ct + ct) * 2 + 1]); }
                        if (hi < lo) { lo = 0; hi = 0; }
                        hi = std::min(hi, c->M);
                        // the tile runs K tiles lo/kb .. hi/kb; its row block b issues MFMAs only in those inside ITS OWN extent
                        // (prune_gemm.hip, block_ranges); the last K tile of the matrix is ragged
                        const int t_lo = lo / kBK, t_hi = hi / kBK;
                        for (int b = row0 / 16; b < row0 / 16 + mi && b * 16 < rows; ++b) {
                            int b_lo = t_lo, b_hi = t_hi;
                            if (e && per_block) {
                                if (b >= nb || e[2 * b + 1] < e[2 * b]) continue;
                                b_lo = std::max(t_lo, e[2 * b] / kBK);
                                b_hi = std::min(t_hi, e[2 * b + 1] / kBK);
                            }
                            if (b_hi < b_lo) continue;
                            const int kk = std::min((b_hi - b_lo + 1) * kBK, c->M + 1 - b_lo * kBK);
                            executed += 2.0 * std::min(16, rows - b * 16) * (double)kk * (have_b ? (double)kBN : (double)cols);
                        }
                    }
                }
            }
        }
        if (per_launch) per_launch->push_back(executed - before);
    }
    return executed;
}

void collect_stats(cafe_ctx* c) {
    if (!c->events_valid) return;
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->stats.ms_matrices = ms;
    if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) c->stats.ms_prune = ms;
    if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) c->stats.ms_reduce = ms;
    double g = 0;
    for (size_t i = 0; i + 1 < c->gemm_ev_used; i += 2)
        if (hipEventElapsedTime(&ms, c->gemm_ev[i], c->gemm_ev[i + 1]) == hipSuccess) g += ms;
    c->stats.ms_gemm = g;
}


}  // namespace

namespace cafe {

cafe_ctx* create_child_for_device_counts(const cafe_ctx* parent, int64_t n_families) {
    cafe_ctx* c = new (std::nothrow) cafe_ctx();
    if (!c) return nullptr;
    cafe_problem pb{};
    std::vector<int32_t> par(parent->parent.begin(), parent->parent.end()), lam(parent->lam_idx.begin(), parent->lam_idx.end()),
        leaf(parent->leaf_taxon.begin(), parent->leaf_taxon.end());
    pb.n_nodes = parent->n_nodes; pb.parent = par.data(); pb.branch_length = parent->blen.data(); pb.lambda_index = lam.data();
    pb.leaf_taxon = leaf.data(); pb.n_taxa = parent->n_taxa; pb.n_families = n_families; pb.counts = nullptr;
    pb.max_family_size = parent->M; pb.max_root_family_size = parent->R; pb.n_lambdas = parent->n_lambdas;
    pb.single_lambda = parent->single_lambda; pb.max_categories = 1; pb.n_deviations = 0; pb.device = parent->device;
    pb.flags = kFlagDeviceCounts; pb.workspace_limit = 0;
    int rc = CAFE_ERR_MEMORY;
    try { rc = create_impl(c, &pb); } catch (const std::exception&) { rc = CAFE_ERR_MEMORY; }
    if (rc != CAFE_OK) { free_device(c); delete c; return nullptr; }
    return c;
}

void destroy_child(cafe_ctx* c) {
    if (!c) return;
    free_device(c);
    delete c;
}

int enqueue_rootmax(cafe_ctx* c, const double* lambdas, hipStream_t s) {
    cafe_params pr{};
    pr.model = CAFE_MODEL_BASE; pr.lambdas = lambdas; pr.n_categories = 1;
    return enqueue(c, &pr, c->d_result, s, true);
}

}  // namespace cafe

extern "C" {

int cafe_abi_version(void) { return CAFE_ABI_VERSION; }

cafe_ctx* cafe_create(const cafe_problem* problem, char* err, size_t errlen) {
    cafe_ctx* c = new (std::nothrow) cafe_ctx();
    if (!c) return nullptr;
    int rc = CAFE_ERR_ARGUMENT;
    try {
        if (problem && (problem->flags & kFlagDeviceCounts)) { set_err(c, "cafe_create: unknown flag"); rc = CAFE_ERR_ARGUMENT; }
        else rc = create_impl(c, problem);
    } catch (const std::exception& e) {
        set_err(c, "cafe_create: %s", e.what());
        rc = CAFE_ERR_MEMORY;
    }
    if (rc != CAFE_OK) {
        if (err && errlen) { std::snprintf(err, errlen, "%s", c->err.c_str()); }
        free_device(c);
        delete c;
        return nullptr;
    }
    if (err && errlen) err[0] = 0;
    return c;
}

void cafe_destroy(cafe_ctx* ctx) {
    if (!ctx) return;
    free_device(ctx);
    delete ctx;
}

const char* cafe_last_error(const cafe_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

double cafe_finish_partial(const double hp[2]) {
    if (std::isnan(hp[1])) return std::numeric_limits<double>::quiet_NaN();     // a shard failed its call (poisoned pair)
    if (hp[1] > 0) return std::numeric_limits<double>::infinity();
    return -hp[0];
}

int cafe_score_partial(cafe_ctx* ctx, const cafe_params* params, double* device_partial, void* hip_stream) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    if (!device_partial) { set_err(ctx, "cafe_score_partial: device_partial is NULL"); return CAFE_ERR_ARGUMENT; }
    int rc;
    try {
        rc = enqueue(ctx, params, device_partial, (hipStream_t)hip_stream);
    } catch (const std::exception& e) {
        set_err(ctx, "cafe_score_partial: %s", e.what());
        rc = CAFE_ERR_MEMORY;
    }
    // the caller owns the collective: a failed shard leaves rejects = NaN in its pair (best effort), so that a reduction over
    // the shards reads NaN on every rank (cafe_finish_partial returns NaN) even if this return code goes unread
    if (rc != CAFE_OK && ctx->h_poison && ctx->device_ready) {
        (void)hipMemcpyAsync(device_partial, ctx->h_poison, 2 * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)hip_stream);
        (void)hipStreamSynchronize((hipStream_t)hip_stream);        // whatever the failed call left in flight (its upload reads h_stage)
        ctx->upload_pending = false;
    }
    return rc;
}

int cafe_score(cafe_ctx* ctx, const cafe_params* params, double* neg_lnl, const cafe_family_out* out) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    if (!neg_lnl) { set_err(ctx, "cafe_score: neg_lnl is NULL"); return CAFE_ERR_ARGUMENT; }
    auto t0 = std::chrono::steady_clock::now();
    int rc;
    try {
        rc = enqueue(ctx, params, ctx->d_result, ctx->stream);
    } catch (const std::exception& e) {
        set_err(ctx, "cafe_score: %s", e.what());
        rc = CAFE_ERR_MEMORY;
    }
    if (rc != CAFE_OK) {
        if (!ctx->comm) {
            if (ctx->device_ready && ctx->stream) (void)hipStreamSynchronize(ctx->stream);   // what the failed call left in flight
            ctx->upload_pending = false;
            return rc;
        }
        // The call is collective: the other ranks are in the all-reduce or on their way into it, whatever went wrong here
        // (a HIP error, an allocation, an exception -- argument errors are the same on every rank, but nothing relies on
        // that).  Enter it with rejects = NaN: every rank then reads NaN and returns an error.  If even that cannot be
        // enqueued, abort the communicator; the others run into their deadline (comm_wait_stream).
        const std::string first = ctx->err;
        bool sent = hipSetDevice(ctx->device) == hipSuccess &&
                    hipMemcpyAsync(ctx->d_result, ctx->h_poison, 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                    comm_allreduce_pair(ctx, ctx->d_result, ctx->stream) == CAFE_OK;
        if (sent) sent = comm_wait_stream(ctx, ctx->stream) == CAFE_OK;
        if (!sent) comm_abort(ctx);
        ctx->upload_pending = false;
        ctx->have_results = false;
        set_err(ctx, "%s [rank %d of %d; the other ranks were %s]", first.c_str(), ctx->comm_rank, ctx->comm_world,
                sent ? "told through the all-reduce" : "NOT told: communicator aborted");
        return rc;
    }
    rc = comm_allreduce_pair(ctx, ctx->d_result, ctx->stream);          // family shards on other GPUs: one RCCL all-reduce
    if (rc != CAFE_OK) { comm_abort(ctx); return rc; }
    // without a communicator K4's last kernel has already written the pair to h_result (a host-only rejection went
    // through a copy into d_result instead)
    if (ctx->comm || ctx->last_rejected)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_result, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    rc = comm_wait_stream(ctx, ctx->stream);
    if (rc != CAFE_OK) return rc;
    ctx->upload_pending = false;
    if (ctx->comm && std::isnan(ctx->h_result[1])) {                    // (rejects is a sum of family weights: never NaN by itself)
        ctx->have_results = false;
        set_err(ctx, "cafe_score: another rank of the communicator failed its call (rank %d of %d read the poisoned pair)", ctx->comm_rank, ctx->comm_world);
        return CAFE_ERR_DEVICE;
    }
    *neg_lnl = cafe_finish_partial(ctx->h_result);
    if (std::isnan(ctx->h_result[0])) ctx->panels_dirty = true;        // NaNs may now sit in the panels (see record_call)
    ctx->stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    collect_stats(ctx);
    if (out) return cafe_family_results(ctx, out);
    return CAFE_OK;
}

int cafe_debug_fail_next(cafe_ctx* ctx, int n) {
    if (!ctx || n < 0) return CAFE_ERR_ARGUMENT;
    ctx->debug_fail_in = n;
    return CAFE_OK;
}

int cafe_family_results(cafe_ctx* ctx, const cafe_family_out* out) {
    if (!ctx || !out) return CAFE_ERR_ARGUMENT;
    if (ctx->last_rejected || !ctx->have_results || ctx->rootmax_last) {
        // the reference leaves `results` empty / stale on a rejected call (gamma_core.cpp:173-179)
        set_err(ctx, "cafe_family_results: the last call was rejected (+inf) or no call was made");
        return CAFE_ERR_STATE;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    const int K = ctx->K_last;
    std::vector<double> tmp((size_t)ctx->F_uniq * std::max(1, K));
    std::vector<int32_t> itmp;
    if (out->family_lnl) {
        HIP_TRY(ctx, hipMemcpy(tmp.data(), ctx->d_fam_out, sizeof(double) * ctx->F_uniq, hipMemcpyDeviceToHost));
        for (int64_t f = 0; f < ctx->F_all; ++f) out->family_lnl[f] = tmp[ctx->ref_of[f]];
    }
    if (ctx->model_last == CAFE_MODEL_GAMMA) {
        if (out->family_likelihood) {
            HIP_TRY(ctx, hipMemcpy(tmp.data(), ctx->d_fam_lik, sizeof(double) * ctx->F_uniq, hipMemcpyDeviceToHost));
            for (int64_t f = 0; f < ctx->F_all; ++f) out->family_likelihood[f] = tmp[ctx->ref_of[f]];
        }
        if (out->category_likelihood) {
            HIP_TRY(ctx, hipMemcpy(tmp.data(), ctx->d_cat_out, sizeof(double) * ctx->F_uniq * K, hipMemcpyDeviceToHost));
            for (int64_t f = 0; f < ctx->F_all; ++f)
                for (int k = 0; k < K; ++k) out->category_likelihood[f * K + k] = tmp[ctx->ref_of[f] * K + k];
        }
    }
    if (out->failed) {
        itmp.resize(ctx->F_uniq);
        HIP_TRY(ctx, hipMemcpy(itmp.data(), ctx->d_failed, sizeof(int32_t) * ctx->F_uniq, hipMemcpyDeviceToHost));
        for (int64_t f = 0; f < ctx->F_all; ++f) out->failed[f] = itmp[ctx->ref_of[f]];
    }
    return CAFE_OK;
}

int cafe_root_max(cafe_ctx* ctx, const cafe_params* params, double* out) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    if (!out) { set_err(ctx, "cafe_root_max: out is NULL"); return CAFE_ERR_ARGUMENT; }
    int rc;
    try {
        rc = enqueue(ctx, params, ctx->d_result, ctx->stream, true);
    } catch (const std::exception& e) {
        set_err(ctx, "cafe_root_max: %s", e.what());
        return CAFE_ERR_MEMORY;
    }
    if (rc != CAFE_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->upload_pending = false;
    collect_stats(ctx);
    if (ctx->last_rejected) {            // an invalid lambda has no matrices: the reference would throw (matrix_cache.cpp:90)
        set_err(ctx, "cafe_root_max: invalid lambda");
        return CAFE_ERR_ARGUMENT;
    }
    std::vector<double> tmp((size_t)ctx->F_uniq);
    HIP_TRY(ctx, hipMemcpy(tmp.data(), ctx->d_fam_out, sizeof(double) * ctx->F_uniq, hipMemcpyDeviceToHost));
    for (int64_t f = 0; f < ctx->F_all; ++f) out[f] = tmp[ctx->ref_of[f]];
    return CAFE_OK;
}

int cafe_reconstruct(cafe_ctx* ctx, const cafe_params* params, const float* root_prior, int32_t* states) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    try {
        return reconstruct_impl(ctx, params, root_prior, states);
    } catch (const std::exception& e) {
        set_err(ctx, "cafe_reconstruct: %s", e.what());
        return CAFE_ERR_MEMORY;
    }
}

int cafe_branch_probabilities(cafe_ctx* ctx, const cafe_params* params, const int32_t* sizes, double* out) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    try {
        return branch_probabilities_impl(ctx, params, sizes, out);
    } catch (const std::exception& e) {
        set_err(ctx, "cafe_branch_probabilities: %s", e.what());
        return CAFE_ERR_MEMORY;
    }
}

int cafe_pvalues(cafe_ctx* ctx, const cafe_params* params, int32_t n_simulations, uint64_t seed, double* pvalues) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    try {
        return pvalues_impl(ctx, params, n_simulations, seed, pvalues);
    } catch (const std::exception& e) {
        set_err(ctx, "cafe_pvalues: %s", e.what());
        return CAFE_ERR_MEMORY;
    }
}

int cafe_matrix_size(const cafe_ctx* ctx) { return ctx ? ctx->N : 0; }

int cafe_debug_stamps(cafe_ctx* ctx, unsigned long long* out, size_t words) {
    if (!ctx || !ctx->d_stamps || !out) return CAFE_ERR_STATE;
    if (words > ctx->stamps_words) words = ctx->stamps_words;
    return hipMemcpy(out, ctx->d_stamps, words * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? CAFE_OK : CAFE_ERR_DEVICE;
}

int cafe_debug_force_tile(cafe_ctx* ctx, int mi) {
    if (!ctx || (mi != 0 && (mi < 2 || mi > 9))) return CAFE_ERR_ARGUMENT;
    ctx->force_mi = mi;
    return CAFE_OK;
}

int cafe_set_profiling(cafe_ctx* ctx, int on) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    ctx->profile = on ? 1 : 0;
    return CAFE_OK;
}

int cafe_get_extents(cafe_ctx* ctx, int32_t node, int32_t category, int32_t* matrix_ext, size_t matrix_ext_len,
                     int32_t* panel_ext, size_t panel_ext_len, int32_t* n_tiles) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    if (!ctx->have_results || ctx->last_rejected) { set_err(ctx, "cafe_get_extents: no completed call"); return CAFE_ERR_STATE; }
    if (node < 0 || node >= ctx->n_nodes || node == ctx->root || category < 0 || category >= ctx->K_last) {
        set_err(ctx, "cafe_get_extents: node/category out of range");
        return CAFE_ERR_ARGUMENT;
    }
    if (!ctx->kpool.ext || !ctx->pool.ext) { set_err(ctx, "cafe_get_extents: extents are switched off"); return CAFE_ERR_STATE; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    const bool leaf = ctx->leaf_taxon[node] >= 0;
    const MatrixPool& mp = leaf ? ctx->pool : ctx->kpool;
    const int slot = ctx->slot_of[(size_t)node * ctx->Kmax + category];
    if (matrix_ext) {
        if (matrix_ext_len < (size_t)2 * mp.ext_blocks) { set_err(ctx, "cafe_get_extents: matrix_ext too small"); return CAFE_ERR_ARGUMENT; }
        HIP_TRY(ctx, hipMemcpy(matrix_ext, mp.ext + (size_t)slot * mp.ext_blocks * 2, sizeof(int32_t) * 2 * mp.ext_blocks, hipMemcpyDeviceToHost));
    }
    if (n_tiles) *n_tiles = 0;
    if (panel_ext && !leaf && ctx->panel_extents && ctx->d_tileext[node]) {
        const int nt = (int)((ctx->subtree_dedup ? ctx->pat_cols[node] : ctx->Fp) / kBN);
        if (panel_ext_len < (size_t)2 * nt) { set_err(ctx, "cafe_get_extents: panel_ext too small"); return CAFE_ERR_ARGUMENT; }
        HIP_TRY(ctx, hipMemcpy(panel_ext, ctx->d_tileext[node] + (size_t)category * nt * 2, sizeof(int32_t) * 2 * nt, hipMemcpyDeviceToHost));
        if (n_tiles) *n_tiles = nt;
    }
    return CAFE_OK;
}

int cafe_debug_leaf_transposes(cafe_ctx* ctx, int32_t* n_branches, int32_t* used_by_last_call) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    if (n_branches) *n_branches = ctx->d_lt ? (int32_t)ctx->lt_pairs.size() : 0;
    if (used_by_last_call) *used_by_last_call = ctx->lt_used_last ? 1 : 0;
    return CAFE_OK;
}

int cafe_debug_column_extents(cafe_ctx* ctx, int32_t node, int32_t category, int32_t* out, size_t out_len, int64_t* n_cols) {
    if (!ctx || !out) return CAFE_ERR_ARGUMENT;
    if (!ctx->have_results || ctx->last_rejected) { set_err(ctx, "cafe_debug_column_extents: no completed call"); return CAFE_ERR_STATE; }
    if (node < 0 || node >= ctx->n_nodes || category < 0 || category >= ctx->K_last || !ctx->panel_extents || !ctx->d_colext[node]) {
        set_err(ctx, "cafe_debug_column_extents: no extents for this node");
        return CAFE_ERR_ARGUMENT;
    }
    const int64_t cols = ctx->subtree_dedup ? ctx->pat_cols[node] : ctx->Fp;
    if (n_cols) *n_cols = cols;
    if (out_len < (size_t)2 * cols) { set_err(ctx, "cafe_debug_column_extents: out too small"); return CAFE_ERR_ARGUMENT; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->d_colext[node] + (size_t)category * cols * 2, sizeof(int32_t) * 2 * cols, hipMemcpyDeviceToHost));
    return CAFE_OK;
}

int cafe_debug_tile_range_flops(cafe_ctx* ctx, double* flops) {
    if (!ctx || !flops) return CAFE_ERR_ARGUMENT;
    if (!ctx->have_results || ctx->gemm_launches_info.empty()) { set_err(ctx, "cafe_debug_tile_range_flops: no completed call that was enqueued launch by launch"); return CAFE_ERR_STATE; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    const double v = count_executed_flops(ctx, nullptr, false);
    if (v < 0) { set_err(ctx, "cafe_debug_tile_range_flops: reading the extents back failed"); return CAFE_ERR_DEVICE; }
    *flops = v;
    return CAFE_OK;
}

int cafe_executed_flops(cafe_ctx* ctx, double* flops) {
    if (!ctx || !flops) return CAFE_ERR_ARGUMENT;
    if (!ctx->have_results) { set_err(ctx, "cafe_executed_flops: no completed call"); return CAFE_ERR_STATE; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    if (ctx->gemm_launches_info.empty()) { set_err(ctx, "cafe_executed_flops: the last call was not enqueued launch by launch (a graph replay keeps no launch list)"); return CAFE_ERR_STATE; }
    const double v = count_executed_flops(ctx);
    if (v < 0) { set_err(ctx, "cafe_executed_flops: reading the extents back failed"); return CAFE_ERR_DEVICE; }
    *flops = v;
    return CAFE_OK;
}

// diagnostic / test: read the tile lists of the last call back and check them against the extents -- every tile of every
// op of every launch exactly once, with the K range the extents give, nothing behind the end of a list.
// *n_planned: K2 launches checked; *worst_load: largest planned workgroup load over the mean load of its XCD.
int cafe_debug_plan_check(cafe_ctx* ctx, int32_t* n_planned, double* worst_load) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    if (n_planned) *n_planned = 0;
    if (worst_load) *worst_load = 1.0;
    const DescSet* ds = ctx->desc_last;
    if (!ds || ctx->plan_launches_last == 0 || ds->plan_desc_sent.empty()) return CAFE_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    const int nb = ctx->kpool.ext_blocks;
    std::vector<int32_t> aext, bext;
    if (ctx->kpool.ext) {
        aext.resize((size_t)2 * ctx->max_kslots * nb);
        HIP_TRY(ctx, hipMemcpy(aext.data(), ctx->kpool.ext, aext.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    std::vector<int2> plan;
    double worst = 1.0;
    for (const PlanLaunch& L : ds->plan_desc_sent) {
        const int nlb = L.blocks_per_xcd;
        plan.resize((size_t)8 * nlb * L.rounds);
        HIP_TRY(ctx, hipMemcpy(plan.data(), L.plan, plan.size() * sizeof(int2), hipMemcpyDeviceToHost));
        const GemmOp* ops = ds->gemm_ops_sent.data() + (L.ops - ds->d_gemm_ops);
        std::vector<std::vector<int32_t>> op_bext(L.n_ops);
        for (int o = 0; o < L.n_ops; ++o) {
            const int nct = L.uniform_ld > 0 ? L.uniform_ld / kBN : ops[o].n_col_tiles;
            if (ops[o].bext && ctx->kpool.ext) {
                op_bext[o].resize((size_t)2 * L.n_categories * nct);
                HIP_TRY(ctx, hipMemcpy(op_bext[o].data(), ops[o].bext, op_bext[o].size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            }
        }
        for (int x = 0; x < 8; ++x) {
            std::vector<std::vector<char>> seen(L.n_ops);
            for (int o = 0; o < L.n_ops; ++o) {
                const int nct = L.uniform_ld > 0 ? L.uniform_ld / kBN : ops[o].n_col_tiles;
                seen[o].assign((size_t)((L.n_categories * nct - x + 7) >> 3) * ops[o].n_row_tiles, 0);
            }
            double total = 0, top = 0;
            for (int w = 0; w < nlb; ++w) {
                bool ended = false;
                double load = 0;
                for (int r = 0; r < L.rounds; ++r) {
                    const int2 e = plan[((size_t)x * nlb + w) * L.rounds + r];
                    if (e.y == 0) { ended = true; if (e.x != 0) goto bad; continue; }
                    const int o = e.x >> 24, t = e.x & 0xFFFFFF;
                    if (ended || o < 0 || o >= L.n_ops || t >= (int)seen[o].size() || seen[o][t]) goto bad;
                    seen[o][t] = 1;
                    const int nrt = ops[o].n_row_tiles, nct = L.uniform_ld > 0 ? L.uniform_ld / kBN : ops[o].n_col_tiles;
                    const int row_tile = t % nrt, pair = x + 8 * (t / nrt);
                    const int ct = pair % nct, cat = pair / nct, b0 = row_tile * L.mi;
                    int lo = 0, hi = L.k_valid - 1, zlo = 0;
                    if (ctx->kpool.ext) {
                        const int32_t* a = aext.data() + ((size_t)ops[o].slot[cat] * nb + b0) * 2;
                        lo = 0x7fffffff; hi = -1;
                        for (int b = 0; b < L.mi && b0 + b < nb; ++b) { lo = std::min(lo, a[2 * b]); hi = std::max(hi, a[2 * b + 1]); }
                        if (!op_bext[o].empty()) {
                            const int32_t* be = op_bext[o].data() + ((size_t)cat * nct + ct) * 2;
                            lo = std::max(lo, be[0]); hi = std::min(hi, be[1]);
                            if (be[1] >= be[0]) zlo = be[0];
                        }
                        if (hi < lo) { lo = zlo; hi = zlo; }
                        hi = std::min(hi, L.k_valid - 1);
                    }
                    if ((e.y >> 16) != lo / L.kb || (e.y & 0xFFFF) != hi / L.kb - lo / L.kb + 1) goto bad;
                    load += (e.y & 0xFFFF) + L.fixed;
                }
                total += load;
                top = std::max(top, load);
            }
            for (auto& sv : seen) for (char v : sv) if (!v) goto bad;
            if (total > 0) worst = std::max(worst, top / (total / nlb));
        }
        if (n_planned) *n_planned += 1;
        continue;
    bad:
        set_err(ctx, "cafe_debug_plan_check: the tile lists of a launch do not match its extents");
        return CAFE_ERR_STATE;
    }
    if (worst_load) *worst_load = worst;
    return CAFE_OK;
}

int cafe_debug_launch_ms(cafe_ctx* ctx, double* ms, size_t n) {      // diagnostic: HIP-event duration of every K2 launch of the last profiled call
    if (!ctx || !ms) return CAFE_ERR_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    for (size_t i = 0; i < n; ++i) {
        float t = 0;
        ms[i] = (2 * i + 1 < ctx->gemm_ev_used && hipEventElapsedTime(&t, ctx->gemm_ev[2 * i], ctx->gemm_ev[2 * i + 1]) == hipSuccess) ? t : 0.0;
    }
    return CAFE_OK;
}

int cafe_debug_launch_flops(cafe_ctx* ctx, double* executed, double* all_k_tiles, int32_t* tile_height, size_t n) {
    if (!ctx || !executed) return CAFE_ERR_ARGUMENT;
    if (!ctx->have_results || ctx->gemm_launches_info.empty()) { set_err(ctx, "cafe_debug_launch_flops: no call enqueued launch by launch"); return CAFE_ERR_STATE; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    std::vector<double> v;
    if (count_executed_flops(ctx, &v) < 0) { set_err(ctx, "cafe_debug_launch_flops: reading the extents back failed"); return CAFE_ERR_DEVICE; }
    for (size_t i = 0; i < n && i < v.size(); ++i) {
        const auto& L = ctx->gemm_launches_info[i];
        executed[i] = v[i];
        if (all_k_tiles) {
            all_k_tiles[i] = 0;
            for (int oi : ctx->groups[L.group].ops) {
                const Op& op = ctx->ops[oi];
                all_k_tiles[i] += 2.0 * (op.to_root ? ctx->R : ctx->M) * (ctx->M + 1) * (double)(ctx->subtree_dedup ? ctx->pat_cols[op.child] : L.cols) * L.K;
            }
        }
        if (tile_height) tile_height[i] = L.mi;
    }
    return (int)std::min(n, v.size()) >= 0 ? CAFE_OK : CAFE_OK;
}

int cafe_set_graphs(cafe_ctx* ctx, int on) {
    if (!ctx) return CAFE_ERR_ARGUMENT;
    ctx->use_graph = on ? 1 : 0;
    return CAFE_OK;
}

int cafe_get_matrix(cafe_ctx* ctx, int32_t node, int32_t category, double* out, size_t out_len) {
    if (!ctx || !out) return CAFE_ERR_ARGUMENT;
    if (!ctx->have_results) { set_err(ctx, "cafe_get_matrix: no completed call"); return CAFE_ERR_STATE; }
    if (node < 0 || node >= ctx->n_nodes || node == ctx->root || category < 0 || category >= ctx->K_last) {
        set_err(ctx, "cafe_get_matrix: node/category out of range");
        return CAFE_ERR_ARGUMENT;
    }
    const size_t n = (size_t)ctx->N;
    if (out_len < n * n) { set_err(ctx, "cafe_get_matrix: out buffer too small"); return CAFE_ERR_ARGUMENT; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    const int slot = ctx->slot_of[(size_t)node * ctx->Kmax + category];
    if (ctx->leaf_taxon[node] >= 0) {
        HIP_TRY(ctx, hipMemcpy2D(out, n * sizeof(double), ctx->pool.base + (int64_t)slot * ctx->pool.stride,
                                 (size_t)ctx->pool.ld * sizeof(double), n * sizeof(double), n, hipMemcpyDeviceToHost));
        return CAFE_OK;
    }
    // interior branch: stored k-major, Pt[c][s-1] = P[s][c] for c <= M, s >= 1; row 0 of P is e_0 and the
    // columns c > M are never materialised (the prune never reads them): reported as 0
    const size_t ldt = (size_t)ctx->kpool.ld, rows = (size_t)ctx->kpool.rows;
    std::vector<double> tmp(rows * ldt);
    HIP_TRY(ctx, hipMemcpy(tmp.data(), ctx->kpool.base + (int64_t)slot * ctx->kpool.stride, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::fill(out, out + n * n, 0.0);
    out[0] = 1.0;
    for (size_t s2 = 1; s2 < n; ++s2)
        for (size_t c2 = 0; c2 < n && c2 < rows; ++c2) out[s2 * n + c2] = tmp[c2 * ldt + (s2 - 1)];
    return CAFE_OK;
}

int cafe_get_root_likelihoods(cafe_ctx* ctx, int64_t family, int32_t category, double* out, size_t out_len) {
    if (!ctx || !out) return CAFE_ERR_ARGUMENT;
    if (!ctx->have_results || ctx->last_rejected) { set_err(ctx, "cafe_get_root_likelihoods: no completed call"); return CAFE_ERR_STATE; }
    if (family < 0 || family >= ctx->F_all || category < 0 || category >= ctx->K_last || out_len < (size_t)ctx->R) {
        set_err(ctx, "cafe_get_root_likelihoods: argument out of range");
        return CAFE_ERR_ARGUMENT;
    }
    const int64_t u = ctx->ref_of[family];
    if (u < ctx->last_chunk_f0 || u >= ctx->last_chunk_f0 + ctx->last_chunk_nf) {
        set_err(ctx, "cafe_get_root_likelihoods: family is not in the last resident chunk");
        return CAFE_ERR_STATE;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    const int64_t cols = std::min<int64_t>(ctx->chunk_cols, ctx->Fp - ctx->last_chunk_f0);
    const cafe::Panel& RP = ctx->panels[ctx->root_panel];
    const double* src = ctx->d_panels + RP.offset + (int64_t)category * RP.kstride + (u - ctx->last_chunk_f0);
    HIP_TRY(ctx, hipMemcpy2D(out, sizeof(double), src, (size_t)cols * sizeof(double), sizeof(double), (size_t)ctx->R, hipMemcpyDeviceToHost));
    return CAFE_OK;
}

int cafe_get_stats(const cafe_ctx* ctx, cafe_stats* stats) {
    if (!ctx || !stats) return CAFE_ERR_ARGUMENT;
    collect_stats(const_cast<cafe_ctx*>(ctx));
    *stats = ctx->stats;
    return CAFE_OK;
}

int cafe_build_matrices(int32_t device, int32_t n, int32_t count, const double* lambdas, const double* ts, int32_t layout, double* out) {
    if (n < 2 || count < 1 || !lambdas || !ts || !out || n > bd_matrix_max_order()) return CAFE_ERR_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CAFE_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return CAFE_ERR_DEVICE;
    MatrixPool pool{};
    pool.n = n;
    if (layout == 0) {
        pool.ld = round_up(n, 16); pool.stride = (int64_t)n * pool.ld; pool.kmajor = 0; pool.rows = n; pool.k_valid = n;
    } else {
        pool.rows = round_up(n, kBK); pool.k_valid = n; pool.kmajor = 1;
        pool.ld = round_up(n - 1, 16) + 16; pool.stride = (int64_t)pool.rows * pool.ld;
    }
    std::vector<SlotParam> sp(count);
    for (int i = 0; i < count; ++i) {
        long lq, tq;
        quantize(lambdas[i], ts[i], &lq, &tq);
        const double lambda_q = double(lq) / 1000000000.0, t_q = double(tq) / 1000.0;
        const double alpha = lambda_q * t_q / (1 + lambda_q * t_q);
        const double coeff = 1 - 2 * alpha;
        sp[i].alpha = alpha; sp[i].oma2 = (1 - alpha) * (1 - alpha); sp[i].zero = !(coeff > 0 && coeff != 1); sp[i].pad = 0;
    }
    SlotParam* d_sp = nullptr;
    int rc = CAFE_OK;
    const size_t bytes = sizeof(double) * pool.stride * count;
    if (hipMalloc(&pool.base, bytes) != hipSuccess) return CAFE_ERR_MEMORY;
    if (hipMalloc(&d_sp, sizeof(SlotParam) * count) != hipSuccess) { (void)hipFree(pool.base); return CAFE_ERR_MEMORY; }
    if (hipMemset(pool.base, 0, bytes) != hipSuccess) rc = CAFE_ERR_DEVICE;
    if (rc == CAFE_OK && hipMemcpy(d_sp, sp.data(), sizeof(SlotParam) * count, hipMemcpyHostToDevice) != hipSuccess) rc = CAFE_ERR_DEVICE;
    if (rc == CAFE_OK && launch_bd_matrix_build(pool, d_sp, count, nullptr) != hipSuccess) rc = CAFE_ERR_DEVICE;
    if (rc == CAFE_OK && hipDeviceSynchronize() != hipSuccess) rc = CAFE_ERR_DEVICE;
    std::vector<double> tmp;
    for (int i = 0; i < count && rc == CAFE_OK; ++i) {
        double* o = out + (size_t)i * n * n;
        if (layout == 0) {
            if (hipMemcpy2D(o, (size_t)n * sizeof(double), pool.base + (int64_t)i * pool.stride, (size_t)pool.ld * sizeof(double),
                            (size_t)n * sizeof(double), n, hipMemcpyDeviceToHost) != hipSuccess)
                rc = CAFE_ERR_DEVICE;
        } else {
            tmp.resize((size_t)pool.stride);
            if (hipMemcpy(tmp.data(), pool.base + (int64_t)i * pool.stride, tmp.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
                rc = CAFE_ERR_DEVICE;
                break;
            }
            std::fill(o, o + (size_t)n * n, 0.0);
            o[0] = 1.0;                                    // P's row 0 = e_0 is implicit in the k-major layout
            for (int s2 = 1; s2 < n; ++s2)
                for (int c2 = 0; c2 < n; ++c2) o[(size_t)s2 * n + c2] = tmp[(size_t)c2 * pool.ld + (s2 - 1)];
        }
    }
    (void)hipFree(pool.base);
    (void)hipFree(d_sp);
    return rc;
}

int cafe_probe_fp64_mfma(int32_t device, double* tflops) {
    if (!tflops) return CAFE_ERR_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CAFE_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return CAFE_ERR_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CAFE_ERR_DEVICE;
    const int blocks = prop.multiProcessorCount * 2, iters = 20000;
    double* d = nullptr;
    if (hipMalloc(&d, sizeof(double) * 256 * blocks) != hipSuccess) return CAFE_ERR_MEMORY;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)launch_mfma_probe(d, 200, blocks, nullptr);      // warm-up
    (void)hipEventRecord(a, nullptr);
    (void)launch_mfma_probe(d, iters, blocks, nullptr);
    (void)hipEventRecord(b, nullptr);
    int rc = hipEventSynchronize(b) == hipSuccess ? CAFE_OK : CAFE_ERR_DEVICE;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    // each wave issues iters * 8 MFMAs of 16*16*4*2 flops
    const double flops = (double)blocks * 4 /*waves*/ * iters * 8.0 * 2048.0;
    *tflops = ms > 0 ? flops / (ms * 1e-3) / 1e12 : 0.0;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
