// compute_pvalues (src/probability.cpp:255-454) above the C ABI.
//
// Reference: for every root size i < R, simulate `number_of_simulations` families down the tree
// (set_weighted_random_family_size, probability.cpp:320-351), prune each and keep max_j L_root[j]
// (get_random_probabilities, :273-317); then for every observed family take the same statistic and report
// max_i pvalue(observed, conditional_distribution[i]) (:379-444).
//
// Here: the draws stay on the host and consume the global engine exactly like the reference (same
// std::discrete_distribution / std::uniform_int_distribution objects, same prefix traversal, same order of
// root sizes and simulations), reading rows of the transition matrices the device built; the R x nsim + F
// prunes are two cafe_root_max calls.  No likelihood arithmetic happens on the host.
#include "cafe_host.h"

#include <algorithm>
#include <cmath>

#include "../../include/cafe_mi355x.h"

namespace cafe {

namespace {

struct ctx_holder {
    cafe_ctx* ctx = nullptr;
    ~ctx_holder() { if (ctx) cafe_destroy(ctx); }
};

std::vector<double> root_max(cafe_ctx* ctx, const lambda* p_lambda, size_t n) {
    std::vector<double> lambdas = p_lambda->values(), out(n);
    cafe_params pr{};
    pr.model = CAFE_MODEL_BASE; pr.lambdas = lambdas.data(); pr.n_categories = 1;
    if (cafe_root_max(ctx, &pr, out.data()) != CAFE_OK) throw std::runtime_error(std::string("cafe_root_max: ") + cafe_last_error(ctx));
    return out;
}

bool is_saturated(double branch_length, double lambda) {          // matrix_cache.cpp:113-118
    const double alpha = lambda * branch_length / (1 + lambda * branch_length);
    return (1 - 2 * alpha) < 0;
}

}  // namespace

std::vector<double> compute_pvalues(const clade* p_tree, const std::vector<gene_family>& families, const lambda* p_lambda,
                                    int number_of_simulations, int max_family_size, int max_root_family_size, int device, pvalue_work* keep) {
    const std::vector<const clade*> order = p_tree->post_order();
    const int n = (int)order.size();
    std::map<const clade*, int> index;
    std::vector<const clade*> leaves;
    for (int i = 0; i < n; ++i) { index[order[i]] = i; if (order[i]->is_leaf()) leaves.push_back(order[i]); }
    const int T = (int)leaves.size();
    std::vector<int> taxon_of(n, -1);
    for (int t = 0; t < T; ++t) taxon_of[index.at(leaves[t])] = t;

    // ---- observed families: max_j L_root[j] (compute_tree_pvalue, :397-399; NULL error model)
    const int64_t F = (int64_t)families.size();
    std::vector<int32_t> counts((size_t)F * T);
    for (int64_t f = 0; f < F; ++f)
        for (int t = 0; t < T; ++t) counts[(size_t)f * T + t] = families[f].get_species_size(leaves[t]->get_taxon_name());
    ctx_holder obs;
    obs.ctx = create_device_context(p_lambda, order, counts.data(), F, max_family_size, max_root_family_size, 1, 0, device);
    const std::vector<double> observed = root_max(obs.ctx, p_lambda, (size_t)F);

    // ---- the matrices those prunes used, row-major on the host, for the draws
    const int N = cafe_matrix_size(obs.ctx);
    std::vector<std::vector<double>> matrix(n);
    for (int v = 0; v < n; ++v) {
        if (order[v]->is_root()) continue;
        matrix[v].resize((size_t)N * N);
        if (cafe_get_matrix(obs.ctx, v, 0, matrix[v].data(), matrix[v].size()) != CAFE_OK)
            throw std::runtime_error(std::string("cafe_get_matrix: ") + cafe_last_error(obs.ctx));
    }

    // ---- simulate (get_random_probabilities :279-296, set_weighted_random_family_size :320-351)
    const int nsim = number_of_simulations, R = max_root_family_size;
    std::vector<const clade*> prefix;
    p_tree->apply_prefix_order([&](const clade* c) { prefix.push_back(c); });
    std::vector<int32_t> sim_counts((size_t)R * nsim * T);
    // a distribution object depends on (branch, parent size) only and keeps no state between draws: build each once
    std::vector<std::map<int, std::discrete_distribution<int>>> dist(n);
    std::vector<int> sizes(n);
    for (int i = 0; i < R; ++i)
        for (int sidx = 0; sidx < nsim; ++sidx) {
            int32_t* row = sim_counts.data() + ((size_t)i * nsim + sidx) * T;
            for (const clade* c : prefix) {
                const int v = index.at(c);
                if (c->is_root()) { sizes[v] = i; continue; }
                const int parent_family_size = sizes[index.at(c->get_parent())];
                int csize = 0;
                if (parent_family_size > 0) {
                    const double lam = p_lambda->get_value_for_clade(c), t = c->get_branch_length();
                    if (is_saturated(t, lam)) {                 // the reference draws and then overwrites the value (:333-337)
                        std::uniform_int_distribution<int> distribution(0, max_family_size - 1);
                        csize = distribution(randomizer_engine);
                    }
                    auto it = dist[v].find(parent_family_size);
                    if (it == dist[v].end()) {
                        const double* p = matrix[v].data() + (size_t)parent_family_size * N;
                        it = dist[v].emplace(parent_family_size, std::discrete_distribution<int>(p, p + max_family_size)).first;
                    }
                    csize = it->second(randomizer_engine);
                }
                sizes[v] = csize;                               // adjust_for_error_model(c, NULL) is the identity (:353-356)
                if (c->is_leaf()) row[taxon_of[v]] = csize;
            }
        }

    // ---- prune the simulated families, sort every root size's row (:309-315)
    std::vector<std::vector<double>> cond(R);
    {
        ctx_holder sim;
        sim.ctx = create_device_context(p_lambda, order, sim_counts.data(), (int64_t)R * nsim, max_family_size, max_root_family_size, 1, 0, device);
        const std::vector<double> lik = root_max(sim.ctx, p_lambda, (size_t)R * nsim);
        for (int i = 0; i < R; ++i) {
            cond[i].assign(lik.begin() + (size_t)i * nsim, lik.begin() + (size_t)(i + 1) * nsim);
            std::sort(cond[i].begin(), cond[i].end());
        }
    }

    // ---- compute_tree_pvalue :401-407
    std::vector<double> result((size_t)F);
    for (int64_t f = 0; f < F; ++f) {
        double best = 0;
        for (int s = 0; s < R; ++s) {
            const double p = pvalue(observed[f], cond[s]);
            if (s == 0 || p > best) best = p;
        }
        result[f] = best;
    }
    if (keep) { keep->conditional_distribution = std::move(cond); keep->observed_max_likelihood = observed; }
    return result;
}

}  // namespace cafe
