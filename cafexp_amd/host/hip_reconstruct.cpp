// Pupko reconstruction and Viterbi branch probabilities of the hip models: the reference's
// model::reconstruct_ancestral_states (base_model.cpp:145, gamma_core.cpp:301) and the compute_viterbi_sum loop of
// estimator::execute (execute.cpp:163-176), forwarded to cafe_reconstruct / cafe_branch_probabilities.
#include "cafe_host.h"

#include <algorithm>
#include <cmath>

#include "../../include/cafe_mi355x.h"

namespace cafe {

std::vector<int32_t> hip_model_base::device_reconstruct(const std::vector<gene_family>& families, root_equilibrium_distribution* p_prior,
                                                        const std::vector<double>* multipliers) {
    if (&families != _p_gene_families && families.size() != _p_gene_families->size())
        throw std::runtime_error("reconstruct_ancestral_states: the family list must be the model's own");
    const int K = multipliers ? (int)multipliers->size() : 1;
    ensure_context(K);
    const int jmax = std::min(_max_family_size, _max_root_family_size);
    std::vector<float> root_prior(jmax + 1);
    for (int j = 0; j <= jmax; ++j) root_prior[j] = p_prior->compute(j);       // as left by the last inference call (execute.cpp:163)
    std::vector<double> lambdas = _p_lambda->values();
    cafe_params pr{};
    pr.model = multipliers ? CAFE_MODEL_GAMMA : CAFE_MODEL_BASE;
    pr.lambdas = lambdas.data();
    pr.n_categories = K;
    pr.multipliers = multipliers ? multipliers->data() : nullptr;
    std::vector<int32_t> states((size_t)K * families.size() * _order.size());
    if (cafe_reconstruct(_ctx, &pr, root_prior.data(), states.data()) != CAFE_OK)
        throw std::runtime_error(std::string("cafe_reconstruct: ") + cafe_last_error(_ctx));
    return states;
}

std::vector<double> hip_model_base::device_pvalues(int number_of_simulations, uint64_t seed) {
    ensure_context(1);
    std::vector<double> lambdas = _p_lambda->values(), out(_p_gene_families->size());
    cafe_params pr{};
    pr.model = CAFE_MODEL_BASE; pr.lambdas = lambdas.data(); pr.n_categories = 1;
    if (cafe_pvalues(_ctx, &pr, number_of_simulations, seed, out.data()) != CAFE_OK)
        throw std::runtime_error(std::string("cafe_pvalues: ") + cafe_last_error(_ctx));
    return out;
}

std::vector<double> hip_model_base::branch_probability_table(const reconstruction& rec, const std::vector<gene_family>& families,
                                                             const std::vector<const clade*>& order) {
    ensure_context(1);
    const size_t n = _order.size(), F = families.size();
    std::vector<int32_t> sizes(F * n);
    for (size_t f = 0; f < F; ++f)
        for (size_t v = 0; v < n; ++v) sizes[f * n + v] = rec.reconstructed_size(families[f], _order[v]);
    std::vector<double> lambdas = _p_lambda->values(), flat(F * n);
    cafe_params pr{};
    pr.model = CAFE_MODEL_BASE; pr.lambdas = lambdas.data(); pr.n_categories = 1;
    if (cafe_branch_probabilities(_ctx, &pr, sizes.data(), flat.data()) != CAFE_OK)
        throw std::runtime_error(std::string("cafe_branch_probabilities: ") + cafe_last_error(_ctx));
    std::map<const clade*, size_t> pos;
    for (size_t v = 0; v < n; ++v) pos[_order[v]] = v;
    std::vector<double> out(F * order.size());
    for (size_t f = 0; f < F; ++f)
        for (size_t i = 0; i < order.size(); ++i) out[f * order.size() + i] = flat[f * n + pos.at(order[i])];
    return out;
}

reconstruction* hip_base_model::reconstruct_ancestral_states(const std::vector<gene_family>& families, root_equilibrium_distribution* p_prior) {
    const std::vector<int32_t> states = device_reconstruct(families, p_prior, nullptr);
    auto result = new base_model_reconstruction();
    const size_t n = _order.size();
    for (size_t f = 0; f < families.size(); ++f) {
        auto& m = result->_reconstructions[families[f].id()];
        for (size_t v = 0; v < n; ++v)
            if (!_order[v]->is_leaf()) m[_order[v]] = states[f * n + v];          // leaves are read from the family (base_model.cpp:183)
    }
    return result;
}

reconstruction* hip_gamma_model::reconstruct_ancestral_states(const std::vector<gene_family>& families, root_equilibrium_distribution* p_prior) {
    if (_category_likelihoods.size() != families.size())
        throw std::runtime_error("reconstruct_ancestral_states: run infer_family_likelihoods first (category likelihoods are copied, gamma_core.cpp:323)");
    const std::vector<int32_t> states = device_reconstruct(families, p_prior, &_lambda_multipliers);
    auto result = new gamma_model_reconstruction(_lambda_multipliers);
    const size_t n = _order.size(), F = families.size(), K = _lambda_multipliers.size();
    for (size_t f = 0; f < F; ++f) {
        auto& r = result->_reconstructions[families[f].id()];
        r._category_likelihoods = _category_likelihoods[f];
        r.category_reconstruction.resize(K);
        for (size_t k = 0; k < K; ++k)
            for (size_t v = 0; v < n; ++v)
                if (!_order[v]->is_leaf()) r.category_reconstruction[k][_order[v]] = states[(k * F + f) * n + v];
        r.reconstruction = get_weighted_averages(r.category_reconstruction, _gamma_cat_probs);
    }
    return result;
}

branch_probabilities compute_branch_probabilities(hip_model_base& mdl, const reconstruction& rec, const std::vector<gene_family>& families,
                                                  const std::vector<double>& pvalues, double test_pvalue, const cladevector& order) {
    branch_probabilities probs;
    bool any = false;
    for (double p : pvalues) any = any || p < test_pvalue;
    if (!any) return probs;
    const std::vector<double> table = mdl.branch_probability_table(rec, families, order);
    for (size_t i = 0; i < families.size(); ++i) {
        if (!(pvalues[i] < test_pvalue)) continue;
        for (size_t j = 0; j < order.size(); ++j) {
            const double v = table[i * order.size() + j];
            probs.set(families[i], order[j], std::isnan(v) ? branch_probabilities::invalid() : branch_probabilities::branch_probability(v));
        }
    }
    return probs;
}

}  // namespace cafe
