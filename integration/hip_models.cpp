// hip_models.cpp -- reference-side binding, see hip_models.h.  Compiled against the reference's headers.
#include "hip_models.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <numeric>
#include <stdexcept>

#include "error_model.h"
#include "gamma.h"
#include "gene_family.h"
#include "gene_family_reconstructor.h"
#include "lambda.h"
#include "matrix_cache.h"
#include "root_distribution.h"
#include "root_equilibrium_distribution.h"

namespace {

[[noreturn]] void fail(const char* what, cafe_ctx* ctx) {
    throw std::runtime_error(std::string(what) + ": " + cafe_last_error(ctx));      // the reference's fatal type (cafexp.cpp:215)
}

// 0-based lambda of the branch above every node.  multiple_lambda keeps its node-name -> index map private
// (lambda.h:69) and calculate_child_factor reads it per child (lambda.cpp:34-36); a clone updated to the values
// {0, 1, 2, ...} answers get_value_for_clade(c) with the index itself.
std::vector<int32_t> lambda_indices(const lambda* p_lambda, const cladevector& order) {
    std::vector<int32_t> idx(order.size(), 0);
    if (!dynamic_cast<const multiple_lambda*>(p_lambda)) return idx;
    std::unique_ptr<lambda> probe(p_lambda->clone());
    std::vector<double> ramp(p_lambda->count());
    std::iota(ramp.begin(), ramp.end(), 0.0);
    probe->update(ramp.data());
    for (size_t i = 0; i < order.size(); ++i)
        if (!order[i]->is_root()) idx[i] = (int32_t)probe->get_value_for_clade(order[i]);
    return idx;
}

std::vector<float> root_prior_for_reconstruction(root_equilibrium_distribution* p_prior, int max_family_size, int max_root_family_size) {
    // the root scan of reconstruct_gene_family (gene_family_reconstructor.cpp:47-62) reads compute(j), j the SIZE,
    // up to min(M, R) inclusive
    const int jmax = std::min(max_family_size, max_root_family_size);
    std::vector<float> v(jmax + 1);
    for (int j = 0; j <= jmax; ++j) v[j] = p_prior->compute(j);
    return v;
}

}  // namespace

hip_device_context::~hip_device_context() {
    if (_ctx) cafe_destroy(_ctx);
    if (_sharded) cafe_sharded_destroy(_sharded);
}

bool hip_device_context::matches(const signature& s, const lambda* p_lambda, const clade* p_tree, const std::vector<gene_family>* p_families,
                                 int categories, const error_model* p_error_model) const {
    return s.families == p_families && s.n_families == p_families->size() && s.tree == p_tree && s.lambda_count == p_lambda->count() &&
           s.multiple == (dynamic_cast<const multiple_lambda*>(p_lambda) != nullptr) && categories <= s.categories &&
           s.n_deviations == (p_error_model ? (int)p_error_model->n_deviations() : 0);
}

void hip_device_context::create(bool sharded, const lambda* p_lambda, const clade* p_tree, const std::vector<gene_family>* p_families,
                                int max_family_size, int max_root_family_size, int categories, const error_model* p_error_model) {
    if (!p_tree || !p_families || p_families->empty()) throw std::runtime_error("hip model: a tree and a non-empty family list are required");
    const bool multiple = dynamic_cast<const multiple_lambda*>(p_lambda) != nullptr;
    const int n_dev = p_error_model ? (int)p_error_model->n_deviations() : 0;
    _order.clear();
    p_tree->apply_reverse_level_order([this](const clade* c) { _order.push_back(c); });     // children before parents (clade.cpp:255)
    const int n = (int)_order.size();
    std::map<const clade*, int> index;
    for (int i = 0; i < n; ++i) index[_order[i]] = i;
    std::vector<int32_t> parent(n), leaf_taxon(n, -1);
    std::vector<double> blen(n);
    std::vector<const clade*> leaves;
    for (int i = 0; i < n; ++i) {
        const clade* c = _order[i];
        parent[i] = c->is_root() ? -1 : index.at(c->get_parent());
        blen[i] = c->get_branch_length();
        if (c->is_leaf()) { leaf_taxon[i] = (int32_t)leaves.size(); leaves.push_back(c); }
    }
    const std::vector<int32_t> lam_idx = lambda_indices(p_lambda, _order);
    const size_t T = leaves.size(), F = p_families->size();
    std::vector<int32_t> counts(F * T);                                                      // gene_family::get_species_size, once
    for (size_t f = 0; f < F; ++f)
        for (size_t t = 0; t < T; ++t) counts[f * T + t] = (*p_families)[f].get_species_size(leaves[t]->get_taxon_name());
    cafe_problem pb = {};
    pb.n_nodes = n; pb.parent = parent.data(); pb.branch_length = blen.data(); pb.lambda_index = lam_idx.data();
    pb.leaf_taxon = leaf_taxon.data(); pb.n_taxa = (int32_t)T; pb.n_families = (int64_t)F; pb.counts = counts.data();
    pb.max_family_size = max_family_size; pb.max_root_family_size = max_root_family_size;
    pb.n_lambdas = p_lambda->count(); pb.single_lambda = multiple ? 0 : 1; pb.max_categories = categories;
    pb.n_deviations = n_dev; pb.device = device;
    char err[512];
    signature sig;
    sig.families = p_families; sig.n_families = F; sig.tree = p_tree; sig.lambda_count = p_lambda->count(); sig.multiple = multiple;
    sig.categories = categories; sig.n_deviations = n_dev;
    if (sharded) {
        if (_sharded) { cafe_sharded_destroy(_sharded); _sharded = nullptr; }
        std::vector<int32_t> devices(std::max(1, n_gpus));
        std::iota(devices.begin(), devices.end(), n_gpus > 1 ? 0 : device);
        _sharded = cafe_create_sharded(&pb, devices.data(), (int32_t)devices.size(), err, sizeof err);
        if (!_sharded) throw std::runtime_error(std::string("cafe_create_sharded: ") + err);
        _sharded_sig = sig;
    } else {
        if (_ctx) { cafe_destroy(_ctx); _ctx = nullptr; }
        _ctx = cafe_create(&pb, err, sizeof err);
        if (!_ctx) throw std::runtime_error(std::string("cafe_create: ") + err);
        _ctx_sig = sig;
    }
}

cafe_ctx* hip_device_context::ensure(const lambda* p_lambda, const clade* p_tree, const std::vector<gene_family>* p_families,
                                     int max_family_size, int max_root_family_size, int categories, const error_model* p_error_model) {
    if (!_ctx || !matches(_ctx_sig, p_lambda, p_tree, p_families, categories, p_error_model))
        create(false, p_lambda, p_tree, p_families, max_family_size, max_root_family_size, categories, p_error_model);
    return _ctx;
}

void hip_device_context::ensure_scorer(const lambda* p_lambda, const clade* p_tree, const std::vector<gene_family>* p_families,
                                       int max_family_size, int max_root_family_size, int categories, const error_model* p_error_model) {
    if (n_gpus <= 1 && !sharded_scorer()) { ensure(p_lambda, p_tree, p_families, max_family_size, max_root_family_size, categories, p_error_model); return; }
    if (!_sharded || !matches(_sharded_sig, p_lambda, p_tree, p_families, categories, p_error_model))
        create(true, p_lambda, p_tree, p_families, max_family_size, max_root_family_size, categories, p_error_model);
}

bool hip_device_context::sharded_scorer() const {
    // several GPUs -- or CAFE_FORCE_SHARDED, which sends a single device through the same plan / worker thread / communicator
    // / gather code (what a one-GPU box can exercise of it)
    return n_gpus > 1 || std::getenv("CAFE_FORCE_SHARDED") != nullptr;
}

void hip_device_context::score(const cafe_params* params, double* neg_lnl) {
    if (sharded_scorer()) {
        if (cafe_sharded_score(_sharded, params, neg_lnl, nullptr) != CAFE_OK) throw std::runtime_error(std::string("cafe_sharded_score: ") + cafe_sharded_last_error(_sharded));
    } else if (cafe_score(_ctx, params, neg_lnl, nullptr) != CAFE_OK) {
        fail("cafe_score", _ctx);
    }
}

bool hip_device_context::family_results(const cafe_family_out* out) {
    return (sharded_scorer() ? cafe_sharded_family_results(_sharded, out) : cafe_family_results(_ctx, out)) == CAFE_OK;
}

void hip_call_inputs::gather(root_equilibrium_distribution* p_prior, const std::map<int, int>& rootdist, const lambda* p_lambda,
                             const error_model* p_error_model, int max_family_size, int max_root_family_size) {
    root_distribution rd;                                          // base_model.cpp:62-72 / gamma_core.cpp:182-192
    if (rootdist.size() > 0) rd.vectorize(rootdist);
    else rd.vectorize_uniform(max_root_family_size);
    p_prior->initialize(&rd);
    prior.resize(max_root_family_size);
    for (int j = 0; j < max_root_family_size; ++j) prior[j] = p_prior->compute(j);      // stays a float, like base_model.cpp:96
    lambdas = get_lambda_values(p_lambda);                         // matrix_cache.cpp:99
    error_table.clear();
    if (p_error_model) {
        // [(M+1)][n_deviations].  get_probs (error_model.cpp:52) is defined up to the error model's own maximum; the
        // rows beyond it are never read (observed counts do not exceed it) and repeat the last one.
        const size_t nd = p_error_model->n_deviations(), rows = p_error_model->get_max_family_size();
        for (int x = 0; x <= max_family_size; ++x) {
            const std::vector<double> p = p_error_model->get_probs(std::min<size_t>((size_t)x, rows - 1));
            error_table.insert(error_table.end(), p.begin(), p.begin() + nd);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- base
double hip_base_model::infer_family_likelihoods(root_equilibrium_distribution* prior, const std::map<int, int>& root_distribution_map,
                                                const lambda*) {
    _monitor.Event_InferenceAttempt_Started();
    if (!_p_lambda->is_valid()) {                                   // base_model.cpp:56-60 (member lambda, not the argument)
        _monitor.Event_InferenceAttempt_InvalidValues();
        return -log(0);
    }
    _dev.ensure_scorer(_p_lambda, _p_tree, _p_gene_families, _max_family_size, _max_root_family_size, 1, _p_error_model);
    hip_call_inputs in;
    in.gather(prior, root_distribution_map, _p_lambda, _p_error_model, _max_family_size, _max_root_family_size);
    cafe_params pr = {};
    pr.model = CAFE_MODEL_BASE; pr.lambdas = in.lambdas.data(); pr.n_categories = 1; pr.prior = in.prior.data();
    pr.error_model = in.error_table.empty() ? nullptr : in.error_table.data();
    double score = 0;
    _dev.score(&pr, &score);
    const size_t F = _p_gene_families->size();
    std::vector<double> lnl(F);
    cafe_family_out out = {};
    out.family_lnl = lnl.data();
    if (!_dev.family_results(&out)) throw std::runtime_error("cafe_family_results failed after a valid base-model call");
    results.resize(F);
    for (size_t i = 0; i < F; ++i)                                  // base_model.cpp:105
        results[i] = family_info_stash(_p_gene_families->at(i).id(), 0.0, 0.0, 0.0, lnl[i], false);
    _monitor.Event_InferenceAttempt_Complete(score);
    return score;
}

reconstruction* hip_base_model::reconstruct_ancestral_states(const std::vector<gene_family>& families, matrix_cache*,
                                                             root_equilibrium_distribution* p_prior) {
    _monitor.Event_Reconstruction_Started("Base");
    if (&families != _p_gene_families && families.size() != _p_gene_families->size())
        throw std::runtime_error("reconstruct_ancestral_states: the family list must be the model's own");
    cafe_ctx* ctx = _dev.ensure(_p_lambda, _p_tree, _p_gene_families, _max_family_size, _max_root_family_size, 1, _p_error_model);
    const std::vector<float> root_prior = root_prior_for_reconstruction(p_prior, _max_family_size, _max_root_family_size);
    const std::vector<double> lambdas = get_lambda_values(_p_lambda);
    cafe_params pr = {};
    pr.model = CAFE_MODEL_BASE; pr.lambdas = lambdas.data(); pr.n_categories = 1;
    const cladevector& order = _dev.order();
    const size_t n = order.size();
    std::vector<int32_t> states(families.size() * n);
    if (cafe_reconstruct(ctx, &pr, root_prior.data(), states.data()) != CAFE_OK) fail("cafe_reconstruct", ctx);
    auto result = new base_model_reconstruction();
    for (size_t f = 0; f < families.size(); ++f) {
        clademap<int>& m = result->_reconstructions[families[f].id()];
        for (size_t v = 0; v < n; ++v)
            if (!order[v]->is_leaf()) m[order[v]] = states[f * n + v];      // leaves are read from the family (base_model.cpp:183)
    }
    _monitor.Event_Reconstruction_Complete();
    return result;
}

// --------------------------------------------------------------------------------------------------------- gamma
void hip_gamma_model::current_categories(std::vector<double>& cat_probs, std::vector<double>& multipliers) const {
    multipliers = get_lambda_multipliers();
    if (_explicit_categories) { cat_probs = _explicit_cat_probs; return; }
    cat_probs.assign(get_gamma_cat_probs_count(), 0.0);
    if (cat_probs.size() > 1) {                                     // gamma_model::set_alpha, gamma_core.cpp:58-63
        std::vector<double> rates(cat_probs.size());
        get_gamma(cat_probs, rates, get_alpha());
    }
}

double hip_gamma_model::infer_family_likelihoods(root_equilibrium_distribution* prior, const std::map<int, int>& root_distribution_map,
                                                 const lambda*) {
    _monitor.Event_InferenceAttempt_Started();
    results.clear();
    if (!can_infer()) {                                             // gamma_core.cpp:123-142, :175-179
        _monitor.Event_InferenceAttempt_InvalidValues();
        return -log(0);
    }
    std::vector<double> cat_probs, multipliers;
    current_categories(cat_probs, multipliers);
    const int K = (int)cat_probs.size();
    _dev.ensure_scorer(_p_lambda, _p_tree, _p_gene_families, _max_family_size, _max_root_family_size, K, _p_error_model);
    hip_call_inputs in;
    in.gather(prior, root_distribution_map, _p_lambda, _p_error_model, _max_family_size, _max_root_family_size);
    cafe_params pr = {};
    pr.model = CAFE_MODEL_GAMMA; pr.lambdas = in.lambdas.data(); pr.n_categories = K; pr.multipliers = multipliers.data();
    pr.cat_probs = cat_probs.data(); pr.alpha = _explicit_categories ? 0.0 : get_alpha(); pr.prior = in.prior.data();
    pr.error_model = in.error_table.empty() ? nullptr : in.error_table.data();
    double score = 0;
    _dev.score(&pr, &score);
    const size_t F = _p_gene_families->size();
    std::vector<double> cat(F * K), fam(F);
    std::vector<int32_t> failed(F);
    cafe_family_out out = {};
    out.category_likelihood = cat.data(); out.family_likelihood = fam.data(); out.failed = failed.data();
    if (!_dev.family_results(&out)) return score;                    // rejected before any family was pruned
    if (std::isinf(score)) {                                        // a category's root vector summed to 0: gamma_core.cpp:227-236
        for (size_t i = 0; i < F; ++i)
            if (failed[i]) _monitor.Event_InferenceAttempt_Saturation(_p_gene_families->at(i).id());
        return score;
    }
    _hip_category_likelihoods.assign(F, std::vector<double>(K));
    for (size_t i = 0; i < F; ++i) {
        double denominator = 0;                                     // get_posterior_probabilities, gamma_core.cpp:97-107
        for (int k = 0; k < K; ++k) denominator += cat[i * K + k] * cat_probs[k];
        for (int k = 0; k < K; ++k) {
            const double post = cat[i * K + k] * cat_probs[k] / denominator;
            results.push_back(family_info_stash(_p_gene_families->at(i).id(), multipliers[k], cat[i * K + k], fam[i], post, post > 0.95));
            _hip_category_likelihoods[i][k] = cat[i * K + k];
        }
    }
    _monitor.Event_InferenceAttempt_Complete(score);
    return score;
}

reconstruction* hip_gamma_model::reconstruct_ancestral_states(const std::vector<gene_family>& families, matrix_cache*,
                                                              root_equilibrium_distribution* p_prior) {
    _monitor.Event_Reconstruction_Started("Gamma");
    if (_hip_category_likelihoods.size() != families.size())
        throw std::runtime_error("reconstruct_ancestral_states: run infer_family_likelihoods first (gamma_core.cpp:322 copies its category likelihoods)");
    std::vector<double> cat_probs, multipliers;
    current_categories(cat_probs, multipliers);
    const size_t K = multipliers.size();
    cafe_ctx* ctx = _dev.ensure(_p_lambda, _p_tree, _p_gene_families, _max_family_size, _max_root_family_size, (int)K, _p_error_model);
    const std::vector<float> root_prior = root_prior_for_reconstruction(p_prior, _max_family_size, _max_root_family_size);
    const std::vector<double> lambdas = get_lambda_values(_p_lambda);
    cafe_params pr = {};
    pr.model = CAFE_MODEL_GAMMA; pr.lambdas = lambdas.data(); pr.n_categories = (int)K; pr.multipliers = multipliers.data();
    const cladevector& order = _dev.order();
    const size_t n = order.size(), F = families.size();
    std::vector<int32_t> states(K * F * n);
    if (cafe_reconstruct(ctx, &pr, root_prior.data(), states.data()) != CAFE_OK) fail("cafe_reconstruct", ctx);
    auto result = new gamma_model_reconstruction(multipliers);
    for (size_t f = 0; f < F; ++f) {
        auto& r = result->_reconstructions[families[f].id()];
        r._category_likelihoods = _hip_category_likelihoods[f];
        r.category_reconstruction.resize(K);
        for (size_t k = 0; k < K; ++k)
            for (size_t v = 0; v < n; ++v)
                if (!order[v]->is_leaf()) r.category_reconstruction[k][order[v]] = states[(k * F + f) * n + v];
        r.reconstruction = get_weighted_averages(r.category_reconstruction, cat_probs);     // gamma_core.cpp:341
    }
    _monitor.Event_Reconstruction_Complete();
    return result;
}

// ------------------------------------------------------------------------------- Viterbi branch probabilities
branch_probabilities hip_compute_branch_probabilities(model* p_model, const hip_device_context& dev, const reconstruction* rec,
                                                      const std::vector<gene_family>& families, const std::vector<double>& pvalues,
                                                      double test_pvalue) {
    branch_probabilities probs;
    cafe_ctx* ctx = dev.get();
    const cladevector& order = dev.order();
    if (!ctx) throw std::runtime_error("hip_compute_branch_probabilities: no device context (run infer_family_likelihoods first)");
    if (std::none_of(pvalues.begin(), pvalues.end(), [test_pvalue](double p) { return p < test_pvalue; })) return probs;
    const size_t n = order.size(), F = families.size();
    std::vector<int32_t> sizes(F * n);
    for (size_t f = 0; f < F; ++f)
        for (size_t v = 0; v < n; ++v) sizes[f * n + v] = rec->reconstructed_size(families[f], order[v]);
    const std::vector<double> lambdas = get_lambda_values(p_model->get_lambda());
    std::vector<double> table(F * n);
    cafe_params pr = {};
    pr.model = CAFE_MODEL_BASE; pr.lambdas = lambdas.data(); pr.n_categories = 1;
    if (cafe_branch_probabilities(ctx, &pr, sizes.data(), table.data()) != CAFE_OK) fail("cafe_branch_probabilities", ctx);
    for (size_t f = 0; f < F; ++f) {
        if (!(pvalues[f] < test_pvalue)) continue;                  // execute.cpp:167
        for (size_t v = 0; v < n; ++v) {
            const double p = table[f * n + v];
            probs.set(families[f], order[v], std::isnan(p) ? branch_probabilities::invalid() : branch_probabilities::branch_probability(p));
        }
    }
    return probs;
}
