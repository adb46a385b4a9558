// hip_base_model / hip_gamma_model: the reference's model interface (core.h:122-186) with
// infer_family_likelihoods forwarded to the C ABI (include/cafe_mi355x.h).  No likelihood
// arithmetic happens on the host; a missing GPU / library surfaces as std::runtime_error,
// like the reference's other fatal paths (matrix_cache.cpp:90-95).
#include "cafe_host.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <limits>

#include "../../include/cafe_mi355x.h"

namespace cafe {

hip_model_base::~hip_model_base() {
    if (_ctx) cafe_destroy(_ctx);
    if (_sharded) cafe_sharded_destroy(_sharded);
}

// devices: nullptr / 0 -> one context on `device` (returned through *ctx); else a sharded scorer over the listed devices
static void create_device_objects(const lambda* lam, const std::vector<const clade*>& order, const int32_t* counts, int64_t n_families,
                                  int max_family_size, int max_root_family_size, int max_categories, int n_deviations, int device,
                                  const std::vector<int>* devices, cafe_ctx** ctx, cafe_sharded** sharded);

cafe_ctx* create_device_context(const lambda* lam, const std::vector<const clade*>& order, const int32_t* counts, int64_t n_families,
                                 int max_family_size, int max_root_family_size, int max_categories, int n_deviations, int device) {
    cafe_ctx* ctx = nullptr;
    create_device_objects(lam, order, counts, n_families, max_family_size, max_root_family_size, max_categories, n_deviations, device, nullptr, &ctx, nullptr);
    return ctx;
}

static void create_device_objects(const lambda* lam, const std::vector<const clade*>& order, const int32_t* counts, int64_t n_families,
                                  int max_family_size, int max_root_family_size, int max_categories, int n_deviations, int device,
                                  const std::vector<int>* devices, cafe_ctx** out_ctx, cafe_sharded** out_sharded) {
    const int n = (int)order.size();
    std::map<const clade*, int> index;
    for (int i = 0; i < n; ++i) index[order[i]] = i;
    std::vector<int32_t> parent(n), lam_idx(n, 0), leaf_taxon(n, -1);
    std::vector<double> blen(n, 0.0);
    const multiple_lambda* ml = dynamic_cast<const multiple_lambda*>(lam);
    int T = 0;
    for (int i = 0; i < n; ++i) {
        const clade* c = order[i];
        parent[i] = c->is_root() ? -1 : index.at(c->get_parent());
        blen[i] = c->get_branch_length();
        if (ml) lam_idx[i] = ml->index_of(c);                    // by node name (lambda.cpp:34-36)
        if (c->is_leaf()) leaf_taxon[i] = T++;
    }
    cafe_problem pb{};
    pb.n_nodes = n; pb.parent = parent.data(); pb.branch_length = blen.data(); pb.lambda_index = lam_idx.data();
    pb.leaf_taxon = leaf_taxon.data(); pb.n_taxa = T; pb.n_families = n_families; pb.counts = counts;
    pb.max_family_size = max_family_size; pb.max_root_family_size = max_root_family_size;
    pb.n_lambdas = lam->count(); pb.single_lambda = ml ? 0 : 1; pb.max_categories = max_categories;
    pb.n_deviations = n_deviations;
    pb.device = device; pb.flags = 0; pb.workspace_limit = 0;
    char err[512];
    if (devices && !devices->empty() && out_sharded) {
        std::vector<int32_t> dev(devices->begin(), devices->end());
        *out_sharded = cafe_create_sharded(&pb, dev.data(), (int32_t)dev.size(), err, sizeof err);
        if (!*out_sharded) throw std::runtime_error(std::string("cafe_create_sharded: ") + err);
        return;
    }
    *out_ctx = cafe_create(&pb, err, sizeof err);
    if (!*out_ctx) throw std::runtime_error(std::string("cafe_create: ") + err);
}

static std::vector<int32_t> flatten_counts(const clade* tree, const std::vector<gene_family>& fams, std::vector<const clade*>& order) {
    order = tree->post_order();
    std::vector<const clade*> leaves;
    for (const clade* c : order) if (c->is_leaf()) leaves.push_back(c);
    const int T = (int)leaves.size();
    const int64_t F = (int64_t)fams.size();
    std::vector<int32_t> counts((size_t)F * T);
    for (int64_t f = 0; f < F; ++f)
        for (int t = 0; t < T; ++t) counts[(size_t)f * T + t] = fams[f].get_species_size(leaves[t]->get_taxon_name());
    return counts;
}

void hip_model_base::ensure_context(int max_categories) {
    const int sig = _p_lambda->count() * 2 + (dynamic_cast<const multiple_lambda*>(_p_lambda) ? 1 : 0);
    if (_ctx && max_categories <= _ctx_categories && sig == _ctx_lambda_sig) return;
    _ctx_lambda_sig = sig;
    if (_ctx) { cafe_destroy(_ctx); _ctx = nullptr; }
    if (!_p_tree || !_p_gene_families || _p_gene_families->empty())
        throw std::runtime_error("hip model: a tree and a non-empty family list are required");
    const std::vector<int32_t> counts = flatten_counts(_p_tree, *_p_gene_families, _order);
    _ctx = create_device_context(_p_lambda, _order, counts.data(), (int64_t)_p_gene_families->size(), _max_family_size, _max_root_family_size,
                                 max_categories, _p_error_model ? (int)_p_error_model->n_deviations() : 0, _device);
    _ctx_categories = max_categories;
}

void hip_model_base::ensure_scorer(int max_categories) {
    // (CAFE_FORCE_SHARDED: the multi-GPU scorer also for one device -- its plan, worker thread, communicator and gather
    // paths on a one-GPU box)
    if (_devices.size() <= 1 && std::getenv("CAFE_FORCE_SHARDED")) { if (_devices.empty()) _devices.push_back(_device); }
    else if (_devices.size() <= 1) { ensure_context(max_categories); return; }
    const int sig = _p_lambda->count() * 2 + (dynamic_cast<const multiple_lambda*>(_p_lambda) ? 1 : 0);
    if (_sharded && max_categories <= _sharded_categories && sig == _sharded_lambda_sig) return;
    _sharded_lambda_sig = sig;
    if (_sharded) { cafe_sharded_destroy(_sharded); _sharded = nullptr; }
    if (!_p_tree || !_p_gene_families || _p_gene_families->empty())
        throw std::runtime_error("hip model: a tree and a non-empty family list are required");
    const std::vector<int32_t> counts = flatten_counts(_p_tree, *_p_gene_families, _order);
    create_device_objects(_p_lambda, _order, counts.data(), (int64_t)_p_gene_families->size(), _max_family_size, _max_root_family_size, max_categories,
                          _p_error_model ? (int)_p_error_model->n_deviations() : 0, _device, &_devices, nullptr, &_sharded);
    _sharded_categories = max_categories;
}

int hip_model_base::score_call(const cafe_params* pr, double* score) {
    return _sharded ? cafe_sharded_score(_sharded, pr, score, nullptr) : cafe_score(_ctx, pr, score, nullptr);
}
int hip_model_base::family_results_call(const cafe_family_out* out) {
    return _sharded ? cafe_sharded_family_results(_sharded, out) : cafe_family_results(_ctx, out);
}
const char* hip_model_base::scorer_error() const {
    return _sharded ? cafe_sharded_last_error(_sharded) : cafe_last_error(_ctx);
}

void hip_model_base::gather_call_inputs(root_equilibrium_distribution* prior, const std::map<int, int>& rootdist, std::vector<float>& prior_f,
                                        std::vector<double>& err_table, std::vector<double>& lambdas) const {
    root_distribution rd;                                       // base_model.cpp:62-72 / gamma_core.cpp:182-192
    if (!rootdist.empty()) rd.vectorize(rootdist);
    else rd.vectorize_uniform(_max_root_family_size);
    prior->initialize(&rd);
    prior_f.resize(_max_root_family_size);
    for (int j = 0; j < _max_root_family_size; ++j) prior_f[j] = prior->compute(j);
    err_table.clear();
    if (_p_error_model) {
        const size_t nd = _p_error_model->n_deviations();
        err_table.resize((size_t)(_max_family_size + 1) * nd);
        for (int x = 0; x <= _max_family_size; ++x) {
            const std::vector<double> p = _p_error_model->get_probs(x);
            std::copy(p.begin(), p.begin() + nd, err_table.begin() + (size_t)x * nd);
        }
    }
    lambdas = _p_lambda->values();
}

// ------------------------------------------------------------------------------------------ base
double hip_base_model::infer_family_likelihoods(root_equilibrium_distribution* prior, const std::map<int, int>& rootdist, const lambda*) {
    // like the reference this uses the member lambda, not the argument (base_model.cpp:56, :78)
    _monitor.attempts++;
    if (!_p_lambda->is_valid()) {
        _monitor.rejects++;
        return std::numeric_limits<double>::infinity();
    }
    ensure_scorer(1);
    std::vector<float> prior_f;
    std::vector<double> err, lambdas;
    gather_call_inputs(prior, rootdist, prior_f, err, lambdas);
    cafe_params pr{};
    pr.model = CAFE_MODEL_BASE; pr.lambdas = lambdas.data(); pr.n_categories = 1; pr.prior = prior_f.data();
    pr.error_model = err.empty() ? nullptr : err.data();
    const size_t F = _p_gene_families->size();
    std::vector<double> lnl(F);
    cafe_family_out out{};
    out.family_lnl = lnl.data();
    double score = 0;
    if (score_call(&pr, &score) != CAFE_OK) throw std::runtime_error(std::string("cafe_score: ") + scorer_error());
    results.resize(F);
    if (family_results_call(&out) == CAFE_OK)
        for (size_t i = 0; i < F; ++i) {                         // results[i] = {id, 0, 0, 0, lnL_i, false} (base_model.cpp:105)
            results[i] = family_info_stash();
            results[i].family_id = (*_p_gene_families)[i].id();
            results[i].posterior_probability = lnl[i];
        }
    return score;
}

void hip_base_model::write_family_likelihoods(std::ostream& ost) {
    ost << "#FamilyID\tLikelihood of Family" << std::endl;
    for (const auto& r : results) ost << r.family_id << "\t" << r.posterior_probability << std::endl;
}

// ------------------------------------------------------------------------------------------ gamma
hip_gamma_model::hip_gamma_model(lambda* l, const clade* tree, const std::vector<gene_family>* fams, int max_family_size, int max_root_family_size,
                                 int n_gamma_cats, double fixed_alpha, error_model* em)
    : hip_model_base(l, tree, fams, max_family_size, max_root_family_size, em) {
    _gamma_cat_probs.resize(n_gamma_cats);
    _lambda_multipliers.resize(n_gamma_cats);
    if (fams) _category_likelihoods.resize(fams->size());
    set_alpha(fixed_alpha);
}

hip_gamma_model::hip_gamma_model(lambda* l, const clade* tree, const std::vector<gene_family>* fams, int max_family_size, int max_root_family_size,
                                 const std::vector<double>& gamma_categories, const std::vector<double>& multipliers, error_model* em)
    : hip_model_base(l, tree, fams, max_family_size, max_root_family_size, em), _lambda_multipliers(multipliers), _gamma_cat_probs(gamma_categories), _alpha(0) {
    if (fams) _category_likelihoods.resize(fams->size());
}

void hip_gamma_model::set_alpha(double alpha) {
    _alpha = alpha;
    if (_gamma_cat_probs.size() > 1) get_gamma(_gamma_cat_probs, _lambda_multipliers, alpha);
}

bool hip_gamma_model::can_infer() const {
    if (!_p_lambda->is_valid()) return false;
    if (_alpha < 0) return false;
    const std::vector<double> v = _p_lambda->values();
    const std::set<double> lengths = _p_tree->get_branch_lengths();
    const double longest = *std::max_element(lengths.begin(), lengths.end());
    const double lm = *std::max_element(_lambda_multipliers.begin(), _lambda_multipliers.end());
    const double ll = *std::max_element(v.begin(), v.end());
    const double lambda = lm * ll;
    const double a = lambda * longest / (1 + lambda * longest);             // matrix_cache::is_saturated
    return !((1 - 2 * a) < 0);
}

double hip_gamma_model::infer_family_likelihoods(root_equilibrium_distribution* prior, const std::map<int, int>& rootdist, const lambda*) {
    _monitor.attempts++;
    results.clear();
    if (!can_infer()) {
        _monitor.rejects++;
        return std::numeric_limits<double>::infinity();
    }
    const int K = (int)_gamma_cat_probs.size();
    ensure_scorer(K);
    std::vector<float> prior_f;
    std::vector<double> err, lambdas;
    gather_call_inputs(prior, rootdist, prior_f, err, lambdas);
    cafe_params pr{};
    pr.model = CAFE_MODEL_GAMMA; pr.lambdas = lambdas.data(); pr.n_categories = K; pr.multipliers = _lambda_multipliers.data();
    pr.cat_probs = _gamma_cat_probs.data(); pr.alpha = _alpha; pr.prior = prior_f.data(); pr.error_model = err.empty() ? nullptr : err.data();
    double score = 0;
    if (score_call(&pr, &score) != CAFE_OK) throw std::runtime_error(std::string("cafe_score: ") + scorer_error());
    const size_t F = _p_gene_families->size();
    std::vector<double> cat((size_t)F * K), fam(F);
    std::vector<int32_t> failed(F);
    cafe_family_out out{};
    out.category_likelihood = cat.data(); out.family_likelihood = fam.data(); out.failed = failed.data();
    if (family_results_call(&out) != CAFE_OK) return score;                 // rejected on the device side without results
    if (std::isinf(score)) {                                                // a category summed to zero: gamma_core.cpp:227-236
        for (size_t i = 0; i < F; ++i)
            if (failed[i]) _monitor.failure_count[(*_p_gene_families)[i].id()]++;
        return score;
    }
    _category_likelihoods.assign(F, std::vector<double>(K));
    for (size_t i = 0; i < F; ++i) {
        double denom = 0;                                                   // get_posterior_probabilities, gamma_core.cpp:97
        for (int k = 0; k < K; ++k) denom += cat[i * K + k] * _gamma_cat_probs[k];
        for (int k = 0; k < K; ++k) {
            family_info_stash s;
            s.family_id = (*_p_gene_families)[i].id();
            s.lambda_multiplier = _lambda_multipliers[k];
            s.category_likelihood = cat[i * K + k];
            s.family_likelihood = fam[i];
            s.posterior_probability = cat[i * K + k] * _gamma_cat_probs[k] / denom;
            s.significant = s.posterior_probability > 0.95;
            results.push_back(s);
            _category_likelihoods[i][k] = cat[i * K + k];
        }
    }
    return score;
}

void hip_gamma_model::write_vital_statistics(std::ostream& ost, double final_likelihood) {
    model::write_vital_statistics(ost, final_likelihood);
    ost << "Alpha: " << _alpha << std::endl;
}

void hip_gamma_model::write_family_likelihoods(std::ostream& ost) {
    ost << "#FamilyID\tGamma Cat Median\tLikelihood of Category\tLikelihood of Family\tPosterior Probability\tSignificant" << std::endl;
    for (const auto& r : results) ost << r << "\n";
}

// scorer selection: base_model.cpp:123-141, gamma_core.cpp:250-280
inference_optimizer_scorer* hip_base_model::get_lambda_optimizer(user_data& data) {
    if (data.p_lambda) return nullptr;
    initialize_lambda(data.p_lambda_tree.get());
    const std::set<double> lengths = _p_tree->get_branch_lengths();
    const double longest = *std::max_element(lengths.begin(), lengths.end());
    if (_p_error_model && !data.p_error_model)
        return new lambda_epsilon_optimizer(this, _p_error_model, data.p_prior.get(), data.rootdist, _p_lambda, longest);
    return new lambda_optimizer(_p_lambda, this, data.p_prior.get(), longest, data.rootdist);
}

inference_optimizer_scorer* hip_gamma_model::get_lambda_optimizer(user_data& data) {
    const bool estimate_lambda = !data.p_lambda;
    const bool estimate_alpha = _alpha <= 0.0;
    const std::set<double> lengths = _p_tree->get_branch_lengths();
    const double longest = *std::max_element(lengths.begin(), lengths.end());
    if (estimate_lambda && estimate_alpha) {
        initialize_lambda(data.p_lambda_tree.get());
        return new gamma_lambda_optimizer(_p_lambda, this, data.p_prior.get(), data.rootdist, longest);
    }
    if (estimate_lambda) {
        initialize_lambda(data.p_lambda_tree.get());
        return new lambda_optimizer(_p_lambda, this, data.p_prior.get(), longest, data.rootdist);
    }
    if (estimate_alpha) {
        _p_lambda = data.p_lambda->clone();
        return new gamma_optimizer(this, data.p_prior.get(), data.rootdist);
    }
    return nullptr;
}

}  // namespace cafe
