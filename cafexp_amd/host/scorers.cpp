// Scorers: flat parameter vector -> model state -> model::infer_family_likelihoods.
// Parameter layouts and initial-guess distributions follow src/optimizer_scorer.cpp:19-170.
#include "cafe_host.h"

#include <cmath>
#include <iostream>
#include <limits>

namespace cafe {

std::mt19937 randomizer_engine(std::random_device{}());

double inference_optimizer_scorer::calculate_score(const double* values) {
    prepare_calculation(values);
    if (!quiet) report_precalculation();
    double score = _p_model->infer_family_likelihoods(_p_distribution, _rootdist_map, _p_lambda);
    if (std::isnan(score)) score = std::numeric_limits<double>::infinity();      // optimizer_scorer.cpp:30
    return score;
}

// 1/longest_branch times a normal draw centred so that lambda starts near 0.002 (optimizer_scorer.cpp:37-52)
std::vector<double> lambda_optimizer::initial_guesses() {
    const double distmean = 0.002 / (1.0 / _longest_branch);
    std::vector<double> result(_p_lambda->count());
    std::normal_distribution<double> distribution(distmean, 0.2);
    for (auto& v : result) {
        v = 1.0 / _longest_branch * distribution(randomizer_engine);
        while (v < 0) v = 1.0 / _longest_branch * distribution(randomizer_engine);
    }
    return result;
}

void lambda_optimizer::report_precalculation() { std::cout << "Lambda: " << _p_lambda->to_string() << std::endl; }

std::vector<double> lambda_epsilon_optimizer::initial_guesses() {
    std::vector<double> result = _lambda_optimizer.initial_guesses();
    current_guesses = _p_error_model->get_epsilons();
    result.insert(result.end(), current_guesses.begin(), current_guesses.end());
    return result;
}

void lambda_epsilon_optimizer::prepare_calculation(const double* values) {
    const double* epsilons = values + _p_lambda->count();
    _lambda_optimizer.prepare_calculation(values);
    std::map<double, double> replacements;                    // old epsilon -> new epsilon, by value
    for (size_t i = 0; i < current_guesses.size(); ++i) {
        replacements[current_guesses[i]] = epsilons[i];
        current_guesses[i] = epsilons[i];
    }
    _p_error_model->replace_epsilons(replacements);
}

void lambda_epsilon_optimizer::report_precalculation() {
    std::cout << "Calculating probability: epsilon=" << _p_error_model->get_epsilons().back() * 2.0 << ", lambda=" << _p_lambda->to_string() << std::endl;
}

void lambda_epsilon_optimizer::finalize(double* results) {
    _lambda_optimizer.finalize(results);
    _p_error_model->update_single_epsilon(results[_p_lambda->count()]);
}

// alpha starts from Gamma(4, 0.25): mean 1 (optimizer_scorer.cpp:116-121)
std::vector<double> gamma_optimizer::initial_guesses() {
    std::gamma_distribution<double> distribution(4.0, 0.25);
    return std::vector<double>({distribution(randomizer_engine)});
}

void gamma_optimizer::report_precalculation() { std::cout << "Attempting alpha: " << _p_gamma_model->get_alpha() << std::endl; }

std::vector<double> gamma_lambda_optimizer::initial_guesses() {
    std::vector<double> values = _lambda_optimizer.initial_guesses();
    std::vector<double> alpha = _gamma_optimizer.initial_guesses();
    values.insert(values.end(), alpha.begin(), alpha.end());
    return values;
}

void gamma_lambda_optimizer::prepare_calculation(const double* values) {
    _lambda_optimizer.prepare_calculation(values);
    _gamma_optimizer.prepare_calculation(values + _p_lambda->count());
}

void gamma_lambda_optimizer::report_precalculation() {
    std::cout << "Attempting lambda: " << _p_lambda->to_string() << ", alpha: " << _gamma_optimizer.get_alpha() << std::endl;
}

void gamma_lambda_optimizer::finalize(double* results) {
    _lambda_optimizer.finalize(results);
    _gamma_optimizer.finalize(results + _p_lambda->count());
}

// ---------------------------------------------------------------- empirical Poisson prior (src/poisson.cpp)
static double poisspdf(int x, double lambda) { return std::exp(x * std::log(lambda) - std::lgamma(x + 1) - lambda); }

poisson_scorer::poisson_scorer(const std::vector<gene_family>& gene_families) {
    for (auto& fam : gene_families)
        for (const auto& species : fam.get_species())
            if (fam.get_species_size(species) > 0) leaf_family_sizes.push_back(fam.get_species_size(species) - 1);
}

std::vector<double> poisson_scorer::initial_guesses() {
    std::uniform_real_distribution<double> distribution(0.0, 1.0);
    return std::vector<double>{distribution(randomizer_engine)};
}

double poisson_scorer::lnLPoisson(const double* plambda) {
    const double lambda = plambda[0];
    double score = 0.0;
    for (int sz : leaf_family_sizes) {
        const double ll = poisspdf(sz, lambda);
        if (std::isnan(ll) || std::isinf(ll) || ll == 0) continue;      // incalculable sizes are skipped (poisson.cpp:69)
        score += std::log(ll);
    }
    return -score;
}

poisson_distribution::poisson_distribution(const std::vector<gene_family>* p_gene_families) {
    poisson_scorer scorer(*p_gene_families);
    optimizer opt(&scorer);
    _lambda = opt.optimize().values[0];
}

}  // namespace cafe
