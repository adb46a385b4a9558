// TEST INFRASTRUCTURE -- not product code.
//
// Driver for the *real* reference (Han9527/CAFExp), compiled in place from
// /root/reference/src/*.cpp by oracle/Makefile into oracle/_ref/ (git-ignored).
// This file is our own code: it only calls the reference's public API
// (src/probability.h, src/matrix_cache.h, src/core.h, src/base_model.h,
// src/gamma_core.h ...) the way the reference's own cafexp.cpp / test.cpp do.
// No reference source is copied into the repo.
//
// Usage: ref_harness <job> [key=value ...]   -> one JSON object on stdout.
// Jobs:
//   bd        lambda= t= s= c=                   the_probability_of_going_from_parent_fam_size_to_c
//   bdlog     s= c= log_alpha= coeff=            birthdeath_rate_with_log_alpha
//   matrix    n= lambda= t=                      matrix_cache::precalculate_matrices + get_matrix (key-quantized)
//   gamma     k= alpha=                          get_gamma (PAML discrete gamma)
//   key       lambda= t=                         matrix_cache_key de-quantized values
//   prune     newick= counts=A:3,B:6 lambda= mult= m= r= [errfile=]   inference_prune root vector
//   score     tree= families= model=base|gamma [lambda=|lambdas=a,b lambda_tree=] [k= alpha=]
//             [errfile=] [prior=uniform|poisson:X] [rootdist=file] [rootfilter=1] [limit=N]
//             [m= r=]  -> -lnL, per-family table, wall seconds
//   time_matrices n= lambda= count= t0=          wall seconds of precalculate_matrices for `count` branch lengths
//   reconstruct tree= families= [lambda=|lambdas= lambda_tree=] [model=gamma k= alpha=] [errfile=] [prior=] [limit=] [m= r=]
//             [nsim= seed= pvalue=0.05 files=1]   the tail of estimator::execute (execute.cpp:147-180): infer once,
//             [compute_pvalues,] reconstruct_ancestral_states, compute_viterbi_sum for every family and node;
//             files=1 adds the text of every report reconstruction::write_results writes
//   search    tree= families= [model=gamma k= [alpha=]] [lambda_tree=] [errfile=|estimate_error=1] [prior=] [limit=] [seed=10]
//             estimator::estimate_missing_variables (execute.cpp:82-105): model::get_lambda_optimizer + the reference's
//             optimizer (Nelder-Mead, optimizer.cpp:539) at a fixed seed of the global engine
//   Every job that builds a model (score, reconstruct, search) takes device=hip when this file is compiled with
//   -DCAFE_HIP_BINDING (oracle/_ref/ref_hip_harness): the model is then integration/hip_models.h's hip_base_model /
//   hip_gamma_model -- the reference's own classes, optimizer and writers running on the MI355X library.
//   cafexp    <the reference program's own command line>   runs cafexp() (src/cafexp.cpp:175) unchanged, e.g.
//             ref_harness cafexp -t tree -i families -k 3 -o /tmp/out ; prints {"rc": .., "seconds": ..}
//   pvalues   tree= families= lambda=|lambdas= lambda_tree= [nsim=1000] [seed=10] [ncond=0] [limit=] [m= r=]
//             compute_pvalues at a fixed seed of the global engine; ncond > 0 also prints the sorted conditional
//             distributions get_random_probabilities returns for root sizes 0..ncond-1 (same seed, same order)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <vector>
#include <omp.h>

#include "src/io.h"
#include "src/core.h"
#include "src/user_data.h"
#ifdef CAFE_HIP_BINDING
#include "hip_models.h"          // includes base_model.h and gamma_core.h (the latter has no include guard)
#else
#include "src/base_model.h"
#include "src/gamma_core.h"
#endif
#include "src/gamma.h"
#include "src/matrix_cache.h"
#include "src/probability.h"
#include "src/root_equilibrium_distribution.h"
#include "src/root_distribution.h"
#include "src/lambda.h"
#include "src/error_model.h"
#include "src/gene_family.h"
#include "src/gene_family_reconstructor.h"
#include "src/optimizer.h"
#include "src/optimizer_scorer.h"

std::mt19937 randomizer_engine(10);   // main.cpp:3 / test.cpp:35 define this global
int cafexp(int argc, char* const argv[]);   // src/cafexp.cpp:175
void init_lgamma_cache();             // probability.cpp:66

typedef std::map<std::string, std::string> kv_t;

static kv_t parse_args(int argc, char** argv, int first) {
    kv_t kv;
    for (int i = first; i < argc; ++i) {
        std::string a(argv[i]);
        size_t eq = a.find('=');
        if (eq == std::string::npos) { kv[a] = "1"; continue; }
        kv[a.substr(0, eq)] = a.substr(eq + 1);
    }
    return kv;
}
static bool has(const kv_t& kv, const char* k) { return kv.find(k) != kv.end(); }
static std::string gets(const kv_t& kv, const char* k, const char* dflt = "") {
    auto it = kv.find(k); return it == kv.end() ? std::string(dflt) : it->second;
}
static double getd(const kv_t& kv, const char* k, double dflt = 0) {
    auto it = kv.find(k); return it == kv.end() ? dflt : std::stod(it->second);
}
static int geti(const kv_t& kv, const char* k, int dflt = 0) {
    auto it = kv.find(k); return it == kv.end() ? dflt : std::stoi(it->second);
}
static void pd(double v) {
    if (std::isinf(v)) printf(v > 0 ? "\"inf\"" : "\"-inf\"");
    else if (std::isnan(v)) printf("\"nan\"");
    else printf("%.17g", v);
}
static void parr(const char* name, const std::vector<double>& v) {
    printf("\"%s\": [", name);
    for (size_t i = 0; i < v.size(); ++i) { if (i) printf(", "); pd(v[i]); }
    printf("]");
}
static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// base_model / gamma_model as build_models (core.cpp:16-50) would construct them; device=hip: the binding's classes
static model* make_model(const kv_t& kv, user_data& d) {
    const std::string mdl = gets(kv, "model", "base");
    const bool hip = gets(kv, "device", "cpu") == "hip";
#ifndef CAFE_HIP_BINDING
    if (hip) throw std::runtime_error("device=hip needs the ref_hip_harness build (-DCAFE_HIP_BINDING)");
#endif
    if (mdl == "gamma") {
#ifdef CAFE_HIP_BINDING
        if (hip) return new hip_gamma_model(d.p_lambda, d.p_tree, &d.gene_families, d.max_family_size, d.max_root_family_size,
                                            geti(kv, "k"), getd(kv, "alpha"), d.p_error_model, 0, geti(kv, "gpus", 1));
#endif
        return new gamma_model(d.p_lambda, d.p_tree, &d.gene_families, d.max_family_size, d.max_root_family_size,
                               geti(kv, "k"), getd(kv, "alpha"), d.p_error_model);
    }
#ifdef CAFE_HIP_BINDING
    if (hip) return new hip_base_model(d.p_lambda, d.p_tree, &d.gene_families, d.max_family_size, d.max_root_family_size, d.p_error_model, 0, geti(kv, "gpus", 1));
#endif
    return new base_model(d.p_lambda, d.p_tree, &d.gene_families, d.max_family_size, d.max_root_family_size, d.p_error_model);
}

static int job_bd(const kv_t& kv) {
    double v = the_probability_of_going_from_parent_fam_size_to_c(getd(kv, "lambda"), getd(kv, "t"), geti(kv, "s"), geti(kv, "c"));
    printf("{\"value\": "); pd(v); printf("}\n");
    return 0;
}
static int job_bdlog(const kv_t& kv) {
    double v = birthdeath_rate_with_log_alpha(geti(kv, "s"), geti(kv, "c"), getd(kv, "log_alpha"), getd(kv, "coeff"));
    printf("{\"value\": "); pd(v); printf("}\n");
    return 0;
}
static int job_key(const kv_t& kv) {
    matrix_cache_key key(1, getd(kv, "lambda"), getd(kv, "t"));
    printf("{\"lambda_q\": "); pd(key.lambda()); printf(", \"t_q\": "); pd(key.branch_length()); printf("}\n");
    return 0;
}
static int job_matrix(const kv_t& kv) {
    int n = geti(kv, "n");
    double lambda = getd(kv, "lambda"), t = getd(kv, "t");
    matrix_cache calc(n);
    calc.precalculate_matrices({ lambda }, std::set<double>{ t });
    const matrix* m = calc.get_matrix(t, lambda);
    std::vector<double> flat((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) flat[(size_t)i * n + j] = m->get(i, j);
    printf("{\"n\": %d, ", n); parr("values", flat); printf("}\n");
    return 0;
}
static int job_gamma(const kv_t& kv) {
    int k = geti(kv, "k");
    std::vector<double> freq(k), rate(k);
    get_gamma(freq, rate, getd(kv, "alpha"));
    printf("{"); parr("cat_probs", freq); printf(", "); parr("multipliers", rate); printf("}\n");
    return 0;
}
static error_model* load_error_model(const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("cannot open " + path);
    error_model* em = new error_model;
    read_error_model_file(f, em);
    return em;
}
static int job_prune(const kv_t& kv) {
    std::unique_ptr<clade> tree(parse_newick(gets(kv, "newick"), false));
    gene_family fam;
    std::stringstream ss(gets(kv, "counts"));
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        size_t c = tok.find(':');
        fam.set_species_size(tok.substr(0, c), std::stoi(tok.substr(c + 1)));
    }
    double lambda = getd(kv, "lambda"), mult = getd(kv, "mult", 1.0);
    int M = geti(kv, "m"), R = geti(kv, "r");
    single_lambda lam(lambda);
    std::unique_ptr<error_model> em;
    if (has(kv, "errfile")) em.reset(load_error_model(gets(kv, "errfile")));
    matrix_cache cache(std::max(M, R) + 1);
    cache.precalculate_matrices({ lambda * mult }, tree->get_branch_lengths());
    auto v = inference_prune(fam, cache, &lam, em.get(), tree.get(), mult, R, M);
    printf("{"); parr("root", v); printf("}\n");
    return 0;
}

static int job_score(const kv_t& kv) {
    input_parameters p;
    p.tree_file_path = gets(kv, "tree");
    p.input_file_path = gets(kv, "families");
    if (has(kv, "lambda")) p.fixed_lambda = getd(kv, "lambda");
    if (has(kv, "lambdas")) { p.fixed_multiple_lambdas = gets(kv, "lambdas"); p.lambda_tree_file_path = gets(kv, "lambda_tree"); }
    if (has(kv, "errfile")) { p.use_error_model = true; p.error_model_file_path = gets(kv, "errfile"); }
    if (has(kv, "rootdist")) p.rootdist = gets(kv, "rootdist");
    user_data d;
    d.read_datafiles(p);
    if (geti(kv, "rootfilter", 1)) {    // cafexp.cpp:189-199
        auto rem = std::remove_if(d.gene_families.begin(), d.gene_families.end(), [&](const gene_family& fam) {
            return !fam.exists_at_root(d.p_tree); });
        d.gene_families.erase(rem, d.gene_families.end());
    }
    if (has(kv, "limit")) {
        size_t lim = (size_t)geti(kv, "limit");
        if (d.gene_families.size() > lim) d.gene_families.resize(lim);
    }
    if (has(kv, "m")) d.max_family_size = geti(kv, "m");
    if (has(kv, "r")) d.max_root_family_size = geti(kv, "r");

    std::unique_ptr<root_equilibrium_distribution> prior;
    std::string pr = gets(kv, "prior", "uniform");
    if (pr == "uniform") prior.reset(new uniform_distribution());
    else prior.reset(new ::poisson_distribution(std::stod(pr.substr(pr.find(':') + 1))));

    std::string mdl = gets(kv, "model", "base");
    std::unique_ptr<model> m(make_model(kv, d));
    std::vector<double> mults;
    if (auto g = dynamic_cast<gamma_model*>(m.get())) mults = g->get_lambda_multipliers();
    int reps = geti(kv, "reps", 1);
    double score = 0, best = 1e300;
    for (int r = 0; r < reps; ++r) {
        double t0 = now();
        score = m->infer_family_likelihoods(prior.get(), d.rootdist, d.p_lambda);
        double dt = now() - t0;
        if (dt < best) best = dt;
    }
    printf("{\"neg_lnl\": "); pd(score);
    printf(", \"n_families\": %zu, \"max_family_size\": %d, \"max_root_family_size\": %d, \"seconds\": %.6f, \"threads\": %d",
        d.gene_families.size(), d.max_family_size, d.max_root_family_size, best, omp_get_max_threads());
    if (!mults.empty()) { printf(", "); parr("multipliers", mults); }
    if (geti(kv, "per_family", 0) && !std::isinf(score)) {
        std::ostringstream ost;
        ost.precision(17);
        m->write_family_likelihoods(ost);
        // base: "id\tlnL"; gamma: "id\tmult\tcatlik\tfamlik\tpost\tsig" one row per (family, category)
        std::istringstream ist(ost.str());
        std::string line;
        std::getline(ist, line);
        std::vector<double> a, b, c;
        while (std::getline(ist, line)) {
            std::vector<std::string> tk = tokenize_str(line, '\t');
            // (strtod, not std::stod: the latter throws on denormal values, which the borderline fixtures contain)
            auto num = [](const std::string& t) { return std::strtod(t.c_str(), nullptr); };
            if (mdl == "gamma") { a.push_back(num(tk[2])); b.push_back(num(tk[3])); c.push_back(num(tk[4])); }
            else a.push_back(num(tk[1]));
        }
        if (mdl == "gamma") { printf(", "); parr("category_likelihood", a); printf(", "); parr("family_likelihood", b); printf(", "); parr("posterior", c); }
        else { printf(", "); parr("family_lnl", a); }
    }
    if (geti(kv, "files", 0)) {
        // the two files estimator::compute writes (execute.cpp:49-54), verbatim, default stream precision
        std::ostringstream vital, fam;
        m->write_vital_statistics(vital, score);
        m->write_family_likelihoods(fam);
        auto esc = [](const std::string& t) {
            std::string o;
            for (char ch : t) {
                if (ch == '\n') o += "\\n"; else if (ch == '\t') o += "\\t"; else if (ch == '"') o += "\\\""; else if (ch == '\\') o += "\\\\"; else o += ch;
            }
            return o;
        };
        printf(", \"results_txt\": \"%s\", \"family_likelihoods_txt\": \"%s\"", esc(vital.str()).c_str(), esc(fam.str()).c_str());
    }
    printf("}\n");
    return 0;
}

static int job_pvalues(const kv_t& kv) {
    input_parameters p;
    p.tree_file_path = gets(kv, "tree");
    p.input_file_path = gets(kv, "families");
    if (has(kv, "lambda")) p.fixed_lambda = getd(kv, "lambda");
    if (has(kv, "lambdas")) { p.fixed_multiple_lambdas = gets(kv, "lambdas"); p.lambda_tree_file_path = gets(kv, "lambda_tree"); }
    user_data d;
    d.read_datafiles(p);
    if (geti(kv, "rootfilter", 1)) {
        auto rem = std::remove_if(d.gene_families.begin(), d.gene_families.end(), [&](const gene_family& fam) {
            return !fam.exists_at_root(d.p_tree); });
        d.gene_families.erase(rem, d.gene_families.end());
    }
    if (has(kv, "limit")) {
        size_t lim = (size_t)geti(kv, "limit");
        if (d.gene_families.size() > lim) d.gene_families.resize(lim);
    }
    if (has(kv, "m")) d.max_family_size = geti(kv, "m");
    if (has(kv, "r")) d.max_root_family_size = geti(kv, "r");
    const int nsim = geti(kv, "nsim", 1000), seed = geti(kv, "seed", 10), ncond = geti(kv, "ncond", 0);
    // execute.cpp:158-161
    matrix_cache cache(std::max(d.max_family_size, d.max_root_family_size) + 1);
    cache.precalculate_matrices(get_lambda_values(d.p_lambda), d.p_tree->get_branch_lengths());
    randomizer_engine.seed(seed);
    double t0 = now();
    auto pv = compute_pvalues(d.p_tree, d.gene_families, d.p_lambda, cache, nsim, d.max_family_size, d.max_root_family_size);
    double dt = now() - t0;
    printf("{\"n_families\": %zu, \"max_family_size\": %d, \"max_root_family_size\": %d, \"nsim\": %d, \"seed\": %d, \"seconds\": %.6f, \"threads\": %d, ",
        d.gene_families.size(), d.max_family_size, d.max_root_family_size, nsim, seed, dt, omp_get_max_threads());
    parr("pvalues", pv);
    if (ncond > 0) {
        randomizer_engine.seed(seed);
        std::vector<double> flat;
        for (int i = 0; i < ncond; ++i) {
            auto c = get_random_probabilities(d.p_tree, nsim, i, d.max_family_size, d.max_root_family_size, d.p_lambda, cache, NULL);
            flat.insert(flat.end(), c.begin(), c.end());
        }
        printf(", \"ncond\": %d, ", ncond);
        parr("cond", flat);
    }
    printf("}\n");
    return 0;
}

static std::string json_escape(const std::string& t) {
    std::string o;
    for (char ch : t) {
        if (ch == '\n') o += "\\n"; else if (ch == '\t') o += "\\t"; else if (ch == '"') o += "\\\""; else if (ch == '\\') o += "\\\\"; else o += ch;
    }
    return o;
}

static int job_reconstruct(const kv_t& kv) {
    input_parameters p;
    p.tree_file_path = gets(kv, "tree");
    p.input_file_path = gets(kv, "families");
    if (has(kv, "lambda")) p.fixed_lambda = getd(kv, "lambda");
    if (has(kv, "lambdas")) { p.fixed_multiple_lambdas = gets(kv, "lambdas"); p.lambda_tree_file_path = gets(kv, "lambda_tree"); }
    if (has(kv, "errfile")) { p.use_error_model = true; p.error_model_file_path = gets(kv, "errfile"); }
    user_data d;
    d.read_datafiles(p);
    if (geti(kv, "rootfilter", 1)) {
        auto rem = std::remove_if(d.gene_families.begin(), d.gene_families.end(), [&](const gene_family& fam) {
            return !fam.exists_at_root(d.p_tree); });
        d.gene_families.erase(rem, d.gene_families.end());
    }
    if (has(kv, "limit")) {
        size_t lim = (size_t)geti(kv, "limit");
        if (d.gene_families.size() > lim) d.gene_families.resize(lim);
    }
    if (has(kv, "m")) d.max_family_size = geti(kv, "m");
    if (has(kv, "r")) d.max_root_family_size = geti(kv, "r");
    std::unique_ptr<root_equilibrium_distribution> prior;
    std::string prs = gets(kv, "prior", "uniform");
    if (prs == "uniform") prior.reset(new uniform_distribution());
    else prior.reset(new ::poisson_distribution(std::stod(prs.substr(prs.find(':') + 1))));
    std::unique_ptr<model> m(make_model(kv, d));
    const double test_pvalue = getd(kv, "pvalue", 0.05);
    const int nsim = geti(kv, "nsim", 0);

    double score = m->infer_family_likelihoods(prior.get(), d.rootdist, d.p_lambda);       // compute(), execute.cpp:49
    matrix_cache cache(std::max(d.max_family_size, d.max_root_family_size) + 1);             // execute.cpp:157-158
    cache.precalculate_matrices(get_lambda_values(m->get_lambda()), d.p_tree->get_branch_lengths());
    std::vector<double> pvalues(d.gene_families.size(), 1.0);
    if (nsim > 0) {
        randomizer_engine.seed(geti(kv, "seed", 10));
        pvalues = compute_pvalues(d.p_tree, d.gene_families, m->get_lambda(), cache, nsim, d.max_family_size, d.max_root_family_size);
    }
    double t0 = now();
    std::unique_ptr<reconstruction> rec(m->reconstruct_ancestral_states(d.gene_families, &cache, prior.get()));
    double dt = now() - t0;

    cladevector order;
    d.p_tree->apply_reverse_level_order([&order](const clade* c) { order.push_back(c); });
    branch_probabilities probs;                                                              // execute.cpp:165-176
    bool probs_done = false;
#ifdef CAFE_HIP_BINDING
    if (auto hb = dynamic_cast<hip_base_model*>(m.get())) { probs = hip_compute_branch_probabilities(hb, hb->device_context(), rec.get(), d.gene_families, pvalues, test_pvalue); probs_done = true; }
    if (auto hg = dynamic_cast<hip_gamma_model*>(m.get())) { probs = hip_compute_branch_probabilities(hg, hg->device_context(), rec.get(), d.gene_families, pvalues, test_pvalue); probs_done = true; }
#endif
    for (size_t i = 0; !probs_done && i < d.gene_families.size(); ++i)
        if (pvalues[i] < test_pvalue)
            for (auto c : order) probs.set(d.gene_families[i], c, compute_viterbi_sum(c, d.gene_families[i], rec.get(), d.max_family_size, cache, m->get_lambda()));

    printf("{\"neg_lnl\": "); pd(score);
    printf(", \"n_families\": %zu, \"max_family_size\": %d, \"max_root_family_size\": %d, \"seconds\": %.6f, \"threads\": %d, \"nodes\": [",
        d.gene_families.size(), d.max_family_size, d.max_root_family_size, dt, omp_get_max_threads());
    for (size_t i = 0; i < order.size(); ++i) printf("%s\"%s\"", i ? ", " : "", order[i]->get_taxon_name().c_str());
    printf("], \"states\": [");
    bool first = true;
    for (auto& gf : d.gene_families)
        for (auto c : order) { printf("%s%d", first ? "" : ",", rec->reconstructed_size(gf, c)); first = false; }
    printf("]");
    if (auto g = dynamic_cast<gamma_model_reconstruction*>(rec.get())) {
        printf(", \"category_states\": [");       // [family][category][node]
        first = true;
        for (auto& gf : d.gene_families) {
            auto& r = g->_reconstructions.at(gf.id());
            for (auto& cat : r.category_reconstruction)
                for (auto c : order) {
                    int v = c->is_leaf() ? gf.get_species_size(c->get_taxon_name()) : cat.at(c);
                    printf("%s%d", first ? "" : ",", v); first = false;
                }
        }
        printf("], \"averages\": [");             // [family][node], the weighted averages before rounding
        first = true;
        for (auto& gf : d.gene_families) {
            auto& r = g->_reconstructions.at(gf.id());
            for (auto c : order) {
                double v = c->is_leaf() ? gf.get_species_size(c->get_taxon_name()) : r.reconstruction.at(c);
                printf("%s", first ? "" : ","); pd(v); first = false;
            }
        }
        printf("]");
    }
    std::vector<double> bp;
    for (auto& gf : d.gene_families)
        for (auto c : order) {
            auto r = compute_viterbi_sum(c, gf, rec.get(), d.max_family_size, cache, m->get_lambda());
            bp.push_back(r._is_valid ? r._value : NAN);
        }
    printf(", "); parr("branch_probabilities", bp);
    if (nsim > 0) { printf(", "); parr("pvalues", pvalues); }
    if (geti(kv, "files", 0)) {
        std::ostringstream asr, count, change, famres, clres, bpf;
        rec->print_reconstructed_states(asr, order, d.gene_families, d.p_tree, test_pvalue, probs);
        rec->print_node_counts(count, order, d.gene_families, d.p_tree);
        rec->print_node_change(change, order, d.gene_families, d.p_tree);
        rec->print_increases_decreases_by_family(famres, order, d.gene_families, pvalues, test_pvalue);
        rec->print_increases_decreases_by_clade(clres, order, d.gene_families);
        print_branch_probabilities(bpf, order, d.gene_families, probs);
        printf(", \"asr_tre\": \"%s\", \"count_tab\": \"%s\", \"change_tab\": \"%s\", \"family_results_txt\": \"%s\", \"clade_results_txt\": \"%s\", \"branch_probabilities_tab\": \"%s\"",
            json_escape(asr.str()).c_str(), json_escape(count.str()).c_str(), json_escape(change.str()).c_str(), json_escape(famres.str()).c_str(),
            json_escape(clres.str()).c_str(), json_escape(bpf.str()).c_str());
        if (auto g = dynamic_cast<gamma_model_reconstruction*>(rec.get())) {
            std::ostringstream cl;
            g->print_category_likelihoods(cl, order, d.gene_families);
            printf(", \"category_likelihoods_txt\": \"%s\"", json_escape(cl.str()).c_str());
        }
    }
    printf("}\n");
    return 0;
}

static int job_search(const kv_t& kv) {
    input_parameters p;
    p.tree_file_path = gets(kv, "tree");
    p.input_file_path = gets(kv, "families");
    if (has(kv, "lambda_tree")) p.lambda_tree_file_path = gets(kv, "lambda_tree");
    if (has(kv, "errfile")) { p.use_error_model = true; p.error_model_file_path = gets(kv, "errfile"); }
    if (has(kv, "lambda")) p.fixed_lambda = getd(kv, "lambda");          // with model=gamma and no alpha: estimate alpha only
    user_data d;
    d.read_datafiles(p);
    if (geti(kv, "rootfilter", 1)) {
        auto rem = std::remove_if(d.gene_families.begin(), d.gene_families.end(), [&](const gene_family& fam) {
            return !fam.exists_at_root(d.p_tree); });
        d.gene_families.erase(rem, d.gene_families.end());
    }
    if (has(kv, "limit")) {
        size_t lim = (size_t)geti(kv, "limit");
        if (d.gene_families.size() > lim) d.gene_families.resize(lim);
    }
    if (has(kv, "m")) d.max_family_size = geti(kv, "m");
    if (has(kv, "r")) d.max_root_family_size = geti(kv, "r");
    std::string prs = gets(kv, "prior", "uniform");
    if (prs == "uniform") d.p_prior.reset(new uniform_distribution());
    else d.p_prior.reset(new ::poisson_distribution(std::stod(prs.substr(prs.find(':') + 1))));
    std::unique_ptr<error_model> default_em;
    const bool estimate_error = geti(kv, "estimate_error", 0) != 0;
    error_model* user_em = d.p_error_model;
    if (estimate_error) {                       // build_models' default error model when -e has no file (core.cpp:39-44)
        default_em.reset(new error_model());
        default_em->set_probabilities(0, { 0, .95, 0.05 });
        default_em->set_probabilities(d.max_family_size, { 0.05, .9, 0.05 });
        d.p_error_model = default_em.get();
    }
    std::unique_ptr<model> m(make_model(kv, d));
    d.p_error_model = estimate_error ? nullptr : user_em;   // user_data keeps "no error model given" (base_model.cpp:133)
    randomizer_engine.seed(geti(kv, "seed", 10));
    std::unique_ptr<inference_optimizer_scorer> scorer(m->get_lambda_optimizer(d));
    if (!scorer) { printf("{\"error\": \"nothing to optimise\"}\n"); return 1; }
    optimizer opt(scorer.get());
    opt.quiet = true;
    optimizer_parameters params;
    double t0 = now();
    auto result = opt.optimize(params);
    double dt = now() - t0;
    scorer->finalize(&result.values[0]);
    printf("{\"score\": "); pd(result.score);
    printf(", \"iterations\": %d, \"seconds\": %.6f, \"threads\": %d, \"n_families\": %zu, ", result.num_iterations, dt, omp_get_max_threads(), d.gene_families.size());
    parr("values", result.values);
    printf("}\n");
    return 0;
}

static int job_time_matrices(const kv_t& kv) {
    int n = geti(kv, "n"), count = geti(kv, "count", 1);
    double lambda = getd(kv, "lambda"), t0v = getd(kv, "t0", 1.0);
    std::set<double> bl;
    for (int i = 0; i < count; ++i) bl.insert(t0v + 0.731 * i);
    matrix_cache calc(n);
    double t0 = now();
    calc.precalculate_matrices({ lambda }, bl);
    double dt = now() - t0;
    printf("{\"seconds\": %.6f, \"count\": %d, \"n\": %d, \"threads\": %d}\n", dt, count, n, omp_get_max_threads());
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_harness <job> key=value...\n"); return 2; }
    std::string job(argv[1]);
    if (job == "cafexp") {                      // the whole program, its own argument parser (argv[0] = "cafexp")
        double t0 = now();
        int rc = cafexp(argc - 1, argv + 1);
        fflush(stdout);
        printf("\n{\"rc\": %d, \"seconds\": %.3f, \"threads\": %d}\n", rc, now() - t0, omp_get_max_threads());
        return rc;
    }
    init_lgamma_cache();
    kv_t kv = parse_args(argc, argv, 2);
    try {
        // the reference prints progress to cout (e.g. "Found root!", "Score (-lnL)") unless built -DSILENT
        if (job == "bd") return job_bd(kv);
        if (job == "bdlog") return job_bdlog(kv);
        if (job == "key") return job_key(kv);
        if (job == "matrix") return job_matrix(kv);
        if (job == "gamma") return job_gamma(kv);
        if (job == "prune") return job_prune(kv);
        if (job == "score") return job_score(kv);
        if (job == "time_matrices") return job_time_matrices(kv);
        if (job == "pvalues") return job_pvalues(kv);
        if (job == "reconstruct") return job_reconstruct(kv);
        if (job == "search") return job_search(kv);
    } catch (std::exception& e) {
        fprintf(stderr, "ref_harness: %s\n", e.what());
        return 1;
    }
    fprintf(stderr, "unknown job %s\n", job.c_str());
    return 2;
}
