// cafexp_hip: thin driver around the host adapter -- the part of `cafexp -t -i [-l|-m -y] [-k] [-a]
// [-e] [-p] [-f] [-z]` (src/cafexp.cpp:175, src/execute.cpp:42-150) that ends in scorer calls.
// With a fixed lambda it evaluates one infer_family_likelihoods call; without, it runs the
// Nelder-Mead search on the GPU scorer.  Output: one JSON object on stdout.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "cafe_host.h"

using namespace cafe;

static void usage() {
    std::fprintf(stderr,
        "usage: cafexp_hip -t TREE -i FAMILIES [-l LAMBDA | -m L1,L2,.. -y LAMBDA_TREE | -y LAMBDA_TREE] [-k K] [-a ALPHA]\n"
        "                  [-e [ERRMODEL]] [-p [POISSON_LAMBDA]] [-f ROOTDIST] [-z] [-s SEED] [-I MAXITER] [-d DEVICE | --gpus N] [--reps N] [--family-out FILE] [-o OUTDIR] [--limit N]\n"
        "                  [--pvalues NSIM [--pvalues-device] [--pvalues-out FILE] [--pvalues-cond FILE:K]] [--sizes M,R]\n"
        "                  [--reconstruct [-P PVALUE]]   (with -o: the reports of reconstruction::write_results)\n"
        "  --gpus N: the scorer calls shard the families over devices 0..N-1 (one host thread per GPU, one RCCL all-reduce per call)\n");
}

static std::string slurp_first_line(const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("Failed to open " + path);
    std::string line;
    std::getline(f, line);
    return line;
}

static void print_num(const char* key, double v, bool comma = true) {
    if (std::isinf(v)) std::printf("\"%s\": \"%sinf\"%s", key, v < 0 ? "-" : "", comma ? ", " : "");
    else if (std::isnan(v)) std::printf("\"%s\": \"nan\"%s", key, comma ? ", " : "");
    else std::printf("\"%s\": %.17g%s", key, v, comma ? ", " : "");
}

int main(int argc, char** argv) {
    std::string tree_path, fam_path, lambda_tree_path, multi, err_path, rootdist_path, family_out, out_dir;
    std::string pvalues_out, pvalues_cond;
    int pvalue_sims = 0, force_m = -1, force_r = -1;
    bool do_reconstruct = false, pvalues_on_device = false;
    double test_pvalue = 0.05;                                   // input_parameters::pvalue default (io.h)
    long limit = -1;
    double fixed_lambda = 0, fixed_alpha = -1, poisson = 0;
    int k = 1, device = 0, max_iter = 300, reps = 1, n_gpus = 1;
    bool use_err = false, use_poisson = false, keep_all = false;
    unsigned seed = 0;
    bool have_seed = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        auto optional = [&]() -> std::string { if (i + 1 < argc && argv[i + 1][0] != '-') return argv[++i]; return ""; };
        if (a == "-t") tree_path = next();
        else if (a == "-i") fam_path = next();
        else if (a == "-l") fixed_lambda = std::stod(next());
        else if (a == "-m") multi = next();
        else if (a == "-y") lambda_tree_path = next();
        else if (a == "-k") k = std::stoi(next());
        else if (a == "-a") fixed_alpha = std::stod(next());
        else if (a == "-e") { use_err = true; err_path = optional(); }
        else if (a == "-p") { use_poisson = true; std::string v = optional(); poisson = v.empty() ? 0 : std::stod(v); }
        else if (a == "-f") rootdist_path = next();
        else if (a == "-z") keep_all = true;
        else if (a == "-s") { seed = (unsigned)std::stoul(next()); have_seed = true; }
        else if (a == "-I") max_iter = std::stoi(next());
        else if (a == "-d") device = std::stoi(next());
        else if (a == "--gpus") n_gpus = std::stoi(next());
        else if (a == "--reps") reps = std::stoi(next());
        else if (a == "--family-out") family_out = next();
        else if (a == "-o") out_dir = next();
        else if (a == "--limit") limit = std::stol(next());
        else if (a == "--sizes") {                               // M,R instead of the data-derived maxima (tests)
            std::string v = next();
            const size_t comma = v.find(',');
            force_m = std::stoi(v.substr(0, comma)); force_r = std::stoi(v.substr(comma + 1));
        }
        else if (a == "--reconstruct") do_reconstruct = true;
        else if (a == "-P") test_pvalue = std::stod(next());
        else if (a == "--pvalues-device") pvalues_on_device = true;
        else if (a == "--pvalues") pvalue_sims = std::stoi(next());
        else if (a == "--pvalues-out") pvalues_out = next();
        else if (a == "--pvalues-cond") pvalues_cond = next();
        else { usage(); return 2; }
    }
    if (tree_path.empty() || fam_path.empty()) { usage(); return 2; }
    if (have_seed) randomizer_engine.seed(seed);
    try {
        user_data d;
        d.p_tree.reset(parse_newick(slurp_first_line(tree_path), false));
        {
            std::ifstream f(fam_path);
            if (!f.is_open()) throw std::runtime_error(fam_path + ": Failed to open. Exiting...");
            read_gene_families(f, d.p_tree.get(), d.gene_families);
        }
        compute_max_sizes(d.gene_families, d.max_family_size, d.max_root_family_size);
        if (force_m > 0) { d.max_family_size = force_m; d.max_root_family_size = force_r; }
        if (!err_path.empty()) {
            std::ifstream f(err_path);
            if (!f.is_open()) throw std::runtime_error("Failed to open " + err_path + ". Exiting...");
            d.p_error_model.reset(new error_model);
            read_error_model_file(f, d.p_error_model.get());
        }
        if (!lambda_tree_path.empty()) {
            d.p_lambda_tree.reset(parse_newick(slurp_first_line(lambda_tree_path), true));
            d.p_tree->validate_lambda_tree(d.p_lambda_tree.get());
        }
        if (fixed_lambda > 0) d.p_lambda.reset(new single_lambda(fixed_lambda));
        if (!multi.empty()) {
            std::vector<double> v;
            std::stringstream ss(multi);
            std::string tok;
            while (std::getline(ss, tok, ',')) v.push_back(std::stod(tok));
            d.p_lambda.reset(new multiple_lambda(d.p_lambda_tree->get_lambda_index_map(), v));
        }
        if (!rootdist_path.empty()) {
            std::ifstream f(rootdist_path);
            if (!f.is_open()) throw std::runtime_error("Failed to open file '" + rootdist_path + "'");
            read_rootdist(f, d.rootdist);
        }
        if (!keep_all) {                                          // cafexp.cpp:189-199
            auto rem = std::remove_if(d.gene_families.begin(), d.gene_families.end(), [&](const gene_family& f) { return !f.exists_at_root(d.p_tree.get()); });
            d.gene_families.erase(rem, d.gene_families.end());
        }
        if (limit >= 0 && (size_t)limit < d.gene_families.size()) d.gene_families.resize(limit);
        if (use_poisson && poisson > 0) d.p_prior.reset(new poisson_distribution(poisson));
        else if (use_poisson) d.p_prior.reset(new poisson_distribution(&d.gene_families));     // fitted to the leaf sizes (root_equilibrium_distribution.cpp:34)
        else d.p_prior.reset(new uniform_distribution());

        // build_models (core.cpp:16-50): gamma iff fixed_alpha > 0 or K > 1; default error model for -e without a file
        std::unique_ptr<error_model> default_em;
        error_model* em = d.p_error_model.get();
        std::unique_ptr<hip_model_base> mdl;
        lambda* start_lambda = d.p_lambda.get();
        if (fixed_alpha > 0 || k > 1) {
            auto g = new hip_gamma_model(start_lambda, d.p_tree.get(), &d.gene_families, d.max_family_size, d.max_root_family_size, k, fixed_alpha, em);
            mdl.reset(g);
        } else {
            if (use_err && !em) {
                default_em.reset(new error_model());
                default_em->set_probabilities(0, {0, .95, 0.05});
                default_em->set_probabilities(d.max_family_size, {0.05, .9, 0.05});
                em = default_em.get();
            }
            mdl.reset(new hip_base_model(start_lambda, d.p_tree.get(), &d.gene_families, d.max_family_size, d.max_root_family_size, em));
        }
        mdl->set_device(device);
        if (n_gpus > 1) {
            std::vector<int> devs(n_gpus);
            for (int g = 0; g < n_gpus; ++g) devs[g] = g;
            mdl->set_devices(devs);
        }

        std::unique_ptr<inference_optimizer_scorer> scorer(mdl->get_lambda_optimizer(d));
        std::unique_ptr<lambda> owned_lambda;
        optimizer_result opt;
        double search_s = 0;
        if (scorer) {                                            // estimate_missing_variables (execute.cpp:78)
            if (!d.p_lambda) owned_lambda.reset(mdl->get_lambda());
            optimizer o(scorer.get());
            o.max_iterations = max_iter;
            auto t0 = std::chrono::steady_clock::now();
            opt = o.optimize();
            search_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            scorer->finalize(opt.values.data());
        }
        double score = 0, best = 1e300;
        for (int r = 0; r < reps; ++r) {                         // compute (execute.cpp:42)
            auto t0 = std::chrono::steady_clock::now();
            score = mdl->infer_family_likelihoods(d.p_prior.get(), d.rootdist, mdl->get_lambda());
            best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        if (!family_out.empty()) {
            std::ofstream f(family_out);
            f.precision(17);
            mdl->write_family_likelihoods(f);
        }
        if (!out_dir.empty()) {                                  // the two files of estimator::compute (execute.cpp:49-54)
            std::ofstream rf(out_dir + "/" + mdl->name() + "_results.txt");
            mdl->write_vital_statistics(rf, score);
            std::ofstream lf(out_dir + "/" + mdl->name() + "_family_likelihoods.txt");
            mdl->write_family_likelihoods(lf);
            if (use_err) {                                       // write_error_model_if_specified (execute.cpp:24-40)
                std::ofstream ef(out_dir + "/" + mdl->name() + "_error_model.txt");
                if (em) write_error_model_file(ef, *em);
                else {                                           // model::write_error_model's stand-in (core.cpp:118-127)
                    error_model none;
                    none.set_probabilities(d.max_family_size, {0, 1, 0});
                    write_error_model_file(ef, none);
                }
            }
        }
        // compute_pvalues with the model's plain lambda (execute.cpp:153-161); 1000 simulations in the reference
        std::vector<double> pvalues;
        double pvalue_s = 0;
        if (pvalue_sims > 0) {
            pvalue_work work;
            auto t0 = std::chrono::steady_clock::now();
            if (pvalues_on_device)      // simulation on the GPU too: same distribution, another random sample (cafe_pvalues)
                pvalues = mdl->device_pvalues(pvalue_sims, have_seed ? seed : 1u);
            else
                pvalues = compute_pvalues(d.p_tree.get(), d.gene_families, mdl->get_lambda(), pvalue_sims, d.max_family_size, d.max_root_family_size,
                                          device, &work);
            pvalue_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (!pvalues_out.empty()) {
                std::ofstream f(pvalues_out);
                f.precision(17);
                f << "#FamilyID\tpvalue\tobserved max likelihood\n";
                for (size_t i = 0; i < pvalues.size(); ++i)
                    f << d.gene_families[i].id() << '\t' << pvalues[i] << '\t'
                      << (work.observed_max_likelihood.empty() ? 0.0 : work.observed_max_likelihood[i]) << '\n';
            }
            if (!pvalues_cond.empty()) {                         // FILE:K -> the first K sorted conditional distributions, one per line
                const size_t colon = pvalues_cond.rfind(':');
                const size_t kdump = std::min<size_t>(work.conditional_distribution.size(), std::stoul(pvalues_cond.substr(colon + 1)));
                std::ofstream f(pvalues_cond.substr(0, colon));
                f.precision(17);
                for (size_t i = 0; i < kdump; ++i) {
                    for (size_t j = 0; j < work.conditional_distribution[i].size(); ++j) f << (j ? "\t" : "") << work.conditional_distribution[i][j];
                    f << '\n';
                }
            }
        }
        // reconstruct_ancestral_states, Viterbi branch probabilities of the significant families, reports (execute.cpp:163-180)
        double reconstruct_s = 0;
        size_t n_with_probs = 0;
        if (do_reconstruct) {
            if (pvalues.empty()) pvalues.assign(d.gene_families.size(), 1.0);
            auto t0 = std::chrono::steady_clock::now();
            std::unique_ptr<reconstruction> rec(mdl->reconstruct_ancestral_states(d.gene_families, d.p_prior.get()));
            cladevector order;
            d.p_tree->apply_reverse_level_order([&order](const clade* c) { order.push_back(c); });
            branch_probabilities probs = compute_branch_probabilities(*mdl, *rec, d.gene_families, pvalues, test_pvalue, order);
            reconstruct_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            for (const auto& gf : d.gene_families) n_with_probs += probs.contains(gf);
            if (!out_dir.empty()) rec->write_results(mdl->name(), out_dir, d.p_tree.get(), d.gene_families, pvalues, test_pvalue, probs);
        }
        std::printf("{\"model\": \"%s\", ", mdl->name().c_str());
        print_num("neg_lnl", score);
        std::printf("\"n_families\": %zu, \"max_family_size\": %d, \"max_root_family_size\": %d, \"seconds_per_call\": %.6f, ",
                    d.gene_families.size(), d.max_family_size, d.max_root_family_size, best);
        std::printf("\"lambda\": [");
        auto lv = mdl->get_lambda()->values();
        for (size_t i = 0; i < lv.size(); ++i) std::printf("%s%.17g", i ? ", " : "", lv[i]);
        std::printf("]");
        if (auto g = dynamic_cast<hip_gamma_model*>(mdl.get())) {
            std::printf(", "); print_num("alpha", g->get_alpha(), false);
            std::printf(", \"multipliers\": [");
            auto mv = g->get_lambda_multipliers();
            for (size_t i = 0; i < mv.size(); ++i) std::printf("%s%.17g", i ? ", " : "", mv[i]);
            std::printf("]");
        }
        if (em) { std::printf(", "); print_num("epsilon", em->get_epsilons().back(), false); }
        if (scorer) {
            std::printf(", \"search\": {\"iterations\": %d, \"scorer_calls\": %d, \"seconds\": %.3f, ", opt.num_iterations, opt.num_scorer_calls, search_s);
            print_num("score", opt.score, false);
            std::printf("}");
        }
        if (pvalue_sims > 0) {
            size_t sig = 0;
            for (double p : pvalues) if (p < 0.05) ++sig;
            std::printf(", \"pvalues\": {\"simulations\": %d, \"seconds\": %.3f, \"significant_at_0.05\": %zu}", pvalue_sims, pvalue_s, sig);
        }
        if (do_reconstruct)
            std::printf(", \"reconstruct\": {\"seconds\": %.3f, \"families_with_branch_probabilities\": %zu}", reconstruct_s, n_with_probs);
        std::printf("}\n");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "cafexp_hip: %s\n", e.what());
        return 1;
    }
    return 0;
}
