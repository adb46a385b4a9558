// Host-side mirror of the reference's scorer/model interface above the C ABI
// (include/cafe_mi355x.h).  Same class names, argument meaning and error behaviour as the
// reference so that its optimizer (src/optimizer.cpp) and its tests read unchanged:
//   optimizer_scorer / inference_optimizer_scorer   src/optimizer_scorer.h:16-60
//   lambda_optimizer ... gamma_lambda_optimizer      src/optimizer_scorer.h:62-150
//   model::infer_family_likelihoods                  src/core.h:171
//   base_model / gamma_model                         src/base_model.h:22, src/gamma_core.h:47
//   lambda, single_lambda, multiple_lambda           src/lambda.h:21-100
//   error_model                                      src/error_model.h:29
//   root_equilibrium_distribution (float compute!)   src/root_equilibrium_distribution.h:12
// Everything is written from the behaviour described in SURVEY.md section 8; no reference code
// is reused.  The likelihood itself is never computed here: hip_base_model / hip_gamma_model
// forward to cafe_score (HIP kernels), and fail loudly when the library or a GPU is missing.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <iosfwd>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

struct cafe_ctx;
struct cafe_sharded;
struct cafe_params;
struct cafe_family_out;
struct cafe_sharded;

namespace cafe {

// ---------------------------------------------------------------- tree (src/clade.{h,cpp})
class clade {
public:
    clade() = default;
    clade(const std::string& name, double length) : _name(name), _length(length) {}
    ~clade();
    clade(const clade&) = delete;
    clade& operator=(const clade&) = delete;

    const clade* get_parent() const { return _parent; }
    bool is_leaf() const { return _children.empty(); }
    bool is_root() const { return _parent == nullptr; }
    double get_branch_length() const;          // throws on a lambda tree (clade.cpp:35)
    int get_lambda_index() const;              // throws on a branch-length tree (clade.cpp:43)
    std::string get_taxon_name() const { return _name; }
    const std::vector<clade*>& descendants() const { return _children; }
    void add_descendant(clade* c);
    const clade* find_descendant(const std::string& name) const;
    std::set<double> get_branch_lengths() const;                    // t > 0, root included (clade.cpp:196)
    std::map<std::string, int> get_lambda_index_map() const;        // name -> index - 1 (clade.cpp:154)
    void validate_lambda_tree(const clade* lambda_tree) const;      // clade.cpp:207
    void apply_prefix_order(const std::function<void(const clade*)>& f) const;
    void apply_reverse_level_order(const std::function<void(const clade*)>& f) const;   // children before parents
    void write_newick(std::ostream& ost, const std::function<std::string(const clade*)>& textwriter) const;   // clade.cpp:166
    std::vector<const clade*> post_order() const;                    // children before parents, root last
    std::vector<const clade*> leaves() const;

    friend clade* parse_newick(const std::string& text, bool parse_to_lambdas);

private:
    void rename_interior();                    // sorted, concatenated leaf names (clade.cpp:125)
    clade* _parent = nullptr;
    std::string _name;
    double _length = 0.0;
    int _lambda_index = 0;
    bool _is_lambda = false;
    std::vector<clade*> _children;
};
clade* parse_newick(const std::string& text, bool parse_to_lambdas = false);

// ---------------------------------------------------------------- families (src/gene_family.{h,cpp})
class gene_family {
public:
    void set_id(const std::string& id) { _id = id; }
    void set_desc(const std::string& d) { _desc = d; }
    std::string id() const { return _id; }
    void set_species_size(const std::string& species, int count);
    int get_species_size(const std::string& species) const;          // throws when absent (gene_family.cpp:36)
    int get_max_size() const;
    std::vector<std::string> get_species() const;
    bool species_size_match(const gene_family& o) const { return _sizes == o._sizes; }
    bool exists_at_root(const clade* tree) const;                    // gene_family.cpp:60
private:
    std::string _id, _desc;
    std::map<std::string, int> _sizes;       // keys lower-cased: the reference's map compares case-insensitively
};

// ---------------------------------------------------------------- error model (src/error_model.{h,cpp})
class error_model {
public:
    error_model() : _deviations{-1, 0, 1} {}
    void set_max_family_size(size_t m) { _max_family_size = m; }
    void set_deviations(const std::vector<int>& d) { _deviations = d; }
    void set_probabilities(size_t fam_size, const std::vector<double>& probs);
    std::vector<double> get_probs(size_t fam_size) const;
    size_t n_deviations() const { return _deviations.size(); }
    size_t get_max_family_size() const { return _dists.size(); }
    std::vector<double> get_epsilons() const;
    void replace_epsilons(const std::map<double, double>& replacements);
    void update_single_epsilon(double eps);
    const std::vector<int>& deviations() const { return _deviations; }
private:
    size_t _max_family_size = 0;
    std::vector<int> _deviations;
    std::vector<std::vector<double>> _dists;
};

// ---------------------------------------------------------------- lambda holders (src/lambda.{h,cpp})
class lambda {
public:
    virtual ~lambda() {}
    virtual void update(const double* values) = 0;
    virtual int count() const = 0;
    virtual bool is_valid() const = 0;
    virtual double get_value_for_clade(const clade* c) const = 0;
    virtual lambda* multiply(double factor) const = 0;
    virtual lambda* clone() const = 0;
    virtual std::string to_string() const = 0;
    virtual std::vector<double> values() const = 0;          // get_lambda_values (matrix_cache.cpp:99)
};
class single_lambda : public lambda {
    double _lambda;
public:
    explicit single_lambda(double v) : _lambda(v) {}
    double get_single_lambda() const { return _lambda; }
    void update(const double* v) override { _lambda = *v; }
    int count() const override { return 1; }
    bool is_valid() const override { return _lambda > 0; }                     // lambda.h:58
    double get_value_for_clade(const clade*) const override { return _lambda; }
    lambda* multiply(double f) const override { return new single_lambda(_lambda * f); }
    lambda* clone() const override { return new single_lambda(_lambda); }
    std::string to_string() const override;
    std::vector<double> values() const override { return {_lambda}; }
};
class multiple_lambda : public lambda {
    std::map<std::string, int> _index;
    std::vector<double> _lambdas;
public:
    multiple_lambda(const std::map<std::string, int>& name_to_index, const std::vector<double>& v) : _index(name_to_index), _lambdas(v) {}
    void update(const double* v) override { std::copy(v, v + _lambdas.size(), _lambdas.begin()); }
    int count() const override { return (int)_lambdas.size(); }
    bool is_valid() const override;                                             // none < 0 (lambda.cpp:59)
    double get_value_for_clade(const clade* c) const override { return _lambdas[_index.at(c->get_taxon_name())]; }
    int index_of(const clade* c) const { return _index.at(c->get_taxon_name()); }
    lambda* multiply(double f) const override;
    lambda* clone() const override { return new multiple_lambda(_index, _lambdas); }
    std::string to_string() const override;
    std::vector<double> values() const override { return _lambdas; }
};

// ---------------------------------------------------------------- root priors
class root_distribution {                                  // src/root_distribution.{h,cpp}
    std::vector<int> _v;
public:
    void vectorize(const std::map<int, int>& rootdist);
    void vectorize_uniform(int max) { _v.assign(max, 1); }
    void vector(const std::vector<int>& v) { _v = v; }
    size_t size() const { return _v.size(); }
    int at(size_t i) const;
    int sum() const;
};
class root_equilibrium_distribution {                      // compute() is a FLOAT in the reference
public:
    virtual ~root_equilibrium_distribution() {}
    virtual float compute(size_t val) const = 0;
    virtual void initialize(const root_distribution* rd) = 0;
};
class uniform_distribution : public root_equilibrium_distribution {
    root_distribution _rd;
    int _sum = 0;
public:
    void initialize(const root_distribution* rd) override { _rd = *rd; _sum = _rd.sum(); }
    float compute(size_t val) const override;
};
class poisson_distribution : public root_equilibrium_distribution {
    std::vector<double> _pdf;
    double _lambda;
public:
    explicit poisson_distribution(double pl) : _lambda(pl) {}
    explicit poisson_distribution(const std::vector<gene_family>* p_gene_families);     // fitted: root_equilibrium_distribution.cpp:34-45
    double poisson_lambda() const { return _lambda; }
    void initialize(const root_distribution* rd) override;
    float compute(size_t val) const override { return val >= _pdf.size() ? 0 : (float)_pdf[val]; }
};

// ---------------------------------------------------------------- discrete gamma (src/gamma.cpp, PAML)
double point_normal(double prob);
double incomplete_gamma(double x, double alpha, double ln_gamma_alpha);
double point_chi2(double prob, double v);
void get_gamma(std::vector<double>& cat_probs, std::vector<double>& multipliers, double alpha);

// ---------------------------------------------------------------- models (src/core.h, base_model.h, gamma_core.h)
struct family_info_stash {
    std::string family_id;
    double lambda_multiplier = 0, category_likelihood = 0, family_likelihood = 0, posterior_probability = 0;
    bool significant = false;
};
std::ostream& operator<<(std::ostream& o, const family_info_stash& r);

class event_monitor {
public:
    int attempts = 0, rejects = 0;
    std::map<std::string, int> failure_count;
    void summarize(std::ostream& ost) const;
};

class inference_optimizer_scorer;
struct user_data;
class reconstruction;

class model {
protected:
    lambda* _p_lambda;
    const clade* _p_tree;
    const std::vector<gene_family>* _p_gene_families;
    int _max_family_size, _max_root_family_size;
    error_model* _p_error_model;
    std::vector<family_info_stash> results;
    event_monitor _monitor;
public:
    model(lambda* l, const clade* tree, const std::vector<gene_family>* fams, int max_family_size, int max_root_family_size, error_model* em)
        : _p_lambda(l), _p_tree(tree), _p_gene_families(fams), _max_family_size(max_family_size), _max_root_family_size(max_root_family_size), _p_error_model(em) {}
    virtual ~model() {}
    lambda* get_lambda() const { return _p_lambda; }
    void set_lambda(lambda* l) { _p_lambda = l; }
    virtual double infer_family_likelihoods(root_equilibrium_distribution* prior, const std::map<int, int>& root_distribution_map, const lambda* p_lambda) = 0;
    virtual std::string name() const = 0;
    virtual void write_family_likelihoods(std::ostream& ost) = 0;
    virtual void write_vital_statistics(std::ostream& ost, double final_likelihood);     // core.cpp:96
    virtual inference_optimizer_scorer* get_lambda_optimizer(user_data& data) = 0;
    // core.h:177 without the matrix_cache argument (the device builds the matrices); default: not supported
    virtual reconstruction* reconstruct_ancestral_states(const std::vector<gene_family>& families, root_equilibrium_distribution* p_prior);
    const std::vector<family_info_stash>& get_results() const { return results; }
    const event_monitor& get_monitor() const { return _monitor; }
    void initialize_lambda(const clade* lambda_tree);                // core.cpp:76
    const clade* tree() const { return _p_tree; }
};

// Flattens (tree in `order` = children before parents, lambda structure, counts[n_families][leaves of `order`]) into a
// cafe_problem and creates the device context; throws std::runtime_error on failure.
cafe_ctx* create_device_context(const lambda* lam, const std::vector<const clade*>& order, const int32_t* counts, int64_t n_families,
                                 int max_family_size, int max_root_family_size, int max_categories, int n_deviations, int device);

// The two models whose infer_family_likelihoods runs on the GPU through the C ABI.
class hip_model_base : public model {
protected:
    cafe_ctx* _ctx = nullptr;
    int _ctx_categories = 0;
    int _ctx_lambda_sig = -1;                                        // lambda kind/count the context was built for
    int _device = 0;
    // several GPUs: the scorer calls go to a cafe_sharded (family shards, one host thread per device, one RCCL
    // all-reduce per call); what runs once after the search (reconstruction, p-values) stays on the first device
    std::vector<int> _devices;
    cafe_sharded* _sharded = nullptr;
    int _sharded_categories = 0, _sharded_lambda_sig = -1;
    std::vector<const clade*> _order;                                // post-order used to flatten
    void ensure_context(int max_categories);                         // single-device context (_ctx)
    void ensure_scorer(int max_categories);                          // what infer_family_likelihoods calls: _sharded or _ctx
    int score_call(const cafe_params* pr, double* score);
    int family_results_call(const cafe_family_out* out);
    const char* scorer_error() const;
    // prior as floats (compute(j), j < R), error-model table, lambdas in index order
    void gather_call_inputs(root_equilibrium_distribution* prior, const std::map<int, int>& rootdist, std::vector<float>& prior_f,
                            std::vector<double>& err_table, std::vector<double>& lambdas) const;
public:
    // Pupko reconstruction on the device: states[k][family][index in _order] (cafe_reconstruct)
    std::vector<int32_t> device_reconstruct(const std::vector<gene_family>& families, root_equilibrium_distribution* p_prior,
                                            const std::vector<double>* multipliers);
public:
    using model::model;
    ~hip_model_base() override;
    void set_device(int d) { _device = d; }
    void set_devices(const std::vector<int>& d) { _devices = d; if (!d.empty()) _device = d[0]; }
    // compute_pvalues with the Monte-Carlo simulation on the device too (cafe_pvalues): statistical agreement with the
    // reference's procedure, not draw for draw
    std::vector<double> device_pvalues(int number_of_simulations, uint64_t seed);
    // compute_viterbi_sum (gene_family_reconstructor.cpp:361) for every family x node of `order` under the model's
    // plain lambda: [family][order index], NaN where the reference returns an invalid branch_probability
    std::vector<double> branch_probability_table(const reconstruction& rec, const std::vector<gene_family>& families,
                                                 const std::vector<const clade*>& order);
};
class hip_base_model : public hip_model_base {
public:
    using hip_model_base::hip_model_base;
    double infer_family_likelihoods(root_equilibrium_distribution* prior, const std::map<int, int>& rootdist, const lambda* p_lambda) override;
    std::string name() const override { return "Base"; }
    void write_family_likelihoods(std::ostream& ost) override;
    inference_optimizer_scorer* get_lambda_optimizer(user_data& data) override;
    reconstruction* reconstruct_ancestral_states(const std::vector<gene_family>& families, root_equilibrium_distribution* p_prior) override;   // base_model.cpp:145
};
class hip_gamma_model : public hip_model_base {
    std::vector<double> _lambda_multipliers, _gamma_cat_probs;
    std::vector<std::vector<double>> _category_likelihoods;
    double _alpha;
public:
    hip_gamma_model(lambda* l, const clade* tree, const std::vector<gene_family>* fams, int max_family_size, int max_root_family_size,
                    int n_gamma_cats, double fixed_alpha, error_model* em);
    hip_gamma_model(lambda* l, const clade* tree, const std::vector<gene_family>* fams, int max_family_size, int max_root_family_size,
                    const std::vector<double>& gamma_categories, const std::vector<double>& multipliers, error_model* em);
    void set_alpha(double alpha);                                    // gamma_core.cpp:58
    double get_alpha() const { return _alpha; }
    std::vector<double> get_lambda_multipliers() const { return _lambda_multipliers; }
    bool can_infer() const;                                          // gamma_core.cpp:123
    double infer_family_likelihoods(root_equilibrium_distribution* prior, const std::map<int, int>& rootdist, const lambda* p_lambda) override;
    std::string name() const override { return "Gamma"; }
    void write_family_likelihoods(std::ostream& ost) override;
    void write_vital_statistics(std::ostream& ost, double final_likelihood) override;   // + "Alpha:" (gamma_core.cpp:43)
    inference_optimizer_scorer* get_lambda_optimizer(user_data& data) override;
    const std::vector<std::vector<double>>& category_likelihoods() const { return _category_likelihoods; }
    reconstruction* reconstruct_ancestral_states(const std::vector<gene_family>& families, root_equilibrium_distribution* p_prior) override;   // gamma_core.cpp:301
};

// ---------------------------------------------------------------- reconstruction reports (src/core.h:34-99,
// src/gene_family_reconstructor.cpp:167-359, base_model.cpp:181-228, gamma_core.cpp:283-299, :347-432)
typedef std::vector<const clade*> cladevector;
std::string clade_index_or_name(const clade* node, const cladevector& order);                     // clade.cpp:185

class branch_probabilities {
public:
    struct branch_probability {
        bool _is_valid;
        double _value;
        branch_probability(double value) : _is_valid(true), _value(value) {
            if (value < 0 || value > 1) throw std::runtime_error("Not a valid probability");
        }
        branch_probability() : _is_valid(false), _value(0.0) {}
    };
    bool contains(const gene_family& fam) const { return _probabilities.find(fam.id()) != _probabilities.end(); }
    branch_probability at(const gene_family& fam, const clade* c) const { return _probabilities.at(fam.id()).at(c); }
    void set(const gene_family& fam, const clade* c, branch_probability p) { _probabilities[fam.id()][c] = p; }
    static branch_probability invalid() { return branch_probability(); }
private:
    std::map<std::string, std::map<const clade*, branch_probability>> _probabilities;
};

class reconstruction {
public:
    typedef const std::vector<gene_family> familyvector;
    void print_node_change(std::ostream& ost, const cladevector& order, familyvector& gene_families, const clade* p_tree);
    void print_node_counts(std::ostream& ost, const cladevector& order, familyvector& gene_families, const clade* p_tree);
    void print_reconstructed_states(std::ostream& ost, const cladevector& order, familyvector& gene_families, const clade* p_tree, double test_pvalue,
                                    const branch_probabilities& branch_probabilities);
    // the reference walks a map keyed by node ADDRESS here, so its line order is an accident of the allocator;
    // this writer lists the same lines in `order`
    void print_increases_decreases_by_clade(std::ostream& ost, const cladevector& order, familyvector& gene_families);
    void print_increases_decreases_by_family(std::ostream& ost, const cladevector& order, familyvector& gene_families, const std::vector<double>& pvalues,
                                             double test_pvalue);
    void print_family_clade_table(std::ostream& ost, const cladevector& order, familyvector& gene_families, const clade* p_tree,
                                  const std::function<std::string(int family_index, const clade* c)>& get_family_clade_value);
    void write_results(const std::string& model_identifier, const std::string& output_prefix, const clade* p_tree, familyvector& families,
                       std::vector<double>& pvalues, double test_pvalue, const branch_probabilities& branch_probabilities);
    virtual int reconstructed_size(const gene_family& family, const clade* clade) const = 0;
    virtual ~reconstruction() {}
private:
    virtual void print_additional_data(const cladevector&, familyvector&, const std::string&) {}
    virtual int get_difference_from_parent(const gene_family* gf, const clade* c) = 0;
    virtual std::string get_reconstructed_state(const gene_family& gf, const clade* node) = 0;
    virtual void write_nexus_extensions(std::ostream&) {}
    virtual int get_node_count(const gene_family& gf, const clade* c) = 0;
};
void print_branch_probabilities(std::ostream& ost, const cladevector& order, const std::vector<gene_family>& gene_families,
                                const branch_probabilities& branch_probabilities);

class base_model_reconstruction : public reconstruction {
public:
    std::map<std::string, std::map<const clade*, int>> _reconstructions;
    int reconstructed_size(const gene_family& family, const clade* clade) const override;
private:
    std::string get_reconstructed_state(const gene_family& gf, const clade* node) override;
    int get_difference_from_parent(const gene_family* gf, const clade* c) override;
    int get_node_count(const gene_family& gf, const clade* c) override;
};

class gamma_model_reconstruction : public reconstruction {
    const std::vector<double> _lambda_multipliers;
    void write_nexus_extensions(std::ostream& ost) override;
    void print_additional_data(const cladevector& order, familyvector& gene_families, const std::string& output_prefix) override;
    std::string get_reconstructed_state(const gene_family& gf, const clade* node) override;
    int get_difference_from_parent(const gene_family* gf, const clade* c) override;
    int get_node_count(const gene_family& gf, const clade* c) override;
public:
    explicit gamma_model_reconstruction(const std::vector<double>& lambda_multipliers) : _lambda_multipliers(lambda_multipliers) {}
    void print_category_likelihoods(std::ostream& ost, const cladevector& order, familyvector& gene_families);
    int reconstructed_size(const gene_family& family, const clade* clade) const override;
    struct gamma_reconstruction {
        std::vector<std::map<const clade*, int>> category_reconstruction;
        std::map<const clade*, double> reconstruction;
        std::vector<double> _category_likelihoods;
    };
    std::map<std::string, gamma_reconstruction> _reconstructions;
};
// get_weighted_averages, gamma_core.cpp:283-299
std::map<const clade*, double> get_weighted_averages(const std::vector<std::map<const clade*, int>>& m, const std::vector<double>& probabilities);
// execute.cpp:163-176: Viterbi branch probabilities for the families with pvalue < test_pvalue
branch_probabilities compute_branch_probabilities(hip_model_base& mdl, const reconstruction& rec, const std::vector<gene_family>& families,
                                                  const std::vector<double>& pvalues, double test_pvalue, const cladevector& order);

// ---------------------------------------------------------------- scorers (src/optimizer_scorer.{h,cpp})
extern std::mt19937 randomizer_engine;

class optimizer_scorer {
public:
    virtual ~optimizer_scorer() {}
    virtual std::vector<double> initial_guesses() = 0;
    virtual double calculate_score(const double* values) = 0;
};
// empirical Poisson prior: -lnL of the leaf sizes shifted by one (src/poisson.cpp:38-77)
class poisson_scorer : public optimizer_scorer {
    std::vector<int> leaf_family_sizes;
public:
    explicit poisson_scorer(const std::vector<gene_family>& gene_families);
    std::vector<double> initial_guesses() override;
    double calculate_score(const double* values) override { return lnLPoisson(values); }
    double lnLPoisson(const double* plambda);
};
class inference_optimizer_scorer : public optimizer_scorer {
protected:
    lambda* _p_lambda;
    model* _p_model;
    root_equilibrium_distribution* _p_distribution;
    const std::map<int, int>& _rootdist_map;
public:
    virtual void prepare_calculation(const double* values) = 0;
    virtual void report_precalculation() = 0;
    inference_optimizer_scorer(lambda* l, model* m, root_equilibrium_distribution* d, const std::map<int, int>& rootdist)
        : _p_lambda(l), _p_model(m), _p_distribution(d), _rootdist_map(rootdist) {}
    double calculate_score(const double* values) override;           // NaN -> +inf (optimizer_scorer.cpp:30)
    virtual void finalize(double* result) = 0;
    bool quiet = true;
};
class lambda_optimizer : public inference_optimizer_scorer {
    double _longest_branch;
public:
    lambda_optimizer(lambda* l, model* m, root_equilibrium_distribution* d, double longest_branch, const std::map<int, int>& rootdist)
        : inference_optimizer_scorer(l, m, d, rootdist), _longest_branch(longest_branch) {}
    std::vector<double> initial_guesses() override;
    void prepare_calculation(const double* values) override { _p_lambda->update(values); }
    void report_precalculation() override;
    void finalize(double* results) override { _p_lambda->update(results); }
};
class lambda_epsilon_optimizer : public inference_optimizer_scorer {
    lambda_optimizer _lambda_optimizer;
    error_model* _p_error_model;
    std::vector<double> current_guesses;
public:
    lambda_epsilon_optimizer(model* m, error_model* em, root_equilibrium_distribution* d, const std::map<int, int>& rootdist, lambda* l, double longest_branch)
        : inference_optimizer_scorer(l, m, d, rootdist), _lambda_optimizer(l, m, d, longest_branch, rootdist), _p_error_model(em) {}
    std::vector<double> initial_guesses() override;
    void prepare_calculation(const double* values) override;
    void report_precalculation() override;
    void finalize(double* results) override;
};
class gamma_optimizer : public inference_optimizer_scorer {
    hip_gamma_model* _p_gamma_model;
public:
    gamma_optimizer(hip_gamma_model* m, root_equilibrium_distribution* d, const std::map<int, int>& rootdist)
        : inference_optimizer_scorer(m->get_lambda(), m, d, rootdist), _p_gamma_model(m) {}
    std::vector<double> initial_guesses() override;
    void prepare_calculation(const double* values) override { _p_gamma_model->set_alpha(*values); }
    void report_precalculation() override;
    void finalize(double* result) override { _p_gamma_model->set_alpha(*result); }
    double get_alpha() const { return _p_gamma_model->get_alpha(); }
};
class gamma_lambda_optimizer : public inference_optimizer_scorer {
    lambda_optimizer _lambda_optimizer;
    gamma_optimizer _gamma_optimizer;
public:
    gamma_lambda_optimizer(lambda* l, hip_gamma_model* m, root_equilibrium_distribution* d, const std::map<int, int>& rootdist, double longest_branch)
        : inference_optimizer_scorer(l, m, d, rootdist), _lambda_optimizer(l, m, d, longest_branch, rootdist), _gamma_optimizer(m, d, rootdist) {}
    std::vector<double> initial_guesses() override;
    void prepare_calculation(const double* values) override;
    void report_precalculation() override;
    void finalize(double* results) override;
};

// ---------------------------------------------------------------- inputs (src/io.cpp, src/user_data.cpp)
struct user_data {
    int max_family_size = -1, max_root_family_size = -1;
    std::unique_ptr<clade> p_tree, p_lambda_tree;
    std::unique_ptr<lambda> p_lambda;
    std::unique_ptr<error_model> p_error_model;
    std::unique_ptr<root_equilibrium_distribution> p_prior;
    std::vector<gene_family> gene_families;
    std::map<int, int> rootdist;
};
void read_gene_families(std::istream& in, const clade* tree, std::vector<gene_family>& out);    // io.cpp:134
void read_error_model_file(std::istream& in, error_model* em);                                  // io.cpp:226
void write_error_model_file(std::ostream& ost, const error_model& em);                          // io.cpp:275
void read_rootdist(std::istream& in, std::map<int, int>& out);                                  // user_data.cpp:103
void compute_max_sizes(const std::vector<gene_family>& fams, int& max_family_size, int& max_root_family_size);   // user_data.cpp:37-46

// ---------------------------------------------------------------- p-values (src/probability.cpp:255-454)
// The Monte-Carlo part follows the reference draw for draw on the global randomizer_engine (same libstdc++
// distributions, same traversal), so a run at the same seed sees the same simulated families; both prune batches
// (root sizes x simulations, and the observed families) run on the GPU through cafe_root_max.
struct pvalue_work {
    std::vector<std::vector<double>> conditional_distribution;     // [root size][simulation], sorted
    std::vector<double> observed_max_likelihood;                   // per family
};
double pvalue(double v, const std::vector<double>& conddist);                                        // probability.cpp:379
std::vector<double> compute_pvalues(const clade* p_tree, const std::vector<gene_family>& families, const lambda* p_lambda,
                                    int number_of_simulations, int max_family_size, int max_root_family_size, int device = 0,
                                    pvalue_work* keep = nullptr);                                     // probability.cpp:418

// ---------------------------------------------------------------- Nelder-Mead driver (src/optimizer.cpp)
struct optimizer_result {
    std::vector<double> values;
    double score = 0;
    int num_iterations = 0;
    int num_scorer_calls = 0;
};
class optimizer {
    optimizer_scorer* _scorer;
public:
    explicit optimizer(optimizer_scorer* s) : _scorer(s) {}
    int max_iterations = 300;                   // optimizer.h:28
    double tolx = 1e-6, tolf = 1e-6;            // OPTIMIZER_HIGH_PRECISION
    int similarity_window = 12;                 // OPTIMIZER_SIMILARITY_CUTOFF_SIZE, 0 = off
    double similarity_precision = 1e-3;         // OPTIMIZER_LOW_PRECISION
    std::vector<double> get_initial_guesses(int& calls);          // <= 100 retries while +inf (optimizer.cpp:345)
    optimizer_result optimize();
};

}  // namespace cafe
