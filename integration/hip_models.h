// hip_models.h -- the reference-side binding of the MI355X likelihood path (include/cafe_mi355x.h).
//
// This is the translation unit a CAFE5 (Han9527/CAFExp) maintainer adds to src/: it includes the REFERENCE's own headers
// (base_model.h, gamma_core.h) and derives from the reference's own classes, so the reference's optimizer, scorers,
// report writers and CLI keep calling the same virtuals.  It is compiled against /root/reference/src and linked with the
// reference objects + libcafe_mi355x.so by `make -C oracle ref_hip` (container only), and the resulting
// oracle/_ref/ref_hip_harness is run on the GPU box by tests/test_reference_binding.py next to the unmodified reference.
//
// Only PUBLIC / PROTECTED members of the reference are used -- no reference file needs an edit for these classes:
//   * per-node lambda index of a multiple_lambda (its name->index map is private, lambda.h:69): recovered through the
//     public virtuals clone() / update() / get_value_for_clade() with the probe values {0, 1, 2, ...};
//   * gamma_model's private _lambda_multipliers / _alpha: get_lambda_multipliers() / get_alpha() (gamma_core.h:67,93);
//     private _gamma_cat_probs: set_alpha (gamma_core.cpp:58-63) fills it with get_gamma(), a public function
//     (gamma.h:17), so the same call on local vectors yields the same values; the explicit-categories constructor's
//     values are kept by the derived constructor;
//   * private _category_likelihoods (gamma_core.h:51), read by gamma_model::reconstruct_ancestral_states
//     (gamma_core.cpp:322): the derived class keeps its own copy and overrides reconstruct_ancestral_states (virtual,
//     core.h:178), which it has to do anyway to run Pupko's reconstruction on the device.
// The one edit the reference needs is in build_models (core.cpp:16-50): `new hip_base_model(...)` / `new
// hip_gamma_model(...)` instead of `new base_model(...)` / `new gamma_model(...)`.
#ifndef HIP_MODELS_H
#define HIP_MODELS_H

#include <cstdint>
#include <vector>

#include "base_model.h"
#include "gamma_core.h"

extern "C" {
#include "cafe_mi355x.h"
}

class root_equilibrium_distribution;
class matrix_cache;

//! Device context of one model: the flattened tree / family table, created at the first scorer call (tree, families,
//! max sizes and the lambda's shape are fixed by then) and rebuilt when one of them is replaced (model::set_families,
//! model::initialize_lambda).
class hip_device_context {
    struct signature {
        const void* families = nullptr; size_t n_families = 0; const void* tree = nullptr;
        int lambda_count = 0, categories = 0, n_deviations = 0; bool multiple = false;
    };
    cafe_ctx* _ctx = nullptr;                // one device: scorer calls and everything after the search
    cafe_sharded* _sharded = nullptr;        // n_gpus > 1: the scorer calls (family shards + one RCCL all-reduce per call)
    signature _ctx_sig, _sharded_sig;
    cladevector _order;                      // children before parents; index = node id of the C ABI
    bool matches(const signature& s, const lambda* p_lambda, const clade* p_tree, const std::vector<gene_family>* p_families,
                 int categories, const error_model* p_error_model) const;
    void create(bool sharded, const lambda* p_lambda, const clade* p_tree, const std::vector<gene_family>* p_families,
                int max_family_size, int max_root_family_size, int categories, const error_model* p_error_model);
public:
    int device = 0;                          // the single device / the first of the shard devices
    int n_gpus = 1;                          // > 1: devices 0 .. n_gpus-1 share the families of every scorer call
    ~hip_device_context();
    //! the single-device context (created on first use)
    cafe_ctx* ensure(const lambda* p_lambda, const clade* p_tree, const std::vector<gene_family>* p_families,
                     int max_family_size, int max_root_family_size, int categories, const error_model* p_error_model);
    //! what a scorer call runs on: the sharded scorer when n_gpus > 1, else the single-device context
    void ensure_scorer(const lambda* p_lambda, const clade* p_tree, const std::vector<gene_family>* p_families,
                       int max_family_size, int max_root_family_size, int categories, const error_model* p_error_model);
    bool sharded_scorer() const;
    void score(const cafe_params* params, double* neg_lnl);              // throws std::runtime_error on a structural error
    bool family_results(const cafe_family_out* out);                     // false: the call was rejected (+inf), no results
    cafe_ctx* get() const { return _ctx; }
    const cladevector& order() const { return _order; }
};

//! What every call uploads: the float prior (root_equilibrium_distribution::compute), lambdas, error-model table.
struct hip_call_inputs {
    std::vector<float> prior;
    std::vector<double> lambdas, error_table;
    void gather(root_equilibrium_distribution* prior, const std::map<int, int>& rootdist, const lambda* p_lambda,
                const error_model* p_error_model, int max_family_size, int max_root_family_size);
};

class hip_base_model : public base_model {
    hip_device_context _dev;
public:
    hip_base_model(lambda* p_lambda, const clade* p_tree, const std::vector<gene_family>* p_gene_families,
                   int max_family_size, int max_root_family_size, error_model* p_error_model, int device = 0, int n_gpus = 1)
        : base_model(p_lambda, p_tree, p_gene_families, max_family_size, max_root_family_size, p_error_model) { _dev.device = device; _dev.n_gpus = n_gpus; }

    double infer_family_likelihoods(root_equilibrium_distribution* prior, const std::map<int, int>& root_distribution_map,
                                    const lambda* p_lambda) override;
    std::string name() const override { return "Base"; }
    reconstruction* reconstruct_ancestral_states(const std::vector<gene_family>& families, matrix_cache* p_calc,
                                                 root_equilibrium_distribution* p_prior) override;
    const hip_device_context& device_context() const { return _dev; }
};

class hip_gamma_model : public gamma_model {
    hip_device_context _dev;
    bool _explicit_categories = false;
    std::vector<double> _explicit_cat_probs;
    std::vector<std::vector<double>> _hip_category_likelihoods;     // [family][category], gamma_core.h:51's role
    void current_categories(std::vector<double>& cat_probs, std::vector<double>& multipliers) const;
public:
    hip_gamma_model(lambda* p_lambda, clade* p_tree, std::vector<gene_family>* p_gene_families, int max_family_size,
                    int max_root_family_size, int n_gamma_cats, double fixed_alpha, error_model* p_error_model, int device = 0, int n_gpus = 1)
        : gamma_model(p_lambda, p_tree, p_gene_families, max_family_size, max_root_family_size, n_gamma_cats, fixed_alpha, p_error_model) { _dev.device = device; _dev.n_gpus = n_gpus; }
    hip_gamma_model(lambda* p_lambda, clade* p_tree, std::vector<gene_family>* p_gene_families, int max_family_size,
                    int max_root_family_size, std::vector<double> gamma_categories, std::vector<double> multipliers,
                    error_model* p_error_model, int device = 0, int n_gpus = 1)
        : gamma_model(p_lambda, p_tree, p_gene_families, max_family_size, max_root_family_size, gamma_categories, multipliers, p_error_model),
          _explicit_categories(true), _explicit_cat_probs(gamma_categories) { _dev.device = device; _dev.n_gpus = n_gpus; }

    double infer_family_likelihoods(root_equilibrium_distribution* prior, const std::map<int, int>& root_distribution_map,
                                    const lambda* p_lambda) override;
    reconstruction* reconstruct_ancestral_states(const std::vector<gene_family>& families, matrix_cache* p_calc,
                                                 root_equilibrium_distribution* p_prior) override;
    const hip_device_context& device_context() const { return _dev; }
};

//! estimator::execute's compute_viterbi_sum loop (execute.cpp:163-176) as one device call for all families and nodes.
branch_probabilities hip_compute_branch_probabilities(model* p_model, const hip_device_context& dev, const reconstruction* rec, const std::vector<gene_family>& families,
                                                      const std::vector<double>& pvalues, double test_pvalue);

#endif
