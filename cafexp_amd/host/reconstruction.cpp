// Ancestral reconstruction above the C ABI: the reference's reconstruction classes and report
// writers (src/core.h:34-99, src/gene_family_reconstructor.cpp:167-359, src/base_model.cpp:145-228,
// src/gamma_core.cpp:283-432).  Host-only: accessors and text formats, token for token the reference's streams;
// the algorithm itself runs on the device (hip_reconstruct.cpp).
#include "cafe_host.h"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <iterator>
#include <sstream>

namespace cafe {

reconstruction* model::reconstruct_ancestral_states(const std::vector<gene_family>&, root_equilibrium_distribution*) {
    throw std::runtime_error("reconstruct_ancestral_states is not implemented by this model");
}

std::map<const clade*, double> get_weighted_averages(const std::vector<std::map<const clade*, int>>& m, const std::vector<double>& probabilities) {
    std::map<const clade*, double> result;
    for (const auto& kv : m[0]) {
        double val = 0.0;
        for (size_t i = 0; i < probabilities.size(); ++i) val += probabilities[i] * double(m[i].at(kv.first));
        result[kv.first] = val;
    }
    return result;
}

// ------------------------------------------------------------------------------------------ base / gamma accessors
int base_model_reconstruction::reconstructed_size(const gene_family& family, const clade* clade) const {
    if (clade->is_leaf()) return family.get_species_size(clade->get_taxon_name());
    auto it = _reconstructions.find(family.id());
    if (it == _reconstructions.end()) throw std::runtime_error("Family " + family.id() + " was not reconstructed");
    auto c = it->second.find(clade);
    if (c == it->second.end()) throw std::runtime_error("Clade '" + clade->get_taxon_name() + "' was not reconstructed for family " + family.id());
    return c->second;
}
std::string base_model_reconstruction::get_reconstructed_state(const gene_family& gf, const clade* node) {
    const int value = node->is_leaf() ? gf.get_species_size(node->get_taxon_name()) : _reconstructions[gf.id()].at(node);
    return std::to_string(value);
}
int base_model_reconstruction::get_difference_from_parent(const gene_family* gf, const clade* c) {
    if (c->is_root()) return 0;
    const int val = c->is_leaf() ? gf->get_species_size(c->get_taxon_name()) : _reconstructions[gf->id()].at(c);
    return val - _reconstructions[gf->id()].at(c->get_parent());
}
int base_model_reconstruction::get_node_count(const gene_family& gf, const clade* c) { return _reconstructions[gf.id()].at(c); }

int gamma_model_reconstruction::reconstructed_size(const gene_family& family, const clade* clade) const {
    if (clade->is_leaf()) return family.get_species_size(clade->get_taxon_name());
    auto it = _reconstructions.find(family.id());
    if (it == _reconstructions.end()) throw std::runtime_error("Family " + family.id() + " was not reconstructed");
    auto c = it->second.reconstruction.find(clade);
    if (c == it->second.reconstruction.end())
        throw std::runtime_error("Clade '" + clade->get_taxon_name() + "' was not reconstructed for family " + family.id());
    return (int)c->second;                                       // the reference narrows the average to int (gamma_core.cpp:420)
}
std::string gamma_model_reconstruction::get_reconstructed_state(const gene_family& gf, const clade* node) {
    std::ostringstream ost;
    if (node->is_leaf()) ost << gf.get_species_size(node->get_taxon_name());
    else ost << std::round(_reconstructions[gf.id()].reconstruction.at(node));
    return ost.str();
}
int gamma_model_reconstruction::get_difference_from_parent(const gene_family* gf, const clade* c) {
    if (c->is_root()) return 0;
    const double val = c->is_leaf() ? gf->get_species_size(c->get_taxon_name()) : _reconstructions[gf->id()].reconstruction.at(c);
    const double parent_val = _reconstructions[gf->id()].reconstruction.at(c->get_parent());
    return int(val - parent_val);
}
int gamma_model_reconstruction::get_node_count(const gene_family& gf, const clade* c) {
    return int(std::round(_reconstructions[gf.id()].reconstruction.at(c)));
}
void gamma_model_reconstruction::write_nexus_extensions(std::ostream& ost) {
    ost << "\nBEGIN LAMBDA_MULTIPLIERS;\n";
    for (auto& lm : _lambda_multipliers) ost << "  " << lm << ";\n";
    ost << "END;\n\n";
}
void gamma_model_reconstruction::print_category_likelihoods(std::ostream& ost, const cladevector&, familyvector& gene_families) {
    ost << "Family ID\t";
    std::ostream_iterator<double> lm(ost, "\t");
    std::copy(_lambda_multipliers.begin(), _lambda_multipliers.end(), lm);
    ost << std::endl;
    for (const auto& gf : gene_families) {
        ost << gf.id() << '\t';
        const auto& rc = _reconstructions[gf.id()];
        std::ostream_iterator<double> ct(ost, "\t");
        std::copy(rc._category_likelihoods.begin(), rc._category_likelihoods.end(), ct);
        ost << std::endl;
    }
}
void gamma_model_reconstruction::print_additional_data(const cladevector& order, familyvector& gene_families, const std::string& output_prefix) {
    std::ofstream cat_likelihoods(output_prefix + "/Gamma_category_likelihoods.txt");
    print_category_likelihoods(cat_likelihoods, order, gene_families);
}

// ------------------------------------------------------------------------------------------ reports
std::string clade_index_or_name(const clade* node, const cladevector& order) {
    const auto id = std::distance(order.begin(), std::find(order.begin(), order.end(), node));
    if (node->is_leaf()) return node->get_taxon_name() + "<" + std::to_string(id) + ">";
    return "<" + std::to_string(id) + ">";
}

static std::string newick_node(const clade* node, const cladevector& order, bool significant, const std::function<std::string(const clade*)>& textwriter) {
    std::ostringstream ost;
    ost << clade_index_or_name(node, order) << (significant ? "*" : "") << "_" << textwriter(node);
    if (!node->is_root()) ost << ':' << node->get_branch_length();
    return ost.str();
}

void reconstruction::print_family_clade_table(std::ostream& ost, const cladevector& order, familyvector& gene_families, const clade*,
                                              const std::function<std::string(int family_index, const clade* c)>& get_family_clade_value) {
    ost << "FamilyID";
    for (auto c : order) ost << "\t" << clade_index_or_name(c, order);
    ost << std::endl;
    for (size_t i = 0; i < gene_families.size(); ++i) {
        ost << gene_families[i].id();
        for (auto node : order) ost << "\t" << get_family_clade_value((int)i, node);
        ost << std::endl;
    }
}

void reconstruction::print_node_change(std::ostream& ost, const cladevector& order, familyvector& gene_families, const clade* p_tree) {
    print_family_clade_table(ost, order, gene_families, p_tree, [this, &gene_families](int family_index, const clade* c) {
        std::ostringstream o;
        o << std::showpos << get_difference_from_parent(&gene_families[family_index], c);
        return o.str();
    });
}

void reconstruction::print_node_counts(std::ostream& ost, const cladevector& order, familyvector& gene_families, const clade* p_tree) {
    print_family_clade_table(ost, order, gene_families, p_tree, [this, &gene_families](int family_index, const clade* c) {
        const auto& gf = gene_families[family_index];
        if (c->is_leaf()) return std::to_string(gf.get_species_size(c->get_taxon_name()));
        return std::to_string(get_node_count(gf, c));
    });
}

void reconstruction::print_increases_decreases_by_family(std::ostream& ost, const cladevector&, familyvector& gene_families, const std::vector<double>& pvalues,
                                                         double test_pvalue) {
    if (gene_families.size() != pvalues.size()) throw std::runtime_error("No pvalues found for family");
    if (gene_families.empty()) { ost << "No increases or decreases recorded\n"; return; }
    ost << "#FamilyID\tpvalue\tSignificant at " << test_pvalue << "\n";
    for (size_t i = 0; i < gene_families.size(); ++i) {
        ost << gene_families[i].id() << '\t' << pvalues[i] << '\t';
        ost << (pvalues[i] < test_pvalue ? 'y' : 'n');
        ost << std::endl;
    }
}

void reconstruction::print_increases_decreases_by_clade(std::ostream& ost, const cladevector& order, familyvector& gene_families) {
    std::map<const clade*, std::pair<int, int>> increase_decrease_map;
    for (size_t j = 0; j < gene_families.size(); ++j)
        for (size_t i = 0; i < order.size(); ++i) {
            const int val = get_difference_from_parent(&gene_families[j], order[i]);
            if (val > 0) increase_decrease_map[order[i]].first++;
            if (val < 0) increase_decrease_map[order[i]].second++;
        }
    ost << "#Taxon_ID\tIncrease\tDecrease\n";
    for (auto c : order) {
        auto it = increase_decrease_map.find(c);
        if (it == increase_decrease_map.end()) continue;
        ost << clade_index_or_name(c, order) << "\t" << it->second.first << "\t" << it->second.second << std::endl;
    }
}

void print_branch_probabilities(std::ostream& ost, const cladevector& order, const std::vector<gene_family>& gene_families,
                                const branch_probabilities& branch_probabilities) {
    ost << "#FamilyID\t";
    for (auto& it : order) ost << clade_index_or_name(it, order) << "\t";
    ost << std::endl;
    for (auto& gf : gene_families) {
        if (!branch_probabilities.contains(gf)) continue;
        ost << gf.id();
        for (auto c : order) {
            ost << '\t';
            const auto p = branch_probabilities.at(gf, c);
            if (p._is_valid) ost << p._value;
            else ost << "N/A";
        }
        ost << std::endl;
    }
}

void reconstruction::print_reconstructed_states(std::ostream& ost, const cladevector& order, familyvector& gene_families, const clade* p_tree, double test_pvalue,
                                                const branch_probabilities& branch_probabilities) {
    ost << "#nexus\nBEGIN TREES;\n";
    for (size_t i = 0; i < gene_families.size(); ++i) {
        const auto& gene_family = gene_families[i];
        auto g = [&gene_family, this](const clade* node) { return get_reconstructed_state(gene_family, node); };
        std::function<std::string(const clade*)> text_func;
        if (branch_probabilities.contains(gene_family)) {
            text_func = [&, g](const clade* node) {
                const auto p = branch_probabilities.at(gene_family, node);
                return newick_node(node, order, p._is_valid ? p._value < test_pvalue : false, g);
            };
        } else {
            text_func = [&, g](const clade* node) { return newick_node(node, order, false, g); };
        }
        ost << "  TREE " << gene_family.id() << " = ";
        p_tree->write_newick(ost, text_func);
        ost << ';' << std::endl;
    }
    ost << "\nEND;\n";
    write_nexus_extensions(ost);
}

void reconstruction::write_results(const std::string& model_identifier, const std::string& output_prefix, const clade* p_tree, familyvector& families,
                                   std::vector<double>& pvalues, double test_pvalue, const branch_probabilities& branch_probabilities) {
    cladevector order;
    p_tree->apply_reverse_level_order([&order](const clade* c) { order.push_back(c); });
    const std::string dir = output_prefix.empty() ? std::string("results") : output_prefix;      // filename(), core.h:196
    std::ofstream ofst(dir + "/" + model_identifier + "_asr.tre");
    print_reconstructed_states(ofst, order, families, p_tree, test_pvalue, branch_probabilities);
    std::ofstream counts(dir + "/" + model_identifier + "_count.tab");
    print_node_counts(counts, order, families, p_tree);
    std::ofstream change(dir + "/" + model_identifier + "_change.tab");
    print_node_change(change, order, families, p_tree);
    std::ofstream family_results(dir + "/" + model_identifier + "_family_results.txt");
    print_increases_decreases_by_family(family_results, order, families, pvalues, test_pvalue);
    std::ofstream clade_results(dir + "/" + model_identifier + "_clade_results.txt");
    print_increases_decreases_by_clade(clade_results, order, families);
    std::ofstream branch_probabilities_file(dir + "/" + model_identifier + "_branch_probabilities.tab");
    print_branch_probabilities(branch_probabilities_file, order, families, branch_probabilities);
    print_additional_data(order, families, dir);
}

}  // namespace cafe
