// Tree, families, error model, lambda holders, root priors and input readers of the host adapter.
// Behaviour follows the reference (citations inline); the code is ours.
#include "cafe_host.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <numeric>
#include <queue>
#include <sstream>
#include <stack>

namespace cafe {

// ------------------------------------------------------------------------------------------ clade
clade::~clade() {
    for (clade* c : _children) delete c;
}

double clade::get_branch_length() const {
    if (_is_lambda) throw std::runtime_error("Requested branch length from lambda tree");
    return _length;
}

int clade::get_lambda_index() const {
    if (!_is_lambda) throw std::runtime_error("Requested lambda index from branch length tree");
    return _lambda_index;
}

std::vector<const clade*> clade::leaves() const {
    std::vector<const clade*> out;
    for (const clade* n : post_order())
        if (n->is_leaf()) out.push_back(n);
    return out;
}

void clade::rename_interior() {
    // interior nodes are named by their sorted, concatenated leaf names (clade.cpp:125-139);
    // the lambda tree is matched to the tree through these names
    if (is_leaf()) return;
    std::vector<std::string> names;
    for (const clade* l : leaves()) names.push_back(l->_name);
    std::sort(names.begin(), names.end());
    _name.clear();
    for (const auto& s : names) _name += s;
}

void clade::add_descendant(clade* c) {
    c->_parent = this;
    _children.push_back(c);
    for (clade* p = this; p; p = p->_parent) p->rename_interior();
}

const clade* clade::find_descendant(const std::string& name) const {
    const clade* found = nullptr;
    apply_prefix_order([&](const clade* c) { if (c->_name == name) found = c; });
    return found;
}

std::set<double> clade::get_branch_lengths() const {
    std::set<double> out;
    apply_prefix_order([&](const clade* c) { if (c->get_branch_length() > 0.0) out.insert(c->get_branch_length()); });
    return out;
}

std::map<std::string, int> clade::get_lambda_index_map() const {
    std::map<std::string, int> out;
    apply_prefix_order([&](const clade* c) { out[c->get_taxon_name()] = c->get_lambda_index() - 1; });
    return out;
}

void clade::validate_lambda_tree(const clade* lambda_tree) const {
    std::set<std::string> mine, theirs;
    apply_prefix_order([&](const clade* c) { mine.insert(c->get_taxon_name()); });
    lambda_tree->apply_prefix_order([&](const clade* c) { theirs.insert(c->get_taxon_name()); });
    if (mine != theirs) throw std::runtime_error("The lambda tree structure does not match that of the tree");
}

void clade::write_newick(std::ostream& ost, const std::function<std::string(const clade*)>& textwriter) const {   // clade.cpp:166-183
    if (is_leaf()) { ost << textwriter(this); return; }
    ost << '(';
    for (size_t i = 0; i + 1 < _children.size(); ++i) {
        _children[i]->write_newick(ost, textwriter);
        ost << ',';
    }
    _children.back()->write_newick(ost, textwriter);
    ost << ')' << textwriter(this);
}

void clade::apply_prefix_order(const std::function<void(const clade*)>& f) const {
    std::vector<const clade*> todo{this};
    while (!todo.empty()) {
        const clade* c = todo.back();
        todo.pop_back();
        for (auto it = c->_children.rbegin(); it != c->_children.rend(); ++it) todo.push_back(*it);
        f(c);
    }
}

void clade::apply_reverse_level_order(const std::function<void(const clade*)>& f) const {
    std::vector<const clade*> level_order;
    std::queue<const clade*> q;
    q.push(this);
    while (!q.empty()) {
        const clade* c = q.front();
        q.pop();
        level_order.push_back(c);
        for (const clade* d : c->_children) q.push(d);
    }
    for (auto it = level_order.rbegin(); it != level_order.rend(); ++it) f(*it);
}

std::vector<const clade*> clade::post_order() const {
    std::vector<const clade*> out;
    std::vector<std::pair<const clade*, size_t>> st{{this, 0}};
    while (!st.empty()) {
        auto& top = st.back();
        if (top.second < top.first->_children.size()) {
            const clade* next = top.first->_children[top.second++];
            st.push_back({next, 0});
        } else {
            out.push_back(top.first);
            st.pop_back();
        }
    }
    return out;
}

namespace {
bool is_structural(char ch) { return ch == '(' || ch == ')' || ch == ',' || ch == ';' || ch == ':'; }
}  // namespace

// Hand-written tokenizer for the grammar the reference's regex accepts (clade.cpp:284): ( ) , ;
// :number  name.  A ":x" is a branch length, or a 1-based lambda index when parse_to_lambdas.
clade* parse_newick(const std::string& text, bool parse_to_lambdas) {
    std::unique_ptr<clade> root(new clade());
    root->_is_lambda = parse_to_lambdas;
    clade* cur = root.get();
    size_t i = 0;
    const size_t n = text.size();
    while (i < n) {
        const char ch = text[i];
        if (std::isspace((unsigned char)ch)) { ++i; continue; }
        if (ch == '(') {
            clade* child = new clade();
            child->_is_lambda = parse_to_lambdas;
            cur->add_descendant(child);
            cur = child;
            ++i;
        } else if (ch == ',') {
            if (cur == root.get()) {                // newick without the outer parentheses
                clade* new_root = new clade();
                new_root->_is_lambda = parse_to_lambdas;
                new_root->add_descendant(root.release());
                root.reset(new_root);
            }
            clade* sib = new clade();
            sib->_is_lambda = parse_to_lambdas;
            cur->_parent->add_descendant(sib);
            cur = sib;
            ++i;
        } else if (ch == ')') {
            cur = cur->_parent;
            if (!cur) throw std::runtime_error("unbalanced parentheses in newick string");
            ++i;
        } else if (ch == ';') {
            break;
        } else if (ch == ':') {
            size_t j = i + 1;
            while (j < n && !is_structural(text[j]) && !std::isspace((unsigned char)text[j])) ++j;
            const std::string num = text.substr(i + 1, j - i - 1);
            if (parse_to_lambdas) {
                cur->_lambda_index = (int)std::strtol(num.c_str(), nullptr, 0);
                cur->_is_lambda = true;
            } else {
                cur->_length = std::atof(num.c_str());
                cur->_is_lambda = false;
            }
            i = j;
        } else {
            size_t j = i;
            while (j < n && !is_structural(text[j]) && !std::isspace((unsigned char)text[j])) ++j;
            cur->_name = text.substr(i, j - i);
            for (clade* p = cur->_parent; p; p = p->_parent) p->rename_interior();
            i = j;
        }
    }
    if (root->_is_lambda) {
        if (root->_lambda_index == 0) root->_lambda_index = 1;      // the root may omit its index (clade.cpp:383)
        for (const clade* c : root->post_order())
            if (c->_lambda_index < 1) throw std::runtime_error("Invalid lambda index set for " + c->get_taxon_name());
    } else {
        for (const clade* c : root->post_order())
            if (!c->is_root() && c->_length <= 0) throw std::runtime_error("Invalid branch length set for " + c->get_taxon_name());
    }
    return root.release();
}

// ------------------------------------------------------------------------------------------ gene_family
namespace {
std::string lower(const std::string& s) {
    std::string o(s);
    for (auto& ch : o) ch = (char)std::tolower((unsigned char)ch);
    return o;
}
}  // namespace

void gene_family::set_species_size(const std::string& species, int count) { _sizes[lower(species)] = count; }

int gene_family::get_species_size(const std::string& species) const {
    auto it = _sizes.find(lower(species));
    if (it == _sizes.end()) throw std::runtime_error(species + " was not found in gene family " + _id);
    return it->second;
}

int gene_family::get_max_size() const {
    int m = 0;
    for (const auto& kv : _sizes) m = std::max(m, kv.second);
    return m;
}

std::vector<std::string> gene_family::get_species() const {
    std::vector<std::string> out;
    for (const auto& kv : _sizes) out.push_back(kv.first);
    return out;
}

bool gene_family::exists_at_root(const clade* tree) const {
    // every child subtree of the root must hold a leaf with a positive count (gene_family.cpp:60-89)
    for (const clade* child : tree->descendants()) {
        bool any = false;
        for (const clade* l : child->leaves())
            if (get_species_size(l->get_taxon_name()) > 0) { any = true; break; }
        if (!any) return false;
    }
    return true;
}

// ------------------------------------------------------------------------------------------ error_model
namespace {
bool nearly_equal(double x, double y) { return std::abs(x - y) <= 0.01 * std::abs(x); }   // error_model.cpp:25
}  // namespace

void error_model::set_probabilities(size_t fam_size, const std::vector<double>& probs) {
    if ((fam_size == 0 || _dists.empty()) && !nearly_equal(probs[0], 0.0))
        throw std::runtime_error("Cannot have a non-zero probability for family size 0 for negative deviation");
    if (!nearly_equal(std::accumulate(probs.begin(), probs.end(), 0.0), 1.0))
        throw std::runtime_error("Sum of probabilities must be equal to one");
    if (_dists.empty()) _dists.push_back(probs);
    if (_dists.size() <= fam_size) _dists.resize(fam_size + 1, _dists.back());     // skipped sizes repeat the last row
    _dists[fam_size] = probs;
}

std::vector<double> error_model::get_probs(size_t fam_size) const {
    if (fam_size >= _dists.size()) return _dists.back();       // (the reference indexes out of range past max; never hit)
    return _dists[fam_size];
}

std::vector<double> error_model::get_epsilons() const {
    std::set<double> uniq;
    for (const auto& d : _dists) uniq.insert(d.back());
    return std::vector<double>(uniq.begin(), uniq.end());
}

void error_model::update_single_epsilon(double eps) {
    auto e = get_epsilons();
    std::map<double, double> repl;
    repl[e.at(0)] = eps;
    replace_epsilons(repl);
}

void error_model::replace_epsilons(const std::map<double, double>& repl) {
    // row 0 becomes {0, 1-e, e}; rows >= 1 become {e, 1-2e, e}; old -> new matched with 1 % tolerance
    // (error_model.cpp:79-108)
    for (size_t i = 0; i < _dists.size(); ++i) {
        std::vector<double> v = _dists[i];
        for (const auto& kv : repl) {
            if (!nearly_equal(kv.first, v.back())) continue;
            v.back() = kv.second;
            if (i == 0) {
                v[1] = 1 - kv.second;
            } else {
                v.front() = kv.second;
                v[1] = 1 - (kv.second * 2);
            }
            set_probabilities(i, v);
            v = _dists[i];
        }
    }
}

// ------------------------------------------------------------------------------------------ lambda
std::string single_lambda::to_string() const {
    std::ostringstream o;
    o << std::setw(15) << std::setprecision(14) << _lambda;
    return o.str();
}

bool multiple_lambda::is_valid() const {
    return std::none_of(_lambdas.begin(), _lambdas.end(), [](double d) { return d < 0; });
}

lambda* multiple_lambda::multiply(double f) const {
    std::vector<double> v(_lambdas);
    for (auto& x : v) x *= f;
    return new multiple_lambda(_index, v);
}

std::string multiple_lambda::to_string() const {
    std::ostringstream o;
    o << std::setw(15) << std::setprecision(14);
    for (size_t i = 0; i < _lambdas.size(); ++i) o << _lambdas[i] << (i + 1 < _lambdas.size() ? ", " : "");
    return o.str();
}

// ------------------------------------------------------------------------------------------ priors
void root_distribution::vectorize(const std::map<int, int>& rootdist) {
    for (const auto& kv : rootdist)
        for (int i = 0; i < kv.second; ++i) _v.push_back(kv.first);
}
int root_distribution::at(size_t i) const {
    if (i >= _v.size()) throw std::out_of_range("Root distribution value out of range");
    return _v[i];
}
int root_distribution::sum() const {
    if (_v.empty()) throw std::runtime_error("Root distribution not created yet");
    return std::accumulate(_v.begin(), _v.end(), 0);
}

float uniform_distribution::compute(size_t val) const {
    if (val >= _rd.size()) return 0;
    return float(_rd.at(val)) / float(_sum);              // float arithmetic, like the reference
}

void poisson_distribution::initialize(const root_distribution* rd) {
    _pdf.resize(rd->size());
    for (size_t i = 0; i < _pdf.size(); ++i)              // poisspdf, poisson.cpp:19
        _pdf[i] = std::exp((double)i * std::log(_lambda) - std::lgamma((double)i + 1) - _lambda);
}

// ------------------------------------------------------------------------------------------ models: shared bits
std::ostream& operator<<(std::ostream& o, const family_info_stash& r) {
    o << r.family_id << "\t" << r.lambda_multiplier << "\t" << r.category_likelihood << "\t" << r.family_likelihood << "\t"
      << r.posterior_probability << "\t" << (r.significant ? "*" : "N/S");
    return o;
}

void event_monitor::summarize(std::ostream& ost) const {
    if (attempts == 0) { ost << "No attempts made\n"; return; }
    ost << attempts << " values were attempted (" << std::round(double(rejects) / double(attempts) * 100) << "% rejected)\n";
    if (failure_count.empty()) return;
    int worst = 0;
    for (const auto& kv : failure_count) worst = std::max(worst, kv.second);
    if (worst * 5 > (attempts - rejects)) {
        ost << "The following families had failure rates >20% of the time:\n";
        for (const auto& kv : failure_count)
            if (kv.second * 5 > (attempts - rejects)) ost << kv.first << " had " << kv.second << " failures\n";
    }
}

void model::write_vital_statistics(std::ostream& ost, double final_likelihood) {
    ost << "Model " << name() << " Final Likelihood (-lnL): " << final_likelihood << std::endl;
    ost << "Lambda: " << _p_lambda->to_string() << std::endl;
    if (_p_error_model) ost << "Epsilon: " << _p_error_model->get_epsilons()[0] << std::endl;
    const std::set<double> lengths = _p_tree->get_branch_lengths();
    ost << "Maximum possible lambda for this topology: " << 1 / *std::max_element(lengths.begin(), lengths.end()) << std::endl;
    _monitor.summarize(ost);
}

void model::initialize_lambda(const clade* lambda_tree) {
    if (lambda_tree) {
        std::set<int> uniq;
        lambda_tree->apply_prefix_order([&](const clade* c) { uniq.insert(c->get_lambda_index()); });
        _p_lambda = new multiple_lambda(lambda_tree->get_lambda_index_map(), std::vector<double>(uniq.size()));
    } else {
        _p_lambda = new single_lambda(0.0);
    }
}

// ------------------------------------------------------------------------------------------ readers
namespace {
std::vector<std::string> split(const std::string& s, char delim) {
    std::vector<std::string> out;
    std::string tok;
    std::istringstream in(s);
    while (std::getline(in, tok, delim)) out.push_back(tok);
    return out;
}
}  // namespace

void read_gene_families(std::istream& in, const clade* tree, std::vector<gene_family>& out) {
    // CAFE format: one header line "Desc<TAB>Family ID<TAB>species...", then one family per line.
    // CAFExp format: "#species" header lines, counts in tree order, id last (io.cpp:134-215).
    std::string line;
    std::vector<std::string> columns;
    std::map<size_t, std::string> leaf_of_index;
    bool header = true;
    size_t hash_index = 0;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (header && line[0] == '#') {
            if (!tree) throw std::runtime_error("No tree was provided.");
            const std::string name = line.substr(1);
            const clade* c = tree->find_descendant(name);
            if (!c) throw std::runtime_error(name + " not located in tree");
            if (c->is_leaf()) leaf_of_index[hash_index] = name;
            ++hash_index;
            continue;
        }
        std::vector<std::string> tk = split(line, '\t');
        if (header && leaf_of_index.empty()) {
            columns = tk;
            header = false;
            continue;
        }
        header = false;
        gene_family fam;
        if (leaf_of_index.empty()) {
            for (size_t i = 0; i < tk.size(); ++i) {
                if (i == 0) fam.set_desc(tk[i]);
                else if (i == 1) fam.set_id(tk[i]);
                else if (i < columns.size()) fam.set_species_size(columns[i], std::atoi(tk[i].c_str()));
            }
        } else {
            for (size_t i = 0; i < tk.size(); ++i) {
                auto it = leaf_of_index.find(i);
                if (it != leaf_of_index.end()) fam.set_species_size(it->second, std::atoi(tk[i].c_str()));
                else if (i + 1 == tk.size()) fam.set_id(tk[i]);
            }
        }
        out.push_back(fam);
    }
    if (out.empty()) throw std::runtime_error("No families found");
}

void read_error_model_file(std::istream& in, error_model* em) {
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.compare(0, 3, "max") == 0) {
            std::string v = split(line, ':').at(1);
            v.erase(std::remove_if(v.begin(), v.end(), ::isspace), v.end());
            em->set_max_family_size(std::stoi(v));
        } else if (line.compare(0, 3, "cnt") == 0) {
            std::vector<std::string> tk = split(line, ' ');
            if (tk.size() % 2 != 0)
                throw std::runtime_error("Number of different count differences in the error model (including 0) is not an odd number. Exiting...");
            std::vector<int> dev;
            for (size_t i = 1; i < tk.size(); ++i) dev.push_back(std::stoi(tk[i]));
            em->set_deviations(dev);
        } else {
            std::vector<std::string> tk = split(line, ' ');
            if (tk.empty() || tk[0].empty()) continue;
            std::vector<double> probs;
            for (size_t i = 1; i < tk.size(); ++i)
                if (!tk[i].empty()) probs.push_back(std::stod(tk[i]));
            em->set_probabilities(std::stoi(tk[0]), probs);
        }
    }
}

void read_rootdist(std::istream& in, std::map<int, int>& out) {
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream is(line);
        int size, count;
        if (is >> size >> count) out[size] = count;
    }
}

void compute_max_sizes(const std::vector<gene_family>& fams, int& max_family_size, int& max_root_family_size) {
    int mx = max_family_size;
    for (const auto& f : fams) mx = std::max(mx, f.get_max_size());
    max_root_family_size = std::max(30, static_cast<int>(std::rint(mx * 1.25)));       // user_data.cpp:45
    max_family_size = mx + std::max(50, mx / 5);                                       // user_data.cpp:46
}

void write_error_model_file(std::ostream& ost, const error_model& errormodel) {        // io.cpp:275-295
    ost << "maxcnt: " << errormodel.get_max_family_size() - 1 << "\n";
    ost << "cntdiff:";
    for (int j : errormodel.deviations()) ost << " " << j;
    ost << "\n";
    std::vector<double> last_probs;
    for (size_t j = 0; j < errormodel.get_max_family_size(); j++) {
        auto probs = errormodel.get_probs(j);
        if (probs == last_probs) continue;
        last_probs = probs;
        ost << j;
        for (auto p : probs) ost << " " << p;
        ost << std::endl;
    }
}

// ---------------------------------------------------------------- p-values
double pvalue(double v, const std::vector<double>& conddist) {                  // probability.cpp:379-389
    int idx = (int)conddist.size() - 1;
    auto bound = std::upper_bound(conddist.begin(), conddist.end(), v);
    if (bound != conddist.end()) idx = (int)(bound - conddist.begin());
    return idx / (double)conddist.size();
}

}  // namespace cafe
