// Discrete gamma rate categories for the gamma model's lambda multipliers.
//
// The scorer must produce bit-identical multipliers to the reference, because lambda * m_k is
// quantized to 1e-9 before a transition matrix is built (matrix_cache.h:47).  The reference uses
// PAML's routines (src/gamma.cpp:15-240): AS70 normal quantile, AS32 incomplete gamma ratio, AS91
// chi-square quantile, then the mean rate of K equal-probability categories of Gamma(alpha, alpha).
// These are published algorithms with fixed constants; they are restated here with structured
// control flow and the same operation order.
#include "cafe_host.h"

#include <cmath>

namespace cafe {

double point_normal(double prob) {                       // Odeh & Evans 1974 (AS70); gamma.cpp:203
    static const double a[5] = {-.322232431088, -1, -.342242088547, -.0204231210245, -.453642210148e-4};
    static const double b[5] = {.0993484626060, .588581570495, .531103462366, .103537752850, .0038560700634};
    const double p1 = prob < 0.5 ? prob : 1 - prob;
    if (p1 < 1e-20) return -9999;
    const double y = std::sqrt(std::log(1 / (p1 * p1)));
    const double num = (((y * a[4] + a[3]) * y + a[2]) * y + a[1]) * y + a[0];
    const double den = (((y * b[4] + b[3]) * y + b[2]) * y + b[1]) * y + b[0];
    const double z = y + num / den;
    return prob < 0.5 ? -z : z;
}

double incomplete_gamma(double x, double alpha, double ln_gamma_alpha) {     // Bhattacharjee 1970 (AS32); gamma.cpp:66
    const double accurate = 1e-8, overflow = 1e30;
    if (x == 0) return 0;
    if (x < 0 || alpha <= 0) return -1;
    const double factor = std::exp(alpha * std::log(x) - x - ln_gamma_alpha);
    if (!(x > 1 && x >= alpha)) {
        // series expansion
        double gin = 1, term = 1, rn = alpha;
        do {
            rn++;
            term *= x / rn;
            gin += term;
        } while (term > accurate);
        return gin * (factor / alpha);
    }
    // continued fraction
    double a = 1 - alpha, b = a + x + 1, term = 0;
    double pn[6] = {1, x, x + 1, x * b, 0, 0};
    double gin = pn[2] / pn[3];
    while (true) {
        a++;
        b += 2;
        term++;
        const double an = a * term;
        pn[4] = b * pn[2] - an * pn[0];
        pn[5] = b * pn[3] - an * pn[1];
        if (pn[5] != 0) {
            const double rn = pn[4] / pn[5];
            const double dif = std::fabs(gin - rn);
            if (dif <= accurate && dif <= accurate * rn) return 1 - factor * gin;
            gin = rn;
        }
        for (int i = 0; i < 4; ++i) pn[i] = pn[i + 2];
        if (std::fabs(pn[4]) >= overflow)
            for (int i = 0; i < 4; ++i) pn[i] /= overflow;
    }
}

double point_chi2(double prob, double v) {               // Best & Roberts 1975 (AS91); gamma.cpp:129
    const double e = .5e-6, aa = .6931471805;
    const double p = prob;
    if (p < .000002 || p > .999998 || v <= 0) return -1;
    const double g = std::lgamma(v / 2);
    const double xx = v / 2, c = xx - 1;
    double ch;
    if (v < -1.24 * std::log(p)) {
        ch = std::pow((p * xx * std::exp(g + xx * aa)), 1 / xx);
        if (ch - e < 0) return ch;
    } else if (v > .32) {
        const double x = point_normal(p);
        const double p1 = 0.222222 / v;
        ch = v * std::pow((x * std::sqrt(p1) + 1 - p1), 3.0);
        if (ch > 2.2 * v + 6) ch = -2 * (std::log(1 - p) - c * std::log(.5 * ch) + g);
    } else {
        ch = 0.4;
        const double a = std::log(1 - p);
        double q;
        do {
            q = ch;
            const double p1 = 1 + ch * (4.67 + ch);
            const double p2 = ch * (6.73 + ch * (6.66 + ch));
            const double t = -0.5 + (4.67 + 2 * ch) / p1 - (6.73 + ch * (13.32 + 3 * ch)) / p2;
            ch -= (1 - std::exp(a + g + .5 * ch + c * aa) * p2 / p1) / t;
        } while (std::fabs(q / ch - 1) - .01 > 0);
    }
    double q;
    do {
        q = ch;
        const double p1 = .5 * ch;
        double t = incomplete_gamma(p1, xx, g);
        if (t < 0) return -1;
        const double p2 = p - t;
        t = p2 * std::exp(xx * aa + g + p1 - c * std::log(ch));
        const double b = t / ch;
        const double a = 0.5 * t - b * c;
        const double s1 = (210 + a * (140 + a * (105 + a * (84 + a * (70 + 60 * a))))) / 420;
        const double s2 = (420 + a * (735 + a * (966 + a * (1141 + 1278 * a)))) / 2520;
        const double s3 = (210 + a * (462 + a * (707 + 932 * a))) / 2520;
        const double s4 = (252 + a * (672 + 1182 * a) + c * (294 + a * (889 + 1740 * a))) / 5040;
        const double s5 = (84 + 264 * a + c * (175 + 606 * a)) / 2520;
        const double s6 = (120 + c * (346 + 127 * c)) / 5040;
        ch += t * (1 + 0.5 * t * s1 - b * c * (s1 - b * (s2 - b * (s3 - b * (s4 - b * (s5 - b * s6))))));
    } while (std::fabs(q / ch - 1) > e);
    return ch;
}

// get_gamma (gamma.cpp:225) = discrete_gamma(freq, rate, alpha, alpha, K, median = 0) (gamma.cpp:15)
void get_gamma(std::vector<double>& cat_probs, std::vector<double>& multipliers, double alpha) {
    const int K = (int)cat_probs.size();
    const double beta = alpha;
    const double factor = alpha / beta * K;
    const double lnga1 = std::lgamma(alpha + 1);
    std::vector<double> cut(K);
    for (int i = 0; i < K - 1; ++i) cut[i] = point_chi2((i + 1.0) / K, 2.0 * (alpha)) / (2.0 * (beta));
    for (int i = 0; i < K - 1; ++i) cut[i] = incomplete_gamma(cut[i] * beta, alpha + 1, lnga1);
    multipliers[0] = cut[0] * factor;
    multipliers[K - 1] = (1 - cut[K - 2]) * factor;
    for (int i = 1; i < K - 1; ++i) multipliers[i] = (cut[i] - cut[i - 1]) * factor;
    for (int i = 0; i < K; ++i) cat_probs[i] = 1.0 / K;
}

}  // namespace cafe
