// numerics.cpp -- see numerics.hpp.  Reference: src/common.cpp, src/LDPC_DE.cpp:1061-1121.
#include "numerics.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <sstream>
#include <stdexcept>

namespace lut_ldpc {

double Qfunc(double x) { return 0.5 * std::erfc(x / 1.41421356237310); }   // literal as in IT++

double sum(const vec &v) {
    double s = 0;
    for (double x : v) s += x;
    return s;
}

vec fliplr(const vec &x) { return vec(x.rbegin(), x.rend()); }

vec kron(const vec &x, const vec &y) {
    vec z(x.size() * y.size());
    size_t k = 0;
    for (double a : x)
        for (double b : y) z[k++] = a * b;
    return z;
}

vec get_gaussian_pmf(double mu, double sig, int N, double delta) {
    vec pmf((size_t)N);
    const double h = N / 2.0;
    pmf[0] = 1 - Qfunc(((-h + 1) * delta - mu) / sig);
    for (int nn = 1; nn < N - 1; nn++)
        pmf[(size_t)nn] = Qfunc(((nn - h) * delta - mu) / sig) - Qfunc(((nn + 1 - h) * delta - mu) / sig);
    pmf[(size_t)N - 1] = Qfunc(((h - 1) * delta - mu) / sig);
    const double s = sum(pmf);
    for (double &p : pmf) p = p / s;
    return pmf;
}

vec get_var_product_pmf(const std::vector<vec> &p_in) {
    vec prod = p_in.back();
    for (size_t ii = p_in.size() - 1; ii-- > 0;) prod = kron(prod, p_in[ii]);
    return prod;
}

int signed_to_unsigned_idx(int idx, const ivec &inres) {
    int out_max = 2;
    for (int r : inres) out_max *= r / 2;
    int parity = 0, out = 0, base = 1;
    for (int r : inres) {
        const int lab = idx % r, half = r / 2;
        idx /= r;
        if (lab < half) { parity ^= 1; out += base * (half - 1 - lab); }
        else out += base * (lab - half);
        base *= half;
    }
    return parity ? out : out_max - 1 - out;
}

vec get_chk_product_pmf(const std::vector<vec> &p_in) {
    ivec res;
    for (const vec &p : p_in) res.push_back((int)p.size());
    vec p0 = p_in.back(), p1 = fliplr(p_in.back());
    for (size_t ii = p_in.size() - 1; ii-- > 0;) {
        const vec f = fliplr(p_in[ii]);
        const vec a = kron(p0, p_in[ii]), b = kron(p1, f), c = kron(p1, p_in[ii]), d = kron(p0, f);
        vec n0(a.size()), n1(a.size());
        for (size_t i = 0; i < a.size(); i++) { n0[i] = .5 * (a[i] + b[i]); n1[i] = .5 * (c[i] + d[i]); }
        p0.swap(n0); p1.swap(n1);
    }
    size_t len = 2;
    for (int r : res) len *= (size_t)(r / 2);
    vec comb(len, 0.0);
    for (size_t mm = 0; mm < p0.size(); mm++) comb[(size_t)signed_to_unsigned_idx((int)mm, res)] += p0[mm];
    return comb;
}

int quant_nonlin(double x, const vec &b) {
    int idx = 0;
    for (double t : b) { if (x > t) idx++; else break; }
    return idx;
}

ivec quant_nonlin(const vec &x, const vec &b) {
    ivec y(x.size());
    for (size_t i = 0; i < x.size(); i++) y[i] = quant_nonlin(x[i], b);
    return y;
}

double rate_to_shannon_thr(double R) { return 1.0 / std::sqrt(std::pow(2.0, 2 * R) - 1); }

static double x_log2_y(double x, double y) {
    if (x == 0) return 0;
    if (x > 0 && y > 0) return x * std::log2(y);
    throw std::domain_error("x_log2_y(): input invalid");
}

double get_mi_bcpmf_sym(const vec &p) {
    const size_t K = p.size();
    double mi = 0;
    for (size_t i = 0; i < K / 2; i++) {
        const double a = p[i], b = p[K - 1 - i];
        mi += a * std::log2(2 * a / (a + b)) + b * std::log2(2 * b / (b + a));
    }
    return mi;
}

vec sym_llr_sort_unique(const vec &p_in, ivec &idx_in, ivec &idx_sorted, double llr_delta) {
    const int M_in = (int)p_in.size();
    vec llr((size_t)M_in);
    for (int i = 0; i < M_in; i++) {
        double l = std::log(p_in[(size_t)i]) - std::log(p_in[(size_t)(M_in - 1 - i)]);
        // 0/0: a label pair with no mass at all (only reachable from the check-node LUT design,
        // which does not strip such pairs).  The reference sorts the resulting NaNs with a
        // comparison sort, i.e. in unspecified order; here the pair is given LLR 0, which keeps the
        // permutation symmetric and the result defined (DESIGN.md "deviations").
        if (std::isnan(l)) l = 0.0;
        llr[(size_t)i] = l;
    }
    idx_in.resize((size_t)M_in);
    std::iota(idx_in.begin(), idx_in.end(), 0);
    // ascending LLR, equal LLRs by ascending index (the reference sorts, then re-sorts every run
    // of equal LLRs by index, :336-343)
    std::stable_sort(idx_in.begin(), idx_in.end(), [&](int a, int b) { return llr[(size_t)a] < llr[(size_t)b]; });
    ivec half((size_t)(M_in / 2));
    half[0] = 0;
    double dupl = llr[(size_t)idx_in[0]];
    int dupl_idx = 0, num_dupl = 0;
    for (int mm = 1; mm < M_in / 2; mm++) {
        const double cur = llr[(size_t)idx_in[(size_t)mm]];
        if (std::abs(cur - dupl) <= llr_delta) num_dupl++; else dupl_idx++;
        half[(size_t)mm] = dupl_idx;
        dupl = cur;
    }
    const int mx = *std::max_element(half.begin(), half.end());
    idx_sorted.resize((size_t)M_in);
    for (int i = 0; i < M_in / 2; i++) {
        idx_sorted[(size_t)i] = half[(size_t)i];
        idx_sorted[(size_t)(M_in / 2 + i)] = 2 * mx + 1 - half[(size_t)(M_in / 2 - 1 - i)];
    }
    vec p_sorted((size_t)(M_in - 2 * num_dupl), 0.0);
    for (int mm = 0; mm < M_in; mm++) p_sorted[(size_t)idx_sorted[(size_t)mm]] += p_in[(size_t)idx_in[(size_t)mm]];
    return p_sorted;
}

double quant_mi_sym(vec &p_out, ivec &Q_out, const vec &p_in, int Nq, bool sorted) {
    const int K = Nq, M_in = (int)p_in.size();
    if (M_in % 2 || K % 2) throw std::invalid_argument("quant_mi_sym(): pmf length and label count must be even");
    vec p_sorted;
    ivec idx_in, idx_sorted;
    int M;
    if (!sorted) {
        p_sorted = sym_llr_sort_unique(p_in, idx_in, idx_sorted);
        M = (int)p_sorted.size();
    } else {
        idx_in.resize((size_t)M_in);
        std::iota(idx_in.begin(), idx_in.end(), 0);
        idx_sorted = idx_in;
        p_sorted = p_in;
        M = M_in;
    }
    Q_out.assign((size_t)M_in, 0);
    p_out.assign((size_t)K, 0.0);
    if (K >= M) {   // trivial case, :257-272
        int outlabel = 0;
        for (int mm = 0; mm < M_in / 2; mm++) {
            if (idx_sorted[(size_t)mm] > outlabel) outlabel++;
            Q_out[(size_t)idx_in[(size_t)(M_in - 1 - mm)]] = K - 1 - outlabel;
            Q_out[(size_t)idx_in[(size_t)mm]] = outlabel;
        }
        for (int mm = 0; mm < M_in; mm++) p_out[(size_t)Q_out[(size_t)mm]] += p_in[(size_t)mm];
        return get_mi_bcpmf_sym(p_in);
    }
    const int H = M / 2, Kh = K / 2, band = (M - K) / 2 + 1;
    // partial mutual information of merging sorted labels ap..a (and their mirror images)
    std::vector<double> g((size_t)H * (size_t)H, 0.0);
    for (int ap = 0; ap < H; ap++) {
        double pp = 0, pm = 0;
        for (int a = ap; a < H; a++) {
            pp += p_sorted[(size_t)(H + a)];
            pm += p_sorted[(size_t)(H - 1 - a)];
            g[(size_t)ap * H + a] = x_log2_y(pp, 2 * pp / (pp + pm)) + x_log2_y(pm, 2 * pm / (pp + pm));
        }
    }
    std::vector<double> S((size_t)H * (size_t)Kh, 0.0);
    std::vector<int> h((size_t)H * (size_t)Kh, 0);
    for (int a = 0; a < band; a++) S[(size_t)a * Kh] = g[(size_t)a];
    for (int zz = 1; zz < Kh; zz++)
        for (int a = zz; a < zz + band; a++) {
            double best = -std::numeric_limits<double>::max();
            int arg = 0;
            for (int ap = zz; ap <= a; ap++) {
                const double t = S[(size_t)(ap - 1) * Kh + (zz - 1)] + g[(size_t)ap * H + a];
                if (t > best) { best = t; arg = ap; }
            }
            S[(size_t)a * Kh + zz] = best;
            h[(size_t)a * Kh + zz] = arg;
        }
    ivec astar((size_t)Kh + 1, 0);
    astar[(size_t)Kh] = H;
    for (int kk = Kh - 1; kk > 0; kk--) astar[(size_t)kk] = h[(size_t)(astar[(size_t)kk + 1] - 1) * Kh + kk];
    int outlabel = 0;
    for (int mm = 0; mm < M_in / 2; mm++) {
        if (idx_sorted[(size_t)(mm + M_in / 2)] - H >= astar[(size_t)outlabel + 1]) outlabel++;
        Q_out[(size_t)idx_in[(size_t)(M_in / 2 + mm)]] = Kh + outlabel;
        Q_out[(size_t)idx_in[(size_t)(M_in / 2 - 1 - mm)]] = Kh - 1 - outlabel;
    }
    for (int mm = 0; mm < M_in; mm++) p_out[(size_t)Q_out[(size_t)mm]] += p_in[(size_t)mm];
    return S[(size_t)(H - 1) * Kh + (Kh - 1)];
}

vec chk_update_minsum(const vec &p_in, int dc) {
    const size_t H = p_in.size() / 2;
    vec ap(H), am(H), cp(H, 0.0), cm(H, 0.0);
    for (size_t n = 0; n < H; n++) { ap[n] = p_in[H + n] + p_in[H - 1 - n]; am[n] = p_in[H + n] - p_in[H - 1 - n]; }
    const vec bp = ap, bm = am;
    for (int dd = 1; dd < dc - 1; dd++) {
        std::fill(cp.begin(), cp.end(), 0.0);
        std::fill(cm.begin(), cm.end(), 0.0);
        for (size_t i = 0; i < H; i++)
            for (size_t j = 0; j < H; j++) {
                const size_t k = std::min(i, j);
                cp[k] += ap[i] * bp[j];
                cm[k] += am[i] * bm[j];
            }
        ap = cp; am = cm;
    }
    vec out(2 * H);
    for (size_t n = 0; n < H; n++) { out[H + n] = .5 * (cp[n] + cm[n]); out[H - 1 - n] = .5 * (cp[n] - cm[n]); }
    return out;
}

vec parse_vec(const std::string &s_in) {
    std::string s = s_in;
    for (char &c : s) if (c == ',' || c == ';' || c == '[' || c == ']') c = ' ';
    vec out;
    std::istringstream is(s);
    std::string tok;
    while (is >> tok) {
        if (tok.find(':') == std::string::npos) { out.push_back(std::stod(tok)); continue; }
        std::vector<double> parts;
        std::istringstream ts(tok);
        std::string p;
        while (std::getline(ts, p, ':')) parts.push_back(std::stod(p));
        double a, step, b;
        if (parts.size() == 2) { a = parts[0]; step = 1; b = parts[1]; }
        else if (parts.size() == 3) { a = parts[0]; step = parts[1]; b = parts[2]; }
        else throw std::invalid_argument("bad range '" + tok + "'");
        if (step == 0) throw std::invalid_argument("zero step in '" + tok + "'");
        // a:step:b with the end point included when it is hit (to within rounding)
        const int n = (int)std::floor((b - a) / step + 1e-9) + 1;
        for (int i = 0; i < n; i++) out.push_back(a + i * step);
    }
    return out;
}

}  // namespace lut_ldpc
