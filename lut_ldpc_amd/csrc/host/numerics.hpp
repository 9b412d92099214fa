// numerics.hpp -- host-side numeric helpers of the LUT design (set-up path, runs once per ber_sim
// invocation).  Mirrors the free functions of the reference's src/common.{hpp,cpp} and the
// pmf-domain min-sum of src/LDPC_DE.cpp:1061-1121 with std::vector in place of itpp::vec.
// The floating-point operation ORDER follows the reference statement by statement: the designed
// tables are integers obtained from arg-max decisions over these doubles.
#pragma once
#include <string>
#include <vector>

namespace lut_ldpc {

using vec = std::vector<double>;
using ivec = std::vector<int>;
using bvec = std::vector<unsigned char>;

double Qfunc(double x);                                              // itpp Qfunc
double sum(const vec &v);                                            // left-to-right, like itpp::sum
vec fliplr(const vec &x);                                            // common.cpp:170-175
vec kron(const vec &x, const vec &y);                                // common.cpp:180-191

vec get_gaussian_pmf(double mu, double sig, int N, double delta);    // common.cpp:140-149
vec get_var_product_pmf(const std::vector<vec> &p_in);               // common.cpp:30-39
vec get_chk_product_pmf(const std::vector<vec> &p_in);               // common.cpp:41-70
int signed_to_unsigned_idx(int idx, const ivec &inres);              // common.cpp:193-228
int quant_nonlin(double x, const vec &boundaries);                   // common.cpp:120-129
ivec quant_nonlin(const vec &x, const vec &boundaries);              // common.cpp:131-138
double rate_to_shannon_thr(double R);                                // common.cpp:152-154
double get_mi_bcpmf_sym(const vec &p);                               // common.cpp:371-380

// common.cpp:230-331 -- MI-optimal symmetric quantiser (dynamic programme)
double quant_mi_sym(vec &p_out, ivec &Q_out, const vec &p_in, int Nq, bool sorted = false);
// common.cpp:333-369
vec sym_llr_sort_unique(const vec &p_in, ivec &idx_in, ivec &idx_sorted, double llr_delta = 0.0);
// LDPC_DE.cpp:1061-1089
vec chk_update_minsum(const vec &p_in, int dc);

// "0:.5:4", "3 3 3 2", "1,2,3" -> vector (the itpp string -> vec conversion used by
// src/LDPC_BER_Sim.cpp:53,399,412)
vec parse_vec(const std::string &s);

}  // namespace lut_ldpc
