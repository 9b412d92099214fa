// ldpc_parity.cpp -- see ldpc_parity.hpp.
#include "ldpc_parity.hpp"

#include <algorithm>
#include <cstdint>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace lut_ldpc {

namespace {
std::vector<long> ints_of_next_line(std::istream &is) {
    std::string line;
    std::vector<long> v;
    while (std::getline(is, line)) {
        std::istringstream ls(line);
        long x;
        while (ls >> x) v.push_back(x);
        if (!v.empty()) break;
    }
    return v;
}
}  // namespace

void LDPC_Parity::load_alist(const std::string &filename) {
    std::ifstream f(filename);
    if (!f) throw std::runtime_error("LDPC_Parity::load_alist(): could not open \"" + filename + "\"");
    auto bad = [&](const std::string &what) { return std::runtime_error("LDPC_Parity::load_alist(): " + what + " in " + filename); };
    auto dims = ints_of_next_line(f);
    if (dims.size() < 2 || dims[0] <= 0 || dims[1] <= 0) throw bad("bad dimensions");
    nvar = (int)dims[0]; ncheck = (int)dims[1];
    if (ints_of_next_line(f).size() < 2) throw bad("missing maximum weights");
    auto cw = ints_of_next_line(f), rw = ints_of_next_line(f);
    if ((int)cw.size() != nvar || (int)rw.size() != ncheck) throw bad("weight lists do not match the dimensions");
    sumX1.assign(cw.begin(), cw.end()); sumX2.assign(rw.begin(), rw.end());
    cols.assign((size_t)nvar, {}); rows.assign((size_t)ncheck, {});
    for (int v = 0; v < nvar; v++) {
        for (long r : ints_of_next_line(f)) {
            if (r == 0) continue;                                   // zero padding
            if (r < 1 || r > ncheck) throw bad("row index out of range");
            cols[(size_t)v].push_back((int)r - 1);
        }
        if ((int)cols[(size_t)v].size() != sumX1[(size_t)v]) throw bad("column " + std::to_string(v + 1) + " has the wrong weight");
        std::sort(cols[(size_t)v].begin(), cols[(size_t)v].end());
    }
    for (int c = 0; c < ncheck; c++) {
        for (long u : ints_of_next_line(f)) {
            if (u == 0) continue;
            if (u < 1 || u > nvar) throw bad("column index out of range");
            rows[(size_t)c].push_back((int)u - 1);
        }
        if ((int)rows[(size_t)c].size() != sumX2[(size_t)c]) throw bad("row " + std::to_string(c + 1) + " has the wrong weight");
        std::sort(rows[(size_t)c].begin(), rows[(size_t)c].end());
    }
    // both halves must describe the same matrix
    std::vector<int> fill((size_t)ncheck, 0);
    for (int v = 0; v < nvar; v++)
        for (int r : cols[(size_t)v]) {
            auto &row = rows[(size_t)r];
            if (!std::binary_search(row.begin(), row.end(), v)) throw bad("column and row lists disagree");
            fill[(size_t)r]++;
        }
    for (int c = 0; c < ncheck; c++) if (fill[(size_t)c] != sumX2[(size_t)c]) throw bad("column and row lists disagree");
}

void LDPC_Parity::save_alist(const std::string &filename) const {
    std::ofstream f(filename);
    if (!f) throw std::runtime_error("LDPC_Parity::save_alist(): could not open \"" + filename + "\"");
    f << nvar << ' ' << ncheck << '\n';
    f << *std::max_element(sumX1.begin(), sumX1.end()) << ' ' << *std::max_element(sumX2.begin(), sumX2.end()) << '\n';
    for (int w : sumX1) f << w << ' ';
    f << '\n';
    for (int w : sumX2) f << w << ' ';
    f << '\n';
    for (auto &c : cols) { for (int r : c) f << r + 1 << ' '; f << '\n'; }
    for (auto &r : rows) { for (int v : r) f << v + 1 << ' '; f << '\n'; }
}

void LDPC_Parity::permute_cols(const std::vector<int> &perm) {
    if ((int)perm.size() != nvar) throw std::invalid_argument("LDPC_Parity::permute_cols(): wrong permutation length");
    std::vector<std::vector<int>> nc((size_t)nvar);
    std::vector<int> nw((size_t)nvar);
    for (int j = 0; j < nvar; j++) { nc[(size_t)j] = cols[(size_t)perm[(size_t)j]]; nw[(size_t)j] = sumX1[(size_t)perm[(size_t)j]]; }
    cols.swap(nc); sumX1.swap(nw);
    for (auto &r : rows) r.clear();
    for (int v = 0; v < nvar; v++) for (int r : cols[(size_t)v]) rows[(size_t)r].push_back(v);
}

int LDPC_Parity::num_edges() const {
    int e = 0;
    for (int w : sumX1) e += w;
    return e;
}

int LDPC_Parity::row_rank() const {
    int rank = 0;
    std::vector<int> live_w(sumX1);
    std::vector<char> row_gone((size_t)ncheck, 0), col_gone((size_t)nvar, 0);
    std::vector<int> todo;
    for (int v = 0; v < nvar; v++) if (live_w[(size_t)v] == 1) todo.push_back(v);
    // a column with exactly one live entry pivots on that row without creating fill
    while (!todo.empty()) {
        const int v = todo.back(); todo.pop_back();
        if (col_gone[(size_t)v] || live_w[(size_t)v] != 1) continue;
        int r = -1;
        for (int x : cols[(size_t)v]) if (!row_gone[(size_t)x]) { r = x; break; }
        if (r < 0) continue;
        rank++; row_gone[(size_t)r] = 1; col_gone[(size_t)v] = 1;
        for (int u : rows[(size_t)r]) if (!col_gone[(size_t)u] && --live_w[(size_t)u] == 1) todo.push_back(u);
    }
    std::vector<int> rmap((size_t)ncheck, -1), cmap((size_t)nvar, -1);
    int Mr = 0, Nr = 0;
    for (int r = 0; r < ncheck; r++) if (!row_gone[(size_t)r]) rmap[(size_t)r] = Mr++;
    for (int v = 0; v < nvar; v++) if (!col_gone[(size_t)v] && live_w[(size_t)v] > 0) cmap[(size_t)v] = Nr++;
    if (!Mr || !Nr) return rank;
    const size_t W = ((size_t)Nr + 63) / 64;
    std::vector<uint64_t> A((size_t)Mr * W, 0);
    for (int r = 0; r < ncheck; r++) {
        if (rmap[(size_t)r] < 0) continue;
        for (int u : rows[(size_t)r]) {
            const int cu = cmap[(size_t)u];
            if (cu >= 0) A[(size_t)rmap[(size_t)r] * W + (size_t)(cu >> 6)] ^= 1ull << (cu & 63);
        }
    }
    int prow = 0;
    for (int col = 0; col < Nr && prow < Mr; col++) {
        const size_t w = (size_t)(col >> 6);
        const uint64_t bit = 1ull << (col & 63);
        int p = -1;
        for (int r = prow; r < Mr; r++) if (A[(size_t)r * W + w] & bit) { p = r; break; }
        if (p < 0) continue;
        if (p != prow) std::swap_ranges(A.begin() + (long)((size_t)p * W + w), A.begin() + (long)((size_t)p * W + W), A.begin() + (long)((size_t)prow * W + w));
        for (int r = prow + 1; r < Mr; r++)
            if (A[(size_t)r * W + w] & bit)
                for (size_t j = w; j < W; j++) A[(size_t)r * W + j] ^= A[(size_t)prow * W + j];
        prow++;
    }
    return rank + prow;
}

}  // namespace lut_ldpc
