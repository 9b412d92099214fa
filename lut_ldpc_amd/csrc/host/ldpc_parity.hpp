// ldpc_parity.hpp -- parity-check matrix container, alist I/O and GF(2) rank.
//
// Stands in for the pieces of the IT++ fork the reference's decode path touches
// (itpp::LDPC_Parity incl. the fork-only public members nvar, ncheck, sumX1, sumX2,
// get_colsum/get_rowsum -- src/LDPC_Code_LUT.cpp:491-502, src/LDPC_Ensemble.cpp:396-404 --
// and GF2mat::row_rank, src/LDPC_Code_LUT.cpp:494).  The IT++ submodule is empty in the
// reference tree, so this is written against the alist format itself.
#pragma once
#include <string>
#include <vector>

namespace lut_ldpc {

class LDPC_Parity {
public:
    int nvar = 0, ncheck = 0;
    std::vector<int> sumX1, sumX2;            // column / row weights

    LDPC_Parity() = default;
    explicit LDPC_Parity(const std::string &alist_filename) { load_alist(alist_filename); }
    void load_alist(const std::string &filename);        // throws std::runtime_error
    void save_alist(const std::string &filename) const;

    int get_nvar() const { return nvar; }
    int get_ncheck() const { return ncheck; }
    const std::vector<int> &get_colsum() const { return sumX1; }
    const std::vector<int> &get_rowsum() const { return sumX2; }
    // ascending row indices of column v / column indices of row c (0-based)
    const std::vector<int> &get_col(int v) const { return cols[(size_t)v]; }
    const std::vector<int> &get_row(int c) const { return rows[(size_t)c]; }
    int num_edges() const;
    // rank over GF(2): fill-free peeling of single-entry columns, then bit-packed elimination
    int row_rank() const;
    // new column j = old column perm[j] (used by LDPC_Generator_Systematic, which reorders the
    // code so that the information bits come first, like the IT++ class of the same name)
    void permute_cols(const std::vector<int> &perm);

private:
    std::vector<std::vector<int>> cols, rows;
};

}  // namespace lut_ldpc
