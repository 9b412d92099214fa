/*
 * oracle.h -- CPU restatement of the lut_ldpc decode path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is the parity checker for the MI355X build.  Nothing in the product
 * (lut_ldpc_amd/, include/) may include, link or call it; only tests/, the smoke test
 * and bench.py's cpu_baseline leg do.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * mmeidlinger/lut_ldpc tree).  Pinning status (see tests/test_oracle_*.py):
 *   - LUT design numerics : pinned by the known-answer of trees/README.md:46-85
 *   - GF(2) rank / rate   : pinned by result-folder names README.md:114,239
 *   - DE evolve           : pinned by the threshold of README.md:173-176
 *   - decode (integer)    : restated from source; the reference ships no decode vectors
 *   - Monte-Carlo front end (IT++ RNG/AWGN): PARITY UNPINNED (IT++ fork absent)
 */
#ifndef LUT_LDPC_ORACLE_H
#define LUT_LDPC_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- small vectors */
typedef struct { double *v; int n; } or_dvec;
typedef struct { int *v; int n; } or_ivec;

or_dvec or_dvec_new(int n);            /* zero-initialised */
or_dvec or_dvec_copy(or_dvec a);
void    or_dvec_free(or_dvec *a);
or_ivec or_ivec_new(int n);
or_ivec or_ivec_copy(or_ivec a);
void    or_ivec_free(or_ivec *a);

/* ---------------------------------------------------------------- numerics (src/common.cpp) */
double  or_qfunc(double x);
or_dvec or_gaussian_pmf(double mu, double sig, int N, double delta);          /* common.cpp:140-149 */
or_dvec or_var_product_pmf(const or_dvec *p_in, int num_inputs);             /* common.cpp:30-39   */
or_dvec or_chk_product_pmf(const or_dvec *p_in, int num_inputs);             /* common.cpp:41-70   */
int     or_signed_to_unsigned_idx(int idx, const int *inres, int num_in);    /* common.cpp:193-228 */
int     or_quant_nonlin(double x, const double *bounds, int nb);             /* common.cpp:120-129 */
/* common.cpp:230-331; p_out (size Nq) and Q_out (size p_in.n) are allocated by the callee */
double  or_quant_mi_sym(or_dvec *p_out, or_ivec *Q_out, or_dvec p_in, int Nq, int sorted);
or_dvec or_sym_llr_sort_unique(or_dvec p_in, or_ivec *idx_in, or_ivec *idx_sorted, double llr_delta);
or_dvec or_chk_update_minsum_pmf(or_dvec p_in, int dc);                      /* LDPC_DE.cpp:1061-1121 */

/* ---------------------------------------------------------------- parity-check matrix */
typedef struct {
    int nvar, nchk, nedges;
    int *dv, *dc;          /* column / row weights */
    int *col_ptr, *col_idx;/* CSC: rows of each column, ascending (LDPC_Code_LUT.cpp:516-517) */
    int *row_ptr, *row_idx;/* CSR: cols of each row, ascending    (LDPC_Code_LUT.cpp:532-533) */
} or_code;

or_code *or_code_load_alist(const char *path);       /* NULL on failure */
void     or_code_free(or_code *c);
int      or_code_gf2_rank(const or_code *c);         /* GF2mat(H).row_rank(), LDPC_Code_LUT.cpp:494 */
/* decoder_parameterization, LDPC_Code_LUT.cpp:488-541: cn_msg_idx has nedges entries */
void     or_code_cn_msg_idx(const or_code *c, int *cn_msg_idx);

/* ---------------------------------------------------------------- degree distributions */
typedef struct {
    int dv_act, dc_act;
    int *degree_lam, *degree_rho;
    double *lam, *rho;
} or_ensemble;
or_ensemble *or_empirical_ensemble(const or_code *c);        /* LDPC_Ensemble.cpp:391-423,46-132 */
or_ensemble *or_ensemble_from_edge_dist(const int *dl, const double *l, int nl,
                                        const int *dr, const double *r, int nr);
double       or_ensemble_rate(const or_ensemble *e);          /* LDPC_Ensemble.cpp:320 */
void         or_ensemble_free(or_ensemble *e);

/* ---------------------------------------------------------------- LUT trees (src/LUT_Tree.*) */
enum { OR_IM = 0, OR_ROOT = 1, OR_MSG = 2, OR_CHA = 3 };     /* LUT_Tree.hpp:188-192 */
enum { OR_VARTREE = 0, OR_CHKTREE = 1, OR_DECTREE = 2 };     /* LUT_Tree.hpp:50-53   */

typedef struct or_node {
    int type;
    int K;                 /* output alphabet */
    or_ivec Q;             /* half table (may be empty) */
    or_dvec p;             /* pmf during design */
    struct or_node **child;
    int nchild;
} or_node;

typedef struct {
    int type;
    int num_leaves;
    or_node *root;
} or_tree;

or_tree *or_tree_parse(const char *tmpl, int type);           /* LUT_Tree.cpp:167-198,579-592 */
or_tree *or_tree_auto(int num_leaves, int type, const char *mode); /* LUT_Tree.cpp:200-294,594-630 */
or_tree *or_tree_copy(const or_tree *t);
void     or_tree_free(or_tree *t);
void     or_tree_set_resolution(or_tree *t, int Nq_in, int Nq_out, int Nq_cha); /* :296-306 */
void     or_tree_set_leaves(or_tree *t, or_dvec p_msg, or_dvec p_cha);          /* :92-103  */
or_dvec  or_tree_update(or_tree *t, int reuse);               /* :683-698,114-130,709-766 */
void     or_tree_reset_pmfs(or_tree *t);
int      or_tree_height(const or_tree *t);
char    *or_tree_template_string(const or_tree *t);           /* :142-165, malloc'd */
/* evaluation, LUT_Tree.cpp:774-820,402-445.  msgs_in has d entries; out has d entries. */
void     or_tree_var_msg_update(const or_tree *t, const int *msgs_in, int d, int llr, int *out);
void     or_tree_chk_msg_update(const or_tree *t, const int *msgs_in, int d, int *out);
int      or_tree_dec_update(const or_tree *t, const int *msgs_in, int d, int llr);
/* serialisation, LUT_Tree.cpp:488-535,847-927 (format spec trees/README.md:87-95) */
char    *or_tree_serialize(const or_tree *t);                 /* malloc'd */
or_tree *or_tree_deserialize(const char **cursor);

/* a 2-D array [iter-set][degree-class] of trees */
typedef struct {
    int n_sets;
    int *n_classes;        /* per set */
    or_tree ***t;          /* t[set][class] */
} or_tree_array;
void     or_tree_array_free(or_tree_array *a);
char    *or_tree_array_serialize(const or_tree_array *a);     /* LUT_Tree.cpp:855-864 */
or_tree_array *or_tree_array_deserialize(const char *txt);    /* LUT_Tree.cpp:893-927 */

/* ---------------------------------------------------------------- density evolution / design */
typedef struct {
    or_ensemble *ens;      /* borrowed */
    int Nq_Cha;
    or_ivec Nq_Msg_vec;    /* per iteration */
    int maxiter_de;
    unsigned char *reuse_vec;
    double thr_prec, Pe_max, LLR_max, thr_min, thr_max;
    int maxiter_bisec, Nq_fine, max_ni_de_iters;
    int min_lut;
    int strategy;          /* 0 individual, 1 joint_level, 2 joint_root */
    or_tree_array *var_templates, *chk_templates;  /* borrowed; indexed [iter][class] */
    or_dvec pmf_cha, pmf_var2chk, pmf_chk2var;
} or_de_lut;

/* get_lut_tree_templates, LDPC_DE.cpp:1124-1290.  tree_method = "auto_bin_balanced" |
 * "auto_bin_high" | "root_only" | "filename=<ini>".  allow_deg1: build-side extension for
 * degree-1 variable nodes, off reproduces the reference abort (returns -1). */
int or_get_lut_tree_templates(const char *tree_method, const or_ensemble *ens,
                              const int *Nq_Msg, int max_iters, int Nq_Cha, int minLUT,
                              int allow_deg1,
                              or_tree_array **var_luts, or_tree_array **chk_luts);

or_de_lut *or_de_lut_new(or_ensemble *ens, int Nq_Cha, const int *Nq_Msg, int maxiter_de,
                         or_tree_array *var_templates, or_tree_array *chk_templates,
                         const unsigned char *reuse_vec, const char *strategy);
void       or_de_lut_free(or_de_lut *de);
/* LDPC_DE.cpp:198-326; returns iteration count / -1 like the reference */
int  or_de_lut_evolve(or_de_lut *de, double thr, int save_luts,
                      or_tree_array **var_trees, or_tree_array **chk_trees);
void or_de_lut_get_quant_bound(const or_de_lut *de, double sig, or_dvec *qb_Cha, or_dvec *qb_Msg); /* :561-601 */
int  or_de_lut_bisec_search(or_de_lut *de, double *thr);      /* LDPC_DE.cpp:49-96 */

/* ---------------------------------------------------------------- codec (src/LDPC_Code_LUT.*) */
typedef struct {
    or_code *code;         /* borrowed */
    int nvar, nchk, nedges, nchk_lin_indep;
    int *cn_msg_idx;
    int max_iters, psc, pisc, minLUT;
    int Nq_Cha;
    or_ivec Nq_Msg;
    unsigned char *reuse_vec;
    or_dvec qb_Cha, qb_Msg;
    or_ivec Nq_Cha_2_Nq_Msg_map;
    or_tree_array *var_trees, *chk_trees;
    int *var_tree_idx_iter, *chk_tree_idx_iter;   /* per iteration */
    int *var_tree_idx_degree, *chk_tree_idx_degree; /* per node */
    int initial_message_mode;                      /* 0 CONT, 1 QCHA */
    int *msgs;                                     /* work buffer, nedges */
    int faithful;                                  /* 1: per-output queue copy + recursive walk (default) */
} or_codec;

or_codec *or_codec_new(or_code *code, int skip_rank);         /* set_code, LDPC_Code_LUT.cpp:471-541 */
void      or_codec_free(or_codec *c);
/* design_luts, LDPC_Code_LUT.cpp:699-746; returns sigma, <0 on failure */
double    or_codec_design_luts(or_codec *c, const char *tree_method, int min_lut, double sigma2,
                               int max_iters, const unsigned char *reuse_vec, int Nq_Cha,
                               const int *Nq_Msg, int allow_deg1);
/* install externally produced trees (LDPC_Code_LUT.cpp:120-169); strings in the
 * Array<Array<LUT_Tree>> text format.  chk_txt may be NULL/empty for min-LUT. */
int       or_codec_set_trees_txt(or_codec *c, const char *var_txt, const char *chk_txt,
                                 int max_iters, const unsigned char *reuse_vec,
                                 int Nq_Cha, const int *Nq_Msg, int minLUT);
void      or_codec_set_exit_conditions(or_codec *c, int max_iters, int psc, int pisc); /* :176-185 */
/* lut_decode, LDPC_Code_LUT.cpp:259-353.  cha/msg0: nvar labels; out: nvar bits (0/1) */
int       or_codec_lut_decode(or_codec *c, const int *cha, const int *msg0, unsigned char *out);
/* the same with the message dumps of output_verbosity = level (2 or 3) appended to the text file `path` */
int       or_codec_lut_decode_dump(or_codec *c, const int *cha, const int *msg0, unsigned char *out, int level, const char *path);
/* decode(vec llr), LDPC_Code_LUT.cpp:204-239: quantise then lut_decode; returns iteration code */
int       or_codec_decode_llr(or_codec *c, const double *llr, unsigned char *out,
                              int *cha_labels, int *msg_labels);
int       or_codec_syndrome_ok(const or_codec *c, const unsigned char *bits);           /* :455-469 */
/* batch helper for tests and the CPU baseline: labels are uint8, frame-major [B][nvar] */
void      or_codec_lut_decode_batch_u8(or_codec *c, const uint8_t *cha, const uint8_t *msg0, int B,
                                       uint8_t *out_bits, int32_t *out_iters);

/* flat-table mode (or_flat.c): the same decode with the trees flattened into arrays and one frame per thread -- the CPU
 * side-by-side leg (ii) of SURVEY.md 8(d).  The handle borrows the codec (exit conditions are read at decode time). */
typedef struct or_flat or_flat;
or_flat *or_flat_new(const or_codec *c);
void     or_flat_free(or_flat *F);
void     or_flat_decode_batch_u8(const or_flat *F, const uint8_t *cha, const uint8_t *msg0, int B,
                                 uint8_t *out_bits, int32_t *out_iters, int n_threads);

/* [BP] comparison decoder (or_bp.c): the specification of include/lut_ldpc_bp.h on the CPU.  PARITY UNPINNED (IT++ fork absent). */
typedef struct or_bp or_bp;
or_bp *or_bp_new(const or_code *code, int d1, int d2, int d3, int d4);
void   or_bp_free(or_bp *b);
void   or_bp_set_exit_conditions(or_bp *b, int max_iters, int psc, int pisc);
int    or_bp_table(const or_bp *b, int *out);
int    or_bp_to_qllr(const or_bp *b, double l);
void   or_bp_decode_qllr_batch(const or_bp *b, const int *qllr, int B, uint8_t *out_bits, int32_t *out_iters, int *out_qllr);
void   or_bp_decode_llr_batch(const or_bp *b, const double *llr, int B, uint8_t *out_bits, int32_t *out_iters, int *out_qllr);

void   or_sim_awgn_llr(uint64_t seed, uint32_t stream, uint64_t frame0, int B, int N, double N0, const uint8_t *codewords, double *llr, int *uncoded);

void or_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
