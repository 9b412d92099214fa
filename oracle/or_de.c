/*
 * or_de.c -- ORACLE (test infrastructure): restatement of the LUT-design half of
 * src/LDPC_DE.cpp -- tree templates (:1124-1290), LDPC_DE_LUT::evolve (:198-326),
 * channel pmf / boundaries (:400-412,561-601), irregular updates (:414-558), the joint
 * root design (:1345-1466) and the bisection search (:49-96).
 */
#include "or_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tree arrays */
static or_tree_array *array_new(int n_sets)
{
    or_tree_array *a = (or_tree_array *)calloc(1, sizeof(or_tree_array));
    a->n_sets = n_sets;
    a->n_classes = (int *)calloc((size_t)(n_sets ? n_sets : 1), sizeof(int));
    a->t = (or_tree ***)calloc((size_t)(n_sets ? n_sets : 1), sizeof(or_tree **));
    return a;
}
static void array_set_len(or_tree_array *a, int set, int n)
{
    a->n_classes[set] = n;
    a->t[set] = (or_tree **)calloc((size_t)(n ? n : 1), sizeof(or_tree *));
}

/* LDPC_DE.cpp:1124-1290 */
int or_get_lut_tree_templates(const char *tree_method, const or_ensemble *ens, const int *Nq_Msg, int max_iters,
                              int Nq_Cha, int minLUT, int allow_deg1, or_tree_array **var_out, or_tree_array **chk_out)
{
    const char *eq = strchr(tree_method, '=');
    char tm[64]; const char *filename = "";
    size_t tl = eq ? (size_t)(eq - tree_method) : strlen(tree_method);
    if (tl >= sizeof tm) return -1;
    memcpy(tm, tree_method, tl); tm[tl] = 0;
    if (eq) filename = eq + 1;
    const int *var_deg = ens->degree_lam, *chk_deg = ens->degree_rho;
    int dv_act = ens->dv_act, dc_act = ens->dc_act;
    *var_out = NULL; *chk_out = NULL;

    if (!strcmp(tm, "filename")) {
        if (!filename[0]) return -1;
        or_ini *ini = or_ini_load(filename);
        if (!ini) return -1;
        if (!or_ini_has_section(ini, "var_iter_000") || !or_ini_has_section(ini, "DT")) { or_ini_free(ini); return -1; }
        or_tree_array *var = array_new(max_iters);
        char sec[32], key[32];
        for (int ii = 0; ii < max_iters - 1; ii++) {      /* :1152-1191 */
            array_set_len(var, ii, dv_act);
            snprintf(sec, sizeof sec, "var_iter_%03d", ii);
            if (ii == 0 || or_ini_has_section(ini, sec)) {
                for (int dd = 0; dd < dv_act; dd++) {
                    snprintf(key, sizeof key, "var_deg_%03d", var_deg[dd]);
                    const char *s = or_ini_get(ini, sec, key);
                    or_tree *t = s ? or_tree_parse(s, OR_VARTREE) : NULL;
                    if (!t || t->num_leaves != var_deg[dd]) { or_tree_free(t); or_tree_array_free(var); or_ini_free(ini); return -1; }
                    or_tree_set_resolution(t, Nq_Msg[ii], Nq_Msg[ii + 1], Nq_Cha);
                    var->t[ii][dd] = t;
                }
            } else {
                for (int dd = 0; dd < dv_act; dd++) var->t[ii][dd] = or_tree_copy(var->t[ii - 1][dd]);
            }
        }
        array_set_len(var, max_iters - 1, dv_act);        /* :1192-1205 */
        for (int dd = 0; dd < dv_act; dd++) {
            snprintf(key, sizeof key, "var_deg_%03d", var_deg[dd]);
            const char *s = or_ini_get(ini, "DT", key);
            or_tree *t = s ? or_tree_parse(s, OR_DECTREE) : NULL;
            if (!t || t->num_leaves != var_deg[dd] + 1) { or_tree_free(t); or_tree_array_free(var); or_ini_free(ini); return -1; }
            or_tree_set_resolution(t, Nq_Msg[0], Nq_Msg[1], Nq_Cha);
            var->t[max_iters - 1][dd] = t;
        }
        *var_out = var;
        if (!minLUT) {                                    /* :1207-1248 */
            if (!or_ini_has_section(ini, "chk_iter_000")) { or_ini_free(ini); return -1; }
            or_tree_array *chk = array_new(max_iters);
            for (int ii = 0; ii < max_iters; ii++) {
                array_set_len(chk, ii, dc_act);
                snprintf(sec, sizeof sec, "chk_iter_%03d", ii);
                if (ii == 0 || or_ini_has_section(ini, sec)) {
                    for (int dd = 0; dd < dc_act; dd++) {
                        snprintf(key, sizeof key, "chk_deg_%03d", chk_deg[dd]);
                        const char *s = or_ini_get(ini, sec, key);
                        or_tree *t = s ? or_tree_parse(s, OR_CHKTREE) : NULL;
                        if (!t || t->num_leaves != chk_deg[dd] - 1) { or_tree_free(t); or_tree_array_free(chk); or_ini_free(ini); return -1; }
                        or_tree_set_resolution(t, Nq_Msg[ii], Nq_Msg[ii < max_iters - 1 ? ii + 1 : ii], Nq_Cha);
                        chk->t[ii][dd] = t;
                    }
                } else {
                    /* the reference loops to dv_act here (:1243); dc_act is what is meant */
                    for (int dd = 0; dd < dc_act; dd++) chk->t[ii][dd] = or_tree_copy(chk->t[ii - 1][dd]);
                }
            }
            *chk_out = chk;
        }
        or_ini_free(ini);
        return 0;
    }
    if ((!strcmp(tm, "auto_bin_balanced") || !strcmp(tm, "auto_bin_high") || !strcmp(tm, "root_only")) && !filename[0]) {
        or_tree_array *var = array_new(max_iters);        /* :1252-1270 */
        for (int ii = 0; ii < max_iters; ii++) {
            array_set_len(var, ii, dv_act);
            for (int dd = 0; dd < dv_act; dd++) {
                or_tree *t;
                if (ii == max_iters - 1) {
                    t = or_tree_auto(var_deg[dd] + 1, OR_DECTREE, tm);
                    if (t) or_tree_set_resolution(t, Nq_Msg[ii], 2, Nq_Cha);
                } else {
                    /* the reference asserts num_leaves >= 2 (LUT_Tree.cpp:202,242,274) */
                    if (var_deg[dd] < 2 && !(allow_deg1 && !strcmp(tm, "auto_bin_balanced"))) { or_tree_array_free(var); return -1; }
                    t = or_tree_auto(var_deg[dd], OR_VARTREE, tm);
                    if (t) or_tree_set_resolution(t, Nq_Msg[ii], Nq_Msg[ii + 1], Nq_Cha);
                }
                if (!t) { or_tree_array_free(var); return -1; }
                var->t[ii][dd] = t;
            }
        }
        *var_out = var;
        if (!minLUT) {                                    /* :1271-1284 */
            or_tree_array *chk = array_new(max_iters);
            for (int ii = 0; ii < max_iters; ii++) {
                array_set_len(chk, ii, dc_act);
                for (int dd = 0; dd < dc_act; dd++) {
                    if (chk_deg[dd] - 1 < 2) { or_tree_array_free(chk); return -1; }
                    or_tree *t = or_tree_auto(chk_deg[dd] - 1, OR_CHKTREE, tm);
                    or_tree_set_resolution(t, Nq_Msg[ii], Nq_Msg[ii], 0);
                    chk->t[ii][dd] = t;
                }
            }
            *chk_out = chk;
        }
        return 0;
    }
    return -1;
}

/* ------------------------------------------------------------------ LDPC_DE_LUT */
or_de_lut *or_de_lut_new(or_ensemble *ens, int Nq_Cha, const int *Nq_Msg, int maxiter_de,
                         or_tree_array *var_templates, or_tree_array *chk_templates,
                         const unsigned char *reuse_vec, const char *strategy)
{
    or_de_lut *de = (or_de_lut *)calloc(1, sizeof(or_de_lut));   /* LDPC_DE.cpp:105-166, defaults LDPC_DE.hpp:134-140 */
    de->ens = ens; de->Nq_Cha = Nq_Cha; de->maxiter_de = maxiter_de;
    de->Nq_Msg_vec = or_ivec_new(maxiter_de);
    memcpy(de->Nq_Msg_vec.v, Nq_Msg, sizeof(int) * (size_t)maxiter_de);
    de->reuse_vec = (unsigned char *)calloc((size_t)maxiter_de, 1);
    if (reuse_vec) memcpy(de->reuse_vec, reuse_vec, (size_t)maxiter_de);
    de->thr_prec = 1e-6; de->Pe_max = 1e-9; de->maxiter_bisec = 30; de->LLR_max = 25; de->Nq_fine = 5000;
    de->var_templates = var_templates; de->chk_templates = chk_templates;
    de->min_lut = !(chk_templates && chk_templates->n_sets > 0);
    de->max_ni_de_iters = 1;
    de->thr_max = 1.0 / sqrt(pow(2.0, 2 * or_ensemble_rate(ens)) - 1);  /* rate_to_shannon_thr, common.cpp:152 */
    de->thr_min = de->thr_max * 1e-4;
    if (!strategy || !strcmp(strategy, "joint_root")) de->strategy = 2;
    else if (!strcmp(strategy, "joint_level")) de->strategy = 1;
    else de->strategy = 0;
    return de;
}
void or_de_lut_free(or_de_lut *de)
{
    if (!de) return;
    or_ivec_free(&de->Nq_Msg_vec); free(de->reuse_vec);
    or_dvec_free(&de->pmf_cha); or_dvec_free(&de->pmf_var2chk); or_dvec_free(&de->pmf_chk2var);
    free(de);
}

static or_dvec fine_channel_pmf(const or_de_lut *de, double sig, double *delta_out)
{
    double delta = 2 * de->LLR_max / de->Nq_fine;
    if (delta_out) *delta_out = delta;
    return or_gaussian_pmf(2 / (sig * sig), 2 / sig, de->Nq_fine, delta);
}

/* LDPC_DE.cpp:400-412 */
static void set_channel_pmf(or_de_lut *de, double sig)
{
    or_dvec fine = fine_channel_pmf(de, sig, NULL);
    or_ivec Q;
    or_dvec_free(&de->pmf_cha); or_dvec_free(&de->pmf_var2chk);
    (void)or_quant_mi_sym(&de->pmf_cha, &Q, fine, de->Nq_Cha, 1); or_ivec_free(&Q);
    (void)or_quant_mi_sym(&de->pmf_var2chk, &Q, fine, de->Nq_Msg_vec.v[0], 1); or_ivec_free(&Q);
    or_dvec_free(&fine);
}

/* LDPC_DE.cpp:561-601 */
void or_de_lut_get_quant_bound(const or_de_lut *de, double sig, or_dvec *qb_Cha, or_dvec *qb_Msg)
{
    double delta;
    or_dvec fine = fine_channel_pmf(de, sig, &delta);
    int M = de->Nq_fine;
    for (int which = 0; which < 2; which++) {
        int K = which == 0 ? de->Nq_Cha : de->Nq_Msg_vec.v[0];
        or_dvec p; or_ivec Q;
        (void)or_quant_mi_sym(&p, &Q, fine, K, 1);
        double *half = (double *)calloc((size_t)(K / 2), sizeof(double));
        int label = 0;
        for (int mm = 0; mm < M / 2; mm++)
            if (Q.v[M - M / 2 + mm] - K / 2 > label) { half[label] = mm * delta; label++; }
        or_dvec qb = or_dvec_new(K - 1);
        for (int i = 0; i < K / 2 - 1; i++) { qb.v[i] = -half[K / 2 - 2 - i]; qb.v[K / 2 + i] = half[i]; }
        qb.v[K / 2 - 1] = 0;
        if (which == 0) *qb_Cha = qb; else *qb_Msg = qb;
        free(half); or_dvec_free(&p); or_ivec_free(&Q);
    }
    or_dvec_free(&fine);
}

static void axpy(or_dvec *y, double a, or_dvec x) { for (int i = 0; i < x.n; i++) y->v[i] = y->v[i] + a * x.v[i]; }

/* LDPC_DE.cpp:1379-1466: one quantiser shared by a set of nodes (one list per degree) */
static void level_lut_tree_update(or_node ***nodes, const int *J, int L, const double *degree_dist, int chk)
{
    int M_tot = 0, No = -1;
    or_dvec **prod = (or_dvec **)malloc(sizeof(or_dvec *) * (size_t)L);
    double **w = (double **)malloc(sizeof(double *) * (size_t)L);
    for (int ll = 0; ll < L; ll++) {
        prod[ll] = (or_dvec *)malloc(sizeof(or_dvec) * (size_t)(J[ll] ? J[ll] : 1));
        w[ll] = (double *)malloc(sizeof(double) * (size_t)(J[ll] ? J[ll] : 1));
        double ws = 0;
        for (int jj = 0; jj < J[ll]; jj++) {
            or_node *n = nodes[ll][jj];
            if (No == -1) No = n->K;
            /* node weight = number of leaves below it (:1398) */
            int nl = 0; { /* count leaves */
                or_node *stack[1024]; int sp = 0; stack[sp++] = n;
                while (sp) { or_node *x = stack[--sp]; if (x->type == OR_MSG || x->type == OR_CHA) nl++; else for (int c = 0; c < x->nchild; c++) stack[sp++] = x->child[c]; }
            }
            w[ll][jj] = nl;
            or_dvec *pc = (or_dvec *)malloc(sizeof(or_dvec) * (size_t)n->nchild);
            for (int c = 0; c < n->nchild; c++) pc[c] = n->child[c]->p;
            prod[ll][jj] = chk ? or_chk_product_pmf(pc, n->nchild) : or_var_product_pmf(pc, n->nchild);
            free(pc);
            M_tot += prod[ll][jj].n;
        }
        for (int jj = 0; jj < J[ll]; jj++) ws += w[ll][jj];
        for (int jj = 0; jj < J[ll]; jj++) w[ll][jj] = w[ll][jj] / ws;
    }
    or_dvec overall = or_dvec_new(M_tot);
    for (int i = 0; i < M_tot; i++) overall.v[i] = -1e9;
    int I = 0;
    for (int ll = 0; ll < L; ll++)
        for (int jj = 0; jj < J[ll]; jj++) {
            int M = prod[ll][jj].n;
            for (int mm = 0; mm < M / 2; mm++) {
                overall.v[I + mm] = w[ll][jj] * degree_dist[ll] * prod[ll][jj].v[mm];
                overall.v[M_tot - 1 - I - mm] = w[ll][jj] * degree_dist[ll] * prod[ll][jj].v[M - 1 - mm];
            }
            I += M / 2;
        }
    { double s = 0; for (int i = 0; i < M_tot; i++) s += overall.v[i]; for (int i = 0; i < M_tot; i++) overall.v[i] = overall.v[i] / s; }
    or_dvec p_out = or_dvec_new(0);
    or_ivec Qall = or__design_skip_zero_mass(&p_out, overall, No);
    I = 0;
    for (int ll = 0; ll < L; ll++)
        for (int jj = 0; jj < J[ll]; jj++) {
            or_node *n = nodes[ll][jj];
            int M = prod[ll][jj].n;
            or_ivec_free(&n->Q);
            n->Q = or_ivec_new(M / 2);
            for (int mm = 0; mm < M / 2; mm++) n->Q.v[mm] = Qall.v[I + mm];
            I += M / 2;
            or_dvec_free(&n->p);
            n->p = or_dvec_new(No);
            for (int mm = 0; mm < M; mm++) {
                if (mm < M / 2) n->p.v[n->Q.v[mm]] += prod[ll][jj].v[mm];
                else n->p.v[No - 1 - n->Q.v[M - 1 - mm]] += prod[ll][jj].v[mm];
            }
        }
    for (int ll = 0; ll < L; ll++) { for (int jj = 0; jj < J[ll]; jj++) or_dvec_free(&prod[ll][jj]); free(prod[ll]); free(w[ll]); }
    free(prod); free(w); or_dvec_free(&overall); or_dvec_free(&p_out); or_ivec_free(&Qall);
}

static void collect_level(or_node *n, int req, int cur, or_node **out, int *cnt)   /* LUT_Tree.cpp:564-572 */
{
    if (req == cur) { out[(*cnt)++] = n; return; }
    for (int i = 0; i < n->nchild; i++) collect_level(n->child[i], req, cur + 1, out, cnt);
}

/* LDPC_DE.cpp:1345-1377 */
static void joint_root_design(or_tree **trees, int L, const double *degree_dist)
{
    if (L <= 0) return;
    for (int dd = 0; dd < L; dd++) { or_dvec p = or_tree_update(trees[dd], 0); or_dvec_free(&p); }
    or_node ***nodes = (or_node ***)malloc(sizeof(or_node **) * (size_t)L);
    int *J = (int *)malloc(sizeof(int) * (size_t)L);
    for (int ll = 0; ll < L; ll++) { nodes[ll] = (or_node **)malloc(sizeof(or_node *)); nodes[ll][0] = trees[ll]->root; J[ll] = 1; }
    level_lut_tree_update(nodes, J, L, degree_dist, trees[0]->type == OR_CHKTREE);
    for (int ll = 0; ll < L; ll++) free(nodes[ll]);
    free(nodes); free(J);
}

/* LDPC_DE.cpp:1293-1343 */
static void joint_level_design(or_tree **trees, int L, const double *degree_dist)
{
    int maxh = 0;
    int *levels = (int *)malloc(sizeof(int) * (size_t)L);
    for (int ll = 0; ll < L; ll++) { levels[ll] = or_tree_height(trees[ll]); if (levels[ll] > maxh) maxh = levels[ll]; }
    or_node ***nodes = (or_node ***)malloc(sizeof(or_node **) * (size_t)L);
    int *J = (int *)malloc(sizeof(int) * (size_t)L);
    for (int cur = maxh - 1; cur >= 0; cur--) {
        for (int ll = 0; ll < L; ll++) {
            nodes[ll] = (or_node **)malloc(sizeof(or_node *) * 4096);
            J[ll] = 0;
            if (levels[ll] > cur) {
                or_node *tmp[4096]; int cnt = 0;
                collect_level(trees[ll]->root, cur, 0, tmp, &cnt);
                for (int i = 0; i < cnt; i++) if (tmp[i]->type == OR_IM || tmp[i]->type == OR_ROOT) nodes[ll][J[ll]++] = tmp[i];
            }
        }
        level_lut_tree_update(nodes, J, L, degree_dist, trees[0]->type == OR_CHKTREE);
        for (int ll = 0; ll < L; ll++) free(nodes[ll]);
    }
    free(nodes); free(J); free(levels);
}

/* shared body of chk_update_irr / var_update_irr's LUT branches (LDPC_DE.cpp:434-487,505-557) */
static void lut_update_irr(or_de_lut *de, int iter, or_tree **prev, or_tree_array *templates, int L,
                           const double *dist, or_dvec p_msg, int Nq_in, int Nq_out, or_dvec *acc)
{
    if (de->reuse_vec[iter]) {
        for (int dd = 0; dd < L; dd++) {
            or_tree_set_leaves(prev[dd], p_msg, de->pmf_cha);
            or_dvec p = or_tree_update(prev[dd], 1);
            axpy(acc, dist[dd], p); or_dvec_free(&p);
        }
        return;
    }
    for (int dd = 0; dd < L; dd++) {
        or_tree *t = or_tree_copy(templates->t[iter][dd]);
        or_tree_set_leaves(t, p_msg, de->pmf_cha);
        or_tree_set_resolution(t, Nq_in, Nq_out, de->Nq_Cha);
        or_tree_free(prev[dd]);
        prev[dd] = t;
    }
    if (de->strategy == 0) {
        for (int dd = 0; dd < L; dd++) { or_dvec p = or_tree_update(prev[dd], 0); axpy(acc, dist[dd], p); or_dvec_free(&p); }
        return;
    }
    if (de->strategy == 1) joint_level_design(prev, L, dist); else joint_root_design(prev, L, dist);
    for (int i = 0; i < acc->n; i++) acc->v[i] = 0;
    for (int dd = 0; dd < L; dd++) { or_dvec p = or_tree_update(prev[dd], 1); axpy(acc, dist[dd], p); or_dvec_free(&p); }
}

/* LDPC_DE.cpp:414-489 */
static void chk_update_irr(or_de_lut *de, int iter, or_tree **prev_chk)
{
    const or_ensemble *e = de->ens;
    or_dvec_free(&de->pmf_chk2var);
    de->pmf_chk2var = or_dvec_new(de->Nq_Msg_vec.v[iter]);
    if (de->min_lut) {
        for (int dd = 0; dd < e->dc_act; dd++) {
            or_dvec p = or_chk_update_minsum_pmf(de->pmf_var2chk, e->degree_rho[dd]);
            axpy(&de->pmf_chk2var, e->rho[dd], p); or_dvec_free(&p);
        }
    } else
        lut_update_irr(de, iter, prev_chk, de->chk_templates, e->dc_act, e->rho, de->pmf_var2chk,
                       de->Nq_Msg_vec.v[iter], de->Nq_Msg_vec.v[iter], &de->pmf_chk2var);
}

/* LDPC_DE.cpp:494-558 */
static void var_update_irr(or_de_lut *de, int iter, or_tree **prev_var)
{
    const or_ensemble *e = de->ens;
    or_dvec_free(&de->pmf_var2chk);
    de->pmf_var2chk = or_dvec_new(de->Nq_Msg_vec.v[iter + 1]);
    lut_update_irr(de, iter, prev_var, de->var_templates, e->dv_act, e->lam, de->pmf_chk2var,
                   de->Nq_Msg_vec.v[iter], de->Nq_Msg_vec.v[iter + 1], &de->pmf_var2chk);
}

static or_tree_array *snapshot(or_tree **trees, int L, or_tree_array *dst, int idx)
{
    array_set_len(dst, idx, L);
    for (int i = 0; i < L; i++) dst->t[idx][i] = or_tree_copy(trees[i]);
    return dst;
}

/* LDPC_DE.cpp:198-326 */
int or_de_lut_evolve(or_de_lut *de, double thr, int save_luts, or_tree_array **var_trees, or_tree_array **chk_trees)
{
    const or_ensemble *e = de->ens;
    /* the output of the last variable node update is binary (:203) */
    or_ivec ext = or_ivec_new(de->Nq_Msg_vec.n + 1);
    memcpy(ext.v, de->Nq_Msg_vec.v, sizeof(int) * (size_t)de->Nq_Msg_vec.n);
    ext.v[de->Nq_Msg_vec.n] = 2;
    or_ivec saved = de->Nq_Msg_vec; de->Nq_Msg_vec = ext;

    double Pe, Pe_old = 1.0; int ni_iters = 0, ret = -1;
    int num_qtrees = 0;
    for (int i = 0; i < de->maxiter_de; i++) num_qtrees += !de->reuse_vec[i];
    set_channel_pmf(de, thr);
    or_tree **var_iter = (or_tree **)calloc((size_t)e->dv_act, sizeof(or_tree *));
    or_tree **chk_iter = (or_tree **)calloc((size_t)e->dc_act, sizeof(or_tree *));
    if (save_luts) {
        *var_trees = array_new(num_qtrees);
        *chk_trees = de->min_lut ? array_new(0) : array_new(num_qtrees);
    }
    int vidx = 0, cidx = 0;
    int max_iter = save_luts ? de->maxiter_de : de->maxiter_de - 1;
    int ii;
    for (ii = 0; ii < max_iter; ii++) {
        Pe = 0;
        for (int k = 0; k < de->Nq_Msg_vec.v[ii] / 2; k++) Pe += de->pmf_var2chk.v[k];
        if (Pe < de->Pe_max && !save_luts) { ret = ii; goto out; }
        if (Pe <= Pe_old) Pe_old = Pe; else ni_iters++;
        if (ni_iters >= de->max_ni_de_iters && !save_luts) { ret = -1; goto out; }
        chk_update_irr(de, ii, chk_iter);
        var_update_irr(de, ii, var_iter);
        if (save_luts && !de->reuse_vec[ii]) {
            snapshot(var_iter, e->dv_act, *var_trees, vidx++);
            if (!de->min_lut) snapshot(chk_iter, e->dc_act, *chk_trees, cidx++);
        }
    }
    if (save_luts) {
        for (int s = 0; s < (*var_trees)->n_sets; s++) for (int d = 0; d < (*var_trees)->n_classes[s]; d++) or_tree_reset_pmfs((*var_trees)->t[s][d]);
        for (int s = 0; s < (*chk_trees)->n_sets; s++) for (int d = 0; d < (*chk_trees)->n_classes[s]; d++) or_tree_reset_pmfs((*chk_trees)->t[s][d]);
        ret = max_iter;
    } else ret = -1;
out:
    for (int i = 0; i < e->dv_act; i++) or_tree_free(var_iter[i]);
    for (int i = 0; i < e->dc_act; i++) or_tree_free(chk_iter[i]);
    free(var_iter); free(chk_iter);
    de->Nq_Msg_vec = saved; or_ivec_free(&ext);
    return ret;
}

/* LDPC_DE.cpp:49-96 (arithmetic mean mode) */
int or_de_lut_bisec_search(or_de_lut *de, double *thr)
{
    int ach = -1, converged = 0, ii = 0;
    double sig = -1.0, lo = de->thr_min, hi = de->thr_max;
    while (!converged && ii < de->maxiter_bisec) {
        sig = (hi + lo) / 2;
        ach = or_de_lut_evolve(de, sig, 0, NULL, NULL);
        if ((hi - lo < de->thr_prec) && ach >= 0) converged = 1;
        if (ach >= 0) lo = sig; else hi = sig;
        ii++;
    }
    if (converged) { *thr = sig; return ii; }
    *thr = 0;
    return -1;
}
