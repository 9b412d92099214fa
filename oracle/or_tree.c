/*
 * or_tree.c -- ORACLE (test infrastructure): restatement of src/LUT_Tree.cpp -- template
 * parsing and generation, per-node quantiser design, evaluation and (de)serialisation.
 */
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ nodes */
static or_node *node_new(int type)
{
    or_node *n = (or_node *)calloc(1, sizeof(or_node));
    n->type = type;
    return n;
}
static void node_add_back(or_node *n, or_node *c)
{
    n->child = (or_node **)realloc(n->child, sizeof(or_node *) * (size_t)(n->nchild + 1));
    n->child[n->nchild++] = c;
}
static void node_add_front(or_node *n, or_node *c)
{
    n->child = (or_node **)realloc(n->child, sizeof(or_node *) * (size_t)(n->nchild + 1));
    memmove(n->child + 1, n->child, sizeof(or_node *) * (size_t)n->nchild);
    n->child[0] = c; n->nchild++;
}
static void node_free(or_node *n)
{
    if (!n) return;
    for (int i = 0; i < n->nchild; i++) node_free(n->child[i]);
    free(n->child); or_ivec_free(&n->Q); or_dvec_free(&n->p); free(n);
}
static or_node *node_copy(const or_node *n)   /* deep_copy, LUT_Tree.cpp:79-90 */
{
    or_node *m = node_new(n->type);
    m->K = n->K; m->Q = or_ivec_copy(n->Q); m->p = or_dvec_copy(n->p);
    for (int i = 0; i < n->nchild; i++) node_add_back(m, node_copy(n->child[i]));
    return m;
}
static int node_is_leaf(const or_node *n) { return n->type == OR_MSG || n->type == OR_CHA; }
static int node_num_leaves(const or_node *n)  /* :378-388 */
{
    if (node_is_leaf(n)) return 1;
    int nl = 0;
    for (int i = 0; i < n->nchild; i++) nl += node_num_leaves(n->child[i]);
    return nl;
}
static int node_height(const or_node *n)      /* :105-112 */
{
    int h = 0;
    for (int i = 0; i < n->nchild; i++) { int t = node_height(n->child[i]); if (t >= h) h = t + 1; }
    return h;
}

/* ------------------------------------------------------------------ templates */
/* LUT_Tree.cpp:167-198 */
static or_node *node_parse(const char **s)
{
    char c = **s;
    if (c == '\0') return NULL;
    (*s)++;
    or_node *n;
    switch (c) {
    case '/': return NULL;
    case 'r': n = node_new(OR_ROOT); break;
    case 'i': n = node_new(OR_IM); break;
    case 'm': n = node_new(OR_MSG); break;
    case 'c': n = node_new(OR_CHA); break;
    default:
        fprintf(stderr, "oracle: tree template: bad character '%c'\n", c);
        return NULL;
    }
    for (;;) {
        or_node *ch = node_parse(s);
        if (!ch) break;
        node_add_back(n, ch);
    }
    return n;
}

or_tree *or_tree_parse(const char *tmpl, int type)
{
    /* :579-592 -- trees other than CHKTREE need a channel leaf */
    if (!strchr(tmpl, 'c') && type != OR_CHKTREE) return NULL;
    char *buf = strdup(tmpl), *w = buf;
    for (const char *r = tmpl; *r; r++) if (*r != ' ' && *r != '\t' && *r != '\r' && *r != '\n') *w++ = *r;
    *w = 0;
    const char *cur = buf;
    or_tree *t = (or_tree *)calloc(1, sizeof(or_tree));
    t->type = type;
    t->root = node_parse(&cur);
    free(buf);
    if (!t->root) { free(t); return NULL; }
    t->num_leaves = node_num_leaves(t->root);
    return t;
}

/* LUT_Tree.cpp:200-237.  n_msg = num_leaves - var message leaves; n_msg == 0 only arises
 * through the degree-1 extension (ROOT with the single CHA child). */
static or_node *gen_bin_balanced(int num_leaves, int var)
{
    int n = num_leaves - (var ? 1 : 0);
    if (n == 0) { or_node *r = node_new(OR_ROOT); node_add_back(r, node_new(OR_CHA)); return r; }
    or_node **fifo = (or_node **)malloc(sizeof(or_node *) * (size_t)(2 * n + 2));
    int head = 0, tail = 0;
    for (int l = 0; l < n; l++) fifo[tail++] = node_new(OR_MSG);
    or_node *res;
    for (;;) {
        if (tail - head == 1) {
            if (var) { res = node_new(OR_ROOT); node_add_back(res, fifo[head]); node_add_back(res, node_new(OR_CHA)); }
            else { res = fifo[head]; res->type = OR_ROOT; }
            break;
        }
        or_node *l = fifo[head++], *r = fifo[head++];
        or_node *im = node_new(OR_IM);
        node_add_back(im, l); node_add_back(im, r);
        fifo[tail++] = im;
    }
    free(fifo);
    return res;
}

/* LUT_Tree.cpp:240-270 */
static or_node *gen_bin_high(int num_leaves, int var)
{
    or_node *root = node_new(OR_ROOT), *cur = root;
    node_add_back(cur, node_new(var ? OR_CHA : OR_MSG));
    int todo = num_leaves - 1;
    while (todo > 1) {
        node_add_front(cur, node_new(OR_IM));
        cur = cur->child[0];
        node_add_back(cur, node_new(OR_MSG));
        todo--;
    }
    node_add_back(cur, node_new(OR_MSG));
    return root;
}

/* LUT_Tree.cpp:272-294 */
static or_node *gen_root_only(int num_leaves, int var)
{
    or_node *root = node_new(OR_ROOT);
    for (int i = 0; i < num_leaves - 1; i++) node_add_back(root, node_new(OR_MSG));
    node_add_back(root, node_new(var ? OR_CHA : OR_MSG));
    return root;
}

/* LUT_Tree.cpp:594-630 */
or_tree *or_tree_auto(int num_leaves, int type, const char *mode)
{
    int var = (type != OR_CHKTREE);
    or_node *root;
    if (!strcmp(mode, "auto_bin_balanced")) root = gen_bin_balanced(num_leaves, var);
    else if (!strcmp(mode, "auto_bin_high")) root = gen_bin_high(num_leaves, var);
    else if (!strcmp(mode, "root_only")) root = gen_root_only(num_leaves, var);
    else return NULL;
    or_tree *t = (or_tree *)calloc(1, sizeof(or_tree));
    t->type = type; t->num_leaves = num_leaves; t->root = root;
    return t;
}

or_tree *or_tree_copy(const or_tree *t)
{
    or_tree *u = (or_tree *)calloc(1, sizeof(or_tree));
    u->type = t->type; u->num_leaves = t->num_leaves; u->root = t->root ? node_copy(t->root) : NULL;
    return u;
}
void or_tree_free(or_tree *t) { if (!t) return; node_free(t->root); free(t); }
int or_tree_height(const or_tree *t) { return node_height(t->root); }

static void tmpl_rec(const or_node *n, char **buf, size_t *len, size_t *cap)
{
    char c = n->type == OR_ROOT ? 'r' : n->type == OR_IM ? 'i' : n->type == OR_MSG ? 'm' : 'c';
    if (*len + 3 > *cap) { *cap *= 2; *buf = (char *)realloc(*buf, *cap); }
    (*buf)[(*len)++] = c;
    for (int i = 0; i < n->nchild; i++) tmpl_rec(n->child[i], buf, len, cap);
    if (*len + 3 > *cap) { *cap *= 2; *buf = (char *)realloc(*buf, *cap); }
    (*buf)[(*len)++] = '/';
}
char *or_tree_template_string(const or_tree *t)   /* :142-165 */
{
    size_t len = 0, cap = 64; char *buf = (char *)malloc(cap);
    tmpl_rec(t->root, &buf, &len, &cap);
    buf[len] = 0;
    return buf;
}

/* ------------------------------------------------------------------ design */
static void set_res_rec(or_node *n, int Nq_in, int Nq_out, int Nq_cha)   /* :296-306 */
{
    if (n->type == OR_ROOT) n->K = Nq_out;
    else if (n->type == OR_CHA) n->K = Nq_cha;
    else n->K = Nq_in;
    for (int i = 0; i < n->nchild; i++) set_res_rec(n->child[i], Nq_in, Nq_out, Nq_cha);
}
void or_tree_set_resolution(or_tree *t, int Nq_in, int Nq_out, int Nq_cha) { set_res_rec(t->root, Nq_in, Nq_out, Nq_cha); }

static void set_leaves_rec(or_node *n, or_dvec p_msg, or_dvec p_cha)    /* :92-103 */
{
    if (n->type == OR_MSG) { or_dvec_free(&n->p); n->p = or_dvec_copy(p_msg); }
    else if (n->type == OR_CHA) { or_dvec_free(&n->p); n->p = or_dvec_copy(p_cha); }
    else for (int i = 0; i < n->nchild; i++) set_leaves_rec(n->child[i], p_msg, p_cha);
}
void or_tree_set_leaves(or_tree *t, or_dvec p_msg, or_dvec p_cha) { set_leaves_rec(t->root, p_msg, p_cha); }

static void reset_rec(or_node *n) { for (int i = 0; i < n->nchild; i++) reset_rec(n->child[i]); or_dvec_free(&n->p); n->p = or_dvec_new(0); }
void or_tree_reset_pmfs(or_tree *t) { if (t && t->root) reset_rec(t->root); }

static void normalise(or_dvec *p)
{
    double s = 0;
    for (int i = 0; i < p->n; i++) s += p->v[i];
    for (int i = 0; i < p->n; i++) p->v[i] = p->v[i] / s;
}

/* reuse branch shared by var_update / chk_update (:715-723, :749-758) */
static void requantise(or_dvec *p_out, const or_ivec *Q, or_dvec prod, int Nq)
{
    int M = prod.n;
    or_dvec_free(p_out);
    *p_out = or_dvec_new(Nq);
    for (int mm = 0; mm < M; mm++) {
        if (mm < M / 2) p_out->v[Q->v[mm]] += prod.v[mm];
        else p_out->v[Nq - 1 - Q->v[M - 1 - mm]] += prod.v[mm];
    }
}

/* Design a symmetric quantiser for `prod` after dropping label pairs of zero mass; the
 * dropped labels get the least confident outputs.  Returns the FULL length-M map.
 * (LUT_Tree.cpp:725-737 and LDPC_DE.cpp:1428-1440 share this construction.) */
or_ivec or__design_skip_zero_mass(or_dvec *p_out, or_dvec prod, int Nq)
{
    int M = prod.n, nnz = 0;
    unsigned char *nz = (unsigned char *)malloc((size_t)M);
    for (int mm = 0; mm < M; mm++) { nz[mm] = (.5 * (prod.v[mm] + prod.v[M - 1 - mm]) != 0); nnz += nz[mm]; }
    or_dvec pnz = or_dvec_new(nnz);
    for (int mm = 0, k = 0; mm < M; mm++) if (nz[mm]) pnz.v[k++] = prod.v[mm];
    or_ivec Qnz;
    or_dvec_free(p_out);
    (void)or_quant_mi_sym(p_out, &Qnz, pnz, Nq, 0);
    or_ivec Q = or_ivec_new(M);
    for (int mm = 0; mm < M; mm++) Q.v[mm] = mm < M / 2 ? Nq / 2 - 1 : Nq / 2;
    for (int mm = 0, k = 0; mm < M; mm++) if (nz[mm]) Q.v[mm] = Qnz.v[k++];
    or_ivec_free(&Qnz); or_dvec_free(&pnz); free(nz);
    return Q;
}

/* LUT_Tree.cpp:709-742 */
static void var_update(or_node *n, const or_dvec *p_in, int nin, int reuse)
{
    or_dvec prod = or_var_product_pmf(p_in, nin);
    if (reuse) requantise(&n->p, &n->Q, prod, n->K);
    else {
        or_ivec Q = or__design_skip_zero_mass(&n->p, prod, n->K);
        or_ivec_free(&n->Q);
        n->Q = or_ivec_new(Q.n / 2);
        memcpy(n->Q.v, Q.v, sizeof(int) * (size_t)(Q.n / 2));
        or_ivec_free(&Q);
    }
    normalise(&n->p);
    or_dvec_free(&prod);
}

/* LUT_Tree.cpp:744-766 */
static void chk_update(or_node *n, const or_dvec *p_in, int nin, int reuse)
{
    or_dvec prod = or_chk_product_pmf(p_in, nin);
    if (reuse) requantise(&n->p, &n->Q, prod, n->K);
    else {
        or_ivec Q;
        or_dvec_free(&n->p);
        (void)or_quant_mi_sym(&n->p, &Q, prod, n->K, 0);
        or_ivec_free(&n->Q);
        n->Q = or_ivec_new(Q.n / 2);
        memcpy(n->Q.v, Q.v, sizeof(int) * (size_t)(Q.n / 2));
        or_ivec_free(&Q);
    }
    normalise(&n->p);
    or_dvec_free(&prod);
}

/* LUT_Tree.cpp:114-130 */
static or_dvec update_rec(or_node *n, int reuse, int chk)
{
    if (node_is_leaf(n)) return n->p;
    or_dvec *pc = (or_dvec *)malloc(sizeof(or_dvec) * (size_t)n->nchild);
    for (int i = 0; i < n->nchild; i++) pc[i] = update_rec(n->child[i], reuse, chk);
    if (chk) chk_update(n, pc, n->nchild, reuse); else var_update(n, pc, n->nchild, reuse);
    free(pc);
    return n->p;
}

/* LUT_Tree.cpp:683-698; the returned vector is a copy owned by the caller */
or_dvec or_tree_update(or_tree *t, int reuse)
{
    return or_dvec_copy(update_rec(t->root, reuse, t->type == OR_CHKTREE));
}

/* ------------------------------------------------------------------ evaluation */
typedef struct { const int *q; int pos; } queue_t;

/* LUT_Tree.cpp:402-418 */
static int var_eval(const or_node *n, queue_t *q)
{
    if (node_is_leaf(n)) return q->q[q->pos++];
    int label = 0, base = 1;
    for (int i = 0; i < n->nchild; i++) { label += base * var_eval(n->child[i], q); base *= n->child[i]->K; }
    if (label < n->Q.n) return n->Q.v[label];
    return n->K - 1 - n->Q.v[2 * n->Q.n - 1 - label];
}

/* LUT_Tree.cpp:420-445 */
static int chk_eval(const or_node *n, queue_t *q)
{
    if (n->type == OR_MSG) return q->q[q->pos++];
    int label = 0, base = 1, parity = 0;
    for (int i = 0; i < n->nchild; i++) {
        int s = chk_eval(n->child[i], q), r = n->child[i]->K;
        if (s < r / 2) { parity ^= 1; label += base * (r / 2 - 1 - s); }
        else label += base * (s - r / 2);
        base *= r / 2;
    }
    return parity == 1 ? n->Q.v[label] : n->K - 1 - n->Q.v[label];
}

/* LUT_Tree.cpp:774-790: for every output the full queue is copied and one element erased */
void or_tree_var_msg_update(const or_tree *t, const int *msgs_in, int d, int llr, int *out)
{
    int all[512], one[512];
    for (int i = 0; i < d; i++) all[i] = msgs_in[i];
    all[d] = llr;
    for (int ii = 0; ii < d; ii++) {
        int k = 0;
        for (int j = 0; j <= d; j++) if (j != ii) one[k++] = all[j];
        queue_t q = { one, 0 };
        out[ii] = var_eval(t->root, &q);
    }
}

/* LUT_Tree.cpp:792-807 */
void or_tree_chk_msg_update(const or_tree *t, const int *msgs_in, int d, int *out)
{
    int one[512];
    for (int ii = 0; ii < d; ii++) {
        int k = 0;
        for (int j = 0; j < d; j++) if (j != ii) one[k++] = msgs_in[j];
        queue_t q = { one, 0 };
        out[ii] = chk_eval(t->root, &q);
    }
}

/* LUT_Tree.cpp:809-820 */
int or_tree_dec_update(const or_tree *t, const int *msgs_in, int d, int llr)
{
    int all[512];
    for (int i = 0; i < d; i++) all[i] = msgs_in[i];
    all[d] = llr;
    queue_t q = { all, 0 };
    return var_eval(t->root, &q);
}

/* ------------------------------------------------------------------ serialisation */
typedef struct { char *s; size_t len, cap; } sbuf;
static void sb_printf_int(sbuf *b, long v, char term)
{
    if (b->len + 32 > b->cap) { b->cap = b->cap * 2 + 64; b->s = (char *)realloc(b->s, b->cap); }
    b->len += (size_t)sprintf(b->s + b->len, "%ld%c", v, term);
}

/* LUT_Tree.cpp:488-517 */
static void ser_rec(const or_node *n, sbuf *b)
{
    sb_printf_int(b, n->nchild, '\n');
    sb_printf_int(b, n->type, ' '); sb_printf_int(b, n->Q.n, ' '); sb_printf_int(b, n->K, '\n');
    if (n->Q.n > 0) for (int i = 0; i < n->Q.n; i++) sb_printf_int(b, n->Q.v[i], i == n->Q.n - 1 ? '\n' : ' ');
    for (int i = 0; i < n->nchild; i++) ser_rec(n->child[i], b);
}
static void ser_tree(const or_tree *t, sbuf *b)     /* :847-853 */
{
    sb_printf_int(b, t->type, ' '); sb_printf_int(b, t->num_leaves, '\n');
    ser_rec(t->root, b);
}
char *or_tree_serialize(const or_tree *t)
{
    sbuf b = { (char *)malloc(64), 0, 64 };
    ser_tree(t, &b);
    b.s[b.len] = 0;
    return b.s;
}
char *or_tree_array_serialize(const or_tree_array *a)   /* :855-864 */
{
    sbuf b = { (char *)malloc(64), 0, 64 };
    sb_printf_int(&b, a ? a->n_sets : 0, '\n');
    if (a) for (int i = 0; i < a->n_sets; i++) {
        sb_printf_int(&b, a->n_classes[i], '\n');
        for (int j = 0; j < a->n_classes[i]; j++) ser_tree(a->t[i][j], &b);
    }
    if (b.len + 1 > b.cap) b.s = (char *)realloc(b.s, b.len + 1);
    b.s[b.len] = 0;
    return b.s;
}

static int next_int(const char **c, long *v)
{
    char *e; *v = strtol(*c, &e, 10);
    if (e == *c) return 0;
    *c = e; return 1;
}
/* LUT_Tree.cpp:448-485,520-535 */
static or_node *deser_rec(const char **c)
{
    long nch, t, inres, outres;
    if (!next_int(c, &nch) || !next_int(c, &t) || !next_int(c, &inres) || !next_int(c, &outres)) return NULL;
    or_node *n = node_new((int)t);
    n->K = (int)outres;
    n->Q = or_ivec_new((int)inres);
    for (int i = 0; i < inres; i++) { long q; if (!next_int(c, &q)) { node_free(n); return NULL; } n->Q.v[i] = (int)q; }
    for (int i = 0; i < nch; i++) {
        or_node *ch = deser_rec(c);
        if (!ch) { node_free(n); return NULL; }
        node_add_back(n, ch);
    }
    return n;
}
or_tree *or_tree_deserialize(const char **cursor)   /* :868-891 */
{
    long t, nl;
    if (!next_int(cursor, &t) || !next_int(cursor, &nl)) return NULL;
    or_tree *tr = (or_tree *)calloc(1, sizeof(or_tree));
    tr->type = (int)t; tr->num_leaves = (int)nl;
    tr->root = deser_rec(cursor);
    if (!tr->root) { free(tr); return NULL; }
    return tr;
}
or_tree_array *or_tree_array_deserialize(const char *txt)   /* :893-927 */
{
    const char *c = txt; long ns;
    if (!txt || !next_int(&c, &ns)) return NULL;
    or_tree_array *a = (or_tree_array *)calloc(1, sizeof(or_tree_array));
    a->n_sets = (int)ns;
    a->n_classes = (int *)calloc((size_t)(ns ? ns : 1), sizeof(int));
    a->t = (or_tree ***)calloc((size_t)(ns ? ns : 1), sizeof(or_tree **));
    for (int i = 0; i < ns; i++) {
        long nc;
        if (!next_int(&c, &nc)) { or_tree_array_free(a); return NULL; }
        a->n_classes[i] = (int)nc;
        a->t[i] = (or_tree **)calloc((size_t)nc, sizeof(or_tree *));
        for (int j = 0; j < nc; j++) {
            a->t[i][j] = or_tree_deserialize(&c);
            if (!a->t[i][j]) { or_tree_array_free(a); return NULL; }
        }
    }
    return a;
}
void or_tree_array_free(or_tree_array *a)
{
    if (!a) return;
    for (int i = 0; i < a->n_sets; i++) {
        if (a->t[i]) for (int j = 0; j < a->n_classes[i]; j++) or_tree_free(a->t[i][j]);
        free(a->t[i]);
    }
    free(a->t); free(a->n_classes); free(a);
}
