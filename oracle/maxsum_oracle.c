/*
 * maxsum_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU restatement of the reference's synchronous Max-Sum
 * (/root/reference/pydcop/algorithms/maxsum.py, v0.1.2a1) on the flat factor
 * graph format of include/maxsum_gpu.h.  It is the checker for the HIP engine:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (pydcop_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py runs it against the
 * reference's own MaxSum{Factor,Variable}Computation objects (driven by
 * oracle/ref_harness.py) in the build container, and tests/golden/ holds
 * vectors generated from the reference by oracle/make_golden.py for the GPU box.
 *
 * Every arithmetic expression follows the reference's evaluation order so the
 * f64 build agrees with the reference to the last bit wherever the reference's
 * own order is deterministic (the one exception: select_value sums the factor
 * messages in dict-arrival order, maxsum.py:609; we use links order).
 *
 * Schedule (verified against the reference): every cycle
 * reads only the messages of the previous cycle (Jacobi), because the BSP
 * barrier of pydcop/infrastructure/computations.py:684-788 delivers in cycle t
 * what was posted in cycle t-1.
 *
 * Build: see oracle/Makefile (-DMSO_REAL=float gives the f32 twin used to check
 * the engine's MXS_DTYPE_F32 mode op for op).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/maxsum_gpu.h"

#ifndef MSO_REAL
#define MSO_REAL double
#endif
typedef MSO_REAL real;

#define SAME_COUNT 4 /* maxsum.py:106 */

typedef struct mso_state {
    int32_t n_vars, n_factors, n_edges;
    int32_t *dom_size, *init_idx, *factor_rowptr, *edge_var, *var_rowptr, *var_edges;
    int64_t *table_off;
    int64_t *cost_off;  /* [n_vars+1] prefix of dom_size */
    int64_t *msg_off;   /* [n_edges+1] */
    real *var_cost, *tables;
    double *var_cost64, *tables64; /* untouched copies for eval_cost */
    uint8_t *var_owned;
    mxs_params p;
    /* double-buffered messages: what each receiver currently holds */
    real *v2f[2], *f2v[2];
    uint8_t *cnt_v[2], *cnt_f[2]; /* send counters, 0 == "prev is None" */
    int cur;
    int32_t *sel;
    real *belief;
    int64_t cycles;
    int threads;
} mso_state;

static void *dup_mem(const void *src, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (src && bytes) memcpy(p, src, bytes);
    return p;
}

/* ---- the five pure functions ------------------------------------------ */

/* factor_costs_for_var, maxsum.py:382-447.  Message of factor f to the variable
 * at scope position `pos`.  `v2f` holds the last message received from every
 * scope variable (zeros == "no message yet", which the reference skips,
 * maxsum.py:430-436: adding 0.0 is exact). */
static void factor_costs_for_var(const mso_state *s, int32_t f, int pos,
                                 const real *v2f, real *out) {
    const int32_t e0 = s->factor_rowptr[f];
    const int arity = s->factor_rowptr[f + 1] - e0;
    const real *tab = s->tables + s->table_off[f];
    int dims[64];
    int64_t stride[64];
    int idx[64];
    int64_t st = 1;
    for (int i = arity - 1; i >= 0; --i) {
        dims[i] = s->dom_size[s->edge_var[e0 + i]];
        stride[i] = st;
        st *= dims[i];
    }
    const int is_max = (s->p.mode == MXS_MODE_MAX);
    for (int d = 0; d < dims[pos]; ++d) {
        real optimal = is_max ? (real)-INFINITY : (real)INFINITY; /* :419 */
        for (int i = 0; i < arity; ++i) idx[i] = 0;
        idx[pos] = d;
        /* every assignment of the other variables (generate_assignment_as_dict,
         * pydcop/dcop/relations.py:1452-1476; enumeration order does not change
         * the optimum) */
        for (;;) {
            int64_t lin = 0;
            for (int i = 0; i < arity; ++i) lin += idx[i] * stride[i];
            const real f_val = tab[lin]; /* factor(**assignment), :423 */
            real sum_cost = 0;           /* :425 */
            for (int i = 0; i < arity; ++i) { /* others, in dimensions order :427 */
                if (i == pos) continue;
                sum_cost += v2f[s->msg_off[e0 + i] + idx[i]]; /* :433 */
            }
            const real current = f_val + sum_cost; /* :438 */
            if ((!is_max && optimal > current) || (is_max && optimal < current))
                optimal = current; /* :439-443 */
            int i = arity - 1;
            for (; i >= 0; --i) {
                if (i == pos) continue;
                if (++idx[i] < dims[i]) break;
                idx[i] = 0;
            }
            if (i < 0) break;
        }
        out[d] = optimal;
    }
}

/* costs_for_factor, maxsum.py:623-676.  Message of variable v to the factor
 * behind var-side slot `kout`.  NB the average only holds the received costs,
 * not the variable's own cost (:653-671). */
static void costs_for_factor(const mso_state *s, int32_t v, int kout,
                             const real *f2v, real *out) {
    const int D = s->dom_size[v];
    const real *c = s->var_cost + s->cost_off[v];
    const int32_t k0 = s->var_rowptr[v], k1 = s->var_rowptr[v + 1];
    real sum_cost = 0; /* :653 */
    for (int d = 0; d < D; ++d) {
        real m = c[d]; /* :651 */
        for (int32_t k = k0; k < k1; ++k) {
            if (k - k0 == kout) continue; /* :656 */
            const real x = f2v[s->msg_off[s->var_edges[k]] + d];
            sum_cost += x; /* :663 */
            m += x;        /* :664 */
        }
        out[d] = m;
    }
    const real avg = sum_cost / (real)D; /* :671 */
    for (int d = 0; d < D; ++d) out[d] = out[d] - avg; /* :672-674 */
}

/* select_value, maxsum.py:584-620: first optimum in domain order. */
static void select_value(const mso_state *s, int32_t v, const real *f2v,
                         int32_t *sel, real *cost) {
    const int D = s->dom_size[v];
    const real *c = s->var_cost + s->cost_off[v];
    const int is_max = (s->p.mode == MXS_MODE_MAX);
    int best = 0;
    real best_c = 0;
    for (int d = 0; d < D; ++d) {
        real b = c[d]; /* :607 */
        for (int32_t k = s->var_rowptr[v]; k < s->var_rowptr[v + 1]; ++k)
            b += f2v[s->msg_off[s->var_edges[k]] + d]; /* :608-610 */
        if (d == 0 || (!is_max && b < best_c) || (is_max && b > best_c)) {
            best = d; /* min()/max() keep the first optimum, :615-618 */
            best_c = b;
        }
    }
    *sel = best;
    *cost = best_c;
}

/* approx_match, maxsum.py:688-710 (prev is not None here). */
static int approx_match(const real *costs, const real *prev, int D, real stability) {
    for (int d = 0; d < D; ++d) {
        const real c = costs[d], prev_c = prev[d];
        if (prev_c != c) {
            const real delta = (real)fabs((double)(prev_c - c));
            if (prev_c + c != 0) {
                if (!(((real)2 * delta / (real)fabs((double)(prev_c + c))) < stability))
                    return 0;
            } else {
                return 0;
            }
        }
    }
    return 1;
}

/* The send rule shared by both on_new_cycle (maxsum.py:349-377, 540-564) with
 * apply_damping (maxsum.py:679-685).  `newm` is overwritten with what the
 * receiver holds after this cycle. */
static void damp_and_filter(const mso_state *s, real *newm, const real *prev,
                            uint8_t cnt, int D, int damp_on, uint8_t *cnt_out) {
    const real damping = (real)s->p.damping;
    if (cnt > 0 && damp_on)
        for (int d = 0; d < D; ++d)
            newm[d] = damping * prev[d] + ((real)1 - damping) * newm[d]; /* :683 */
    const int match = cnt > 0 && approx_match(newm, prev, D, (real)s->p.stability);
    if (!match) {
        *cnt_out = 1; /* sent, :363-364 */
    } else if (cnt < SAME_COUNT) {
        *cnt_out = (uint8_t)(cnt + 1); /* sent again, :371-372 */
    } else {
        for (int d = 0; d < D; ++d) newm[d] = prev[d]; /* not sent: receiver keeps */
        *cnt_out = cnt;
    }
}

/* ---- cycles ----------------------------------------------------------- */

static void factor_cycle(mso_state *s, int32_t f, int start) {
    const int c = s->cur, n = c ^ 1;
    const int32_t e0 = s->factor_rowptr[f], e1 = s->factor_rowptr[f + 1];
    const int arity = e1 - e0;
    const int damp_on = (s->p.damping_nodes & MXS_DAMP_FACTORS) != 0;
    real buf[4096];
    for (int32_t e = e0; e < e1; ++e) {
        const int D = s->dom_size[s->edge_var[e]];
        real *out = s->f2v[n] + s->msg_off[e];
        if (start) { /* on_start, maxsum.py:305-328: counters stay 0 */
            const int sends = (arity == 1 && s->p.start_messages != MXS_START_ALL) ||
                              s->p.start_messages == MXS_START_ALL;
            if (sends) factor_costs_for_var(s, f, e - e0, s->v2f[c], out);
            else memset(out, 0, sizeof(real) * D);
            s->cnt_f[n][e] = 0;
            continue;
        }
        factor_costs_for_var(s, f, e - e0, s->v2f[c], buf);
        damp_and_filter(s, buf, s->f2v[c] + s->msg_off[e], s->cnt_f[c][e], D, damp_on,
                        &s->cnt_f[n][e]);
        memcpy(out, buf, sizeof(real) * D);
    }
}

static void variable_cycle(mso_state *s, int32_t v, int start) {
    const int c = s->cur, n = c ^ 1;
    const int32_t k0 = s->var_rowptr[v], k1 = s->var_rowptr[v + 1];
    const int deg = k1 - k0;
    const int D = s->dom_size[v];
    const int damp_on = (s->p.damping_nodes & MXS_DAMP_VARS) != 0;
    real buf[4096];
    if (s->var_owned && !s->var_owned[v]) { /* ghost: messages come from its owner */
        for (int32_t k = k0; k < k1; ++k) {
            const int32_t e = s->var_edges[k];
            memcpy(s->v2f[n] + s->msg_off[e], s->v2f[c] + s->msg_off[e], sizeof(real) * D);
            s->cnt_v[n][e] = s->cnt_v[c][e];
        }
        return;
    }
    if (start) { /* on_start, maxsum.py:495-523 */
        if (s->init_idx && s->init_idx[v] >= 0) {
            s->sel[v] = s->init_idx[v]; /* value_selection(initial_value) :498 */
            s->belief[v] = 0;
        } else {
            select_value(s, v, s->f2v[c], &s->sel[v], &s->belief[v]); /* :500 */
        }
        const int sends = (deg == 1 && s->p.start_messages == MXS_START_LEAFS) ||
                          s->p.start_messages != MXS_START_LEAFS;
        for (int32_t k = k0; k < k1; ++k) {
            const int32_t e = s->var_edges[k];
            real *out = s->v2f[n] + s->msg_off[e];
            if (sends) costs_for_factor(s, v, k - k0, s->f2v[c], out);
            else memset(out, 0, sizeof(real) * D);
            s->cnt_v[n][e] = 0;
        }
        return;
    }
    /* A variable without factor never cycles (no neighbour, no barrier). */
    if (deg == 0) return;
    select_value(s, v, s->f2v[c], &s->sel[v], &s->belief[v]); /* :532 */
    for (int32_t k = k0; k < k1; ++k) {
        const int32_t e = s->var_edges[k];
        costs_for_factor(s, v, k - k0, s->f2v[c], buf); /* :537 */
        damp_and_filter(s, buf, s->v2f[c] + s->msg_off[e], s->cnt_v[c][e], D, damp_on,
                        &s->cnt_v[n][e]);
        memcpy(s->v2f[n] + s->msg_off[e], buf, sizeof(real) * D);
    }
}

static void one_cycle(mso_state *s, int start) {
    int32_t i;
#pragma omp parallel for schedule(static) num_threads(s->threads)
    for (i = 0; i < s->n_factors; ++i) factor_cycle(s, i, start);
#pragma omp parallel for schedule(static) num_threads(s->threads)
    for (i = 0; i < s->n_vars; ++i) variable_cycle(s, i, start);
    s->cur ^= 1;
    if (!start) s->cycles++;
}

/* ---- C entry points (bound by oracle/maxsum_oracle.py) ---------------- */

void mso_reset(mso_state *s) {
    const int64_t nm = s->msg_off[s->n_edges];
    for (int b = 0; b < 2; ++b) {
        memset(s->v2f[b], 0, sizeof(real) * (nm ? nm : 1));
        memset(s->f2v[b], 0, sizeof(real) * (nm ? nm : 1));
        memset(s->cnt_v[b], 0, s->n_edges ? s->n_edges : 1);
        memset(s->cnt_f[b], 0, s->n_edges ? s->n_edges : 1);
    }
    s->cur = 0;
    s->cycles = 0;
    one_cycle(s, 1); /* cycle 0 == start(), computations.py:741-753 */
}

mso_state *mso_create(const mxs_graph *g, const mxs_params *p) {
    mso_state *s = (mso_state *)calloc(1, sizeof(*s));
    s->n_vars = g->n_vars;
    s->n_factors = g->n_factors;
    s->n_edges = g->n_edges;
    s->p = *p;
    s->threads = 1;
    s->dom_size = (int32_t *)dup_mem(g->dom_size, sizeof(int32_t) * g->n_vars);
    s->init_idx = g->init_idx ? (int32_t *)dup_mem(g->init_idx, sizeof(int32_t) * g->n_vars) : NULL;
    s->var_owned = g->var_owned ? (uint8_t *)dup_mem(g->var_owned, g->n_vars) : NULL;
    s->factor_rowptr = (int32_t *)dup_mem(g->factor_rowptr, sizeof(int32_t) * (g->n_factors + 1));
    s->edge_var = (int32_t *)dup_mem(g->edge_var, sizeof(int32_t) * g->n_edges);
    s->table_off = (int64_t *)dup_mem(g->table_off, sizeof(int64_t) * (g->n_factors + 1));
    s->var_rowptr = (int32_t *)dup_mem(g->var_rowptr, sizeof(int32_t) * (g->n_vars + 1));
    s->var_edges = (int32_t *)dup_mem(g->var_edges, sizeof(int32_t) * g->n_edges);
    s->cost_off = (int64_t *)malloc(sizeof(int64_t) * (g->n_vars + 1));
    s->cost_off[0] = 0;
    for (int32_t v = 0; v < g->n_vars; ++v) s->cost_off[v + 1] = s->cost_off[v] + g->dom_size[v];
    s->msg_off = (int64_t *)malloc(sizeof(int64_t) * (g->n_edges + 1));
    s->msg_off[0] = 0;
    for (int32_t e = 0; e < g->n_edges; ++e)
        s->msg_off[e + 1] = s->msg_off[e] + g->dom_size[g->edge_var[e]];
    const int64_t nc = s->cost_off[g->n_vars], nt = g->table_off[g->n_factors];
    s->var_cost64 = (double *)dup_mem(g->eval_var_cost ? g->eval_var_cost : g->var_cost, sizeof(double) * nc);
    s->tables64 = (double *)dup_mem(g->tables, sizeof(double) * nt);
    s->var_cost = (real *)malloc(sizeof(real) * (nc ? nc : 1));
    s->tables = (real *)malloc(sizeof(real) * (nt ? nt : 1));
    for (int64_t i = 0; i < nc; ++i) s->var_cost[i] = (real)g->var_cost[i];
    for (int64_t i = 0; i < nt; ++i) s->tables[i] = (real)g->tables[i];
    const int64_t nm = s->msg_off[g->n_edges];
    for (int b = 0; b < 2; ++b) {
        s->v2f[b] = (real *)malloc(sizeof(real) * (nm ? nm : 1));
        s->f2v[b] = (real *)malloc(sizeof(real) * (nm ? nm : 1));
        s->cnt_v[b] = (uint8_t *)malloc(g->n_edges ? g->n_edges : 1);
        s->cnt_f[b] = (uint8_t *)malloc(g->n_edges ? g->n_edges : 1);
    }
    s->sel = (int32_t *)calloc(g->n_vars ? g->n_vars : 1, sizeof(int32_t));
    s->belief = (real *)calloc(g->n_vars ? g->n_vars : 1, sizeof(real));
    mso_reset(s);
    return s;
}

void mso_set_threads(mso_state *s, int n) { s->threads = n < 1 ? 1 : n; }

void mso_run(mso_state *s, int32_t n_cycles) {
    for (int32_t t = 0; t < n_cycles; ++t) one_cycle(s, 0);
}

int64_t mso_cycle_count(const mso_state *s) { return s->cycles; }

void mso_get_assignment(const mso_state *s, int32_t *idx, double *belief) {
    for (int32_t v = 0; v < s->n_vars; ++v) {
        if (idx) idx[v] = s->sel[v];
        if (belief) belief[v] = (double)s->belief[v];
    }
}

void mso_get_messages(const mso_state *s, double *v2f, double *f2v, uint8_t *cv, uint8_t *cf) {
    const int64_t nm = s->msg_off[s->n_edges];
    for (int64_t i = 0; i < nm; ++i) {
        if (v2f) v2f[i] = (double)s->v2f[s->cur][i];
        if (f2v) f2v[i] = (double)s->f2v[s->cur][i];
    }
    if (cv) memcpy(cv, s->cnt_v[s->cur], s->n_edges);
    if (cf) memcpy(cf, s->cnt_f[s->cur], s->n_edges);
}

/* The inverse of mso_get_messages / mso_get_assignment (mxs_set_state on the device):
 * checkpoint / resume, and carrying a run over to a changed graph. */
void mso_set_state(mso_state *s, const double *v2f, const double *f2v, const uint8_t *cv,
                   const uint8_t *cf, const int32_t *idx, const double *belief, int64_t cycles) {
    const int64_t nm = s->msg_off[s->n_edges];
    for (int64_t i = 0; i < nm; ++i) {
        if (v2f) s->v2f[s->cur][i] = (real)v2f[i];
        if (f2v) s->f2v[s->cur][i] = (real)f2v[i];
    }
    if (cv) memcpy(s->cnt_v[s->cur], cv, s->n_edges);
    if (cf) memcpy(s->cnt_f[s->cur], cf, s->n_edges);
    for (int32_t v = 0; v < s->n_vars; ++v) {
        if (idx) s->sel[v] = idx[v];
        if (belief) s->belief[v] = (real)belief[v];
    }
    s->cycles = cycles;
}

/* Overwrite the V->F message a ghost edge holds (what mxs_halo_* does on the
 * device); lets the sharding logic be tested on CPU. */
void mso_set_v2f(mso_state *s, int32_t e, const double *msg, uint8_t cnt) {
    const int D = s->dom_size[s->edge_var[e]];
    for (int d = 0; d < D; ++d) s->v2f[s->cur][s->msg_off[e] + d] = (real)msg[d];
    s->cnt_v[s->cur][e] = cnt;
}

/* change_factor_function, pydcop/algorithms/maxsum_dynamic.py:80-104: same scope,
 * new costs; the messages carry on. */
void mso_update_table(mso_state *s, int32_t f, const double *table) {
    const int64_t lo = s->table_off[f], hi = s->table_off[f + 1];
    for (int64_t k = lo; k < hi; ++k) {
        s->tables64[k] = table[k - lo];
        s->tables[k] = (real)table[k - lo];
    }
}

/* solution_cost, pydcop/dcop/dcop.py:319-367: always f64, host tables. */
void mso_eval_cost(const mso_state *s, const int32_t *idx, double infinity, double *cost,
                   int64_t *violations) {
    if (!idx) idx = s->sel;
    double soft = 0;
    int64_t hard = 0;
    for (int32_t f = 0; f < s->n_factors; ++f) {
        const int32_t e0 = s->factor_rowptr[f], e1 = s->factor_rowptr[f + 1];
        int64_t lin = 0;
        for (int32_t e = e0; e < e1; ++e)
            lin = lin * s->dom_size[s->edge_var[e]] + idx[s->edge_var[e]];
        const double r = s->tables64[s->table_off[f] + lin];
        if (r != infinity) soft += r; else hard += 1; /* :352-355 */
    }
    for (int32_t v = 0; v < s->n_vars; ++v) {
        if (s->var_owned && !s->var_owned[v]) continue;
        const double c = s->var_cost64[s->cost_off[v] + idx[v]];
        if (c != infinity) soft += c; else hard += 1; /* :359-365 */
    }
    *cost = soft;
    *violations = hard;
}

void mso_destroy(mso_state *s) {
    if (!s) return;
    free(s->dom_size); free(s->init_idx); free(s->var_owned); free(s->factor_rowptr);
    free(s->edge_var); free(s->table_off); free(s->var_rowptr); free(s->var_edges);
    free(s->cost_off); free(s->msg_off); free(s->var_cost64); free(s->tables64);
    free(s->var_cost); free(s->tables);
    for (int b = 0; b < 2; ++b) { free(s->v2f[b]); free(s->f2v[b]); free(s->cnt_v[b]); free(s->cnt_f[b]); }
    free(s->sel); free(s->belief);
    free(s);
}

int mso_real_bytes(void) { return (int)sizeof(real); }

/* Test hook: approx_match on its own (the reference's unit tests pin it directly,
 * tests/unit/test_algorithms_amaxsum.py:160-203). */
int mso_approx_match(const double *costs, const double *prev, int32_t D, double stability) {
    real c[64], p[64];
    if (D > 64) return -1;
    for (int d = 0; d < D; ++d) {
        c[d] = (real)costs[d];
        p[d] = (real)prev[d];
    }
    return approx_match(c, p, D, (real)stability);
}
