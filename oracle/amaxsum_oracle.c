/*
 * amaxsum_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's ASYNCHRONOUS Max-Sum
 * (/root/reference/pydcop/algorithms/amaxsum.py, v0.1.2a1) on the flat factor graph format of
 * include/maxsum_gpu.h, under ONE defined delivery order: every computation is started in graph
 * order (variables, then factors: ComputationsFactorGraph, factor_graph.py:223-235) and every
 * message goes through ONE first-in-first-out queue -- exactly what oracle/ref_harness.py does
 * with the reference's own computation objects (run_reference_amaxsum).  The threaded runtime
 * of the reference delivers in an order that depends on thread timing; a FIFO is the one order it
 * can be pinned on, and the order the GPU engine reproduces generation by generation
 * (generation 0 = the start messages, generation g + 1 = the messages sent while handling those of
 * generation g: a FIFO handles all of generation g before any of g + 1).
 *
 * Parity status: PINNED by tests/test_amaxsum_oracle_vs_reference.py (the reference's own
 * amaxsum computations, build container) and tests/golden/amaxsum_*.npz (oracle/make_golden.py).
 *
 * Arithmetic follows the reference expression by expression:
 *   factor  _on_maxsum_msg   amaxsum.py:191-250  (waits for all variables :206; skips the sender)
 *   variable _on_maxsum_msg  amaxsum.py:366-424  (select value, then every factor but the sender)
 *   on_start                 amaxsum.py:140-160, 295-332
 *   factor_costs_for_var / costs_for_factor / select_value / apply_damping / approx_match
 *                            maxsum.py:382-447, 623-676, 584-620, 679-685, 688-710
 * select_value sums the held factor costs in `costs.values()` order (maxsum.py:609), i.e. in the
 * order the factors FIRST sent to the variable -- tracked here (v_order).
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

typedef struct amso_state {
    int32_t n_vars, n_factors, n_edges;
    int32_t *dom_size, *init_idx, *factor_rowptr, *edge_var, *var_rowptr, *var_edges;
    int32_t *edge_factor;  /* [n_edges] */
    int32_t *edge_vpos;    /* [n_edges] position of the edge in its variable's links */
    int64_t *table_off, *cost_off, *msg_off;
    real *var_cost, *tables;
    double *var_cost64, *tables64;
    mxs_params p;
    /* factor side, per edge: what the factor holds from the variable / last sent to it */
    real *f_cost, *f_prev;
    uint8_t *f_has, *f_cnt;
    int32_t *f_nhas;       /* [n_factors] variables heard from */
    /* variable side, per edge */
    real *v_cost, *v_prev;
    uint8_t *v_has, *v_cnt;
    int32_t *v_order;      /* [n_edges] per variable: its edges in first-arrival order */
    int32_t *v_narr;       /* [n_vars] */
    int32_t *sel;
    real *belief;
    /* the queue */
    int32_t *q_edge;       /* edge id */
    uint8_t *q_dir;        /* 0: variable -> factor, 1: factor -> variable */
    int32_t *q_gen;
    int64_t *q_off;        /* payload offset */
    real *q_pay;
    int64_t q_head, q_tail, q_cap, pay_len, pay_cap;
    int64_t delivered;
    int32_t generation;    /* generation of the last delivered message, -1 before */
    int64_t *gen_sizes;    /* messages per generation */
    int32_t gen_cap;
} amso_state;

static void *dup_mem(const void *src, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (src && bytes) memcpy(p, src, bytes);
    return p;
}

static void q_push(amso_state *s, int32_t e, int dir, int gen, const real *msg, int D) {
    if (s->q_tail == s->q_cap) {
        s->q_cap = s->q_cap ? 2 * s->q_cap : 1024;
        s->q_edge = (int32_t *)realloc(s->q_edge, sizeof(int32_t) * s->q_cap);
        s->q_dir = (uint8_t *)realloc(s->q_dir, s->q_cap);
        s->q_gen = (int32_t *)realloc(s->q_gen, sizeof(int32_t) * s->q_cap);
        s->q_off = (int64_t *)realloc(s->q_off, sizeof(int64_t) * s->q_cap);
    }
    if (s->pay_len + D > s->pay_cap) {
        s->pay_cap = s->pay_cap ? 2 * s->pay_cap + D : 4096 + D;
        s->q_pay = (real *)realloc(s->q_pay, sizeof(real) * s->pay_cap);
    }
    s->q_edge[s->q_tail] = e;
    s->q_dir[s->q_tail] = (uint8_t)dir;
    s->q_gen[s->q_tail] = gen;
    s->q_off[s->q_tail] = s->pay_len;
    memcpy(s->q_pay + s->pay_len, msg, sizeof(real) * D);
    s->pay_len += D;
    s->q_tail += 1;
    if (gen >= s->gen_cap) {
        const int32_t nc = gen * 2 + 64;
        s->gen_sizes = (int64_t *)realloc(s->gen_sizes, sizeof(int64_t) * nc);
        for (int32_t g = s->gen_cap; g < nc; ++g) s->gen_sizes[g] = 0;
        s->gen_cap = nc;
    }
    s->gen_sizes[gen] += 1;
}

/* approx_match, maxsum.py:688-710 (prev is not None) */
static int approx_match(const real *c, const real *prev, int D, real stability) {
    for (int d = 0; d < D; ++d) {
        if (prev[d] != c[d]) {
            const real delta = (real)fabs((double)(prev[d] - c[d]));
            if (prev[d] + c[d] != 0) {
                if (!(((real)2 * delta / (real)fabs((double)(prev[d] + c[d]))) < stability)) return 0;
            } else {
                return 0;
            }
        }
    }
    return 1;
}

/* factor_costs_for_var, maxsum.py:382-447: costs of factor f for the variable at scope
 * position pos, from the costs the factor holds (a variable not heard from contributes 0). */
static void factor_costs_for_var(const amso_state *s, int32_t f, int pos, real *out) {
    const int32_t e0 = s->factor_rowptr[f];
    const int arity = s->factor_rowptr[f + 1] - e0;
    const real *tab = s->tables + s->table_off[f];
    int dims[64], idx[64];
    int64_t stride[64], st = 1;
    for (int i = arity - 1; i >= 0; --i) {
        dims[i] = s->dom_size[s->edge_var[e0 + i]];
        stride[i] = st;
        st *= dims[i];
    }
    const int is_max = (s->p.mode == MXS_MODE_MAX);
    for (int d = 0; d < dims[pos]; ++d) {
        real optimal = is_max ? (real)-INFINITY : (real)INFINITY;
        for (int i = 0; i < arity; ++i) idx[i] = 0;
        idx[pos] = d;
        for (;;) {
            int64_t lin = 0;
            for (int i = 0; i < arity; ++i) lin += idx[i] * stride[i];
            const real f_val = tab[lin];
            real sum_cost = 0;
            for (int i = 0; i < arity; ++i) {
                if (i == pos) continue;
                if (s->f_has[e0 + i]) sum_cost += s->f_cost[s->msg_off[e0 + i] + idx[i]]; /* :430-436 */
            }
            const real current = f_val + sum_cost;
            if ((!is_max && optimal > current) || (is_max && optimal < current)) optimal = current;
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

/* costs_for_factor, maxsum.py:623-676: message of variable v for the factor behind its slot kout;
 * factors in links order, those not heard from skipped, ONE running sum_cost. */
static void costs_for_factor(const amso_state *s, int32_t v, int kout, real *out) {
    const int D = s->dom_size[v];
    const int32_t k0 = s->var_rowptr[v], k1 = s->var_rowptr[v + 1];
    const real *c = s->var_cost + s->cost_off[v];
    real sum_cost = 0;
    for (int d = 0; d < D; ++d) {
        real m = c[d];
        for (int32_t k = k0; k < k1; ++k) {
            const int32_t e = s->var_edges[k];
            if (k == kout || !s->v_has[e]) continue;
            const real x = s->v_cost[s->msg_off[e] + d];
            sum_cost += x;
            m += x;
        }
        out[d] = m;
    }
    const real avg = sum_cost / (real)D;
    for (int d = 0; d < D; ++d) out[d] = out[d] - avg;
}

/* select_value, maxsum.py:584-620: held costs summed in first-arrival order; first index wins ties */
static void select_value(amso_state *s, int32_t v) {
    const int D = s->dom_size[v];
    const real *c = s->var_cost + s->cost_off[v];
    const int32_t k0 = s->var_rowptr[v];
    const int is_max = (s->p.mode == MXS_MODE_MAX);
    int best = 0;
    real best_c = 0;
    for (int d = 0; d < D; ++d) {
        real b = c[d];
        for (int r = 0; r < s->v_narr[v]; ++r) b += s->v_cost[s->msg_off[s->v_order[k0 + r]] + d];
        if (d == 0 || (!is_max && b < best_c) || (is_max && b > best_c)) {
            best = d;
            best_c = b;
        }
    }
    s->sel[v] = best;
    s->belief[v] = best_c;
}

/* damping + send rule shared by both sides (amaxsum.py:213-244, 386-424).
 * msg: computed costs, damped in place; returns 1 if it is sent (prev / count updated). */
static int damp_and_decide(const amso_state *s, real *msg, real *prev, uint8_t *cnt, int D, int damp_on) {
    const real damping = (real)s->p.damping;
    if (*cnt > 0 && damp_on) /* apply_damping: identity when prev is None */
        for (int d = 0; d < D; ++d) msg[d] = damping * prev[d] + ((real)1 - damping) * msg[d];
    const int match = *cnt > 0 && approx_match(msg, prev, D, (real)s->p.stability);
    if (!match) {
        memcpy(prev, msg, sizeof(real) * D);
        *cnt = 1;
        return 1;
    }
    if (*cnt < SAME_COUNT) {
        memcpy(prev, msg, sizeof(real) * D);
        *cnt = (uint8_t)(*cnt + 1);
        return 1;
    }
    return 0; /* same and already sent SAME_COUNT times */
}

static void deliver(amso_state *s, int64_t qi) {
    const int32_t e = s->q_edge[qi];
    const int gen = s->q_gen[qi];
    const int32_t v = s->edge_var[e], f = s->edge_factor[e];
    const int D = s->dom_size[v];
    const real *pay = s->q_pay + s->q_off[qi];
    real out[4096];
    if (s->q_dir[qi] == 0) { /* variable -> factor: amaxsum.py:191-250 */
        memcpy(s->f_cost + s->msg_off[e], pay, sizeof(real) * D);
        if (!s->f_has[e]) {
            s->f_has[e] = 1;
            s->f_nhas[f] += 1;
        }
        const int32_t e0 = s->factor_rowptr[f], e1 = s->factor_rowptr[f + 1];
        if (s->f_nhas[f] != e1 - e0) return; /* still waiting for some variable, :206 */
        const int damp_on = (s->p.damping_nodes == MXS_DAMP_FACTORS || s->p.damping_nodes == MXS_DAMP_BOTH);
        for (int32_t e2 = e0; e2 < e1; ++e2) {
            if (e2 == e) continue; /* not back to the sender, :208 */
            const int D2 = s->dom_size[s->edge_var[e2]];
            factor_costs_for_var(s, f, e2 - e0, out);
            if (damp_and_decide(s, out, s->f_prev + s->msg_off[e2], &s->f_cnt[e2], D2, damp_on))
                q_push(s, e2, 1, gen + 1, out, D2);
        }
    } else { /* factor -> variable: amaxsum.py:366-424 */
        memcpy(s->v_cost + s->msg_off[e], pay, sizeof(real) * D);
        if (!s->v_has[e]) {
            s->v_has[e] = 1;
            s->v_order[s->var_rowptr[v] + s->v_narr[v]] = e;
            s->v_narr[v] += 1;
        }
        select_value(s, v);
        const int damp_on = (s->p.damping_nodes == MXS_DAMP_VARS || s->p.damping_nodes == MXS_DAMP_BOTH);
        for (int32_t k = s->var_rowptr[v]; k < s->var_rowptr[v + 1]; ++k) {
            const int32_t e2 = s->var_edges[k];
            if (e2 == e) continue;
            costs_for_factor(s, v, k, out);
            if (damp_and_decide(s, out, s->v_prev + s->msg_off[e2], &s->v_cnt[e2], D, damp_on))
                q_push(s, e2, 0, gen + 1, out, D);
        }
    }
}

void amso_reset(amso_state *s) {
    const int64_t nm = s->msg_off[s->n_edges];
    memset(s->f_cost, 0, sizeof(real) * (nm ? nm : 1));
    memset(s->f_prev, 0, sizeof(real) * (nm ? nm : 1));
    memset(s->v_cost, 0, sizeof(real) * (nm ? nm : 1));
    memset(s->v_prev, 0, sizeof(real) * (nm ? nm : 1));
    memset(s->f_has, 0, s->n_edges ? s->n_edges : 1);
    memset(s->f_cnt, 0, s->n_edges ? s->n_edges : 1);
    memset(s->v_has, 0, s->n_edges ? s->n_edges : 1);
    memset(s->v_cnt, 0, s->n_edges ? s->n_edges : 1);
    memset(s->f_nhas, 0, sizeof(int32_t) * (s->n_factors ? s->n_factors : 1));
    memset(s->v_narr, 0, sizeof(int32_t) * (s->n_vars ? s->n_vars : 1));
    s->q_head = s->q_tail = 0;
    s->pay_len = 0;
    s->delivered = 0;
    s->generation = -1;
    for (int32_t g = 0; g < s->gen_cap; ++g) s->gen_sizes[g] = 0;
    real out[4096];
    /* start(): variables first, then factors (graph order) */
    for (int32_t v = 0; v < s->n_vars; ++v) { /* amaxsum.py:295-332 */
        if (s->init_idx && s->init_idx[v] >= 0) {
            s->sel[v] = s->init_idx[v];
            s->belief[v] = 0; /* value_selection(initial_value, None) */
        } else {
            select_value(s, v);
        }
        const int32_t k0 = s->var_rowptr[v], k1 = s->var_rowptr[v + 1];
        const int deg = k1 - k0;
        const int D = s->dom_size[v];
        if ((deg == 1 && s->p.start_messages == MXS_START_LEAFS) ||
            (s->p.start_messages == MXS_START_LEAFS_VARS || s->p.start_messages == MXS_START_ALL)) {
            for (int32_t k = k0; k < k1; ++k) {
                costs_for_factor(s, v, k, out);
                q_push(s, s->var_edges[k], 0, 0, out, D); /* post_msg only: _prev_messages untouched */
            }
        }
    }
    for (int32_t f = 0; f < s->n_factors; ++f) { /* amaxsum.py:140-160 */
        const int32_t e0 = s->factor_rowptr[f], e1 = s->factor_rowptr[f + 1];
        const int unary_leaf = (e1 - e0 == 1) && (s->p.start_messages == MXS_START_LEAFS ||
                                                  s->p.start_messages == MXS_START_LEAFS_VARS);
        if (unary_leaf || s->p.start_messages == MXS_START_ALL) {
            for (int32_t e = e0; e < e1; ++e) {
                factor_costs_for_var(s, f, e - e0, out);
                q_push(s, e, 1, 0, out, s->dom_size[s->edge_var[e]]);
            }
        }
    }
}

amso_state *amso_create(const mxs_graph *g, const mxs_params *p) {
    amso_state *s = (amso_state *)calloc(1, sizeof(*s));
    s->n_vars = g->n_vars;
    s->n_factors = g->n_factors;
    s->n_edges = g->n_edges;
    s->p = *p;
    s->dom_size = (int32_t *)dup_mem(g->dom_size, sizeof(int32_t) * g->n_vars);
    s->init_idx = g->init_idx ? (int32_t *)dup_mem(g->init_idx, sizeof(int32_t) * g->n_vars) : NULL;
    s->factor_rowptr = (int32_t *)dup_mem(g->factor_rowptr, sizeof(int32_t) * (g->n_factors + 1));
    s->edge_var = (int32_t *)dup_mem(g->edge_var, sizeof(int32_t) * g->n_edges);
    s->table_off = (int64_t *)dup_mem(g->table_off, sizeof(int64_t) * (g->n_factors + 1));
    s->var_rowptr = (int32_t *)dup_mem(g->var_rowptr, sizeof(int32_t) * (g->n_vars + 1));
    s->var_edges = (int32_t *)dup_mem(g->var_edges, sizeof(int32_t) * g->n_edges);
    s->edge_factor = (int32_t *)malloc(sizeof(int32_t) * (g->n_edges ? g->n_edges : 1));
    for (int32_t f = 0; f < g->n_factors; ++f)
        for (int32_t e = g->factor_rowptr[f]; e < g->factor_rowptr[f + 1]; ++e) s->edge_factor[e] = f;
    s->edge_vpos = (int32_t *)malloc(sizeof(int32_t) * (g->n_edges ? g->n_edges : 1));
    for (int32_t v = 0; v < g->n_vars; ++v)
        for (int32_t k = g->var_rowptr[v]; k < g->var_rowptr[v + 1]; ++k)
            s->edge_vpos[g->var_edges[k]] = k - g->var_rowptr[v];
    s->cost_off = (int64_t *)malloc(sizeof(int64_t) * (g->n_vars + 1));
    s->cost_off[0] = 0;
    for (int32_t v = 0; v < g->n_vars; ++v) s->cost_off[v + 1] = s->cost_off[v] + g->dom_size[v];
    s->msg_off = (int64_t *)malloc(sizeof(int64_t) * (g->n_edges + 1));
    s->msg_off[0] = 0;
    for (int32_t e = 0; e < g->n_edges; ++e) s->msg_off[e + 1] = s->msg_off[e] + g->dom_size[g->edge_var[e]];
    const int64_t nc = s->cost_off[g->n_vars], nt = g->table_off[g->n_factors], nm = s->msg_off[g->n_edges];
    s->var_cost64 = (double *)dup_mem(g->eval_var_cost ? g->eval_var_cost : g->var_cost, sizeof(double) * nc);
    s->tables64 = (double *)dup_mem(g->tables, sizeof(double) * nt);
    s->var_cost = (real *)malloc(sizeof(real) * (nc ? nc : 1));
    s->tables = (real *)malloc(sizeof(real) * (nt ? nt : 1));
    for (int64_t i = 0; i < nc; ++i) s->var_cost[i] = (real)g->var_cost[i];
    for (int64_t i = 0; i < nt; ++i) s->tables[i] = (real)g->tables[i];
    s->f_cost = (real *)malloc(sizeof(real) * (nm ? nm : 1));
    s->f_prev = (real *)malloc(sizeof(real) * (nm ? nm : 1));
    s->v_cost = (real *)malloc(sizeof(real) * (nm ? nm : 1));
    s->v_prev = (real *)malloc(sizeof(real) * (nm ? nm : 1));
    const size_t ne = g->n_edges ? g->n_edges : 1;
    s->f_has = (uint8_t *)malloc(ne);
    s->f_cnt = (uint8_t *)malloc(ne);
    s->v_has = (uint8_t *)malloc(ne);
    s->v_cnt = (uint8_t *)malloc(ne);
    s->v_order = (int32_t *)malloc(sizeof(int32_t) * ne);
    s->f_nhas = (int32_t *)malloc(sizeof(int32_t) * (g->n_factors ? g->n_factors : 1));
    s->v_narr = (int32_t *)malloc(sizeof(int32_t) * (g->n_vars ? g->n_vars : 1));
    s->sel = (int32_t *)calloc(g->n_vars ? g->n_vars : 1, sizeof(int32_t));
    s->belief = (real *)calloc(g->n_vars ? g->n_vars : 1, sizeof(real));
    amso_reset(s);
    return s;
}

/* Deliver every queued message of generations < max_generations (all of them when
 * max_generations < 0), at most max_messages (< 0: no limit).  Returns the number delivered. */
int64_t amso_run(amso_state *s, int32_t max_generations, int64_t max_messages) {
    int64_t n = 0;
    while (s->q_head < s->q_tail) {
        const int64_t qi = s->q_head;
        if (max_generations >= 0 && s->q_gen[qi] >= max_generations) break;
        if (max_messages >= 0 && n >= max_messages) break;
        s->q_head += 1;
        s->generation = s->q_gen[qi];
        deliver(s, qi);
        n += 1;
    }
    s->delivered += n;
    return n;
}

int64_t amso_delivered(const amso_state *s) { return s->delivered; }
int64_t amso_pending(const amso_state *s) { return s->q_tail - s->q_head; }
int32_t amso_generation(const amso_state *s) { return s->generation; }

/* messages per generation so far (sent, delivered or not) */
int32_t amso_generation_sizes(const amso_state *s, int64_t *out, int32_t cap) {
    int32_t n = 0;
    for (int32_t g = 0; g < s->gen_cap; ++g)
        if (s->gen_sizes[g]) n = g + 1;
    for (int32_t g = 0; g < n && g < cap; ++g) out[g] = s->gen_sizes[g];
    return n;
}

void amso_get_assignment(const amso_state *s, int32_t *idx, double *belief) {
    for (int32_t v = 0; v < s->n_vars; ++v) {
        if (idx) idx[v] = s->sel[v];
        if (belief) belief[v] = (double)s->belief[v];
    }
}

/* what every receiver holds (0 where nothing was received) and the last sent messages + counters */
void amso_get_messages(const amso_state *s, double *f_cost, double *v_cost, double *f_prev, double *v_prev,
                       uint8_t *f_has, uint8_t *v_has, uint8_t *f_cnt, uint8_t *v_cnt) {
    const int64_t nm = s->msg_off[s->n_edges];
    for (int64_t i = 0; i < nm; ++i) {
        if (f_cost) f_cost[i] = (double)s->f_cost[i];
        if (v_cost) v_cost[i] = (double)s->v_cost[i];
        if (f_prev) f_prev[i] = (double)s->f_prev[i];
        if (v_prev) v_prev[i] = (double)s->v_prev[i];
    }
    if (f_has) memcpy(f_has, s->f_has, s->n_edges);
    if (v_has) memcpy(v_has, s->v_has, s->n_edges);
    if (f_cnt) memcpy(f_cnt, s->f_cnt, s->n_edges);
    if (v_cnt) memcpy(v_cnt, s->v_cnt, s->n_edges);
}

void amso_eval_cost(const amso_state *s, const int32_t *idx, double infinity, double *cost, int64_t *violations) {
    if (!idx) idx = s->sel;
    double soft = 0;
    int64_t hard = 0;
    for (int32_t f = 0; f < s->n_factors; ++f) {
        int64_t lin = 0;
        for (int32_t e = s->factor_rowptr[f]; e < s->factor_rowptr[f + 1]; ++e)
            lin = lin * s->dom_size[s->edge_var[e]] + idx[s->edge_var[e]];
        const double r = s->tables64[s->table_off[f] + lin];
        if (r != infinity) soft += r; else hard += 1;
    }
    for (int32_t v = 0; v < s->n_vars; ++v) {
        const double c = s->var_cost64[s->cost_off[v] + idx[v]];
        if (c != infinity) soft += c; else hard += 1;
    }
    *cost = soft;
    *violations = hard;
}

/* change_factor_function with the same scope (maxsum_dynamic.py:80-104: `self.factor = fn`, nothing
 * is sent): the factor's table is replaced between two deliveries, held costs and last-sent
 * messages carry on.  `table`: row-major in the factor's own dimension order. */
void amso_update_table(amso_state *s, int32_t factor, const double *table) {
    const int64_t lo = s->table_off[factor], hi = s->table_off[factor + 1];
    for (int64_t i = lo; i < hi; ++i) {
        s->tables64[i] = table[i - lo];
        s->tables[i] = (real)table[i - lo];
    }
}

void amso_destroy(amso_state *s) {
    if (!s) return;
    free(s->dom_size); free(s->init_idx); free(s->factor_rowptr); free(s->edge_var); free(s->var_rowptr);
    free(s->var_edges); free(s->edge_factor); free(s->edge_vpos); free(s->table_off); free(s->cost_off);
    free(s->msg_off); free(s->var_cost); free(s->tables); free(s->var_cost64); free(s->tables64);
    free(s->f_cost); free(s->f_prev); free(s->v_cost); free(s->v_prev);
    free(s->f_has); free(s->f_cnt); free(s->v_has); free(s->v_cnt); free(s->v_order);
    free(s->f_nhas); free(s->v_narr); free(s->sel); free(s->belief);
    free(s->q_edge); free(s->q_dir); free(s->q_gen); free(s->q_off); free(s->q_pay); free(s->gen_sizes);
    free(s);
}
