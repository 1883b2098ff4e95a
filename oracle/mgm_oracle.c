/*
 * mgm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's MGM (/root/reference/pydcop/algorithms/mgm.py,
 * v0.1.2a1) on the flat factor graph format of include/maxsum_gpu.h (factors = constraints of the
 * constraints hypergraph, mgm.py:68).  MGM is bulk-synchronous by construction -- a computation
 * handles the values of a round only once ALL its neighbours' values are in, then the gains
 * likewise, and parks early messages (mgm.py:311-333, 476-497) -- so its result does not depend on
 * the delivery order and one lock-step round over all variables restates it exactly.
 *
 * Parity status: PINNED by tests/test_mgm_oracle_vs_reference.py (the reference's own
 * MgmComputation objects under oracle/ref_harness.run_reference_mgm) with the reference's three
 * uses of the unseeded `random` module made deterministic the same way on both sides:
 *   random.choice(domain)      -> the first value   (initial value, mgm.py:299)
 *   random.choice(best values) -> the first of them (mgm.py:381)
 *   random.random()            -> unused ("random" tie breaking never triggers in the reference:
 *                                 `self.break_mode == random` compares a string with the module,
 *                                 mgm.py:543 -- ties are always broken by name)
 * One order the reference itself does not fix: the variable costs of `concerned_vars` -- a SET of
 * Variable objects whose hash includes the name string, i.e. PYTHONHASHSEED -- are summed in set
 * order (mgm.py:366-370, 448-452).  Here: ascending variable index.  Identical whenever the sums are
 * exact (the pinned cases use costs on a binary grid); otherwise within rounding of each other,
 * like two runs of the reference.
 *
 * Quirks restated as they are (each cited below): the cost a variable holds is only updated when
 * the variable itself moves; the candidate evaluation adds the variable's own cost at its CURRENT
 * value; the winner of a neighbourhood is the LARGEST gain also in max mode.

 * A variable WITHOUT neighbours starts at the optimum of its own costs: optimal_cost_value takes min /
 * max over (cost, value) tuples (relations.py:1661-1665), i.e. cost ties break on the domain VALUE
 * (smallest for min, largest for max) -- `value_rank` carries the order of the values (round 3;
 * NULL = written in ascending order, then first index for min, last for max), pinned against the
 * reference with an unsorted string domain.
 * Deliberate deviations left, on BOTH sides of every test (oracle, engines), in corners no comparison
 * reaches: (1) such a variable without cost function gets random.choice and cost None in the
 * reference, index 0 and cost 0 here.
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

typedef struct mgmo_state {
    int32_t n_vars, n_factors, n_edges;
    int32_t *dom_size, *init_idx, *factor_rowptr, *edge_var, *var_rowptr, *var_edges, *edge_factor;
    int32_t *name_rank;   /* [n_vars] rank of the variable's name in sorted order (lexic ties) */
    int32_t *value_rank;  /* NULL: the domains are written in ascending order */
    int64_t *table_off, *cost_off;
    real *var_cost, *tables;
    double *var_cost64, *tables64;
    int is_max;
    int32_t *cur, *newv, *n_neigh;
    uint8_t *has_cost;
    real *cost, *gain;
    int64_t rounds;
} mgmo_state;

static void *dup_mem(const void *src, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (src && bytes) memcpy(p, src, bytes);
    return p;
}

/* c.slice(neighbours' values)(x): the table entry with v at x and every other scope variable at
 * its current value (mgm.py:352-356, 436-440) */
static real constraint_at(const mgmo_state *s, int32_t f, int32_t v, int x) {
    int64_t lin = 0;
    for (int32_t e = s->factor_rowptr[f]; e < s->factor_rowptr[f + 1]; ++e) {
        const int32_t u = s->edge_var[e];
        lin = lin * s->dom_size[u] + (u == v ? x : s->cur[u]);
    }
    return s->tables[s->table_off[f] + lin];
}

/* functools.reduce(operator.add, [f(x) for f in reduced_cs]): utilities order, no initial 0 */
static real utilities_at(const mgmo_state *s, int32_t v, int x) {
    real acc = 0;
    int first = 1;
    for (int32_t k = s->var_rowptr[v]; k < s->var_rowptr[v + 1]; ++k) {
        const real f = constraint_at(s, s->edge_factor[s->var_edges[k]], v, x);
        acc = first ? f : acc + f;
        first = 0;
    }
    return acc;
}

/* visit the distinct variables of v's constraints (v included) in ascending index */
typedef void (*visit_fn)(const mgmo_state *, int32_t v, int32_t u, void *ctx);
static void for_concerned(const mgmo_state *s, int32_t v, visit_fn fn, void *ctx) {
    /* small neighbourhoods: selection by repeated minimum, no allocation */
    int32_t last = -1;
    for (;;) {
        int32_t best = INT32_MAX;
        for (int32_t k = s->var_rowptr[v]; k < s->var_rowptr[v + 1]; ++k) {
            const int32_t f = s->edge_factor[s->var_edges[k]];
            for (int32_t e = s->factor_rowptr[f]; e < s->factor_rowptr[f + 1]; ++e) {
                const int32_t u = s->edge_var[e];
                if (u > last && u < best) best = u;
            }
        }
        if (best == INT32_MAX) break;
        fn(s, v, best, ctx);
        last = best;
    }
}

static void add_var_cost(const mgmo_state *s, int32_t v, int32_t u, void *ctx) {
    (void)v;
    *(real *)ctx += s->var_cost[s->cost_off[u] + s->cur[u]]; /* own: current value; neighbours: theirs */
}

/* phase 1 of a round for one variable: mgm.py:335-391 (_handle_value_message, all values in) */
static void compute_gain(mgmo_state *s, int32_t v) {
    if (s->n_neigh[v] == 0) return;
    if (!s->has_cost[v]) { /* first round: the cost of the current value, :349-372 */
        real cost = utilities_at(s, v, s->cur[v]);
        for_concerned(s, v, add_var_cost, &cost);
        s->cost[v] = cost;
        s->has_cost[v] = 1;
    }
    /* _compute_best_value, :428-454 + find_arg_optimal, relations.py:1554-1591 */
    real best = 0;
    int best_x = -1;
    for (int x = 0; x < s->dom_size[v]; ++x) {
        const real r = utilities_at(s, v, x);
        if (best_x < 0 || (s->is_max ? best < r : best > r)) { /* strictly better: new list */
            best = r;
            best_x = x; /* random.choice(best values) -> the first of them */
        }
    }
    real val_cost = best;
    for_concerned(s, v, add_var_cost, &val_cost); /* own cost at the CURRENT value, :449-450 */
    s->gain[v] = s->cost[v] - val_cost; /* :375 */
    if ((!s->is_max && s->gain[v] > 0) || (s->is_max && s->gain[v] < 0)) s->newv[v] = best_x; /* :376-381 */
    else s->newv[v] = s->cur[v];
}

/* phase 2: mgm.py:499-540 (_handle_gain_message, all gains in) + _break_ties :542-588 */
static void decide(const mgmo_state *s, int32_t v, int32_t *cur_out, real *cost_out) {
    *cur_out = s->cur[v];
    *cost_out = s->cost[v];
    if (s->n_neigh[v] == 0) return;
    real max_n = 0;
    int first = 1, wins_tie = 1;
    for (int32_t k = s->var_rowptr[v]; k < s->var_rowptr[v + 1]; ++k) {
        const int32_t f = s->edge_factor[s->var_edges[k]];
        for (int32_t e = s->factor_rowptr[f]; e < s->factor_rowptr[f + 1]; ++e) {
            const int32_t u = s->edge_var[e];
            if (u == v) continue;
            if (first || s->gain[u] > max_n) max_n = s->gain[u]; /* max() also in max mode, :513 */
            first = 0;
        }
    }
    for (int32_t k = s->var_rowptr[v]; k < s->var_rowptr[v + 1]; ++k) {
        const int32_t f = s->edge_factor[s->var_edges[k]];
        for (int32_t e = s->factor_rowptr[f]; e < s->factor_rowptr[f + 1]; ++e) {
            const int32_t u = s->edge_var[e];
            if (u != v && s->gain[u] == max_n && s->name_rank[u] < s->name_rank[v]) wins_tie = 0;
        }
    }
    if (s->gain[v] > max_n || (s->gain[v] == max_n && wins_tie)) { /* :514-525, lexic ties :566-588 */
        *cur_out = s->newv[v];
        *cost_out = s->cost[v] - s->gain[v];
    }
}

void mgmo_reset(mgmo_state *s) {
    s->rounds = 0;
    for (int32_t v = 0; v < s->n_vars; ++v) {
        s->has_cost[v] = 0;
        s->cost[v] = 0;
        s->gain[v] = 0;
        if (s->n_neigh[v] == 0) { /* on_start without neighbours: optimal_cost_value, :279-290 */
            const real *c = s->var_cost + s->cost_off[v];
            const int32_t *rk = s->value_rank ? s->value_rank + s->cost_off[v] : NULL;
            int best = 0;
            for (int d = 1; d < s->dom_size[v]; ++d) { /* min / max over (cost, value) tuples, relations.py:1661-1665 */
                const int rd = rk ? rk[d] : d, rb = rk ? rk[best] : best;
                if (s->is_max ? (c[d] > c[best] || (c[d] == c[best] && rd > rb))
                              : (c[d] < c[best] || (c[d] == c[best] && rd < rb))) best = d;
            }
            s->cur[v] = best;
            s->cost[v] = c[best];
            s->has_cost[v] = 1;
        } else { /* initial value, else random.choice(domain) -> the first value, :296-305 */
            s->cur[v] = (s->init_idx && s->init_idx[v] >= 0) ? s->init_idx[v] : 0;
        }
        s->newv[v] = s->cur[v];
    }
}

/* the order of every variable's domain values (include/maxsum_gpu.h, mxs_mgm_set_value_rank); resets */
void mgmo_set_value_rank(mgmo_state *s, const int32_t *rank) {
    free(s->value_rank);
    s->value_rank = rank ? (int32_t *)dup_mem(rank, sizeof(int32_t) * (size_t)s->cost_off[s->n_vars]) : NULL;
    mgmo_reset(s);
}

mgmo_state *mgmo_create(const mxs_graph *g, const mxs_params *p, const int32_t *name_rank) {
    mgmo_state *s = (mgmo_state *)calloc(1, sizeof(*s));
    s->n_vars = g->n_vars;
    s->n_factors = g->n_factors;
    s->n_edges = g->n_edges;
    s->is_max = p->mode == MXS_MODE_MAX;
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
    s->name_rank = (int32_t *)malloc(sizeof(int32_t) * (g->n_vars ? g->n_vars : 1));
    for (int32_t v = 0; v < g->n_vars; ++v) s->name_rank[v] = name_rank ? name_rank[v] : v;
    s->cost_off = (int64_t *)malloc(sizeof(int64_t) * (g->n_vars + 1));
    s->cost_off[0] = 0;
    for (int32_t v = 0; v < g->n_vars; ++v) s->cost_off[v + 1] = s->cost_off[v] + g->dom_size[v];
    const int64_t nc = s->cost_off[g->n_vars], nt = g->table_off[g->n_factors];
    s->var_cost64 = (double *)dup_mem(g->eval_var_cost ? g->eval_var_cost : g->var_cost, sizeof(double) * nc);
    s->tables64 = (double *)dup_mem(g->tables, sizeof(double) * nt);
    s->var_cost = (real *)malloc(sizeof(real) * (nc ? nc : 1));
    s->tables = (real *)malloc(sizeof(real) * (nt ? nt : 1));
    for (int64_t i = 0; i < nc; ++i) s->var_cost[i] = (real)g->var_cost[i];
    for (int64_t i = 0; i < nt; ++i) s->tables[i] = (real)g->tables[i];
    const size_t nv = g->n_vars ? g->n_vars : 1;
    s->cur = (int32_t *)calloc(nv, sizeof(int32_t));
    s->newv = (int32_t *)calloc(nv, sizeof(int32_t));
    s->n_neigh = (int32_t *)calloc(nv, sizeof(int32_t));
    s->has_cost = (uint8_t *)calloc(nv, 1);
    s->cost = (real *)calloc(nv, sizeof(real));
    s->gain = (real *)calloc(nv, sizeof(real));
    for (int32_t v = 0; v < g->n_vars; ++v) /* a variable has neighbours iff a non-unary constraint */
        for (int32_t k = g->var_rowptr[v]; k < g->var_rowptr[v + 1]; ++k) {
            const int32_t f = s->edge_factor[g->var_edges[k]];
            if (g->factor_rowptr[f + 1] - g->factor_rowptr[f] > 1) s->n_neigh[v] = 1;
        }
    mgmo_reset(s);
    return s;
}

/* n rounds = values, gains, decisions (the reference with stop_cycle = n + 1, mgm.py:407-411) */
void mgmo_run(mgmo_state *s, int32_t n_rounds) {
    int32_t *cur2 = (int32_t *)malloc(sizeof(int32_t) * (s->n_vars ? s->n_vars : 1));
    real *cost2 = (real *)malloc(sizeof(real) * (s->n_vars ? s->n_vars : 1));
    for (int32_t r = 0; r < n_rounds; ++r) {
        for (int32_t v = 0; v < s->n_vars; ++v) compute_gain(s, v);
        for (int32_t v = 0; v < s->n_vars; ++v) decide(s, v, &cur2[v], &cost2[v]);
        memcpy(s->cur, cur2, sizeof(int32_t) * s->n_vars);
        memcpy(s->cost, cost2, sizeof(real) * s->n_vars);
        s->rounds += 1;
    }
    free(cur2);
    free(cost2);
}

int64_t mgmo_rounds(const mgmo_state *s) { return s->rounds; }

void mgmo_get_state(const mgmo_state *s, int32_t *idx, double *cost, uint8_t *has_cost, double *gain, int32_t *newv) {
    for (int32_t v = 0; v < s->n_vars; ++v) {
        if (idx) idx[v] = s->cur[v];
        if (cost) cost[v] = (double)s->cost[v];
        if (has_cost) has_cost[v] = s->has_cost[v];
        if (gain) gain[v] = (double)s->gain[v];
        if (newv) newv[v] = s->newv[v];
    }
}

void mgmo_eval_cost(const mgmo_state *s, const int32_t *idx, double infinity, double *cost, int64_t *violations) {
    if (!idx) idx = s->cur;
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

void mgmo_destroy(mgmo_state *s) {
    if (!s) return;
    free(s->dom_size); free(s->init_idx); free(s->factor_rowptr); free(s->edge_var); free(s->var_rowptr);
    free(s->var_edges); free(s->edge_factor); free(s->name_rank); free(s->value_rank); free(s->table_off); free(s->cost_off);
    free(s->var_cost); free(s->tables); free(s->var_cost64); free(s->tables64);
    free(s->cur); free(s->newv); free(s->n_neigh); free(s->has_cost); free(s->cost); free(s->gain);
    free(s);
}
