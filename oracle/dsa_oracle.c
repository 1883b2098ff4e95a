/*
 * dsa_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's DSA (/root/reference/pydcop/algorithms/dsa.py,
 * v0.1.2a1: variants A, B, C; Zhang & al. 2005) on the flat factor graph format of
 * include/maxsum_gpu.h (factors = constraints of the constraints hypergraph, dsa.py:121).  DSA is
 * bulk-synchronous by construction: a computation evaluates a cycle once ALL its neighbours'
 * values of that cycle are in and parks the values of the next one (dsa.py:300-317), so one
 * lock-step pass over all variables per cycle restates it exactly.
 *
 * The reference draws from Python's unseeded `random` module: the initial value
 * (random_value_selection, dsa.py:291), the move test `probability > random.random()` and the
 * choice among the best values (dsa.py:413-414).  To have anything to pin, BOTH sides of every
 * comparison draw from ONE counter-based generator instead -- dsa_uniform(seed, variable, cycle,
 * draw) below, restated in oracle/ref_harness.py and patched into the reference's `random` for the
 * duration of a run (the harness knows which computation is handling a message):
 *   draw 0 of cycle 0      the initial value: domain[floor(u * |domain|)]
 *   draw 1 of cycle c + 1  the move test of the evaluation made at cycle_count == c
 *   draw 2 of cycle c + 1  the choice among the best values: best[floor(u * |best|)]
 * Parity status: PINNED under that generator by tests/test_dsa_oracle_vs_reference.py (the
 * reference's own DsaComputation objects, build container).  Against the unpatched reference only
 * the statistics can agree.
 *
 * Quirks restated as they are: variable costs never enter (find_optimal tests
 * hasattr(variable, "cost_for_value") -- the attribute is cost_for_val -- relations.py:1630);
 * Variable.initial_value is ignored (dsa.py:291); the cost a computation holds is 0 until its first
 * move (value_selection's default, computations.py:1057), then the best cost of that move.

 * A variable WITHOUT neighbours starts at the optimum of its own costs: optimal_cost_value takes min /
 * max over (cost, value) tuples (relations.py:1661-1665), i.e. cost ties break on the domain VALUE
 * (smallest for min, largest for max) -- `value_rank` carries the order of the values (round 3;
 * NULL = written in ascending order, then first index for min, last for max), pinned against the
 * reference with an unsorted string domain.
 * Deliberate deviations left, on BOTH sides of every test (oracle, engines), in corners no comparison
 * reaches: (1) such a variable without cost function gets random.choice and cost None in the
 * reference, index 0 and cost 0 here; (2) p_mode "arity" with a variable whose constraints are all
 * unary divides by zero in the reference (dsa.py:256-259), here it falls back to `probability`.
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

/* splitmix64 finaliser over a key built from (seed, variable, cycle, draw): the SAME function
 * in oracle/ref_harness.py (python ints masked to 64 bits) and in pydcop_amd/csrc/dsa.hip */
static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
double dsa_uniform(uint64_t seed, int32_t variable, int64_t cycle, int32_t draw) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)(uint32_t)variable + 1);
    z = mix64(z) + 0x9E3779B97F4A7C15ull * ((uint64_t)cycle + 1);
    z = mix64(z) + (uint64_t)(uint32_t)draw;
    return (double)(mix64(z) >> 11) * (1.0 / 9007199254740992.0); /* [0, 1), 53 bits */
}

typedef struct dsao_state {
    int32_t n_vars, n_factors, n_edges;
    int32_t *dom_size, *factor_rowptr, *edge_var, *var_rowptr, *var_edges, *edge_factor;
    int64_t *table_off, *cost_off;
    real *var_cost, *tables, *f_opt; /* f_opt[f]: find_optimum of the constraint (variant B) */
    double *var_cost64, *tables64;
    int is_max, variant;             /* variant 0 = A, 1 = B, 2 = C */
    int arity_mode;
    double probability;
    uint64_t seed;
    int32_t *cur, *n_neigh;
    int32_t *value_rank;             /* NULL: the domains are written in ascending order */
    real *cost;
    double *prob;                    /* per variable (p_mode arity: 1.2 / sum(arity - 1)) */
    int64_t cycles;
} dsao_state;

static void *dup_mem(const void *src, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (src && bytes) memcpy(p, src, bytes);
    return p;
}

static real constraint_at(const dsao_state *s, const int32_t *cur, int32_t f, int32_t v, int x) {
    int64_t lin = 0;
    for (int32_t e = s->factor_rowptr[f]; e < s->factor_rowptr[f + 1]; ++e) {
        const int32_t u = s->edge_var[e];
        lin = lin * s->dom_size[u] + (u == v ? x : cur[u]);
    }
    return s->tables[s->table_off[f] + lin];
}

/* assignment_cost, relations.py:1513-1533: cost = 0; cost += c(...) in constraints order */
static real assignment_cost(const dsao_state *s, const int32_t *cur, int32_t v, int x) {
    real cost = 0;
    for (int32_t k = s->var_rowptr[v]; k < s->var_rowptr[v + 1]; ++k)
        cost += constraint_at(s, cur, s->edge_factor[s->var_edges[k]], v, x);
    return cost;
}

void dsao_reset(dsao_state *s) {
    s->cycles = 0;
    for (int32_t v = 0; v < s->n_vars; ++v) {
        if (s->n_neigh[v] == 0) { /* optimal_cost_value, dsa.py:278-289 */
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
        } else { /* random_value_selection, dsa.py:291 */
            s->cur[v] = (int32_t)(dsa_uniform(s->seed, v, 0, 0) * s->dom_size[v]);
            s->cost[v] = 0;
        }
    }
}

/* the order of every variable's domain values (include/maxsum_gpu.h, mxs_dsa_set_value_rank); resets */
void dsao_set_value_rank(dsao_state *s, const int32_t *rank) {
    free(s->value_rank);
    s->value_rank = rank ? (int32_t *)dup_mem(rank, sizeof(int32_t) * (size_t)s->cost_off[s->n_vars]) : NULL;
    dsao_reset(s);
}

dsao_state *dsao_create(const mxs_graph *g, const mxs_params *p, int32_t variant, double probability,
                        int32_t arity_mode, uint64_t seed) {
    dsao_state *s = (dsao_state *)calloc(1, sizeof(*s));
    s->n_vars = g->n_vars;
    s->n_factors = g->n_factors;
    s->n_edges = g->n_edges;
    s->is_max = p->mode == MXS_MODE_MAX;
    s->variant = variant;
    s->probability = probability;
    s->arity_mode = arity_mode;
    s->seed = seed;
    s->dom_size = (int32_t *)dup_mem(g->dom_size, sizeof(int32_t) * g->n_vars);
    s->factor_rowptr = (int32_t *)dup_mem(g->factor_rowptr, sizeof(int32_t) * (g->n_factors + 1));
    s->edge_var = (int32_t *)dup_mem(g->edge_var, sizeof(int32_t) * g->n_edges);
    s->table_off = (int64_t *)dup_mem(g->table_off, sizeof(int64_t) * (g->n_factors + 1));
    s->var_rowptr = (int32_t *)dup_mem(g->var_rowptr, sizeof(int32_t) * (g->n_vars + 1));
    s->var_edges = (int32_t *)dup_mem(g->var_edges, sizeof(int32_t) * g->n_edges);
    s->edge_factor = (int32_t *)malloc(sizeof(int32_t) * (g->n_edges ? g->n_edges : 1));
    for (int32_t f = 0; f < g->n_factors; ++f)
        for (int32_t e = g->factor_rowptr[f]; e < g->factor_rowptr[f + 1]; ++e) s->edge_factor[e] = f;
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
    s->f_opt = (real *)malloc(sizeof(real) * (g->n_factors ? g->n_factors : 1));
    for (int32_t f = 0; f < g->n_factors; ++f) { /* find_optimum, relations.py:1367-1401 */
        real opt = s->tables[g->table_off[f]];
        for (int64_t k = g->table_off[f] + 1; k < g->table_off[f + 1]; ++k)
            if (s->is_max ? s->tables[k] > opt : s->tables[k] < opt) opt = s->tables[k];
        s->f_opt[f] = opt;
    }
    const size_t nv = g->n_vars ? g->n_vars : 1;
    s->cur = (int32_t *)calloc(nv, sizeof(int32_t));
    s->n_neigh = (int32_t *)calloc(nv, sizeof(int32_t));
    s->cost = (real *)calloc(nv, sizeof(real));
    s->prob = (double *)calloc(nv, sizeof(double));
    for (int32_t v = 0; v < g->n_vars; ++v) {
        int64_t n_count = 0;
        for (int32_t k = g->var_rowptr[v]; k < g->var_rowptr[v + 1]; ++k) {
            const int32_t f = s->edge_factor[g->var_edges[k]];
            const int ar = g->factor_rowptr[f + 1] - g->factor_rowptr[f];
            if (ar > 1) s->n_neigh[v] = 1;
            n_count += ar - 1;
        }
        s->prob[v] = (arity_mode && n_count > 0) ? 1.0 / (double)n_count * 1.2 : probability; /* dsa.py:256-259 */
    }
    dsao_reset(s);
    return s;
}

/* evaluate_cycle, dsa.py:319-359, for one variable; returns its value after the cycle */
static int32_t evaluate(dsao_state *s, const int32_t *cur, int32_t v, real *cost_io) {
    if (s->n_neigh[v] == 0) return cur[v];
    const int D = s->dom_size[v];
    /* find_optimal, relations.py:1622-1638: equality first, then strictly better */
    real best_cost = s->is_max ? (real)-INFINITY : (real)INFINITY;
    int n_best = 0, first_best = -1, has_cur = 0;
    for (int x = 0; x < D; ++x) {
        const real c = assignment_cost(s, cur, v, x);
        if (c == best_cost) {
            n_best += 1;
            if (x == cur[v]) has_cur = 1;
        } else if ((!s->is_max && c < best_cost) || (s->is_max && c > best_cost)) {
            best_cost = c;
            n_best = 1;
            first_best = x;
            has_cur = (x == cur[v]);
        }
    }
    const real current_cost = assignment_cost(s, cur, v, cur[v]);
    const real delta = (real)fabs((double)(current_cost - best_cost));
    int attempt = 0, drop_cur = 0;
    if (delta > 0) {
        attempt = 1;
    } else if (delta == 0) {
        if (s->variant == 1) { /* B: some constraint not at its optimum, dsa.py:421-433 */
            for (int32_t k = s->var_rowptr[v]; k < s->var_rowptr[v + 1] && !attempt; ++k) {
                const int32_t f = s->edge_factor[s->var_edges[k]];
                if (constraint_at(s, cur, f, v, cur[v]) != s->f_opt[f]) attempt = 1;
            }
        } else if (s->variant == 2) {
            attempt = 1;
        }
        if (attempt && n_best > 1 && has_cur) drop_cur = 1; /* best_values.remove(current_value) */
    }
    if (!attempt) return cur[v];
    if (!(s->prob[v] > dsa_uniform(s->seed, v, s->cycles + 1, 1))) return cur[v]; /* dsa.py:413 */
    /* random.choice(best_values): the j-th best value in domain order, the current one skipped */
    const int n = n_best - drop_cur;
    int j = (int)(dsa_uniform(s->seed, v, s->cycles + 1, 2) * n);
    int32_t pick = first_best;
    for (int x = 0; x < D; ++x) {
        if (assignment_cost(s, cur, v, x) != best_cost) continue;
        if (drop_cur && x == cur[v]) continue;
        if (j-- == 0) {
            pick = x;
            break;
        }
    }
    *cost_io = best_cost; /* value_selection(choice, best_cost) */
    return pick;
}

void dsao_run(dsao_state *s, int32_t n_cycles) {
    int32_t *next = (int32_t *)malloc(sizeof(int32_t) * (s->n_vars ? s->n_vars : 1));
    for (int32_t c = 0; c < n_cycles; ++c) {
        for (int32_t v = 0; v < s->n_vars; ++v) next[v] = evaluate(s, s->cur, v, &s->cost[v]);
        memcpy(s->cur, next, sizeof(int32_t) * s->n_vars);
        s->cycles += 1;
    }
    free(next);
}

int64_t dsao_cycles(const dsao_state *s) { return s->cycles; }

void dsao_get_state(const dsao_state *s, int32_t *idx, double *cost) {
    for (int32_t v = 0; v < s->n_vars; ++v) {
        if (idx) idx[v] = s->cur[v];
        if (cost) cost[v] = (double)s->cost[v];
    }
}

void dsao_eval_cost(const dsao_state *s, const int32_t *idx, double infinity, double *cost, int64_t *violations) {
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

void dsao_destroy(dsao_state *s) {
    if (!s) return;
    free(s->dom_size); free(s->factor_rowptr); free(s->edge_var); free(s->var_rowptr); free(s->var_edges);
    free(s->edge_factor); free(s->table_off); free(s->cost_off); free(s->var_cost); free(s->tables);
    free(s->f_opt); free(s->var_cost64); free(s->tables64); free(s->cur); free(s->n_neigh); free(s->cost);
    free(s->prob); free(s->value_rank);
    free(s);
}
