// dsa.hip -- the reference's DSA (pydcop/algorithms/dsa.py: variants A, B, C; Zhang & al. 2005)
// on gfx950, on the same flat factor-graph arrays as the Max-Sum engine (SURVEY.md section
// 8(f).4).  DSA is bulk-synchronous by construction (a computation evaluates a cycle once ALL
// its neighbours' values of that cycle are in and parks the next ones, dsa.py:300-317): one cycle =
// ONE launch, reading the neighbours' values of the previous cycle and the variable's constraints'
// tables at them -- on the PACKED view (local_search.h: one lane per (variable, constraint), unary /
// binary constraints over domains of at most four values) where the instance allows it, thread per
// variable on the slot view or the CSR walk otherwise; bit-identical results.  The dynamic state lives
// in packed order (Dev::q).
//
// The reference draws from Python's unseeded `random` module (initial value, move test, choice
// among the best values).  Here every draw comes from a counter-based generator keyed on (seed,
// variable, cycle, draw) -- dsa_uniform, the same function in oracle/dsa_oracle.c and, patched into
// the reference's `random` for the duration of a run, in oracle/ref_harness.py -- so that the
// stochastic algorithm has a pinned parity: bit for bit the reference's own DsaComputation objects
// under that generator (tests/test_dsa_oracle_vs_reference.py), independent of scheduling.  The
// reference's quirks are restated as they are (variable costs never enter, initial values are
// ignored, the held cost is 0 until the first move): see oracle/dsa_oracle.c.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/maxsum_gpu.h"
#include "local_search.h"

extern "C" __attribute__((visibility("hidden"))) void mxs_set_last_error(const char* msg);  // engine.hip

namespace dsa {

constexpr int TPB = 64;  // one wave per block: 100k variables spread over every CU (latency-bound CSR walks)

static int fail(int code, const std::string& msg) {
    mxs_set_last_error(msg.c_str());
    return code;
}
#define DSA_TRY(call)                                                                     \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) return fail(MXS_E_HIP, std::string(#call) + " failed");     \
    } while (0)

template <typename U>
struct Buf {
    U* p = nullptr;
    size_t n = 0;
    hipError_t upload(const std::vector<U>& h, hipStream_t st) {
        release();
        n = h.size();
        hipError_t e = hipMalloc((void**)&p, (n ? n : 1) * sizeof(U));
        if (e != hipSuccess || h.empty()) return e;
        e = hipMemcpyAsync(p, h.data(), n * sizeof(U), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(st);
    }
    hipError_t alloc(size_t count) {
        n = count;
        return hipMalloc((void**)&p, (n ? n : 1) * sizeof(U));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~Buf() {
        if (p) (void)hipFree(p);
    }
};

// splitmix64 finaliser over a key of (seed, variable, cycle, draw): oracle/dsa_oracle.c, bit for bit
__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline double uniform(uint64_t seed, int32_t variable, int64_t cycle, int32_t draw) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)(uint32_t)variable + 1);
    z = mix64(z) + 0x9E3779B97F4A7C15ull * ((uint64_t)cycle + 1);
    z = mix64(z) + (uint64_t)(uint32_t)draw;
    return (double)(mix64(z) >> 11) * (1.0 / 9007199254740992.0);
}

// the same generator from its first stage, mix64(seed + G * (variable + 1)): a constant of the variable,
// computed once on the host for the packed kernel (two 64-bit multiplies per draw fewer)
__host__ __device__ inline uint64_t uniform_key(uint64_t seed, int32_t variable) {
    return mix64(seed + 0x9E3779B97F4A7C15ull * ((uint64_t)(uint32_t)variable + 1));
}
__host__ __device__ inline double uniform_from_key(uint64_t key, int64_t cycle, int32_t draw) {
    uint64_t z = key + 0x9E3779B97F4A7C15ull * ((uint64_t)cycle + 1);
    z = mix64(z) + (uint64_t)(uint32_t)draw;
    return (double)(mix64(z) >> 11) * (1.0 / 9007199254740992.0);
}

template <typename T>
struct Dev {
    int32_t n_vars, is_max, variant;
    uint64_t seed;
    int64_t cycle;        // cycle_count of the evaluation being made
    const int32_t *dom_size, *factor_rowptr, *edge_var, *edge_factor, *var_rowptr, *var_edges, *n_neigh;
    const int64_t* table_off;
    const T *tables, *f_opt;
    const double* prob;
    const int32_t* cur;
    int32_t* cur_out;
    T* cost;
    lsearch::Slots slots;
    lsearch::Pack pack;          // the packed view (local_search.h): lane per (variable, constraint)
    const T* pack_fopt;          // [lanes] optimum of the lane's constraint (variant B)
    const int32_t* var_list;     // the variables a thread-per-variable launch works on (NULL: all)
    int32_t n_list;
    // The dynamic state (cur, cost) is stored in PACKED ORDER, position q[v] = the variable's rank in the
    // packed view's wave order (the others after them): a packed wave's variables are q = first ..
    // first + nv - 1, what lane k = 0 of each reads and writes of its own state is one line per array
    // instead of one per variable (mgm.hip, Dev::q).  The packed view holds positions (nb); the
    // thread-per-variable kernels translate through q[]; the random draws stay keyed on graph indices.
    const int32_t* q;
    const int32_t* pack_dom;               // [packed variables] dom_size, in packed order
    const uint64_t* pack_key;              // [packed variables] uniform_key(seed, graph index): the random draws' key
    const double* pack_prob;               // [packed variables] the change probability
};

template <typename T>
__device__ T constraint_at(const Dev<T>& g, int f, int v, int x) {
    int64_t lin = 0;
    for (int e = g.factor_rowptr[f]; e < g.factor_rowptr[f + 1]; ++e) {
        const int u = g.edge_var[e];
        lin = lin * g.dom_size[u] + (u == v ? x : g.cur[g.q[u]]);
    }
    return g.tables[g.table_off[f] + lin];
}

// assignment_cost (relations.py:1513-1533): cost = 0; cost += c(...) in constraints order
template <typename T>
__device__ T assignment_cost(const Dev<T>& g, int v, int x) {
    T cost = (T)0;
    for (int k = g.var_rowptr[v]; k < g.var_rowptr[v + 1]; ++k)
        cost += constraint_at(g, g.edge_factor[g.var_edges[k]], v, x);
    return cost;
}

// evaluate_cycle, dsa.py:319-359 + variant_a/b/c :361-409 + probabilistic_change :411-419
template <typename T>
__global__ void __launch_bounds__(TPB) k_dsa_cycle(Dev<T> g) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.n_list) return;
    const int v = g.var_list ? g.var_list[tid] : tid;
    const int qv = g.q[v];
    const int mine = g.cur[qv];
    int out = mine;
    if (g.n_neigh[v] != 0) {
        const int D = g.dom_size[v];
        T best_cost = g.is_max ? -(T)INFINITY : (T)INFINITY;   // find_optimal, relations.py:1622-1638
        int n_best = 0, first_best = -1;
        bool has_cur = false;
        for (int x = 0; x < D; ++x) {
            const T c = assignment_cost(g, v, x);
            if (c == best_cost) {
                n_best += 1;
                if (x == mine) has_cur = true;
            } else if ((!g.is_max && c < best_cost) || (g.is_max && c > best_cost)) {
                best_cost = c;
                n_best = 1;
                first_best = x;
                has_cur = x == mine;
            }
        }
        const T current_cost = assignment_cost(g, v, mine);
        const T diff = current_cost - best_cost;
        const T delta = diff < (T)0 ? -diff : diff;
        bool attempt = false, drop_cur = false;
        if (delta > (T)0) {
            attempt = true;
        } else if (delta == (T)0) {
            if (g.variant == 1) {  // B: some constraint is not at its optimum (dsa.py:421-433)
                for (int k = g.var_rowptr[v]; k < g.var_rowptr[v + 1] && !attempt; ++k) {
                    const int f = g.edge_factor[g.var_edges[k]];
                    if (constraint_at(g, f, v, mine) != g.f_opt[f]) attempt = true;
                }
            } else if (g.variant == 2) {
                attempt = true;
            }
            if (attempt && n_best > 1 && has_cur) drop_cur = true;  // best_values.remove(current_value)
        }
        if (attempt && g.prob[v] > uniform(g.seed, v, g.cycle + 1, 1)) {
            const int n = n_best - (drop_cur ? 1 : 0);
            int j = (int)(uniform(g.seed, v, g.cycle + 1, 2) * n);
            int pick = first_best;
            for (int x = 0; x < D; ++x) {  // the j-th best value in domain order, the current one skipped
                if (assignment_cost(g, v, x) != best_cost) continue;
                if (drop_cur && x == mine) continue;
                if (j-- == 0) {
                    pick = x;
                    break;
                }
            }
            out = pick;
            g.cost[qv] = best_cost;  // value_selection(choice, best_cost)
        }
    }
    g.cur_out[qv] = out;
}

// the same cycle on the slot view (local_search.h): the D costs in registers, one pass over the
// variable's constraints instead of 2D+1 CSR walks; domains of at most MAXD values
template <typename T, int MAXD>
__global__ void __launch_bounds__(TPB) k_dsa_cycle_slots(Dev<T> g) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.n_list) return;
    const int v = g.var_list ? g.var_list[tid] : tid;
    const int qv = g.q[v];
    const int mine = g.cur[qv];
    int out = mine;
    if (g.n_neigh[v] != 0) {
        const int D = g.dom_size[v];
        const int s0 = g.var_rowptr[v], s1 = g.var_rowptr[v + 1];
        T c[MAXD];
        if (g.slots.rows != nullptr && s0 < s1 && g.slots.row_base[s0] >= 0) {  // contiguous rows (local_search.h)
            if (g.slots.rows_int8) lsearch::costs_of_values_rows<T, int8_t, MAXD>(g.slots, g.cur, s0, s1, D, true, c);
            else lsearch::costs_of_values_rows<T, T, MAXD>(g.slots, g.cur, s0, s1, D, true, c);
        } else {
            lsearch::costs_of_values<T, MAXD>(g.slots, g.tables, g.cur, s0, s1, D, true, c);
        }
        T best_cost = g.is_max ? -(T)INFINITY : (T)INFINITY;
        int n_best = 0, first_best = -1;
        bool has_cur = false;
#pragma unroll
        for (int x = 0; x < MAXD; ++x)
            if (x < D) {
                if (c[x] == best_cost) {
                    n_best += 1;
                    if (x == mine) has_cur = true;
                } else if ((!g.is_max && c[x] < best_cost) || (g.is_max && c[x] > best_cost)) {
                    best_cost = c[x];
                    n_best = 1;
                    first_best = x;
                    has_cur = x == mine;
                }
            }
        const T diff = lsearch::pick<T, MAXD>(c, mine) - best_cost;
        const T delta = diff < (T)0 ? -diff : diff;
        bool attempt = false, drop_cur = false;
        if (delta > (T)0) {
            attempt = true;
        } else if (delta == (T)0) {
            if (g.variant == 1) {
                for (int s = s0; s < s1 && !attempt; ++s) {
                    int64_t off = g.slots.base[s] + (int64_t)mine * g.slots.stride_v[s];
                    for (int k = g.slots.nb_rowptr[s]; k < g.slots.nb_rowptr[s + 1]; ++k)
                        off += (int64_t)g.cur[g.slots.nb_var[k]] * g.slots.nb_stride[k];
                    if (g.tables[off] != g.f_opt[g.edge_factor[g.var_edges[s]]]) attempt = true;
                }
            } else if (g.variant == 2) {
                attempt = true;
            }
            if (attempt && n_best > 1 && has_cur) drop_cur = true;
        }
        if (attempt && g.prob[v] > uniform(g.seed, v, g.cycle + 1, 1)) {
            const int n = n_best - (drop_cur ? 1 : 0);
            int j = (int)(uniform(g.seed, v, g.cycle + 1, 2) * n);
            int pick = first_best;
            bool done = false;
#pragma unroll
            for (int x = 0; x < MAXD; ++x)
                if (x < D && !done && c[x] == best_cost && !(drop_cur && x == mine)) {
                    if (j-- == 0) {
                        pick = x;
                        done = true;
                    }
                }
            out = pick;
            g.cost[qv] = best_cost;
        }
    }
    g.cur_out[qv] = out;
}

// the same cycle on the PACKED view (local_search.h): one lane per (variable, constraint), the
// constraint's entries for the neighbour's current value from the lane's private transposed
// record, the D costs by cross-lane sums in slot order; every lane of a variable then takes the
// same decision, lane k = 0 writes it.  TT = int8_t (records of small integers) or T.
constexpr int PACK_TPB = 256;
template <typename T, typename TT>
__global__ void __launch_bounds__(PACK_TPB) k_dsa_cycle_pack(Dev<T> g) {
    constexpr int MAXD = lsearch::PACK_D;
    const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= g.pack.n_lanes) return;  // whole waves (n_lanes is a multiple of 64)
    const lsearch::PackWave wm = g.pack.waves[__builtin_amdgcn_readfirstlane((int)(pos >> 6))];
    const uint32_t dn = (uint32_t)wm.deg_nv;
    const int deg = (int)(dn & 255u), nv = (int)((dn >> 8) & 255u);
    const int l = (int)threadIdx.x & 63;
    const int var = (int)(((uint32_t)l * (dn >> 16)) >> 15);  // l / deg (exact for l < 64)
    const int k = l - var * deg;
    const bool has = var < nv;
    const int qv = wm.first + (has ? var : 0);  // packed position: the index into the dynamic state
    const int seg = l - k;
    const int mine = g.cur[qv];
    const int D = g.pack_dom[qv];
    const uint64_t key = g.pack_key[qv];     // the variable's key of the random draws (its graph index inside)
    const double prob = g.pack_prob[qv];     // requested with the others, used after the decision
    T t[MAXD], c[MAXD];
    lsearch::pack_costs<T, TT>(g.pack, g.cur, pos, deg, seg, true, t, c);
    T best_cost = g.is_max ? -(T)INFINITY : (T)INFINITY;   // find_optimal, relations.py:1622-1638
    int n_best = 0, first_best = -1;
    bool has_cur = false;
    // (selects, no branches: "equal" and "better" exclude each other)
#pragma unroll
    for (int x = 0; x < MAXD; ++x) {
        const bool in = x < D;
        const bool lt = c[x] < best_cost, gt = c[x] > best_cost;
        const bool eq = in & (c[x] == best_cost);
        const bool better = in & (g.is_max ? gt : lt);
        n_best = better ? 1 : n_best + (eq ? 1 : 0);
        first_best = better ? x : first_best;
        has_cur = better ? x == mine : (has_cur || (eq && x == mine));
        best_cost = better ? c[x] : best_cost;
    }
    const T diff = lsearch::pick<T, MAXD>(c, mine) - best_cost;
    const T delta = diff < (T)0 ? -diff : diff;
    // variant B (dsa.py:421-433): some constraint of the variable is not at its optimum -- the
    // lanes of the variable vote (a padding lane has no constraint)
    bool off_opt = false;
    if (g.variant == 1 && has) off_opt = lsearch::pick<T, MAXD>(t, mine) != g.pack_fopt[pos];
    const unsigned long long votes = __ballot(off_opt ? 1 : 0);
    const unsigned long long mine_mask = (deg >= 64 ? ~0ull : ((1ull << deg) - 1ull)) << seg;
    bool attempt = false, drop_cur = false;
    if (delta > (T)0) {
        attempt = true;
    } else if (delta == (T)0) {
        if (g.variant == 1) attempt = (votes & mine_mask) != 0ull;
        else if (g.variant == 2) attempt = true;
        if (attempt && n_best > 1 && has_cur) drop_cur = true;  // best_values.remove(current_value)
    }
    int out = mine;
    bool moved = false;
    if (attempt && prob > uniform_from_key(key, g.cycle + 1, 1)) {
        const int n = n_best - (drop_cur ? 1 : 0);
        int j = (int)(uniform_from_key(key, g.cycle + 1, 2) * n);
        int pick = first_best;
        bool done = false;
#pragma unroll
        for (int x = 0; x < MAXD; ++x)
            if (x < D && !done && c[x] == best_cost && !(drop_cur && x == mine)) {
                if (j-- == 0) {
                    pick = x;
                    done = true;
                }
            }
        out = pick;
        moved = true;
    }
    if (has && k == 0) {
        g.cur_out[qv] = out;
        if (moved) g.cost[qv] = best_cost;  // value_selection(choice, best_cost)
    }
}

struct Base {
    virtual ~Base() {}
    virtual int init(const mxs_graph& G, const mxs_params& p, int variant, double probability, int arity_mode,
                     uint64_t seed, int device) = 0;
    virtual int reset() = 0;
    virtual int set_value_rank(const int32_t* rank) = 0;
    virtual int run(int32_t n) = 0;
    virtual int get_state(int32_t* idx, double* cost) = 0;
    virtual int eval_cost(const int32_t* idx, double infinity, double* cost, int64_t* viol) = 0;
    int64_t cycles = 0;
};

template <typename T>
struct Engine : Base {
    int device = 0;
    hipStream_t stream = nullptr;
    Dev<T> g{};
    int which = 0;
    uint64_t seed = 0;
    std::vector<int32_t> h_dom, h_frow, h_evar, h_nn, h_q, h_vrank;
    std::vector<int64_t> h_toff, h_coff;
    std::vector<double> h_tables, h_eval_cost, h_var_cost;
    Buf<int32_t> dom_size, factor_rowptr, edge_var, edge_factor, var_rowptr, var_edges, n_neigh;
    Buf<int32_t> cur[2];
    Buf<int64_t> table_off;
    Buf<T> tables, f_opt, cost;
    Buf<double> prob;
    Buf<int64_t> sl_base;
    Buf<int32_t> sl_stride_v, sl_nb_rowptr, sl_nb_var, sl_nb_stride, sl_nb0_var, sl_nb0_stride, sl_conc_rowptr, sl_conc_var;
    Buf<uint8_t> sl_rows;           // the row view of the variables the pack cannot take (local_search.h, Slots::rows)
    Buf<int64_t> sl_row_base;
    Buf<int32_t> sl_row_nb_stride, sl_row_nb0_stride;
    bool have_rows = false;
    Buf<lsearch::PackWave> pk_waves;
    Buf<int32_t> pk_nb, pk_slot, pk_rest, pk_dom, qmap;
    Buf<uint64_t> pk_key;
    Buf<double> pk_prob;
    Buf<int8_t> pk_rec8;
    Buf<T> pk_recT, pk_fopt;
    bool pack_int8 = false;
    int n_rest = 0;
    int max_dom = 0;

    ~Engine() override {
        if (stream) (void)hipStreamDestroy(stream);
    }

    int init(const mxs_graph& G, const mxs_params& p, int variant, double probability, int arity_mode,
             uint64_t sd, int dev) override {
        device = dev;
        seed = sd;
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return fail(MXS_E_NODEVICE, "no HIP device visible: the engine has no CPU fallback");
        if (dev < 0 || dev >= count) return fail(MXS_E_INVALID, "device index out of range");
        if (variant < 0 || variant > 2) return fail(MXS_E_INVALID, "variant must be 0 (A), 1 (B) or 2 (C)");
        if (p.mode != MXS_MODE_MIN && p.mode != MXS_MODE_MAX) return fail(MXS_E_INVALID, "invalid mode");
        DSA_TRY(hipSetDevice(dev));
        DSA_TRY(hipStreamCreateWithFlags(&stream, 0));
        const int nV = G.n_vars, nF = G.n_factors, nE = G.n_edges;
        if (nV < 0 || nF < 0 || nE < 0) return fail(MXS_E_INVALID, "negative size");
        h_dom.assign(G.dom_size, G.dom_size + nV);
        h_frow.assign(G.factor_rowptr, G.factor_rowptr + nF + 1);
        h_evar.assign(G.edge_var, G.edge_var + nE);
        h_toff.assign(G.table_off, G.table_off + nF + 1);
        h_coff.assign(nV + 1, 0);
        for (int v = 0; v < nV; ++v) {
            if (h_dom[v] < 1) return fail(MXS_E_INVALID, "empty domain");
            h_coff[v + 1] = h_coff[v] + h_dom[v];
        }
        std::vector<int32_t> efac(nE), vrow(G.var_rowptr, G.var_rowptr + nV + 1), vedges(G.var_edges, G.var_edges + nE);
        for (int f = 0; f < nF; ++f) {
            if (h_frow[f + 1] <= h_frow[f]) return fail(MXS_E_INVALID, "factor without variable");
            for (int e = h_frow[f]; e < h_frow[f + 1]; ++e) {
                if (h_evar[e] < 0 || h_evar[e] >= nV) return fail(MXS_E_INVALID, "edge_var out of range");
                efac[e] = f;
            }
        }
        h_nn.assign(nV, 0);
        std::vector<int64_t> n_count(nV, 0);
        for (int f = 0; f < nF; ++f) {
            const int ar = h_frow[f + 1] - h_frow[f];
            for (int e = h_frow[f]; e < h_frow[f + 1]; ++e) {
                if (ar > 1) h_nn[h_evar[e]] = 1;
                n_count[h_evar[e]] += ar - 1;
            }
        }
        std::vector<double> pr(nV);
        for (int v = 0; v < nV; ++v)  // p_mode arity: 1 / sum(arity - 1) * 1.2 (dsa.py:256-259)
            pr[v] = (arity_mode && n_count[v] > 0) ? 1.0 / (double)n_count[v] * 1.2 : probability;
        h_tables.assign(G.tables, G.tables + h_toff[nF]);
        h_var_cost.assign(G.var_cost, G.var_cost + h_coff[nV]);
        const double* ev = G.eval_var_cost ? G.eval_var_cost : G.var_cost;
        h_eval_cost.assign(ev, ev + h_coff[nV]);
        std::vector<T> tt(h_tables.size()), fo(nF);
        for (size_t i = 0; i < tt.size(); ++i) tt[i] = (T)h_tables[i];
        for (int f = 0; f < nF; ++f) {  // find_optimum (relations.py:1367-1401): variant B
            T opt = tt[h_toff[f]];
            for (int64_t k = h_toff[f] + 1; k < h_toff[f + 1]; ++k)
                if (p.mode == MXS_MODE_MAX ? tt[k] > opt : tt[k] < opt) opt = tt[k];
            fo[f] = opt;
        }
        lsearch::HostSlots hs;
        const std::string bad = hs.build(nV, nF, h_dom, h_frow, h_evar, h_toff, vrow, vedges);
        if (!bad.empty()) return fail(MXS_E_INVALID, bad);
        max_dom = 0;
        for (int v = 0; v < nV; ++v) max_dom = h_dom[v] > max_dom ? h_dom[v] : max_dom;
        {   // the packed view of the variables it can take (local_search.h); the others -- constraints of
            // arity > 2, larger domains, degrees above 64 -- stay on the thread-per-variable kernel
            lsearch::HostPack hp;
            hp.build(nV, h_dom, vrow, h_nn, hs, h_tables);
            std::vector<T> fopt_lane(hp.slot.size(), (T)0);
            for (size_t i = 0; i < hp.slot.size(); ++i)
                if (hp.slot[i] >= 0) fopt_lane[i] = fo[efac[vedges[hp.slot[i]]]];
            pack_int8 = hp.int8_exact;
            if (pack_int8) {
                std::vector<int8_t> r8(hp.rec.size());
                for (size_t i = 0; i < r8.size(); ++i) r8[i] = (int8_t)hp.rec[i];
                DSA_TRY(pk_rec8.upload(r8, stream));
            } else {
                std::vector<T> rt(hp.rec.size());
                for (size_t i = 0; i < rt.size(); ++i) rt[i] = (T)hp.rec[i];
                DSA_TRY(pk_recT.upload(rt, stream));
            }
            DSA_TRY(pk_waves.upload(hp.waves, stream));
            // packed positions (Dev::q): the packed variables in wave order, then the others
            h_q.assign(nV, -1);
            int nq = 0;
            for (int v : hp.vars) h_q[v] = nq++;
            const int n_packed = nq;
            for (int v = 0; v < nV; ++v)
                if (h_q[v] < 0) h_q[v] = nq++;
            auto to_q = [&](std::vector<int32_t> a) {
                for (auto& x : a)
                    if (x >= 0) x = h_q[x];
                return a;
            };
            std::vector<int32_t> pdom(n_packed);
            std::vector<uint64_t> pkey(n_packed);
            std::vector<double> pprob(n_packed);
            for (int v : hp.vars) {
                pdom[h_q[v]] = h_dom[v];
                pkey[h_q[v]] = uniform_key(seed, v);
                pprob[h_q[v]] = pr[v];
            }
            DSA_TRY(qmap.upload(h_q, stream));
            DSA_TRY(pk_dom.upload(pdom, stream));
            DSA_TRY(pk_key.upload(pkey, stream));
            DSA_TRY(pk_prob.upload(pprob, stream));
            DSA_TRY(sl_nb_var.upload(to_q(hs.nb_var), stream));
            DSA_TRY(sl_nb0_var.upload(to_q(hs.nb0_var), stream));
            g.q = qmap.p;
            g.pack_dom = pk_dom.p;
            g.pack_key = pk_key.p;
            g.pack_prob = pk_prob.p;
            DSA_TRY(pk_nb.upload(to_q(hp.nb), stream));
            DSA_TRY(pk_slot.upload(hp.slot, stream));
            DSA_TRY(pk_rest.upload(hp.rest, stream));
            DSA_TRY(pk_fopt.upload(fopt_lane, stream));
            n_rest = (int)hp.rest.size();
            // the row view for them (domains of at most 32 values; $MAXSUM_LOCAL_SEARCH_ROWS=0 leaves it out, the
            // budget in bytes can be set: A/B runs and tests)
            {
                const int64_t budget = lsearch::HostSlots::rows_budget();
                have_rows = budget > 0 && max_dom <= 32 && hs.build_rows(hp.rest, h_dom, vrow, h_toff, h_tables, (int)sizeof(T), 32, budget);
                if (have_rows) {
                    // an upload that fails (device memory) leaves the strided path: free what was allocated and carry on
                    const bool ok = sl_rows.upload(hs.rows, stream) == hipSuccess && sl_row_base.upload(hs.row_base, stream) == hipSuccess &&
                                    sl_row_nb_stride.upload(hs.row_nb_stride, stream) == hipSuccess &&
                                    sl_row_nb0_stride.upload(hs.row_nb0_stride, stream) == hipSuccess;
                    if (!ok) {
                        (void)hipGetLastError();
                        sl_rows.release(), sl_row_base.release(), sl_row_nb_stride.release(), sl_row_nb0_stride.release();
                        have_rows = false;
                    }
                    hs.rows.clear();
                    hs.rows.shrink_to_fit();
                }
            }
            g.pack = lsearch::Pack{pk_waves.p, pk_nb.p, pk_slot.p,
                                   pack_int8 ? (const void*)pk_rec8.p : (const void*)pk_recT.p, (int32_t)hp.nb.size()};
            g.pack_fopt = pk_fopt.p;
        }
        DSA_TRY(sl_base.upload(hs.base, stream));
        DSA_TRY(sl_stride_v.upload(hs.stride_v, stream));
        DSA_TRY(sl_nb_rowptr.upload(hs.nb_rowptr, stream));
        DSA_TRY(sl_nb_stride.upload(hs.nb_stride, stream));
        DSA_TRY(sl_nb0_stride.upload(hs.nb0_stride, stream));
        DSA_TRY(sl_conc_rowptr.upload(hs.conc_rowptr, stream));
        DSA_TRY(sl_conc_var.upload(hs.conc_var, stream));
        g.slots = lsearch::Slots{sl_base.p, sl_stride_v.p, sl_nb_rowptr.p, sl_nb_var.p, sl_nb_stride.p,
                                 sl_nb0_var.p, sl_nb0_stride.p, sl_conc_rowptr.p, sl_conc_var.p,
                                 have_rows ? sl_rows.p : nullptr, sl_row_base.p, sl_row_nb_stride.p, sl_row_nb0_stride.p,
                                 hs.rows_int8 ? 1 : 0};
        DSA_TRY(dom_size.upload(h_dom, stream));
        DSA_TRY(factor_rowptr.upload(h_frow, stream));
        DSA_TRY(edge_var.upload(h_evar, stream));
        DSA_TRY(edge_factor.upload(efac, stream));
        DSA_TRY(var_rowptr.upload(vrow, stream));
        DSA_TRY(var_edges.upload(vedges, stream));
        DSA_TRY(n_neigh.upload(h_nn, stream));
        DSA_TRY(table_off.upload(h_toff, stream));
        DSA_TRY(tables.upload(tt, stream));
        DSA_TRY(f_opt.upload(fo, stream));
        DSA_TRY(prob.upload(pr, stream));
        for (int b = 0; b < 2; ++b) DSA_TRY(cur[b].alloc(nV));
        DSA_TRY(cost.alloc(nV));
        g.n_vars = nV;
        g.is_max = p.mode == MXS_MODE_MAX;
        g.variant = variant;
        g.seed = seed;
        g.dom_size = dom_size.p; g.factor_rowptr = factor_rowptr.p; g.edge_var = edge_var.p;
        g.edge_factor = edge_factor.p; g.var_rowptr = var_rowptr.p; g.var_edges = var_edges.p;
        g.n_neigh = n_neigh.p; g.table_off = table_off.p; g.tables = tables.p; g.f_opt = f_opt.p;
        g.prob = prob.p; g.cost = cost.p;
        return reset();
    }

    // the order of every variable's domain values (include/maxsum_gpu.h): cost ties of a variable without
    // neighbours break on the value, as the reference's optimal_cost_value does
    int set_value_rank(const int32_t* rank) override {
        if (rank) h_vrank.assign(rank, rank + h_coff[g.n_vars]);
        else h_vrank.clear();
        return reset();
    }

    int reset() override {
        DSA_TRY(hipSetDevice(device));
        const int nV = g.n_vars;
        std::vector<int32_t> c0(nV);
        std::vector<T> k0(nV, (T)0);
        for (int v = 0; v < nV; ++v) {
            if (h_nn[v] == 0) {  // optimal_cost_value (dsa.py:278-289)
                const int32_t* rk = h_vrank.empty() ? nullptr : h_vrank.data() + h_coff[v];
                int best = 0;
                for (int d = 1; d < h_dom[v]; ++d) {  // min / max over (cost, value) tuples, relations.py:1661-1665
                    const T a = (T)h_var_cost[h_coff[v] + d], b = (T)h_var_cost[h_coff[v] + best];
                    const int rd = rk ? rk[d] : d, rb = rk ? rk[best] : best;
                    if (g.is_max ? (a > b || (a == b && rd > rb)) : (a < b || (a == b && rd < rb))) best = d;
                }
                c0[h_q[v]] = best;  // (the state lives in packed order, Dev::q)
                k0[h_q[v]] = (T)h_var_cost[h_coff[v] + best];
            } else {  // random_value_selection (dsa.py:291): draw 0 of cycle 0
                c0[h_q[v]] = (int32_t)(uniform(seed, v, 0, 0) * h_dom[v]);
            }
        }
        which = 0;
        if (nV) {
            DSA_TRY(hipMemcpyAsync(cur[0].p, c0.data(), 4 * nV, hipMemcpyHostToDevice, stream));
            // (both buffers: the packed launch writes only the variables that have neighbours)
            DSA_TRY(hipMemcpyAsync(cur[1].p, c0.data(), 4 * nV, hipMemcpyHostToDevice, stream));
            DSA_TRY(hipMemcpyAsync(cost.p, k0.data(), sizeof(T) * nV, hipMemcpyHostToDevice, stream));
            DSA_TRY(hipStreamSynchronize(stream));
        }
        cycles = 0;
        return MXS_OK;
    }

    int run(int32_t n) override {
        DSA_TRY(hipSetDevice(device));
        const int nV = g.n_vars;
        if (nV == 0) {
            cycles += n > 0 ? n : 0;
            return MXS_OK;
        }
        const char* env = std::getenv("MAXSUM_LOCAL_SEARCH_GENERIC");  // =1: the CSR-walk kernel, =2: the slot
        const bool generic = env && env[0] == '1';                     // kernel for every variable (A/B, tests)
        const bool packed = !generic && !(env && env[0] == '2') && g.pack.n_lanes > 0;
        for (int32_t r = 0; r < n; ++r) {
            g.cur = cur[which].p;
            g.cur_out = cur[which ^ 1].p;
            g.cycle = cycles;
            g.var_list = nullptr;
            g.n_list = nV;
            if (packed) {
                const dim3 pgrid((unsigned)((g.pack.n_lanes + PACK_TPB - 1) / PACK_TPB)), pblock(PACK_TPB);
                if (pack_int8) hipLaunchKernelGGL((k_dsa_cycle_pack<T, int8_t>), pgrid, pblock, 0, stream, g);
                else hipLaunchKernelGGL((k_dsa_cycle_pack<T, T>), pgrid, pblock, 0, stream, g);
                DSA_TRY(hipGetLastError());
                g.var_list = pk_rest.p;
                g.n_list = n_rest;
            }
            if (g.n_list > 0) {
                const dim3 grid((unsigned)((g.n_list + TPB - 1) / TPB)), block(TPB);
                if (generic || max_dom > 32) hipLaunchKernelGGL((k_dsa_cycle<T>), grid, block, 0, stream, g);
                else if (max_dom <= 4) hipLaunchKernelGGL((k_dsa_cycle_slots<T, 4>), grid, block, 0, stream, g);
                else if (max_dom <= 8) hipLaunchKernelGGL((k_dsa_cycle_slots<T, 8>), grid, block, 0, stream, g);
                else if (max_dom <= 16) hipLaunchKernelGGL((k_dsa_cycle_slots<T, 16>), grid, block, 0, stream, g);
                else hipLaunchKernelGGL((k_dsa_cycle_slots<T, 32>), grid, block, 0, stream, g);
                DSA_TRY(hipGetLastError());
            }
            which ^= 1;
            cycles += 1;
        }
        DSA_TRY(hipStreamSynchronize(stream));
        return MXS_OK;
    }

    int get_state(int32_t* idx, double* cst) override {
        DSA_TRY(hipSetDevice(device));
        const int nV = g.n_vars;
        if (!nV) return MXS_OK;
        std::vector<T> hc(nV);
        std::vector<int32_t> hi(nV);
        DSA_TRY(hipMemcpyAsync(hi.data(), cur[which].p, 4 * nV, hipMemcpyDeviceToHost, stream));
        DSA_TRY(hipMemcpyAsync(hc.data(), cost.p, sizeof(T) * nV, hipMemcpyDeviceToHost, stream));
        DSA_TRY(hipStreamSynchronize(stream));
        for (int v = 0; v < nV; ++v) {  // the state lives in packed order (Dev::q)
            if (idx) idx[v] = hi[h_q[v]];
            if (cst) cst[v] = (double)hc[h_q[v]];
        }
        return MXS_OK;
    }

    int eval_cost(const int32_t* idx, double infinity, double* cst, int64_t* viol) override {
        std::vector<int32_t> c;
        if (!idx) {
            c.resize(g.n_vars);
            int rc = get_state(c.data(), nullptr);
            if (rc) return rc;
            idx = c.data();
        }
        double soft = 0;
        int64_t hard = 0;
        const int nF = (int)h_frow.size() - 1;
        for (int f = 0; f < nF; ++f) {
            int64_t lin = 0;
            for (int e = h_frow[f]; e < h_frow[f + 1]; ++e) {
                const int v = h_evar[e];
                if (idx[v] < 0 || idx[v] >= h_dom[v]) return fail(MXS_E_INVALID, "assignment index out of the domain");
                lin = lin * h_dom[v] + idx[v];
            }
            const double r = h_tables[h_toff[f] + lin];
            if (r != infinity) soft += r; else hard += 1;
        }
        for (int v = 0; v < g.n_vars; ++v) {
            const double x = h_eval_cost[h_coff[v] + idx[v]];
            if (x != infinity) soft += x; else hard += 1;
        }
        if (cst) *cst = soft;
        if (viol) *viol = hard;
        return MXS_OK;
    }
};

}  // namespace dsa

struct mxs_dsa {
    dsa::Base* impl;
};

extern "C" {

int mxs_dsa_create(const mxs_graph* g, const mxs_params* p, int32_t variant, double probability, int32_t arity_mode,
                   uint64_t seed, int32_t device, mxs_dsa** out) {
    if (!g || !p || !out) return dsa::fail(MXS_E_INVALID, "null argument");
    *out = nullptr;
    try {
        dsa::Base* impl = p->dtype == MXS_DTYPE_F32 ? (dsa::Base*)new dsa::Engine<float>() : (dsa::Base*)new dsa::Engine<double>();
        int rc = impl->init(*g, *p, variant, probability, arity_mode, seed, device);
        if (rc) {
            delete impl;
            return rc;
        }
        *out = new mxs_dsa{impl};
        return MXS_OK;
    } catch (const std::exception& ex) {
        return dsa::fail(MXS_E_NOMEM, ex.what());
    }
}
int mxs_dsa_reset(mxs_dsa* e) { return e ? e->impl->reset() : dsa::fail(MXS_E_INVALID, "null handle"); }
int mxs_dsa_set_value_rank(mxs_dsa* e, const int32_t* value_rank) {
    return e ? e->impl->set_value_rank(value_rank) : dsa::fail(MXS_E_INVALID, "null handle");
}
int mxs_dsa_run(mxs_dsa* e, int32_t n_cycles) {
    if (!e) return dsa::fail(MXS_E_INVALID, "null handle");
    if (n_cycles < 0) return dsa::fail(MXS_E_INVALID, "negative cycle count");
    return e->impl->run(n_cycles);
}
int mxs_dsa_cycles(const mxs_dsa* e, int64_t* cycles) {
    if (!e) return dsa::fail(MXS_E_INVALID, "null handle");
    if (cycles) *cycles = e->impl->cycles;
    return MXS_OK;
}
int mxs_dsa_get_state(mxs_dsa* e, int32_t* idx, double* cost) {
    return e ? e->impl->get_state(idx, cost) : dsa::fail(MXS_E_INVALID, "null handle");
}
int mxs_dsa_eval_cost(mxs_dsa* e, const int32_t* idx, double infinity, double* cost, int64_t* violations) {
    return e ? e->impl->eval_cost(idx, infinity, cost, violations) : dsa::fail(MXS_E_INVALID, "null handle");
}
int mxs_dsa_destroy(mxs_dsa* e) {
    if (e) {
        delete e->impl;
        delete e;
    }
    return MXS_OK;
}

}  // extern "C"
