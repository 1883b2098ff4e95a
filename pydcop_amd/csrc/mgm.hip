// mgm.hip -- the reference's MGM (pydcop/algorithms/mgm.py: Maheswaran, Pearce, Tambe 2004) on
// gfx950, on the same flat factor-graph arrays as the Max-Sum engine (SURVEY.md section 8(f).4:
// "per-variable segmented argmin over neighbour values -- same layout, different semiring").
//
// MGM is bulk-synchronous by construction (a computation handles a round's values only when ALL
// its neighbours' values are in, then the gains; early messages are parked, mgm.py:311-333,
// 476-497): one round = two launches over all variables,
//   k_mgm_gain   values in  -> the best unilateral move and its gain   (mgm.py:335-391, 428-454)
//   k_mgm_move   gains in   -> the largest gain of a neighbourhood moves, ties by name (:499-588)
// Three families of the two kernels, bit-identical results: the PACKED view (local_search.h: one lane
// per (variable, constraint) for unary / binary constraints over domains of at most four values -- what
// runs by default where the instance allows it), the slot view (thread per variable, register arrays
// for domains up to 32 values) for the other variables, and the CSR walk (anything).  The dynamic
// state lives in packed order (Dev::q); a variable's gain, new value and name rank are one 16-byte
// record (GainRec), its own cost at the current value is kept (Dev::vcc).  The reference's quirks
// are restated as they are and listed in oracle/mgm_oracle.c, whose arithmetic this file follows
// expression for expression (the oracle is pinned against the reference's own MgmComputation).
// The reference's draws from the unseeded `random` module are fixed the way the oracle fixes them:
// first domain value at start, first of equally good values.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/maxsum_gpu.h"
#include "local_search.h"

extern "C" __attribute__((visibility("hidden"))) void mxs_set_last_error(const char* msg);  // engine.hip

namespace mgm {

constexpr int TPB = 64;  // one wave per block: 100k variables spread over every CU (latency-bound CSR walks)

static int fail(int code, const std::string& msg) {
    mxs_set_last_error(msg.c_str());
    return code;
}
#define MGM_TRY(call)                                                                     \
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

template <typename T>
struct alignas(16) GainRec {
    T gain;
    int32_t newv, rank;  // rank: written once at init
};

template <typename T>
struct Dev {
    int32_t n_vars, is_max;
    const int32_t *dom_size, *factor_rowptr, *edge_var, *edge_factor, *var_rowptr, *var_edges, *init_idx, *name_rank, *n_neigh;
    const int64_t *table_off, *cost_off;
    const T *tables, *var_cost;
    const int32_t* cur;    // values of the round being handled
    const T* cost;
    int32_t* cur_out;      // k_mgm_move: after the round
    T* cost_out;
    uint8_t* has_cost;
    GainRec<T>* grec;      // [n_vars] gain, new value and name rank of a variable in ONE 16-byte record: what a
                           // neighbour's move reads of it is one random request instead of three
    const T* vcc;          // [n_vars] var_cost at the variable's value in `cur` (kept by the move kernels):
    T* vcc_out;            // the concerned-variable sums read one word instead of cost_off -> cur -> var_cost
    const T* vc4;          // [n_vars][PACK_D] var_cost of the packed variables, addressed by the variable alone
    lsearch::Slots slots;
    lsearch::Pack pack;          // the packed view (local_search.h): lane per (variable, constraint)
    const int32_t* pack_conc;    // [lanes] element k of the variable's concerned-variables list, -1 = none
    const int32_t* pack_conc_x;  // [lanes] lane k = 0: element `deg` of that list (it has at most deg + 1), else -1
    const int32_t* var_list;     // the variables a thread-per-variable launch works on (NULL: all)
    int32_t n_list;
    // The DYNAMIC per-variable state -- cur, cost, has_cost, grec, vcc -- is stored in PACKED ORDER:
    // position q[v] = the variable's rank in the packed view's wave order (the other variables after them).
    // A packed wave's variables are then q = first .. first + nv - 1: what lane k = 0 of each of them
    // reads and writes of its own state is one line per array instead of one line per variable
    // (scattered 8-byte stores of ~8 lanes per wave cost 0.9 us per store instruction at 100k variables,
    // profiles/r03_local_search_kernels_v2.txt).  The packed view holds q directly (nb, conc, vars); the
    // thread-per-variable kernels translate graph indices through q[].  Everything the semantics depends
    // on -- sum orders, name ranks, the concerned lists' ascending order -- stays on graph indices.
    const int32_t* q;
    const int32_t* pack_dom;     // [packed variables] dom_size in packed order
};

// c.slice(neighbours' values)(x): the table entry with v at x, every other scope variable at its value
template <typename T>
__device__ T constraint_at(const Dev<T>& g, int f, int v, int x) {
    int64_t lin = 0;
    for (int e = g.factor_rowptr[f]; e < g.factor_rowptr[f + 1]; ++e) {
        const int u = g.edge_var[e];
        lin = lin * g.dom_size[u] + (u == v ? x : g.cur[g.q[u]]);
    }
    return g.tables[g.table_off[f] + lin];
}

// functools.reduce(operator.add, [f(x) for f in reduced_cs]): utilities order, no initial 0
template <typename T>
__device__ T utilities_at(const Dev<T>& g, int v, int x) {
    T acc = (T)0;
    bool first = true;
    for (int k = g.var_rowptr[v]; k < g.var_rowptr[v + 1]; ++k) {
        const T f = constraint_at(g, g.edge_factor[g.var_edges[k]], v, x);
        acc = first ? f : acc + f;
        first = false;
    }
    return acc;
}

// acc += cost_for_val of every distinct variable of v's constraints (v included) at its current
// value, in ascending variable index (the reference iterates a set, see oracle/mgm_oracle.c)
template <typename T>
__device__ T add_concerned_costs(const Dev<T>& g, int v, T acc) {
    int last = -1;
    for (;;) {
        int best = INT32_MAX;
        for (int k = g.var_rowptr[v]; k < g.var_rowptr[v + 1]; ++k) {
            const int f = g.edge_factor[g.var_edges[k]];
            for (int e = g.factor_rowptr[f]; e < g.factor_rowptr[f + 1]; ++e) {
                const int u = g.edge_var[e];
                if (u > last && u < best) best = u;
            }
        }
        if (best == INT32_MAX) break;
        acc += g.vcc[g.q[best]];  // = var_cost[cost_off[best] + cur[best]], kept by the move kernels
        last = best;
    }
    return acc;
}

template <typename T>
__global__ void __launch_bounds__(TPB) k_mgm_gain(Dev<T> g, T* cost_rw) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.n_list) return;
    const int v = g.var_list ? g.var_list[tid] : tid;
    if (g.n_neigh[v] == 0) return;
    const int qv = g.q[v];
    T cost = cost_rw[qv];
    if (!g.has_cost[qv]) {  // first round: the cost of the current value (mgm.py:349-372)
        cost = add_concerned_costs(g, v, utilities_at(g, v, g.cur[qv]));
        cost_rw[qv] = cost;
        g.has_cost[qv] = 1;
    }
    T best = (T)0;
    int best_x = -1;
    for (int x = 0; x < g.dom_size[v]; ++x) {  // find_arg_optimal: strictly better starts a new list
        const T r = utilities_at(g, v, x);
        if (best_x < 0 || (g.is_max ? best < r : best > r)) {
            best = r;
            best_x = x;
        }
    }
    const T val_cost = add_concerned_costs(g, v, best);  // own cost at the CURRENT value (:449-450)
    const T gain = cost - val_cost;
    const int nvl = ((!g.is_max && gain > (T)0) || (g.is_max && gain < (T)0)) ? best_x : g.cur[qv];
    g.grec[qv].gain = gain;
    g.grec[qv].newv = nvl;
}

template <typename T>
__global__ void __launch_bounds__(TPB) k_mgm_move(Dev<T> g) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.n_list) return;
    const int v = g.var_list ? g.var_list[tid] : tid;
    const int qv = g.q[v];
    int cur = g.cur[qv];
    T cost = g.cost[qv];
    if (g.n_neigh[v] != 0) {
        T max_n = (T)0;
        bool first = true;
        for (int k = g.var_rowptr[v]; k < g.var_rowptr[v + 1]; ++k) {
            const int f = g.edge_factor[g.var_edges[k]];
            for (int e = g.factor_rowptr[f]; e < g.factor_rowptr[f + 1]; ++e) {
                const int u = g.edge_var[e];
                if (u == v) continue;
                const T gu = g.grec[g.q[u]].gain;
                if (first || gu > max_n) max_n = gu;  // max() also in max mode (:513)
                first = false;
            }
        }
        bool wins_tie = true;
        for (int k = g.var_rowptr[v]; k < g.var_rowptr[v + 1]; ++k) {
            const int f = g.edge_factor[g.var_edges[k]];
            for (int e = g.factor_rowptr[f]; e < g.factor_rowptr[f + 1]; ++e) {
                const int u = g.edge_var[e];
                if (u != v && g.grec[g.q[u]].gain == max_n && g.name_rank[u] < g.name_rank[v]) wins_tie = false;
            }
        }
        const T gain = g.grec[qv].gain;
        if (gain > max_n || (gain == max_n && wins_tie)) {  // :514-525, lexic ties :566-588
            cur = g.grec[qv].newv;
            cost = cost - gain;
        }
    }
    g.cur_out[qv] = cur;
    g.cost_out[qv] = cost;
    g.vcc_out[qv] = g.var_cost[g.cost_off[v] + cur];
}

// ---- the same two kernels on the slot view (local_search.h) ---------------------------------
// the costs of the D values in registers from one pass over the variable's constraints; the
// distinct variables of those constraints from a list sorted on the host instead of the
// repeated minimum search of add_concerned_costs; domains of at most MAXD values
// (the variable references of the slot view -- nb0_var, nb_var, conc_var -- are uploaded as packed
// positions q: they index the dynamic state directly)
template <typename T>
__device__ T add_concerned_costs_listed(const Dev<T>& g, int v, T acc) {
    for (int k = g.slots.conc_rowptr[v]; k < g.slots.conc_rowptr[v + 1]; ++k)
        acc += g.vcc[g.slots.conc_var[k]];  // = var_cost[cost_off[u] + cur[u]], kept by the move kernels
    return acc;
}

template <typename T, int MAXD>
__global__ void __launch_bounds__(TPB) k_mgm_gain_slots(Dev<T> g, T* cost_rw) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.n_list) return;
    const int v = g.var_list ? g.var_list[tid] : tid;
    if (g.n_neigh[v] == 0) return;
    const int D = g.dom_size[v];
    T c[MAXD];
    {
        const int s0 = g.var_rowptr[v], s1 = g.var_rowptr[v + 1];
        if (g.slots.rows != nullptr && s0 < s1 && g.slots.row_base[s0] >= 0) {  // contiguous rows (local_search.h)
            if (g.slots.rows_int8) lsearch::costs_of_values_rows<T, int8_t, MAXD>(g.slots, g.cur, s0, s1, D, false, c);
            else lsearch::costs_of_values_rows<T, T, MAXD>(g.slots, g.cur, s0, s1, D, false, c);
        } else {
            lsearch::costs_of_values<T, MAXD>(g.slots, g.tables, g.cur, s0, s1, D, false, c);
        }
    }
    const int qv = g.q[v];
    T cost = cost_rw[qv];
    if (!g.has_cost[qv]) {
        cost = add_concerned_costs_listed(g, v, lsearch::pick<T, MAXD>(c, g.cur[qv]));
        cost_rw[qv] = cost;
        g.has_cost[qv] = 1;
    }
    T best = c[0];
    int best_x = 0;
#pragma unroll
    for (int x = 1; x < MAXD; ++x)
        if (x < D && (g.is_max ? best < c[x] : best > c[x])) {
            best = c[x];
            best_x = x;
        }
    const T val_cost = add_concerned_costs_listed(g, v, best);
    const T gain = cost - val_cost;
    const int nvl = ((!g.is_max && gain > (T)0) || (g.is_max && gain < (T)0)) ? best_x : g.cur[qv];
    g.grec[qv].gain = gain;
    g.grec[qv].newv = nvl;
}

template <typename T>
__global__ void __launch_bounds__(TPB) k_mgm_move_listed(Dev<T> g) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= g.n_list) return;
    const int v = g.var_list ? g.var_list[tid] : tid;
    const int qv = g.q[v];
    int cur = g.cur[qv];
    T cost = g.cost[qv];
    if (g.n_neigh[v] != 0) {
        const int k0 = g.slots.conc_rowptr[v], k1 = g.slots.conc_rowptr[v + 1];
        const GainRec<T> me = g.grec[qv];
        T max_n = (T)0;
        bool first = true, wins_tie = true;
        for (int k = k0; k < k1; ++k) {  // one pass: the largest gain and whether a lower name holds it
            const int uq = g.slots.conc_var[k];
            if (uq == qv) continue;
            const GainRec<T> r = g.grec[uq];
            const bool lower = r.rank < me.rank;
            if (first || r.gain > max_n) {
                max_n = r.gain;
                wins_tie = !lower;
            } else if (r.gain == max_n && lower) {
                wins_tie = false;
            }
            first = false;
        }
        if (me.gain > max_n || (me.gain == max_n && wins_tie)) {
            cur = me.newv;
            cost = cost - me.gain;
        }
    }
    g.cur_out[qv] = cur;
    g.cost_out[qv] = cost;
    g.vcc_out[qv] = g.var_cost[g.cost_off[v] + cur];
}

// ---- the same two kernels on the PACKED view (local_search.h): one lane per (variable, constraint) ----
// What a lane holds of its variable's concerned-variables list (ascending, the variable included; at most
// deg + 1 entries): element k, and on lane k = 0 also element deg.
constexpr int PACK_TPB = 256;
struct PackLane {
    int deg, nv, var, k, seg, q;  // q: the variable's packed position = its index into the dynamic state
    bool has;
};
template <typename T>
__device__ inline PackLane pack_lane(const Dev<T>& g, int64_t pos) {
    const lsearch::PackWave wm = g.pack.waves[__builtin_amdgcn_readfirstlane((int)(pos >> 6))];
    const uint32_t dn = (uint32_t)wm.deg_nv;
    PackLane p;
    p.deg = (int)(dn & 255u);
    p.nv = (int)((dn >> 8) & 255u);
    const int l = (int)threadIdx.x & 63;
    p.var = (int)(((uint32_t)l * (dn >> 16)) >> 15);
    p.k = l - p.var * p.deg;
    p.has = p.var < p.nv;
    p.seg = l - p.k;
    p.q = wm.first + (p.has ? p.var : 0);
    return p;
}

// acc + cost_for_val of every concerned variable at its current value, in list order: each lane
// fetches ITS element (the fetches of a variable's lanes are in flight together), the additions
// then run in order through cross-lane reads -- every lane of the variable ends with the same sum
template <typename T>
__device__ inline T pack_add_concerned(const Dev<T>& g, const PackLane& p, int64_t pos, T acc) {
    const int u = g.pack_conc[pos], ux = g.pack_conc_x[pos];
    T w = (T)0, wx = (T)0;
    if (u >= 0) w = g.vcc[u];
    if (ux >= 0) wx = g.vcc[ux];
    // which lanes hold an element: one ballot each instead of a lane exchange per step
    const unsigned long long live = __ballot(u >= 0 ? 1 : 0), live_x = __ballot(ux >= 0 ? 1 : 0);
    for (int i = 0; i < p.deg; ++i) {
        const T e = __shfl(w, p.seg + i, 64);
        acc = ((live >> (p.seg + i)) & 1ull) ? acc + e : acc;
    }
    const T ex = __shfl(wx, p.seg, 64);
    acc = ((live_x >> p.seg) & 1ull) ? acc + ex : acc;
    return acc;
}

template <typename T, typename TT>
__global__ void __launch_bounds__(PACK_TPB) k_mgm_gain_pack(Dev<T> g, T* cost_rw) {
    constexpr int MAXD = lsearch::PACK_D;
    const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= g.pack.n_lanes) return;  // whole waves
    const PackLane p = pack_lane(g, pos);
    const int v = p.q, D = g.pack_dom[v], mine = g.cur[v];  // (v: packed position; nb / conc hold positions too)
    T t[MAXD], c[MAXD];
    lsearch::pack_costs<T, TT>(g.pack, g.cur, pos, p.deg, p.seg, false, t, c);
    T cost = cost_rw[v];
    const bool first_round = !g.has_cost[v];
    // (wave-uniform in practice: every variable gets its cost in the first round)
    if (__ballot(first_round ? 1 : 0) != 0ull) {
        const T c0 = pack_add_concerned(g, p, pos, lsearch::pick<T, MAXD>(c, mine));
        if (first_round) cost = c0;
    }
    T best = c[0];
    int best_x = 0;
#pragma unroll
    for (int x = 1; x < MAXD; ++x) {
        const bool lt = best < c[x], gt = best > c[x];
        const bool better = (x < D) & (g.is_max ? lt : gt);
        best = better ? c[x] : best;
        best_x = better ? x : best_x;
    }
    const T val_cost = pack_add_concerned(g, p, pos, best);  // own cost at the CURRENT value (mgm.py:449-450)
    const T gain = cost - val_cost;
    if (p.has && p.k == 0) {
        if (first_round) {
            cost_rw[v] = cost;
            g.has_cost[v] = 1;
        }
        const int nvl = ((!g.is_max && gain > (T)0) || (g.is_max && gain < (T)0)) ? best_x : mine;
        g.grec[v].gain = gain;
        g.grec[v].newv = nvl;
    }
}

template <typename T>
__global__ void __launch_bounds__(PACK_TPB) k_mgm_move_pack(Dev<T> g) {
    const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= g.pack.n_lanes) return;
    const PackLane p = pack_lane(g, pos);
    const int v = p.q;  // packed position (conc / conc_x hold positions too)
    // The largest gain among the OTHER concerned variables and whether a lower name holds it (max() also in
    // max mode, mgm.py:513; lexic ties :566-588).  Order-independent -- the reference's scan keeps (largest so
    // far, "no lower name holds it") -- so the lanes of a variable reduce their elements pairwise in
    // ceil(log2(deg)) steps instead of every lane scanning all of them; lane k = 0 ends with the result.
    const int u = g.pack_conc[pos], ux = g.pack_conc_x[pos];
    const GainRec<T> me = g.grec[v];
    const int cur0 = g.cur[v];
    const T cost0 = g.cost[v];
    const T vc0 = g.vcc[v];
    // the own costs of the variable's values in packed order (through cost_off[] + the new value they
    // would be one dependent load later); the new value's is picked in registers
    T vc[lsearch::PACK_D];
#pragma unroll
    for (int x = 0; x < lsearch::PACK_D; ++x) vc[x] = g.vc4[(int64_t)v * lsearch::PACK_D + x];
    T e = (T)0, ex = (T)0;
    unsigned fl = 0u, fx = 0u;  // bit 0: an element (a concerned variable other than v), bit 1: a lower name holds it
    if (u >= 0 && u != v) {
        const GainRec<T> r = g.grec[u];
        e = r.gain;
        fl = 1u | (r.rank < me.rank ? 2u : 0u);
    }
    if (ux >= 0 && ux != v) {
        const GainRec<T> r = g.grec[ux];
        ex = r.gain;
        fx = 1u | (r.rank < me.rank ? 2u : 0u);
    }
    auto merge = [&](T e2, unsigned f2) {  // (e, fl) <- the larger of the two; equal gains: either's lower name counts
        const bool la = (fl & 1u) != 0u, lb = (f2 & 1u) != 0u;
        const bool b_wins = lb & (!la | (e2 > e));
        const bool tie = la & lb & (e2 == e);
        fl = b_wins ? f2 : (tie ? fl | (f2 & 2u) : fl);
        e = b_wins ? e2 : e;
    };
    const int l = (int)threadIdx.x & 63;
    for (int s = 1; s < p.deg; s <<= 1) {
        const T e2 = __shfl(e, (l + s) & 63, 64);
        const unsigned f2 = (unsigned)__shfl((int)fl, (l + s) & 63, 64);
        merge(e2, p.k + s < p.deg ? f2 : 0u);
    }
    merge(ex, fx);  // element deg of the list: lane k = 0 holds it
    if (p.has && p.k == 0) {
        const bool any = (fl & 1u) != 0u;
        const T max_n = any ? e : (T)0;
        const bool wins_tie = !any || (fl & 2u) == 0u;
        const bool moves = me.gain > max_n || (me.gain == max_n && wins_tie);  // :514-525
        g.cur_out[v] = moves ? me.newv : cur0;
        g.cost_out[v] = moves ? cost0 - me.gain : cost0;
        g.vcc_out[v] = moves ? lsearch::pick<T, lsearch::PACK_D>(vc, me.newv) : vc0;
    }
}

struct Base {
    virtual ~Base() {}
    virtual int init(const mxs_graph& G, const mxs_params& p, const int32_t* rank, int device) = 0;
    virtual int reset() = 0;
    virtual int set_value_rank(const int32_t* rank) = 0;
    virtual int run(int32_t n) = 0;
    virtual int get_state(int32_t* idx, double* cost, uint8_t* has, double* gain, int32_t* newv) = 0;
    virtual int eval_cost(const int32_t* idx, double infinity, double* cost, int64_t* viol) = 0;
    int64_t rounds = 0;
};

template <typename T>
struct Engine : Base {
    int device = 0;
    hipStream_t stream = nullptr;
    Dev<T> g{};
    int which = 0;
    std::vector<int32_t> h_dom, h_frow, h_evar, h_init, h_nn, h_rank, h_q, h_vrank;
    std::vector<int64_t> h_toff, h_coff;
    std::vector<double> h_tables, h_eval_cost, h_var_cost;
    bool has_init = false;
    Buf<int32_t> dom_size, factor_rowptr, edge_var, edge_factor, var_rowptr, var_edges, init_idx, name_rank, n_neigh;
    Buf<int32_t> cur[2];
    Buf<int64_t> table_off, cost_off;
    Buf<T> tables, var_cost;
    Buf<T> cost[2], vcc[2], vc4;
    Buf<GainRec<T>> grec;
    Buf<uint8_t> has_cost;
    Buf<int64_t> sl_base;
    Buf<int32_t> sl_stride_v, sl_nb_rowptr, sl_nb_var, sl_nb_stride, sl_nb0_var, sl_nb0_stride, sl_conc_rowptr, sl_conc_var;
    Buf<uint8_t> sl_rows;           // the row view of the variables the pack cannot take (local_search.h, Slots::rows)
    Buf<int64_t> sl_row_base;
    Buf<int32_t> sl_row_nb_stride, sl_row_nb0_stride;
    bool have_rows = false;
    Buf<lsearch::PackWave> pk_waves;
    Buf<int32_t> pk_nb, pk_slot, pk_rest, pk_conc, pk_conc_x, pk_dom, qmap;
    Buf<int8_t> pk_rec8;
    Buf<T> pk_recT;
    bool pack_int8 = false;
    int n_rest = 0;
    int max_dom = 0;

    ~Engine() override {
        if (stream) (void)hipStreamDestroy(stream);
    }

    int init(const mxs_graph& G, const mxs_params& p, const int32_t* rank, int dev) override {
        device = dev;
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return fail(MXS_E_NODEVICE, "no HIP device visible: the engine has no CPU fallback");
        if (dev < 0 || dev >= count) return fail(MXS_E_INVALID, "device index out of range");
        MGM_TRY(hipSetDevice(dev));
        MGM_TRY(hipStreamCreateWithFlags(&stream, 0));
        const int nV = G.n_vars, nF = G.n_factors, nE = G.n_edges;
        if (nV < 0 || nF < 0 || nE < 0) return fail(MXS_E_INVALID, "negative size");
        if (p.mode != MXS_MODE_MIN && p.mode != MXS_MODE_MAX) return fail(MXS_E_INVALID, "invalid mode");
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
        for (int f = 0; f < nF; ++f)
            if (h_frow[f + 1] - h_frow[f] > 1)
                for (int e = h_frow[f]; e < h_frow[f + 1]; ++e) h_nn[h_evar[e]] = 1;
        std::vector<int32_t> rk(nV);
        for (int v = 0; v < nV; ++v) rk[v] = rank ? rank[v] : v;
        has_init = G.init_idx != nullptr;
        h_init.assign(nV, -1);
        if (has_init)
            for (int v = 0; v < nV; ++v) {
                if (G.init_idx[v] >= h_dom[v]) return fail(MXS_E_INVALID, "init_idx out of the domain");
                h_init[v] = G.init_idx[v];
            }
        h_tables.assign(G.tables, G.tables + h_toff[nF]);
        h_var_cost.assign(G.var_cost, G.var_cost + h_coff[nV]);
        const double* ev = G.eval_var_cost ? G.eval_var_cost : G.var_cost;
        h_eval_cost.assign(ev, ev + h_coff[nV]);
        std::vector<T> tt(h_tables.size()), vc(h_var_cost.size());
        for (size_t i = 0; i < tt.size(); ++i) tt[i] = (T)h_tables[i];
        for (size_t i = 0; i < vc.size(); ++i) vc[i] = (T)h_var_cost[i];
        lsearch::HostSlots hs;
        const std::string bad = hs.build(nV, nF, h_dom, h_frow, h_evar, h_toff, vrow, vedges);
        if (!bad.empty()) return fail(MXS_E_INVALID, bad);
        max_dom = 0;
        for (int v = 0; v < nV; ++v) max_dom = h_dom[v] > max_dom ? h_dom[v] : max_dom;
        MGM_TRY(sl_base.upload(hs.base, stream));
        MGM_TRY(sl_stride_v.upload(hs.stride_v, stream));
        MGM_TRY(sl_nb_rowptr.upload(hs.nb_rowptr, stream));
        MGM_TRY(sl_nb_stride.upload(hs.nb_stride, stream));
        MGM_TRY(sl_nb0_stride.upload(hs.nb0_stride, stream));
        {   // the packed view of the variables it can take (local_search.h)
            lsearch::HostPack hp;
            hp.build(nV, h_dom, vrow, h_nn, hs, h_tables);
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
            MGM_TRY(qmap.upload(h_q, stream));
            MGM_TRY(sl_nb_var.upload(to_q(hs.nb_var), stream));
            MGM_TRY(sl_nb0_var.upload(to_q(hs.nb0_var), stream));
            MGM_TRY(sl_conc_var.upload(to_q(hs.conc_var), stream));
            std::vector<int32_t> pdom(n_packed);
            std::vector<T> v4((size_t)n_packed * lsearch::PACK_D, (T)0);
            for (int v : hp.vars) {
                pdom[h_q[v]] = h_dom[v];
                for (int x = 0; x < h_dom[v]; ++x) v4[(size_t)h_q[v] * lsearch::PACK_D + x] = vc[h_coff[v] + x];
            }
            MGM_TRY(pk_dom.upload(pdom, stream));
            MGM_TRY(vc4.upload(v4, stream));
            g.q = qmap.p;
            g.pack_dom = pk_dom.p;
            g.vc4 = vc4.p;
            std::vector<int32_t> conc(hp.nb.size(), -1), conc_x(hp.nb.size(), -1);
            for (size_t i = 0; i < hp.nb.size(); ++i) {
                const int v = hp.lane_var[i];
                if (v < 0) continue;
                const int c0 = hs.conc_rowptr[v], n_conc = hs.conc_rowptr[v + 1] - c0, k = hp.lane_k[i], deg = hp.lane_deg[i];
                if (n_conc > deg + 1) return fail(MXS_E_STATE, "concerned-variables list longer than the degree + 1");
                if (k < n_conc) conc[i] = hs.conc_var[c0 + k];
                if (k == 0 && n_conc > deg) conc_x[i] = hs.conc_var[c0 + deg];
            }
            pack_int8 = hp.int8_exact;
            if (pack_int8) {
                std::vector<int8_t> r8(hp.rec.size());
                for (size_t i = 0; i < r8.size(); ++i) r8[i] = (int8_t)hp.rec[i];
                MGM_TRY(pk_rec8.upload(r8, stream));
            } else {
                std::vector<T> rt(hp.rec.size());
                for (size_t i = 0; i < rt.size(); ++i) rt[i] = (T)hp.rec[i];
                MGM_TRY(pk_recT.upload(rt, stream));
            }
            MGM_TRY(pk_waves.upload(hp.waves, stream));
            MGM_TRY(pk_nb.upload(to_q(hp.nb), stream));
            MGM_TRY(pk_slot.upload(hp.slot, stream));
            MGM_TRY(pk_rest.upload(hp.rest, stream));
            MGM_TRY(pk_conc.upload(to_q(conc), stream));
            MGM_TRY(pk_conc_x.upload(to_q(conc_x), stream));
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
            g.pack_conc = pk_conc.p;
            g.pack_conc_x = pk_conc_x.p;
        }
        MGM_TRY(sl_conc_rowptr.upload(hs.conc_rowptr, stream));
        g.slots = lsearch::Slots{sl_base.p, sl_stride_v.p, sl_nb_rowptr.p, sl_nb_var.p, sl_nb_stride.p,
                                 sl_nb0_var.p, sl_nb0_stride.p, sl_conc_rowptr.p, sl_conc_var.p,
                                 have_rows ? sl_rows.p : nullptr, sl_row_base.p, sl_row_nb_stride.p, sl_row_nb0_stride.p,
                                 hs.rows_int8 ? 1 : 0};
        MGM_TRY(dom_size.upload(h_dom, stream));
        MGM_TRY(factor_rowptr.upload(h_frow, stream));
        MGM_TRY(edge_var.upload(h_evar, stream));
        MGM_TRY(edge_factor.upload(efac, stream));
        MGM_TRY(var_rowptr.upload(vrow, stream));
        MGM_TRY(var_edges.upload(vedges, stream));
        MGM_TRY(name_rank.upload(rk, stream));
        MGM_TRY(n_neigh.upload(h_nn, stream));
        MGM_TRY(table_off.upload(h_toff, stream));
        MGM_TRY(cost_off.upload(h_coff, stream));
        MGM_TRY(tables.upload(tt, stream));
        MGM_TRY(var_cost.upload(vc, stream));
        for (int b = 0; b < 2; ++b) {
            MGM_TRY(cur[b].alloc(nV));
            MGM_TRY(cost[b].alloc(nV));
            MGM_TRY(vcc[b].alloc(nV));
        }
        h_rank = rk;
        MGM_TRY(grec.alloc(nV));
        MGM_TRY(has_cost.alloc(nV));
        g.n_vars = nV;
        g.is_max = p.mode == MXS_MODE_MAX;
        g.dom_size = dom_size.p; g.factor_rowptr = factor_rowptr.p; g.edge_var = edge_var.p;
        g.edge_factor = edge_factor.p; g.var_rowptr = var_rowptr.p; g.var_edges = var_edges.p;
        g.init_idx = nullptr; g.name_rank = name_rank.p; g.n_neigh = n_neigh.p;
        g.table_off = table_off.p; g.cost_off = cost_off.p; g.tables = tables.p; g.var_cost = var_cost.p;
        g.has_cost = has_cost.p; g.grec = grec.p;
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
        MGM_TRY(hipSetDevice(device));
        const int nV = g.n_vars;
        std::vector<int32_t> c0(nV);  // (all in packed order, Dev::q)
        std::vector<T> k0(nV, (T)0), v0(nV, (T)0);
        std::vector<uint8_t> h0(nV, 0);
        std::vector<GainRec<T>> gr(nV);  // no gain yet, the "new value" is the initial one
        for (int v = 0; v < nV; ++v) {
            const int qv = h_q[v];
            if (h_nn[v] == 0) {  // on_start without neighbours: optimal_cost_value (mgm.py:279-290)
                const int32_t* rk = h_vrank.empty() ? nullptr : h_vrank.data() + h_coff[v];
                int best = 0;
                for (int d = 1; d < h_dom[v]; ++d) {  // min / max over (cost, value) tuples, relations.py:1661-1665
                    const T a = (T)h_var_cost[h_coff[v] + d], b = (T)h_var_cost[h_coff[v] + best];
                    const int rd = rk ? rk[d] : d, rb = rk ? rk[best] : best;
                    if (g.is_max ? (a > b || (a == b && rd > rb)) : (a < b || (a == b && rd < rb))) best = d;
                }
                c0[qv] = best;
                k0[qv] = (T)h_var_cost[h_coff[v] + best];
                h0[qv] = 1;
            } else {  // the initial value, else the first of the domain (random.choice fixed)
                c0[qv] = h_init[v] >= 0 ? h_init[v] : 0;
            }
            v0[qv] = (T)h_var_cost[h_coff[v] + c0[qv]];
            gr[qv] = GainRec<T>{(T)0, c0[qv], h_rank[v]};
        }
        which = 0;
        if (nV) {
            for (int b = 0; b < 2; ++b) {  // both buffers: the packed launches write only the variables with neighbours
                MGM_TRY(hipMemcpyAsync(cur[b].p, c0.data(), 4 * nV, hipMemcpyHostToDevice, stream));
                MGM_TRY(hipMemcpyAsync(cost[b].p, k0.data(), sizeof(T) * nV, hipMemcpyHostToDevice, stream));
                MGM_TRY(hipMemcpyAsync(vcc[b].p, v0.data(), sizeof(T) * nV, hipMemcpyHostToDevice, stream));
            }
            MGM_TRY(hipMemcpyAsync(has_cost.p, h0.data(), nV, hipMemcpyHostToDevice, stream));
            MGM_TRY(hipMemcpyAsync(grec.p, gr.data(), sizeof(GainRec<T>) * nV, hipMemcpyHostToDevice, stream));
            MGM_TRY(hipStreamSynchronize(stream));
        }
        rounds = 0;
        return MXS_OK;
    }

    int run(int32_t n) override {
        MGM_TRY(hipSetDevice(device));
        const int nV = g.n_vars;
        if (nV == 0) {
            rounds += n > 0 ? n : 0;
            return MXS_OK;
        }
        const char* env = std::getenv("MAXSUM_LOCAL_SEARCH_GENERIC");  // =1: the CSR-walk kernels, =2: the slot
        const bool generic = env && env[0] == '1';                     // kernels for every variable (A/B, tests)
        const bool packed = !generic && !(env && env[0] == '2') && g.pack.n_lanes > 0;
        const dim3 pgrid((unsigned)((g.pack.n_lanes + PACK_TPB - 1) / PACK_TPB)), pblock(PACK_TPB);
        for (int32_t r = 0; r < n; ++r) {
            g.cur = cur[which].p;
            g.cost = cost[which].p;
            g.cur_out = cur[which ^ 1].p;
            g.cost_out = cost[which ^ 1].p;
            g.vcc = vcc[which].p;
            g.vcc_out = vcc[which ^ 1].p;
            T* const cw = cost[which].p;
            g.var_list = packed ? pk_rest.p : nullptr;
            g.n_list = packed ? n_rest : nV;
            const dim3 grid((unsigned)((g.n_list + TPB - 1) / TPB)), block(TPB);
            if (packed) {
                if (pack_int8) hipLaunchKernelGGL((k_mgm_gain_pack<T, int8_t>), pgrid, pblock, 0, stream, g, cw);
                else hipLaunchKernelGGL((k_mgm_gain_pack<T, T>), pgrid, pblock, 0, stream, g, cw);
                MGM_TRY(hipGetLastError());
            }
            if (g.n_list > 0) {
                if (generic || max_dom > 32) hipLaunchKernelGGL((k_mgm_gain<T>), grid, block, 0, stream, g, cw);
                else if (max_dom <= 4) hipLaunchKernelGGL((k_mgm_gain_slots<T, 4>), grid, block, 0, stream, g, cw);
                else if (max_dom <= 8) hipLaunchKernelGGL((k_mgm_gain_slots<T, 8>), grid, block, 0, stream, g, cw);
                else if (max_dom <= 16) hipLaunchKernelGGL((k_mgm_gain_slots<T, 16>), grid, block, 0, stream, g, cw);
                else hipLaunchKernelGGL((k_mgm_gain_slots<T, 32>), grid, block, 0, stream, g, cw);
                MGM_TRY(hipGetLastError());
            }
            if (packed) {
                hipLaunchKernelGGL((k_mgm_move_pack<T>), pgrid, pblock, 0, stream, g);
                MGM_TRY(hipGetLastError());
            }
            if (g.n_list > 0) {
                if (generic) hipLaunchKernelGGL((k_mgm_move<T>), grid, block, 0, stream, g);
                else hipLaunchKernelGGL((k_mgm_move_listed<T>), grid, block, 0, stream, g);
                MGM_TRY(hipGetLastError());
            }
            which ^= 1;
            rounds += 1;
        }
        MGM_TRY(hipStreamSynchronize(stream));
        return MXS_OK;
    }

    int get_state(int32_t* idx, double* cst, uint8_t* has, double* gn, int32_t* nv) override {
        MGM_TRY(hipSetDevice(device));
        const int nV = g.n_vars;
        if (!nV) return MXS_OK;
        std::vector<T> hc(nV);
        std::vector<GainRec<T>> hg(nV);
        std::vector<int32_t> hi(nV);
        std::vector<uint8_t> hh(nV);
        MGM_TRY(hipMemcpyAsync(hi.data(), cur[which].p, 4 * nV, hipMemcpyDeviceToHost, stream));
        MGM_TRY(hipMemcpyAsync(hh.data(), has_cost.p, nV, hipMemcpyDeviceToHost, stream));
        MGM_TRY(hipMemcpyAsync(hc.data(), cost[which].p, sizeof(T) * nV, hipMemcpyDeviceToHost, stream));
        MGM_TRY(hipMemcpyAsync(hg.data(), grec.p, sizeof(GainRec<T>) * nV, hipMemcpyDeviceToHost, stream));
        MGM_TRY(hipStreamSynchronize(stream));
        for (int v = 0; v < nV; ++v) {  // the state lives in packed order (Dev::q)
            const int qv = h_q[v];
            if (idx) idx[v] = hi[qv];
            if (has) has[v] = hh[qv];
            if (cst) cst[v] = (double)hc[qv];
            if (gn) gn[v] = (double)hg[qv].gain;
            if (nv) nv[v] = hg[qv].newv;
        }
        return MXS_OK;
    }

    int eval_cost(const int32_t* idx, double infinity, double* cst, int64_t* viol) override {
        std::vector<int32_t> c;
        if (!idx) {
            c.resize(g.n_vars);
            int rc = get_state(c.data(), nullptr, nullptr, nullptr, nullptr);
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

}  // namespace mgm

struct mxs_mgm {
    mgm::Base* impl;
};

extern "C" {

int mxs_mgm_create(const mxs_graph* g, const mxs_params* p, const int32_t* name_rank, int32_t device, mxs_mgm** out) {
    if (!g || !p || !out) return mgm::fail(MXS_E_INVALID, "null argument");
    *out = nullptr;
    try {
        mgm::Base* impl = p->dtype == MXS_DTYPE_F32 ? (mgm::Base*)new mgm::Engine<float>() : (mgm::Base*)new mgm::Engine<double>();
        int rc = impl->init(*g, *p, name_rank, device);
        if (rc) {
            delete impl;
            return rc;
        }
        *out = new mxs_mgm{impl};
        return MXS_OK;
    } catch (const std::exception& ex) {
        return mgm::fail(MXS_E_NOMEM, ex.what());
    }
}
int mxs_mgm_reset(mxs_mgm* e) { return e ? e->impl->reset() : mgm::fail(MXS_E_INVALID, "null handle"); }
int mxs_mgm_set_value_rank(mxs_mgm* e, const int32_t* value_rank) {
    return e ? e->impl->set_value_rank(value_rank) : mgm::fail(MXS_E_INVALID, "null handle");
}
int mxs_mgm_run(mxs_mgm* e, int32_t n_rounds) {
    if (!e) return mgm::fail(MXS_E_INVALID, "null handle");
    if (n_rounds < 0) return mgm::fail(MXS_E_INVALID, "negative round count");
    return e->impl->run(n_rounds);
}
int mxs_mgm_rounds(const mxs_mgm* e, int64_t* rounds) {
    if (!e) return mgm::fail(MXS_E_INVALID, "null handle");
    if (rounds) *rounds = e->impl->rounds;
    return MXS_OK;
}
int mxs_mgm_get_state(mxs_mgm* e, int32_t* idx, double* cost, uint8_t* has_cost, double* gain, int32_t* new_value) {
    return e ? e->impl->get_state(idx, cost, has_cost, gain, new_value) : mgm::fail(MXS_E_INVALID, "null handle");
}
int mxs_mgm_eval_cost(mxs_mgm* e, const int32_t* idx, double infinity, double* cost, int64_t* violations) {
    return e ? e->impl->eval_cost(idx, infinity, cost, violations) : mgm::fail(MXS_E_INVALID, "null handle");
}
int mxs_mgm_destroy(mxs_mgm* e) {
    if (e) {
        delete e->impl;
        delete e;
    }
    return MXS_OK;
}

}  // extern "C"
