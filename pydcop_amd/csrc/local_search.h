// local_search.h -- what dsa.hip and mgm.hip share: the flat "slot" view of a variable's
// constraints.  Both algorithms evaluate, per variable v and per cycle, the cost of EVERY value x
// of v under the neighbours' current values (dsa.py:319-333 via assignment_cost,
// relations.py:1513-1533; mgm.py:425-451 via slices + find_arg_optimal).  Walking the factor CSR
// once per (x, constraint) is a chain of six dependent loads repeated (D+1)*deg times; here each
// (variable, constraint) pair is a SLOT prepared once on the host:
//
//   entry(x) = tables[ base + x * stride_v + sum over the OTHER scope variables u of cur[u] * stride_u ]
//
// so a thread computes the offset of a slot once per cycle (one load per other variable) and
// then reads the D entries, independent loads.  The sums over constraints keep the reference's
// order (the variable's constraints in var_edges order), so the results are bit for bit those of
// the CSR walk (kept in the two files as the generic path for domains larger than the register array).
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <atomic>
#include <new>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace lsearch {

struct Slots {  // device pointers; slot s of variable v: var_rowptr[v] <= s < var_rowptr[v+1]
    const int64_t* base;        // [n_slots] table_off of the slot's constraint
    const int32_t* stride_v;    // [n_slots] sum of the strides of the positions v holds in the scope
    const int32_t* nb_rowptr;   // [n_slots+1] the other scope variables of the slot
    const int32_t* nb_var;
    const int32_t* nb_stride;
    const int32_t* nb0_var;     // [n_slots] the first of them again (variable 0, stride 0 when there is
    const int32_t* nb0_stride;  //           none): a binary constraint costs no walk of the list
    const int32_t* conc_rowptr; // [n_vars+1] distinct variables of v's constraints (v included), ascending
    const int32_t* conc_var;
    // ROW view (HostSlots::build_rows; NULL when not built): a PRIVATE copy of the constraint's table per slot with
    // the variable's OWN axis last -- the D entries a variable needs of a constraint, for the neighbours' current
    // values, are ONE contiguous row of round_up(D * elem, 8) bytes instead of D entries a stride apart (24 cache
    // lines per constraint on the 24^3 tables of meeting_50k).  int8 when every entry is a small integer, else T.
    const uint8_t* rows;
    const int64_t* row_base;        // [n_slots] byte offset of the slot's copy, -1: none
    const int32_t* row_nb_stride;   // per entry of nb_var: that variable's stride among the OTHER variables, in rows
    const int32_t* row_nb0_stride;  // [n_slots] the first one's again
    int32_t rows_int8;
};

struct HostSlots {
    std::vector<int64_t> base;
    std::vector<int32_t> stride_v, nb_rowptr, nb_var, nb_stride, nb0_var, nb0_stride, conc_rowptr, conc_var;
    // the row view (Slots::rows), for the variables of `vars` -- built when it fits `budget` bytes
    std::vector<uint8_t> rows;
    std::vector<int64_t> row_base;
    std::vector<int32_t> row_nb_stride, row_nb0_stride;
    bool rows_int8 = false;

    static bool small_int(double e) { return e >= -128.0 && e <= 127.0 && e == (double)(int)e && !(e == 0.0 && std::signbit(e)); }

    // word: sizeof(T) of the engine.  Returns true when the view was built.
    bool build_rows(const std::vector<int32_t>& vars, const std::vector<int32_t>& dom, const std::vector<int32_t>& vrow,
                    const std::vector<int64_t>& toff, const std::vector<double>& tables, int word, int max_dom,
                    int64_t budget) {
        const size_t nS = base.size();
        row_base.assign(nS, -1);
        row_nb_stride.assign(nb_var.size(), 0);
        row_nb0_stride.assign(nS, 0);
        rows.clear();
        if (vars.empty()) return false;
        // every table a slot of these variables reads: small integers?
        rows_int8 = true;
        for (int v : vars)
            if (dom[v] > max_dom) return false;
        // (which tables: found through base[] == table_off of the constraint)
        {
            for (int v : vars)
                for (int s = vrow[v]; s < vrow[v + 1] && rows_int8; ++s) {
                    int64_t n = dom[v];
                    for (int k = nb_rowptr[s]; k < nb_rowptr[s + 1]; ++k) n *= dom[nb_var[k]];
                    const double* t = tables.data() + base[s];
                    for (int64_t i = 0; i < n; ++i)
                        if (!small_int(t[i])) {
                            rows_int8 = false;
                            break;
                        }
                }
        }
        const int elem = rows_int8 ? 1 : word;
        int64_t bytes = 0;
        for (int v : vars) {
            const int64_t rs = ((int64_t)dom[v] * elem + 7) / 8 * 8;
            for (int s = vrow[v]; s < vrow[v + 1]; ++s) {
                int64_t R = 1;
                for (int k = nb_rowptr[s]; k < nb_rowptr[s + 1]; ++k) R *= dom[nb_var[k]];
                row_base[s] = bytes;
                bytes += R * rs;
            }
        }
        if (bytes > budget) {
            row_base.assign(nS, -1);
            return false;
        }
        try {
            rows.assign((size_t)bytes + 8, 0);
        } catch (const std::bad_alloc&) {  // a small host: no row view, the strided path works everywhere
            rows.clear();
            row_base.assign(nS, -1);
            return false;
        }
        // fill, a few host threads over the variables (nothing may leave a worker: an exception there would terminate)
        std::atomic<bool> failed{false};
        auto fill = [&](size_t lo, size_t hi) {
            try {
            std::vector<int> digit;
            for (size_t vi = lo; vi < hi; ++vi) {
                const int v = vars[vi], D = dom[v];
                const int64_t rs = ((int64_t)D * elem + 7) / 8 * 8;
                for (int s = vrow[v]; s < vrow[v + 1]; ++s) {
                    const int k0 = nb_rowptr[s], k1 = nb_rowptr[s + 1], no = k1 - k0;
                    // nb_var lists the other variables from the LAST scope position to the first (build()): the row
                    // index keeps that order of significance -- first listed = fastest
                    int64_t R = 1;
                    for (int k = k0; k < k1; ++k) {
                        row_nb_stride[k] = (int32_t)R;
                        R *= dom[nb_var[k]];
                    }
                    row_nb0_stride[s] = no > 0 ? row_nb_stride[k0] : 0;
                    digit.assign(no, 0);
                    int64_t src = base[s];
                    uint8_t* dst = rows.data() + row_base[s];
                    for (int64_t r = 0; r < R; ++r) {
                        for (int x = 0; x < D; ++x) {
                            const double e = tables[src + (int64_t)x * stride_v[s]];
                            if (elem == 1) ((int8_t*)dst)[x] = (int8_t)e;
                            else if (elem == 4) ((float*)dst)[x] = (float)e;
                            else ((double*)dst)[x] = e;
                        }
                        dst += rs;
                        for (int k = 0; k < no; ++k) {  // next combination of the others
                            src += nb_stride[k0 + k];
                            if (++digit[k] < dom[nb_var[k0 + k]]) break;
                            src -= (int64_t)nb_stride[k0 + k] * digit[k];
                            digit[k] = 0;
                        }
                    }
                }
            }
            } catch (...) {
                failed = true;
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nt = std::max<size_t>(1, std::min<size_t>({(size_t)(hw ? hw : 1), (size_t)32, vars.size() / 64 + 1}));
        std::vector<std::thread> pool;
        for (size_t t = 0; t < nt; ++t) pool.emplace_back(fill, vars.size() * t / nt, vars.size() * (t + 1) / nt);
        for (std::thread& th : pool) th.join();
        if (failed) {
            rows.clear();
            rows.shrink_to_fit();
            row_base.assign(nS, -1);
            return false;
        }
        return true;
    }

    // The budget of the row view in bytes: $MAXSUM_LOCAL_SEARCH_ROWS (0 = none; A/B runs and tests), else at most 6 GiB
    // and at most half of the device memory that is free NOW -- the view is an optimisation, an instance that ran without
    // it must not fail to load because of it.
    static int64_t rows_budget() {
        if (const char* renv = std::getenv("MAXSUM_LOCAL_SEARCH_ROWS")) return std::atoll(renv);
        int64_t budget = (int64_t)6 << 30;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min<int64_t>(budget, (int64_t)(free_b / 2));
        return budget;
    }

    // "" or what is wrong with the instance
    std::string build(int nV, int nF, const std::vector<int32_t>& dom, const std::vector<int32_t>& frow,
                      const std::vector<int32_t>& evar, const std::vector<int64_t>& toff,
                      const std::vector<int32_t>& vrow, const std::vector<int32_t>& vedges) {
        const int nE = (int)evar.size();
        std::vector<int32_t> efac(nE);
        for (int f = 0; f < nF; ++f)
            for (int e = frow[f]; e < frow[f + 1]; ++e) efac[e] = f;
        base.resize(nE);
        stride_v.resize(nE);
        nb_rowptr.assign(nE + 1, 0);
        nb0_var.assign(nE, 0);
        nb0_stride.assign(nE, 0);
        nb_var.clear();
        nb_stride.clear();
        conc_rowptr.assign(nV + 1, 0);
        conc_var.clear();
        std::vector<int32_t> seen;
        for (int v = 0; v < nV; ++v) {
            seen.clear();
            seen.push_back(v);
            for (int s = vrow[v]; s < vrow[v + 1]; ++s) {
                if (vedges[s] < 0 || vedges[s] >= nE) return "var_edges out of range";
                const int f = efac[vedges[s]];
                if (toff[f + 1] - toff[f] > INT32_MAX) return "table too large for the local-search kernels";
                base[s] = toff[f];
                int64_t stride = 1, sv = 0;
                for (int e = frow[f + 1] - 1; e >= frow[f]; --e) {  // row-major: last position is contiguous
                    const int u = evar[e];
                    if (u == v) {
                        sv += stride;
                    } else {
                        nb_var.push_back(u);
                        nb_stride.push_back((int32_t)stride);
                        seen.push_back(u);
                    }
                    stride *= dom[u];
                }
                stride_v[s] = (int32_t)sv;
                nb_rowptr[s + 1] = (int32_t)nb_var.size();
                if (nb_rowptr[s + 1] > nb_rowptr[s]) {
                    nb0_var[s] = nb_var[nb_rowptr[s]];
                    nb0_stride[s] = nb_stride[nb_rowptr[s]];
                }
            }
            std::sort(seen.begin(), seen.end());
            seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
            conc_var.insert(conc_var.end(), seen.begin(), seen.end());
            conc_rowptr[v + 1] = (int32_t)conc_var.size();
        }
        return "";
    }
};

// c[x] = sum over v's slots, in order, of the slot's entry at x (x < D <= MAXD; c[x] is unspecified
// for x >= D); `from_zero`: DSA starts from 0 (assignment_cost), MGM folds without an initial
// value (functools.reduce).  CH slots at a time with every load of a level issued together
// (clamped indices instead of branches): a variable of degree <= CH pays the four dependent levels
// (row pointers, slot, neighbour's value, table entries) once instead of once per constraint.
template <typename T, int MAXD>
__device__ inline void costs_of_values(const Slots& sl, const T* __restrict__ tables, const int32_t* __restrict__ cur,
                                       int s0, int s1, int D, bool from_zero, T (&c)[MAXD]) {
    constexpr int CH = MAXD <= 4 ? 4 : (MAXD <= 8 ? 2 : 1);
#pragma unroll
    for (int x = 0; x < MAXD; ++x) c[x] = (T)0;
    for (int s = s0; s < s1; s += CH) {
        int64_t off[CH];
        int sv[CH], k0[CH], k1[CH], u0[CH], st0[CH];
        int more = 0;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int si = s + i < s1 ? s + i : s1 - 1;
            off[i] = sl.base[si];
            sv[i] = sl.stride_v[si];
            k0[i] = sl.nb_rowptr[si] + 1;
            k1[i] = sl.nb_rowptr[si + 1];
            u0[i] = sl.nb0_var[si];
            st0[i] = sl.nb0_stride[si];
            more = k1[i] - k0[i] > more ? k1[i] - k0[i] : more;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) off[i] += (int64_t)cur[u0[i]] * st0[i];
        for (int q = 0; q < more; ++q) {  // arity > 2
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const bool in = k0[i] + q < k1[i];
                const int k = in ? k0[i] + q : k0[i] - 1;
                off[i] += in ? (int64_t)cur[sl.nb_var[k]] * sl.nb_stride[k] : (int64_t)0;
            }
        }
        T t[CH][MAXD];
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int x = 0; x < MAXD; ++x) t[i][x] = tables[off[i] + (int64_t)(x < D ? x : D - 1) * sv[i]];
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (s + i < s1) {
                const bool first = !from_zero && s + i == s0;
#pragma unroll
                for (int x = 0; x < MAXD; ++x) c[x] = first ? t[i][x] : c[x] + t[i][x];
            }
    }
}

// The same costs from the ROW view (Slots::rows): per slot one index (the neighbours' current values in the
// slot's private copy) and ONE contiguous row of D entries, read in 8-byte pieces -- two slots' rows in flight
// together.  int8 rows: integer sums, converted once (every entry is an integer in [-128, 127], at most a few
// thousand of them per variable: every partial sum of the reference's left-to-right fold is an integer far below
// 2^24, exact in f32 and f64 whatever the order -- the argument of pack_costs).
template <typename T, typename TT, int MAXD>
__device__ inline void costs_of_values_rows(const Slots& sl, const int32_t* __restrict__ cur, int s0, int s1, int D,
                                            bool from_zero, T (&c)[MAXD]) {
    constexpr int EPP = 8 / (int)sizeof(TT);             // entries per 8-byte piece
    constexpr int NP = (MAXD + EPP - 1) / EPP;
    const int np = (D + EPP - 1) / EPP;
    const int64_t rs = (int64_t)np * 8;
    int ci[MAXD];
#pragma unroll
    for (int x = 0; x < MAXD; ++x) {
        c[x] = (T)0;
        ci[x] = 0;
    }
    for (int s = s0; s < s1; s += 2) {
        uint64_t w[2][NP];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int si = s + i < s1 ? s + i : s1 - 1;
            int64_t ridx = (int64_t)cur[sl.nb0_var[si]] * sl.row_nb0_stride[si];
            for (int k = sl.nb_rowptr[si] + 1; k < sl.nb_rowptr[si + 1]; ++k) ridx += (int64_t)cur[sl.nb_var[k]] * sl.row_nb_stride[k];
            const uint64_t* r = (const uint64_t*)(sl.rows + sl.row_base[si] + ridx * rs);
#pragma unroll
            for (int q = 0; q < NP; ++q) w[i][q] = r[q < np ? q : np - 1];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (s + i < s1) {
                const bool first = !from_zero && s + i == s0;
#pragma unroll
                for (int x = 0; x < MAXD; ++x) {
                    if constexpr (sizeof(TT) == 1) {
                        ci[x] += (int)(int8_t)(uint8_t)(w[i][x / 8] >> (8 * (x % 8)));
                    } else {
                        TT e;
                        const uint64_t piece = w[i][x / EPP] >> (8 * (int)sizeof(TT) * (x % EPP));
                        if constexpr (sizeof(TT) == 8) {
                            __builtin_memcpy(&e, &piece, 8);
                        } else {
                            const uint32_t lo = (uint32_t)piece;
                            __builtin_memcpy(&e, &lo, 4);
                        }
                        c[x] = first ? (T)e : c[x] + (T)e;
                    }
                }
            }
    }
    if constexpr (sizeof(TT) == 1) {
#pragma unroll
        for (int x = 0; x < MAXD; ++x) c[x] = (T)ci[x];
    }
}

// ---------------------------------------------------------------------------------------------
// The PACKED view: one LANE per (variable, constraint) slot, for variables whose every constraint
// is unary or binary over domains of at most PACK_D values (graph colouring, Ising, most
// benchmark families).  Like the packed variable class of the Max-Sum sweep (kernels.h,
// variable_pack): the variables are grouped by degree, a wave holds floor(64 / deg) of them side
// by side (lane = var_in_wave * deg + k), and everything a lane needs is laid out in LANE order:
//   * nb[lane]      the other variable of the lane's constraint (-1: none),
//   * a PRIVATE, TRANSPOSED copy of the constraint's table, rec[lane][y][x] = the entry for the
//     neighbour's value y and the own value x -- the PACK_D entries the lane needs for the
//     neighbour's current value are ONE aligned row (4 bytes when every entry of every table is a
//     small integer and the records are int8, else PACK_D elements of T), inside a record that the
//     wave's neighbouring lanes stream through anyway.
// The thread-per-variable kernels gather, per constraint, a slot record, the neighbour's value and
// D table entries at a stride -- three to five random 64-byte requests; here the only random access
// left is the neighbour's value (an array of 4 bytes per variable: L2-resident).  Sums over a
// variable's constraints are cross-lane reads in slot order: the reference's order, bit for bit.
constexpr int PACK_D = 4;
struct PackWave {
    int32_t first;    // index of the wave's first variable in HostPack::vars = its position in the engines' state arrays
    int32_t deg_nv;   // deg | nv << 8 | ceil(2^15 / deg) << 16
};
struct Pack {  // device pointers
    const PackWave* waves;
    const int32_t* nb;     // [lanes] the other variable's index INTO THE STATE ARRAYS (the engines store their dynamic
                           // state in packed order -- position = index into HostPack::vars -- and upload nb so)
    const int32_t* slot;   // [lanes] CSR slot of the lane, -1 = padding lane
    const void* rec;       // [lanes][PACK_D][PACK_D] int8 or T
    int32_t n_lanes;
};
struct HostPack {
    std::vector<PackWave> waves;
    std::vector<int32_t> vars, nb, slot, rest;  // rest: variables with neighbours the pack cannot take
    std::vector<int32_t> lane_var, lane_k, lane_deg;  // [lanes] the lane's variable (-1: padding), position, degree
    std::vector<double> rec;                    // [lanes * 16], as doubles; narrowed at upload
    bool int8_exact = true;

    void build(int nV, const std::vector<int32_t>& dom, const std::vector<int32_t>& vrow, const std::vector<int32_t>& n_neigh,
               const HostSlots& hs, const std::vector<double>& tables) {
        std::vector<std::vector<int>> by_deg(65);
        for (int v = 0; v < nV; ++v) {
            if (n_neigh[v] == 0) continue;  // never moves (dsa.py:278-289, mgm.py:335): no work
            const int deg = vrow[v + 1] - vrow[v];
            bool ok = dom[v] <= PACK_D && deg >= 1 && deg <= 64;
            for (int s = vrow[v]; ok && s < vrow[v + 1]; ++s) {
                const int n_nb = hs.nb_rowptr[s + 1] - hs.nb_rowptr[s];
                if (n_nb > 1 || (n_nb == 1 && dom[hs.nb0_var[s]] > PACK_D)) ok = false;
            }
            if (ok) by_deg[deg].push_back(v);
            else rest.push_back(v);
        }
        for (int deg = 1; deg <= 64; ++deg) {
            const std::vector<int>& vs = by_deg[deg];
            const int per_wave = 64 / deg;
            for (size_t x = 0; x < vs.size(); x += per_wave) {
                const int nv = (int)std::min<size_t>(per_wave, vs.size() - x);
                waves.push_back(PackWave{(int32_t)vars.size(), (int32_t)((uint32_t)deg | ((uint32_t)nv << 8) |
                                                                          ((uint32_t)((32768 + deg - 1) / deg) << 16))});
                for (int i = 0; i < nv; ++i) vars.push_back(vs[x + i]);
                for (int lane = 0; lane < 64; ++lane) {
                    const int var = lane / deg, k = lane % deg;
                    lane_var.push_back(var < nv ? vs[x + var] : -1);
                    lane_k.push_back(k);
                    lane_deg.push_back(deg);
                    if (var >= nv) {
                        nb.push_back(-1);
                        slot.push_back(-1);
                        rec.insert(rec.end(), PACK_D * PACK_D, 0.0);
                        continue;
                    }
                    const int v = vs[x + var], s = vrow[v] + k;
                    const bool has_nb = hs.nb_rowptr[s + 1] > hs.nb_rowptr[s];
                    const int u = has_nb ? hs.nb0_var[s] : -1;
                    nb.push_back(u);
                    slot.push_back(s);
                    for (int y = 0; y < PACK_D; ++y)
                        for (int xx = 0; xx < PACK_D; ++xx) {
                            double e = 0.0;
                            if (xx < dom[v] && y < (has_nb ? dom[u] : 1))
                                e = tables[hs.base[s] + (int64_t)xx * hs.stride_v[s] + (has_nb ? (int64_t)y * hs.nb0_stride[s] : 0)];
                            rec.push_back(e);
                            if (!(e >= -128.0 && e <= 127.0 && e == (double)(int)e) || (e == 0.0 && std::signbit(e)))
                                int8_exact = false;
                        }
                }
            }
        }
    }
};

// The PACK_D costs of the lane's variable, c[x] = sum over its constraints in slot order of the
// entry at (x, neighbour's current value); `from_zero` as in costs_of_values.  t[] = the lane's own
// row (what its constraint contributes).  All 64 lanes take part (padding lanes read record zeros).
struct alignas(16) Rec16 {
    uint32_t x, y, z, w;
};
template <typename T, typename TT>
__device__ inline void pack_costs(const Pack& pk, const int32_t* __restrict__ cur, int64_t pos, int deg, int seg,
                                  bool from_zero, T (&t)[PACK_D], T (&c)[PACK_D]) {
    const int u = pk.nb[pos];
#pragma unroll
    for (int x = 0; x < PACK_D; ++x) c[x] = (T)0;
    if constexpr (sizeof(TT) == 1) {
        // the whole 16-byte record is requested with the neighbour's index (the wave streams through these
        // lines anyway); the row is picked in registers once the neighbour's value is in: one dependent
        // level less than a load at rec[pos][y]
        const Rec16 r = *(const Rec16*)((const uint8_t*)pk.rec + pos * (PACK_D * PACK_D));
        const int y = u >= 0 ? cur[u] : 0;
        const uint32_t w = y == 0 ? r.x : (y == 1 ? r.y : (y == 2 ? r.z : r.w));
#pragma unroll
        for (int x = 0; x < PACK_D; ++x) t[x] = (T)(int)(int8_t)(uint8_t)(w >> (8 * x));
        // The row travels between lanes as the one packed word it is, and the sums run in INTEGER
        // arithmetic: every entry is an integer in [-128, 127] (int8_exact, no -0.0), at most 64 of them are
        // added, so every partial sum of the reference's left-to-right floating-point fold is an integer far
        // below 2^24 -- exactly representable in f32 and f64, whatever the order and whether the fold starts
        // from 0 or from the first term: the integer sum converted once IS that fold, bit for bit.
        int ci[PACK_D];
#pragma unroll
        for (int x = 0; x < PACK_D; ++x) ci[x] = 0;
        for (int kk = 0; kk < deg; ++kk) {
            const uint32_t wk = (uint32_t)__shfl((int)w, seg + kk, 64);
#pragma unroll
            for (int x = 0; x < PACK_D; ++x) ci[x] += (int)(int8_t)(uint8_t)(wk >> (8 * x));
        }
#pragma unroll
        for (int x = 0; x < PACK_D; ++x) c[x] = (T)ci[x];
    } else {
        const int y = u >= 0 ? cur[u] : 0;
        const T* r = (const T*)pk.rec + pos * (PACK_D * PACK_D) + y * PACK_D;
#pragma unroll
        for (int x = 0; x < PACK_D; ++x) t[x] = r[x];
        for (int kk = 0; kk < deg; ++kk) {  // the PACK_D exchanges of a step in flight together
            T e[PACK_D];
#pragma unroll
            for (int x = 0; x < PACK_D; ++x) e[x] = __shfl(t[x], seg + kk, 64);
            const bool first = !from_zero && kk == 0;
#pragma unroll
            for (int x = 0; x < PACK_D; ++x) c[x] = first ? e[x] : c[x] + e[x];
        }
    }
}

// c[i] with a run-time i, the array staying in registers.  A plain chain of `x == i ? c[x] : r`
// is rewritten by the optimiser into ONE load at a selected address -- a dynamically indexed
// private array, i.e. scratch memory, from 128 bytes on; the empty asm makes every element a value
// of its own before the selects.
template <typename T, int MAXD>
__device__ inline T pick(const T (&c)[MAXD], int i) {
    T r = c[0];
#pragma unroll
    for (int x = 1; x < MAXD; ++x) {
        T cx = c[x];
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(cx));
#endif
        r = x == i ? cx : r;
    }
    return r;
}

}  // namespace lsearch
