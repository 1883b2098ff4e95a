// layout.cpp -- see layout.h.  Pure host C++ (no HIP): O(E) compilation of the
// caller's flat factor graph into classes, records and block descriptors.
#include "layout.h"

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <numeric>
#include <map>
#include <sstream>
#include <tuple>

namespace mxs {

LayoutOptions options_from_params(const mxs_params& p) {
    LayoutOptions o;
    o.word = (p.dtype == MXS_DTYPE_F32) ? 4 : 8;
    const int f = p.layout_flags;
    // bits 0/1 selected record paddings of the v1 layout; ignored since the
    // gather-only layout (messages are always padded to a sector-friendly size)
    o.no_specialise = (f & 4) != 0;   // bit2: generic kernels only
    o.sort_by_degree = !(f & 8);      // bit3: keep the caller's variable order
    o.nary = !(f & 16);               // bit4: no workgroup-per-factor kernel
    o.sort_factors = MXS_SORT_FACTORS_DEFAULT != 0;
    if (f & 128) o.sort_factors = true;   // bit7: factor order follows the variable order
    if (f & 256) o.sort_factors = false;  // bit8: factors of a class keep the caller's order
    o.factors_second = MXS_FACTORS_SECOND_DEFAULT != 0;
    if (f & 512) o.factors_second = true;    // bit9: shard: every factor class in the second launch
    if (f & 1024) o.factors_second = false;  // bit10: shard: only the cut factor classes
    o.schedule = MXS_SCHEDULE_DEFAULT != 0;
    if (f & 4096) o.schedule = true;   // bit12: co-scheduled, XCD-contiguous block order
    if (f & 2048) o.schedule = false;  // bit11: blocks in class order
    o.compact_tables = MXS_COMPACT_TABLES_DEFAULT != 0;
    if (f & 16384) o.compact_tables = true;   // bit14: narrow storage of exactly-representable tables
    if (f & 8192) o.compact_tables = false;   // bit13: full-width tables
    o.box = !(f & 32768);                     // bit15: no one-wave-per-factor box kernel (lane-packed instead)
    o.half_cut = !(f & 65536);                // bit16: a shard's cut binary factors compute both messages (round 3)
    o.pack8_fused = !(f & 2097152);           // bit21: the lane-per-edge class of 5..8 values in a launch of its own
    o.pack8 = !(f & 1048576);                 // bit20: variables of 5..8 values stay in the wide (workgroup-per-run) class
    o.bin2 = !(f & 524288);                   // bit19: no lane-grid kernel for binary / unary factors (generic instead)
    o.merge_types = !(f & 16777216);          // bit24: lane-grid groups of one shape stay split by storage type (A/B runs)
    o.nary_multi = !(f & 33554432);           // bit25: no multi-pass workgroup kernel (tables beyond 1 024 entries per value of the first variable and arity 6: generic)
    o.small = !(f & 8388608);                 // bit23: no small-domain lane-group kernel (workgroup per factor instead)
    o.hub = !(f & 4194304);                   // bit22: no wave-per-64-edges class for hub variables (thread per variable instead)
    o.tile_bytes = -1;                                   // tiled factor order: decided per instance (build_layout)
    if (f & 131072) o.tile_bytes = MXS_TILE_BYTES;       // bit17: always tiled
    if (f & 262144) o.tile_bytes = 0;                    // bit18: never tiled
    // $MAXSUM_TILE_KB (A/B runs and the parity tests of the tiled order; 0 = off): only where the caller's flags
    // leave the choice open, and only a number counts
    if (!(f & (131072 | 262144)))
        if (const char* tb = std::getenv("MAXSUM_TILE_KB")) {
            char* end = nullptr;
            const long long kb = std::strtoll(tb, &end, 10);
            if (end != tb && *end == '\0' && kb >= 0) o.tile_bytes = (int64_t)kb * 1024;
        }
    return o;
}

namespace {

struct FKey {
    int kind, D;
    int cut = 0;  // 1: the factor reads a ghost variable's message (sharded operation)
    int own = 0;  // cut binary register factors: 1 / 2 = only scope position 0 / 1 is an owned variable (ClassInfo::own_pos)
    bool operator<(const FKey& o) const {
        return cut != o.cut ? cut < o.cut : kind != o.kind ? kind < o.kind : D != o.D ? D < o.D : own < o.own;
    }
    bool operator==(const FKey& o) const { return kind == o.kind && D == o.D && cut == o.cut && own == o.own; }
};

std::string validate(const mxs_graph& g) {
    std::ostringstream err;
    if (g.n_vars < 0 || g.n_factors < 0 || g.n_edges < 0) return "negative size";
    if (g.n_vars && (!g.dom_size || !g.var_cost || !g.var_rowptr)) return "null variable arrays";
    if (g.n_factors && (!g.factor_rowptr || !g.table_off || !g.tables)) return "null factor arrays";
    if (g.n_edges && (!g.edge_var || !g.var_edges)) return "null edge arrays";
    for (int32_t v = 0; v < g.n_vars; ++v)
        if (g.dom_size[v] < 1 || g.dom_size[v] > 4096) {
            err << "variable " << v << ": domain size " << g.dom_size[v] << " not in 1..4096";
            return err.str();
        }
    if (g.n_factors) {
        if (g.factor_rowptr[0] != 0 || g.factor_rowptr[g.n_factors] != g.n_edges)
            return "factor_rowptr does not span the edges";
        if (g.table_off[0] != 0) return "table_off[0] must be 0";
    } else if (g.n_edges) {
        return "edges without factors";
    }
    for (int32_t f = 0; f < g.n_factors; ++f) {
        const int32_t e0 = g.factor_rowptr[f], e1 = g.factor_rowptr[f + 1];
        if (e1 <= e0) { err << "factor " << f << " has no variable"; return err.str(); }
        if (e1 - e0 > MAX_ARITY) { err << "factor " << f << ": arity > " << MAX_ARITY << " not supported"; return err.str(); }
        int64_t size = 1;
        for (int32_t e = e0; e < e1; ++e) {
            const int32_t v = g.edge_var[e];
            if (v < 0 || v >= g.n_vars) { err << "edge " << e << ": variable out of range"; return err.str(); }
            for (int32_t e2 = e0; e2 < e; ++e2)
                if (g.edge_var[e2] == v) { err << "factor " << f << " lists variable " << v << " twice"; return err.str(); }
            size *= g.dom_size[v];
            if (size > ((int64_t)1 << 40)) { err << "factor " << f << ": table too large"; return err.str(); }
        }
        if (g.table_off[f + 1] - g.table_off[f] != size) {
            err << "factor " << f << ": table_off inconsistent with scope (" << size << " entries expected)";
            return err.str();
        }
    }
    if (g.n_vars) {
        if (g.var_rowptr[0] != 0 || g.var_rowptr[g.n_vars] != g.n_edges)
            return "var_rowptr does not span the edges";
        std::vector<uint8_t> seen(g.n_edges, 0);
        for (int32_t v = 0; v < g.n_vars; ++v) {
            if (g.var_rowptr[v + 1] < g.var_rowptr[v]) return "var_rowptr not monotone";
            for (int32_t k = g.var_rowptr[v]; k < g.var_rowptr[v + 1]; ++k) {
                const int32_t e = g.var_edges[k];
                if (e < 0 || e >= g.n_edges || seen[e] || g.edge_var[e] != v) {
                    err << "var_edges[" << k << "] inconsistent";
                    return err.str();
                }
                seen[e] = 1;
            }
            if (g.init_idx && g.init_idx[v] >= g.dom_size[v]) return "init_idx out of the domain";
        }
    }
    return "";
}

}  // namespace

int narrowest_tab_type(const double* v, int64_t n, int word) {
    bool i8 = true, i16 = true, f32 = true;
    for (int64_t k = 0; k < n; ++k) {
        const double x = v[k];
        const bool integral = std::isfinite(x) && x == std::floor(x) && !(x == 0.0 && std::signbit(x));
        if (!integral || std::fabs(x) > 127.0) i8 = false;
        if (!integral || std::fabs(x) > 32767.0) i16 = false;
        if (!((double)(float)x == x || (x != x))) f32 = false;  // (NaN stays NaN; its payload is not compared)
        if (x != x) i8 = i16 = false;
        if (!i16 && !f32) break;
    }
    if (i8) return TAB_I8;
    if (i16) return TAB_I16;
    if (f32 && word == 8) return TAB_F32;
    return TAB_FULL;
}

void encode_tab_record(const double* v, int entries, int t, uint8_t* dst) {
    for (int k = 0; k < entries; ++k) {
        if (t == TAB_I8) {
            const int8_t x = (int8_t)v[k];
            std::memcpy(dst + k, &x, 1);
        } else if (t == TAB_I16) {
            const int16_t x = (int16_t)v[k];
            std::memcpy(dst + 2 * k, &x, 2);
        } else {
            const float x = (float)v[k];
            std::memcpy(dst + 4 * k, &x, 4);
        }
    }
}

void encode_tab_entry(double v, int t, int word, uint8_t* dst) {
    if (t != TAB_FULL) {
        encode_tab_record(&v, 1, t, dst);
    } else if (word == 8) {
        std::memcpy(dst, &v, 8);
    } else {
        const float x = (float)v;
        std::memcpy(dst, &x, 4);
    }
}

std::string build_layout(const mxs_graph& g, const mxs_params& p, Layout& L) {
    {
        std::string e = validate(g);
        if (!e.empty()) return e;
    }
    if (p.mode != MXS_MODE_MIN && p.mode != MXS_MODE_MAX) return "invalid mode";
    if (p.damping_nodes < 0 || p.damping_nodes > 3) return "invalid damping_nodes";
    if (p.start_messages < 0 || p.start_messages > 2) return "invalid start_messages";
    if (p.dtype != MXS_DTYPE_F64 && p.dtype != MXS_DTYPE_F32) return "invalid dtype";

    L = Layout();
    L.opt = options_from_params(p);
    L.n_vars = g.n_vars;
    L.n_factors = g.n_factors;
    L.n_edges = g.n_edges;
    L.is_max = (p.mode == MXS_MODE_MAX);
    const double sign = L.is_max ? -1.0 : 1.0;  // max-sum == min-sum on negated costs
    const int nV = g.n_vars, nF = g.n_factors, nE = g.n_edges;

    // ---- classify and order variables ---------------------------------------
    std::vector<int> vsort(nV);
    for (int v = 0; v < nV; ++v) {
        const int deg = g.var_rowptr[v + 1] - g.var_rowptr[v];
        const int D = g.dom_size[v];
        const bool own = !g.var_owned || g.var_owned[v];
        int kind, sub;
        if (!own) { kind = 90; sub = 0; }                       // ghost: never swept
        else if (deg == 0) { kind = 80; sub = 0; }              // isolated: cycle 0 only
        else if (!L.opt.no_specialise && D >= 2 && D <= MAX_REG_D && deg <= MAX_PACK_DEG) {
            kind = K_V_PACK; sub = D;
        } else if (!L.opt.no_specialise && L.opt.pack8 && D > MAX_REG_D && D <= MAX_PACK8_D && deg <= MAX_PACK_DEG) {
            kind = K_V_PACK8;  // lane per edge on 8-element records, D at run time
            sub = 0;
        } else if (!L.opt.no_specialise && D <= 256 && (int64_t)deg * D <= 1024 && deg <= 256 &&
                   !(L.opt.hub && D <= MAX_PACK8_D && deg > MAX_PACK_DEG)) {  // (deg > 64 on a packed-class domain: a hub, below)
            kind = K_V_WIDE;  // workgroup per run of variables of one D, messages staged in LDS
            sub = 0;
        } else if (!L.opt.no_specialise && L.opt.hub) {
            kind = K_V_HUB;   // a wave per 64 outgoing edges, a lane per edge (round 6; before: a thread per variable)
            sub = 0;
        } else { kind = K_V_GEN; sub = 0; }
        // sort key: class, then degree (the packed class needs equal degrees side
        // by side; bit3 of layout_flags keeps the caller's order elsewhere).  The wide class
        // is ordered by domain size instead: its workgroups take runs of ONE D (WideBlock).
        const bool by_deg = L.opt.sort_by_degree || kind == K_V_PACK || kind == K_V_PACK8;
        // (wide class: by domain size, then by degree in steps of four -- the chains of a wave then
        // take the same path through wide_sum_cost, kernels.h)
        vsort[v] = (kind * 1024 + sub) * 4096 +
                   (kind == K_V_WIDE ? D * 8 + std::min((deg + 3) / 4, 7) : by_deg ? std::min(deg, 4095) : 0);
    }
    L.var_i2e.resize(nV);
    std::iota(L.var_i2e.begin(), L.var_i2e.end(), 0);
    std::stable_sort(L.var_i2e.begin(), L.var_i2e.end(),
                     [&](int a, int b) { return vsort[a] < vsort[b]; });
    L.var_e2i.resize(nV);
    for (int vi = 0; vi < nV; ++vi) L.var_e2i[L.var_i2e[vi]] = vi;

    // ---- classify factors ------------------------------------------------
    std::vector<FKey> fkey(nF);
    // the narrow storage type factor f takes in the small-domain kernel (small_box.h), TAB_FULL = it does not qualify.  All
    // qualifying factors of one arity share ONE type -- the widest any of them needs (their tables are a few hundred bytes: an
    // int8 table stored as int16 costs nothing next to a launch of its own, 8 us of a 67-us cycle on secp_100k).
    auto small_own_type = [&](int f, int e0, int ar) {
        if (!L.opt.nary || !L.opt.small || !L.opt.compact_tables || ar < 3 || ar > 5) return (int)TAB_FULL;
        for (int i = 0; i < ar; ++i)
            if (g.dom_size[g.edge_var[e0 + i]] > SMALL_P) return (int)TAB_FULL;
        return narrowest_tab_type(g.tables + g.table_off[f], g.table_off[f + 1] - g.table_off[f], L.opt.word);
    };
    std::vector<int8_t> small_own(nF, (int8_t)TAB_FULL);
    int small_widest[6] = {TAB_I8, TAB_I8, TAB_I8, TAB_I8, TAB_I8, TAB_I8};  // (TAB_I8 > TAB_I16 > TAB_F32: smaller = wider)
    if (!L.opt.no_specialise)
        for (int f = 0; f < nF; ++f) {
            const int e0 = g.factor_rowptr[f], ar = g.factor_rowptr[f + 1] - e0;
            small_own[f] = (int8_t)small_own_type(f, e0, ar);
            if (small_own[f] != TAB_FULL) small_widest[ar] = std::min(small_widest[ar], (int)small_own[f]);
        }
    auto small_type = [&](int f, int, int ar) { return small_own[f] == TAB_FULL ? (int)TAB_FULL : small_widest[ar]; };
    for (int f = 0; f < nF; ++f) {
        const int e0 = g.factor_rowptr[f], ar = g.factor_rowptr[f + 1] - e0;
        const int D0 = g.dom_size[g.edge_var[e0]];
        FKey k{K_F_GEN, 0};
        if (!L.opt.no_specialise) {
            if (ar == 1 && D0 >= 2 && D0 <= MAX_REG_D) k = FKey{K_F_UNARY, D0};
            else if (ar == 2 && D0 >= 2 && D0 <= MAX_REG_D && g.dom_size[g.edge_var[e0 + 1]] == D0)
                k = FKey{K_F_BIN, D0};
            else if (small_type(f, e0, ar) != TAB_FULL) {
                // arity 3..5, every domain <= SMALL_P, a narrow storage type: a lane group per factor (small_box.h)
                k = FKey{K_F_NARY, nary_group_code(SMALL_BASE, ar, 0, SMALL_WAVES) * 4 + small_type(f, e0, ar)};
            } else if (L.opt.nary && ar >= 2 && ar <= NARY_MULTI_MAX_ARITY) {
                // workgroup-per-factor kernel: 64 <= R <= 1024 (R = product of the
                // dimensions after the first), staged messages fit its LDS arrays
                int64_t R = 1, sumd = 0;
                for (int i = 0; i < ar; ++i) {
                    const int Di = g.dom_size[g.edge_var[e0 + i]];
                    sumd += Di;
                    if (i) R *= Di;
                }
                // one launch group per (arity, R / BLOCK rounded up): compile-time loop bounds
                // (round 5: arity >= 3 also below 64 entries per value of the first variable -- one wave with idle lanes beats
                // the thread-per-edge scalar loops of the generic class by far -- as long as the table has 64 entries at all)
                if (L.opt.nary_multi && (R > 1024 || ar > 5) && R <= NARY_MULTI_MAX_R && sumd <= 1024 && (int64_t)D0 * R >= 64) {
                    // beyond one pass of the workgroup's lanes (arity 3 over more than 32 values, arity 4 over more than 10,
                    // arity 5 over more than 5) or arity 6: the full-width kernel in passes of 1 024 entries per value of
                    // the first variable (kernels.h, k_factor_nary<.., MULTI>)
                    // integer tables an int8 / int16 holds: a narrow row-major image (an eighth / a quarter of the bytes)
                    int t = TAB_FULL;
                    if (L.opt.compact_tables) {
                        const int cand = narrowest_tab_type(g.tables + g.table_off[f], g.table_off[f + 1] - g.table_off[f], L.opt.word);
                        if (cand == TAB_I8 || cand == TAB_I16) t = cand;
                    }
                    k = FKey{K_F_NARY, nary_group_code(0, ar, NARY_NJ_MULTI, BLOCK / 64) * 4 + t};
                } else if (ar <= 5 && (R >= 64 || (ar >= 3 && (int64_t)D0 * R >= 64)) && R <= 1024 && sumd <= 1024) {
                    const int nj = nary_classic_nj(R);
                    const int waves = nary_classic_waves(R);  // 1..4
                    // storage type of this factor's table (one launch group = one kernel
                    // instantiation per type): narrow + lane-packed when that is at most three
                    // quarters of the full-width bytes
                    int t = TAB_FULL, box = 0;
                    if (L.opt.compact_tables) {
                        const int64_t ne = g.table_off[f + 1] - g.table_off[f];
                        const int cand = narrowest_tab_type(g.tables + g.table_off[f], ne, L.opt.word);
                        if (cand != TAB_FULL) {
                            const int64_t packed = (int64_t)D0 * (waves * 64) * nary_slot_bytes(nj, tab_elem_bytes(cand));
                            if (4 * packed <= 3 * ne * L.opt.word) t = cand;
                            // integer tables of arity 3 whose dimensions a box shape divides: one wave per
                            // factor, every running minimum in registers (layout.h, nary_box.h)
                            if (L.opt.box && ar == 3 && (cand == TAB_I8 || cand == TAB_I16)) {
                                box = nary_box_shape(D0, g.dom_size[g.edge_var[e0 + 1]], g.dom_size[g.edge_var[e0 + 2]],
                                                     tab_elem_bytes(cand));
                                if (box) t = cand;
                            }
                        }
                    }
                    k = FKey{K_F_NARY, (box ? nary_group_code(box, ar, 0, BOX_WAVES) : nary_group_code(0, ar, nj, waves)) * 4 + t};
                }
            }
            // binary / unary factors none of the above takes (a domain of more than MAX_REG_D values, two different
            // domain sizes; fewer than 64 entries per value of the first variable), up to 64 x 64: a group of 4 / 16 /
            // 64 lanes per factor (layout.h Bin2Shape, bin_box.h) -- launch groups by (shape, storage type)
            if (k.kind == K_F_GEN && L.opt.bin2 && ar <= 2) {
                const int D1 = ar == 2 ? g.dom_size[g.edge_var[e0 + 1]] : 1;
                const int box = bin2_shape_for(D0, D1, ar == 1);
                if (box) {
                    int t = TAB_FULL;
                    if (L.opt.compact_tables)
                        t = narrowest_tab_type(g.tables + g.table_off[f], g.table_off[f + 1] - g.table_off[f], L.opt.word);
                    k = FKey{K_F_NARY, nary_group_code(box, ar, 0, BIN2_WAVES) * 4 + t};
                }
            }
        }
        // Sharded operation: a factor that reads a ghost variable's V->F message has to
        // wait for the halo exchange; such "cut" factors form classes of their own, swept in
        // a second launch (see engine.hip, step_compute).
        if (g.var_owned)
            for (int i = 0; i < ar; ++i)
                if (!g.var_owned[g.edge_var[e0 + i]]) k.cut = 1;
        // A cut binary factor is replicated on both shards that own one of its variables; each replica's message to
        // the OTHER shard's variable is never read by anybody (ghost variables are storage only): the replica computes
        // only the message to its own variable -- half the records moved per replicated factor.
        if (k.cut && k.kind == K_F_BIN && L.opt.half_cut) {
            const bool o0 = g.var_owned[g.edge_var[e0]] != 0, o1 = g.var_owned[g.edge_var[e0 + 1]] != 0;
            k.own = (o0 && !o1) ? 1 : ((!o0 && o1) ? 2 : 0);
        }
        fkey[f] = k;
    }
    // Lane-grid groups (bin_box.h) of one SHAPE that differ only in storage type: a group whose launch would be latency-bound
    // takes the next wider type a sibling group uses and rides in that launch -- every narrow type widens exactly.  The unary
    // factors of a SECP instance came as three launches (double 20.8 us with the variable class riding, float 4.8, int8 4.7 of a
    // 62-us cycle, profiles/r06_kernel_stats_secp_100k_f64_v4.csv) for tables of five entries.  The price is the wider image:
    // merged while it costs at most BIN2_MERGE_BYTES more (2 us of streaming); the large PEAV groups stay apart.
    if (L.opt.bin2 && L.opt.compact_tables && L.opt.merge_types) {
        struct Stat { int64_t entries[4] = {0, 0, 0, 0}; int to[4] = {0, 1, 2, 3}; };
        std::map<std::tuple<int, int, int>, Stat> stat;  // (group code, cut, own) -> table entries per storage type
        auto bin2_key = [&](int f, std::tuple<int, int, int>& key, int& t) {
            const FKey& k = fkey[f];
            if (k.kind != K_F_NARY || !is_bin2(nary_code_box(k.D >> 2))) return false;
            key = std::make_tuple(k.D >> 2, k.cut, k.own);
            t = k.D & 3;
            return true;
        };
        std::tuple<int, int, int> key;
        int t = 0;
        for (int f = 0; f < nF; ++f)
            if (bin2_key(f, key, t)) stat[key].entries[t] += g.table_off[f + 1] - g.table_off[f];
        auto bytes_of = [&](int tt) { return tt == TAB_FULL ? (int64_t)L.opt.word : (int64_t)tab_elem_bytes(tt); };
        bool any = false;
        for (auto& kv : stat) {
            Stat& st = kv.second;
            for (int n = TAB_I8; n > TAB_FULL; --n) {  // narrowest first: what moved up is weighed again with its new sibling
                if (!st.entries[n]) continue;
                int w = n - 1;
                while (w >= TAB_FULL && !st.entries[w]) --w;
                if (w < TAB_FULL) continue;
                if (st.entries[n] * (bytes_of(w) - bytes_of(n)) > BIN2_MERGE_BYTES) continue;
                st.entries[w] += st.entries[n];
                st.entries[n] = 0;
                for (int q = TAB_FULL; q <= TAB_I8; ++q)
                    if (st.to[q] == n) st.to[q] = w;
                any = true;
            }
        }
        if (any)
            for (int f = 0; f < nF; ++f)
                if (bin2_key(f, key, t)) fkey[f].D = (fkey[f].D & ~3) | stat[key].to[t];
    }
    L.factor_i2e.resize(nF);
    std::iota(L.factor_i2e.begin(), L.factor_i2e.end(), 0);
    // Inside a class the factors follow the internal order of their FIRST scope variable
    // (sort_factors): the V->F records of that variable are then gathered by neighbouring
    // lanes, and -- the F->V records of a binary class being split by scope position, see
    // below -- the variable side reads the position-0 records as one dense stream.  One of
    // the two gathers per edge end turns from a random 64-byte request into streaming.
    // Any order gives the same arithmetic (every message is computed on its own).
    std::vector<int64_t> floc(nF, 0);
    if (L.opt.sort_factors)
        for (int f = 0; f < nF; ++f)
            if (g.factor_rowptr[f + 1] > g.factor_rowptr[f]) floc[f] = L.var_e2i[g.edge_var[g.factor_rowptr[f]]];
    // Tiled order (measurement: profiles/r04_tiled_order_ab_v1.jsonl): the binary factors of a class by (bucket of the first
    // variable, bucket of the second, first variable) -- a bucket = a run of the internal variable order whose V->F records
    // are about tile_bytes.  The factors of a tile gather both their variables' messages from two L2-sized windows, and a
    // variable block finds the position-1 records it gathers in one chunk per tile of its bucket's column instead of anywhere
    // in the array: every random 12- / 24-byte gather of the sweep falls into a window the L2 holds.  The price: the
    // position-0 records of a variable are no longer one run -- its gathers of them are spread over its bucket's row of tiles.
    // Measured, that trade pays on random graphs when the records are 4-byte words (-7 % at 100k variables, -10 % at 1M) or the
    // instance is resident in the Infinity Cache (-3 % on coloring_100k f64), costs 8-11 % on an HBM-resident f64 instance
    // (coloring_1m) and does nothing where the caller's order is already local (Ising grid: +-1 %).  tile_bytes < 0 applies
    // exactly that rule; layout flags 131072 / 262144 force it on / off.
    if (L.opt.sort_factors && L.opt.tile_bytes != 0 && nV > 0) {
        const int64_t tile = L.opt.tile_bytes > 0 ? L.opt.tile_bytes : MXS_TILE_BYTES;
        const int64_t per_var = std::max<int64_t>(1, (int64_t)nE * 3 * L.opt.word / std::max(1, nV));  // ~ V->F bytes per variable
        const int64_t W = std::max<int64_t>(256, tile / per_var);
        const int64_t NB = (nV + W - 1) / W;
        bool on = NB > 1;
        if (on && L.opt.tile_bytes < 0) {
            // (a) is the caller's order local already?  (b) the cycle's bytes (the formula of section 8d, from the graph)
            int64_t n_bin = 0, n_near = 0, bytes = 0;
            for (int f = 0; f < nF; ++f) {
                const int e0 = g.factor_rowptr[f];
                int64_t cells = 1;
                for (int e = e0; e < g.factor_rowptr[f + 1]; ++e) cells *= g.dom_size[g.edge_var[e]];
                bytes += cells * L.opt.word;
                if (fkey[f].kind != K_F_BIN) continue;
                const int64_t b0 = L.var_e2i[g.edge_var[e0]] / W, b1 = L.var_e2i[g.edge_var[e0 + 1]] / W;
                ++n_bin;
                n_near += b0 == b1;  // (a random graph: 1 / NB of them)
            }
            for (int e = 0; e < nE; ++e) bytes += 6 * (int64_t)g.dom_size[g.edge_var[e]] * L.opt.word + 8;
            for (int v = 0; v < nV; ++v) bytes += (int64_t)g.dom_size[v] * L.opt.word + 8 + L.opt.word;
            on = n_bin > 0 && 2 * n_near < n_bin && (L.opt.word == 4 || bytes <= MXS_TILE_RESIDENT_BYTES);
        }
        if (on)
            for (int f = 0; f < nF; ++f) {
                const int e0 = g.factor_rowptr[f];
                if (fkey[f].kind != K_F_BIN) continue;
                const int64_t v0 = L.var_e2i[g.edge_var[e0]], v1 = L.var_e2i[g.edge_var[e0 + 1]];
                floc[f] = ((v0 / W) * NB + v1 / W) * (int64_t)nV + v0;
            }
        L.tiled = on;
    }
    std::stable_sort(L.factor_i2e.begin(), L.factor_i2e.end(), [&](int a, int b) {
        return fkey[a] < fkey[b] || (fkey[a] == fkey[b] && floc[a] < floc[b]);
    });

    L.factor_e2i.resize(nF);
    for (int fi = 0; fi < nF; ++fi) L.factor_e2i[L.factor_i2e[fi]] = fi;
    L.f_tab_base.assign(nF, 0);
    L.f_tab_stride.assign(nF, 1);
    L.f_ctab_off.assign(nF, -1);
    L.f_class.assign(nF, -1);
    L.f_ndesc.assign(nF, -1);
    L.f_tab_type.assign(nF, (uint8_t)TAB_FULL);

    // ---- internal edges, F2V array (factor-major) ------------------------------
    L.edge_i2e.resize(nE);
    L.edge_e2i.resize(nE);
    L.frowptr.assign(nF + 1, 0);
    L.f2v_off.resize(nE);
    L.v2f_off.assign(nE, 0);
    L.edge_dom.resize(nE);
    L.edge_half.resize(nE);
    L.edge_gen_factor.assign(nE, -1);
    L.edge_fcim.assign(nE, 0);
    L.edge_vcim.assign(nE, 0);
    {
        int ei = 0;
        int64_t off = 0;
        const int align = 32 / L.opt.word;  // every class starts on a 32-byte boundary
        for (int fi = 0; fi < nF;) {
            const FKey key = fkey[L.factor_i2e[fi]];
            int fj = fi;
            while (fj < nF && fkey[L.factor_i2e[fj]] == key) ++fj;
            off = (off + align - 1) / align * align;
            // A binary register class keeps its records split by scope position: record
            // (j, 0) at base + j*H, record (j, 1) at base1 + j*H -- two dense streams for
            // the factor side, and with sort_factors the first one is in variable order.
            const bool split = key.kind == K_F_BIN;
            const int Hc = split ? L.half(key.D) : 0;
            const int64_t half_len = split ? ((int64_t)(fj - fi) * Hc + align - 1) / align * align : 0;
            for (int f2 = fi; f2 < fj; ++f2) {
                const int f = L.factor_i2e[f2];
                L.frowptr[f2] = ei;
                for (int e = g.factor_rowptr[f]; e < g.factor_rowptr[f + 1]; ++e, ++ei) {
                    L.edge_i2e[ei] = e;
                    L.edge_e2i[e] = ei;
                    const int D = g.dom_size[g.edge_var[e]];
                    const int H = L.half(D);
                    L.edge_dom[ei] = D;
                    L.edge_half[ei] = H;
                    int64_t at = off;
                    if (split) {
                        at = off + (int64_t)(e - g.factor_rowptr[f]) * half_len + (int64_t)(f2 - fi) * Hc;
                    } else {
                        // every record on the alignment its readers assume (kernels.h Msg::ALIGN: 16 bytes when its length is a
                        // multiple of 16, else 8, else the word) -- behind a tight 3-element record a scope such as (D = 3, D = 7)
                        // would put the 8-element one on a 12- / 24-byte offset
                        const int bytes = H * L.opt.word;
                        const int al = (bytes % 16 == 0 ? 16 : bytes % 8 == 0 ? 8 : L.opt.word) / L.opt.word;
                        off = (off + al - 1) / al * al;
                        at = off;
                        off += H;
                    }
                    if (at > ((int64_t)1 << 31) - 8192) return "message buffer exceeds 2^31 elements";
                    L.f2v_off[ei] = (int32_t)at;
                }
            }
            if (split) off += 2 * half_len;
            if (off > ((int64_t)1 << 31) - 8192) return "message buffer exceeds 2^31 elements";
            fi = fj;
        }
        L.frowptr[nF] = ei;
        // an all-zero block nobody writes, gathered through the padding slots of the
        // packed variable classes (32-byte aligned)
        off = (off + align - 1) / align * align;
        L.null_f2v = off;
        off += std::max(L.half(MAX_REG_D), L.half(MAX_PACK8_D));
        L.f2v_elems = off;
    }

    L.vrowptr.assign(nV + 1, 0);
    L.vslot_edge.resize(nE);
    L.vslot_f2v.resize(nE);
    L.vslot_v2f.assign(nE, 0);
    L.vslot_cv.resize(nE);
    L.vdom.resize(nV);
    L.vhalf.resize(nV);
    L.vcost_off.resize(nV);
    L.init_idx.assign(nV, -1);
    L.owned.assign(nV, 1);
    L.edge_var_int.resize(nE);
    {
        std::vector<int64_t> ext_cost_off(nV + 1, 0);
        for (int v = 0; v < nV; ++v) ext_cost_off[v + 1] = ext_cost_off[v] + g.dom_size[v];
        L.var_cost.resize(ext_cost_off[nV]);
        L.eval_var_cost.resize(ext_cost_off[nV]);
        int k = 0;
        int64_t coff = 0;
        for (int vi = 0; vi < nV; ++vi) {
            const int v = L.var_i2e[vi];
            L.vrowptr[vi] = k;
            for (int kk = g.var_rowptr[v]; kk < g.var_rowptr[v + 1]; ++kk, ++k) {
                const int ei = L.edge_e2i[g.var_edges[kk]];
                L.vslot_edge[k] = ei;
                L.vslot_f2v[k] = L.f2v_off[ei];
                L.edge_var_int[ei] = vi;
            }
            L.vdom[vi] = g.dom_size[v];
            L.vhalf[vi] = L.half(g.dom_size[v]);
            L.vcost_off[vi] = coff;
            for (int d = 0; d < g.dom_size[v]; ++d) {
                L.var_cost[coff + d] = sign * g.var_cost[ext_cost_off[v] + d];
                L.eval_var_cost[coff + d] = (g.eval_var_cost ? g.eval_var_cost : g.var_cost)[ext_cost_off[v] + d];
            }
            coff += g.dom_size[v];
            if (g.init_idx) L.init_idx[vi] = g.init_idx[v];
            if (g.var_owned) L.owned[vi] = g.var_owned[v] ? 1 : 0;
        }
        L.vrowptr[nV] = k;
    }

    // ---- factor classes, tables ---------------------------------------------------
    L.eval_tab_off.assign(nF + 1, 0);
    for (int fi = 0; fi < nF; ++fi) {
        const int f = L.factor_i2e[fi];
        L.eval_tab_off[fi + 1] = L.eval_tab_off[fi] + (g.table_off[f + 1] - g.table_off[f]);
    }
    L.fowned.assign(nF, 1);
    if (g.factor_owned)
        for (int fi = 0; fi < nF; ++fi) L.fowned[fi] = g.factor_owned[L.factor_i2e[fi]] ? 1 : 0;
    L.eval_tables.resize(L.eval_tab_off[nF]);
    L.tables.resize(L.eval_tab_off[nF]);
    for (int fi = 0; fi < nF; ++fi) {
        const int f = L.factor_i2e[fi];
        std::copy(g.tables + g.table_off[f], g.tables + g.table_off[f + 1],
                  L.eval_tables.begin() + L.eval_tab_off[fi]);
    }
    // A shard sweeps a cycle in two launches: the second one waits for the halo exchange.  The
    // cut factor classes have to be in it; with factors_second the interior factor classes
    // join them, so that the first launch (the variables) fits the 2048 resident workgroup
    // slots in one generation and both launches are of similar size.
    const bool second = g.var_owned != nullptr && L.opt.factors_second;
    auto sweep_class = [&](int cls, int per_block, int cut = 0) {  // blocks are derived from blockIdx
        L.classes[cls].per_block = per_block;
        (cut ? L.sweep_order2 : L.sweep_order).push_back(cls);
    };
    for (int fi = 0; fi < nF;) {
        const FKey key = fkey[L.factor_i2e[fi]];
        int fj = fi;
        while (fj < nF && fkey[L.factor_i2e[fj]] == key) ++fj;
        const int n = fj - fi;
        ClassInfo ci{};
        ci.kind = key.kind;
        ci.D = key.kind == K_F_NARY ? 0 : key.D;
        ci.H = ci.D ? L.half(ci.D) : 0;
        ci.first = fi;
        ci.count = n;
        ci.edge_base = L.frowptr[fi];
        ci.f2v_base = nE ? L.f2v_off[L.frowptr[fi]] : 0;
        ci.f2v_base1 = key.kind == K_F_BIN ? L.f2v_off[L.frowptr[fi] + 1] : 0;  // record (0, 1)
        ci.own_pos = key.own;
        ci.tab_base = L.eval_tab_off[fi];
        const int cls = (int)L.classes.size();
        if (key.kind == K_F_UNARY || key.kind == K_F_BIN) {
            // entry-major (SoA) tables: entry k of the class's j-th factor at k*n+j
            const int entries = (key.kind == K_F_UNARY) ? key.D : key.D * key.D;
            for (int j = 0; j < n; ++j) {
                const double* src = L.eval_tables.data() + L.eval_tab_off[fi + j];
                for (int k = 0; k < entries; ++k)
                    L.tables[ci.tab_base + (int64_t)k * n + j] = sign * src[k];
                L.f_tab_base[fi + j] = ci.tab_base + j;
                L.f_tab_stride[fi + j] = n;
            }
            if (ci.H > ci.D)  // counters ride in the records' padding (kernels.h, Msg::CNT_IN_MSG)
                for (int e = L.frowptr[fi]; e < L.frowptr[fj]; ++e) L.edge_fcim[e] = 1;
            // compact storage: the narrowest type every entry of the class fits exactly
            if (L.opt.compact_tables) {
                const int64_t lo = L.eval_tab_off[fi], hi = L.eval_tab_off[fj];
                const int t = narrowest_tab_type(L.eval_tables.data() + lo, hi - lo, L.opt.word);
                if (t != TAB_FULL) {
                    ci.tab_type = t;
                    ci.ctab_rec = tab_record_bytes(entries, tab_elem_bytes(t));
                    ci.ctab_base = (int64_t)((L.ctables.size() + 255) / 256 * 256);
                    L.ctables.resize((size_t)(ci.ctab_base + (int64_t)n * ci.ctab_rec), 0);
                    for (int j = 0; j < n; ++j) {
                        L.f_tab_type[fi + j] = (uint8_t)t;
                        L.f_ctab_off[fi + j] = ci.ctab_base + (int64_t)j * ci.ctab_rec;
                        encode_tab_record(L.eval_tables.data() + L.eval_tab_off[fi + j], entries, t,
                                          L.ctables.data() + L.f_ctab_off[fi + j]);
                    }
                }
            }
            for (int j = 0; j < n; ++j) L.f_class[fi + j] = cls;
            L.classes.push_back(ci);
            sweep_class(cls, BLOCK, key.cut || second);
        } else {
            const int gen_base = (int)L.fgen.size();
            for (int j = 0; j < n; ++j) {
                const int f2 = fi + j;
                FactorGen fg{L.frowptr[f2], L.frowptr[f2 + 1] - L.frowptr[f2], L.eval_tab_off[f2]};
                for (int e = L.frowptr[f2]; e < L.frowptr[f2 + 1]; ++e)
                    L.edge_gen_factor[e] = gen_base + j;
                L.fgen.push_back(fg);
                L.f_tab_base[f2] = L.eval_tab_off[f2];
                L.f_tab_stride[f2] = 1;
                for (int64_t k = L.eval_tab_off[f2]; k < L.eval_tab_off[f2 + 1]; ++k)
                    L.tables[k] = sign * L.eval_tables[k];
            }
            if (key.kind == K_F_GEN) {
                ci.count = L.frowptr[fj] - L.frowptr[fi];  // thread per edge
                L.classes.push_back(ci);
                sweep_class(cls, BLOCK, key.cut || second);
            } else {  // K_F_NARY: one workgroup per factor, one launch per (arity, nj) group
                const int code = key.D / 4, t = key.D % 4;
                NaryLaunch nl{(code / 256) % 16, (code / 16) % 16, (code % 16) * 64, (int32_t)L.ndesc.size(), n, key.cut, t,
                              code / 4096};
                for (int j = 0; j < n; ++j) {
                    const int f2 = fi + j;
                    NaryDesc d{};
                    d.tab_off = L.eval_tab_off[f2];
                    d.edge_base = L.frowptr[f2];
                    d.arity = L.frowptr[f2 + 1] - L.frowptr[f2];
                    for (int i = 0; i < NARY_DESC_ARITY; ++i) {
                        const bool in = i < d.arity;
                        d.dom[i] = in ? L.edge_dom[d.edge_base + i] : 1;
                        d.f2v_off[i] = in ? L.f2v_off[d.edge_base + i] : 0;
                        d.v2f_off[i] = 0;  // filled in once the variable side is laid out
                        d.magic[i] = d.dom[i] > 1 ? (uint32_t)((((uint64_t)1 << 32) + d.dom[i] - 1) / d.dom[i]) : 0u;
                    }
                    // compact storage of THIS factor's table (row-major, like the full-width image)
                    if (t != TAB_FULL || is_bin2(nl.box) || is_small(nl.box)) {  // narrow image: lane-packed slots or a box record per lane (layout.h);
                        // a lane-grid group reads its image at every width
                        const NaryPlace pl = nary_place(nl, d, L.opt.word);
                        const int64_t ne = L.eval_tab_off[f2 + 1] - L.eval_tab_off[f2];
                        const int64_t al = is_bin2(nl.box) ? 16 : 256;
                        const int64_t at = (int64_t)((L.ctables.size() + al - 1) / al * al);
                        L.ctables.resize((size_t)(at + nary_place_bytes(pl, d.dom[0])), 0);
                        const double* src = L.eval_tables.data() + L.eval_tab_off[f2];
                        for (int64_t k = 0; k < ne; ++k)
                            encode_tab_entry(src[k], t, L.opt.word, L.ctables.data() + at + nary_place_pos(pl, k));
                        L.f_tab_type[f2] = (uint8_t)t;
                        L.f_ctab_off[f2] = at;
                        d.tab_off = at;
                    }
                    L.f_ndesc[f2] = (int32_t)L.ndesc.size();
                    L.ndesc.push_back(d);
                }
                L.nary_launches.push_back(nl);
            }
        }
        fi = fj;
    }

    // ---- variable classes, V2F array (variable-major) -----------------------------
    // Send counters: positions [0, nE) follow the CSR slot order (generic class);
    // the packed classes keep theirs in lane order behind.
    L.n_cv = nE;
    for (int k = 0; k < nE; ++k) L.vslot_cv[k] = k;
    int64_t voff = 0;
    for (int vi = 0; vi < nV;) {
        const int v0 = L.var_i2e[vi];
        const int key = vsort[v0] / 4096;
        int vj = vi;
        while (vj < nV && vsort[L.var_i2e[vj]] / 4096 == key) ++vj;
        const int kind = key / 1024, sub = key % 1024;
        ClassInfo ci{};
        ci.first = vi;
        ci.count = vj - vi;
        ci.cost_base = L.vcost_off[vi];
        bool swept = true;
        {
            const int64_t align = 32 / L.opt.word;  // every class starts on a 32-byte boundary
            voff = (voff + align - 1) / align * align;
        }
        if (kind == K_V_PACK) {
            ci.kind = K_V_PACK;
            ci.D = sub;
            ci.H = L.half(ci.D);
        } else if (kind == K_V_PACK8) {
            ci.kind = K_V_PACK8;
            ci.D = MAX_PACK8_D;  // the record length; a variable's own domain size is read from vdom
            ci.H = L.half(ci.D);
            ci.uni_D = L.vdom[vi];
            for (int w = vi; w < vj; ++w)
                if (L.vdom[w] != ci.uni_D) ci.uni_D = 0;
        } else if (kind == K_V_GEN) {
            ci.kind = K_V_GEN;
        } else if (kind == K_V_HUB) {
            // its workgroups: HUB_EDGES outgoing edges of one variable each, the belief lane behind the last edge; the longest
            // chains (highest degree x domain) first -- they are the first workgroups of the sweep's grid
            ci.kind = K_V_HUB;
            std::vector<int> order(vj - vi);
            std::iota(order.begin(), order.end(), vi);
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
                return (int64_t)(L.vrowptr[x + 1] - L.vrowptr[x]) * L.vdom[x] > (int64_t)(L.vrowptr[y + 1] - L.vrowptr[y]) * L.vdom[y];
            });
            ci.first = (int32_t)L.hub_blocks.size();
            for (int w : order) {
                const int deg = L.vrowptr[w + 1] - L.vrowptr[w];
                const uint32_t row = 8u + (uint32_t)(deg < HUB_TILE ? ((deg + 7) & ~7) : HUB_TILE);
                for (int ko0 = 0; ko0 <= deg; ko0 += HUB_EDGES)
                    L.hub_blocks.push_back(HubBlock{w, ko0, L.vdom[w], deg, L.vrowptr[w],
                                                    (uint32_t)((((uint64_t)1 << 32) + row - 1) / row), L.vcost_off[w]});
            }
            ci.count = (int32_t)L.hub_blocks.size() - ci.first;
        } else if (kind == K_V_WIDE) {
            ci.kind = K_V_WIDE;
            // its workgroups: runs of one D that fit the kernel's LDS arrays
            for (int w = vi; w < vj;) {
                const int D = L.vdom[w];
                WideBlock wb{w, 0, D, L.vrowptr[w], 0, (uint32_t)((((uint64_t)1 << 32) + D - 1) / D), L.vcost_off[w]};
                while (w < vj && L.vdom[w] == D) {
                    const int deg = L.vrowptr[w + 1] - L.vrowptr[w];
                    if (wb.n_vars > 0 && ((int64_t)(wb.n_slots + deg) * D > WIDE_CAPB || wb.n_slots + deg > WIDE_MAX_SLOTS ||
                                          wb.n_vars + 1 > WIDE_MAX_VARS || (int64_t)(wb.n_vars + 1) * D > WIDE_MAX_COSTS))
                        break;
                    wb.n_vars += 1;
                    wb.n_slots += deg;
                    ++w;
                }
                L.wide_blocks.push_back(wb);
            }
        } else if (kind == 80) {
            ci.kind = K_V_GEN;
            ci.start_only = 1;
        } else {  // ghosts: storage only
            swept = false;
        }
        if (ci.kind == K_V_PACK || ci.kind == K_V_PACK8) {
            // One lane per edge.  The variables are sorted by degree; a wave holds
            // floor(64/deg) variables of one degree side by side (lane = var*deg + k)
            // and its tail lanes are padding.  Per lane: F2V offset of its edge (-1 =
            // padding), its variable, k; its own V->F message sits at v2f_base + lane*H.
            ci.ell_base = (int64_t)L.vell.size();
            ci.cv_base = L.n_cv;
            ci.v2f_base = voff;
            int64_t lanes = 0;
            for (int w = vi; w < vj;) {
                const int deg = L.vrowptr[w + 1] - L.vrowptr[w];
                int w2 = w;
                while (w2 < vj && L.vrowptr[w2 + 1] - L.vrowptr[w2] == deg) ++w2;
                const int per_wave = 64 / deg;
                for (int x = w; x < w2; x += per_wave) {  // one wave
                    const int nv = std::min(per_wave, w2 - x);
                    L.vwave.push_back(WaveMeta{x, (int32_t)((uint32_t)deg | ((uint32_t)nv << 8) |
                                                          ((uint32_t)((32768 + deg - 1) / deg) << 16))});
                    for (int lane = 0; lane < 64; ++lane) {
                        const int var = lane / deg, k = lane % deg;
                        if (var < nv) {
                            const int ks = L.vrowptr[x + var] + k;
                            L.vell.push_back(L.vslot_f2v[ks]);
                            L.vslot_cv[ks] = ci.cv_base + lanes;
                            L.vslot_v2f[ks] = (int32_t)(ci.v2f_base + lanes * ci.H);
                            if (ci.H > ci.D) L.edge_vcim[L.vslot_edge[ks]] = 1;
                        } else {
                            L.vell.push_back(-1);
                        }
                        ++lanes;
                    }
                }
                w = w2;
            }
            ci.count = (int32_t)lanes;
            L.n_cv += lanes;
            voff += lanes * ci.H;
        } else {  // generic, isolated and ghost variables: CSR slots
            for (int w = vi; w < vj; ++w)
                for (int k = L.vrowptr[w]; k < L.vrowptr[w + 1]; ++k) {
                    L.vslot_v2f[k] = (int32_t)voff;
                    voff += L.vhalf[w];
                }
        }
        if (voff > ((int64_t)1 << 31) - 8192) return "message buffer exceeds 2^31 elements";
        if (swept) {
            const int cls = (int)L.classes.size();
            L.classes.push_back(ci);
            if (ci.kind == K_V_WIDE) {
                L.classes[cls].per_block = 1;  // (its grid is wide_blocks.size())
                L.wide_classes.push_back(cls);
            } else if (ci.kind == K_V_PACK8) {
                L.classes[cls].per_block = BLOCK;  // (its own launch: count / BLOCK workgroups)
                L.pack8_classes.push_back(cls);
            } else {
                sweep_class(cls, ci.kind == K_V_HUB ? 1 : BLOCK);
            }
        }
        vi = vj;
    }
    L.v2f_elems = voff;
    for (int k = 0; k < nE; ++k) L.v2f_off[L.vslot_edge[k]] = L.vslot_v2f[k];
    for (NaryDesc& d : L.ndesc)
        for (int i = 0; i < (d.arity & 255); ++i) d.v2f_off[i] = L.v2f_off[d.edge_base + i];

    // Launch order of the sweep classes: the longest per-thread chains first
    // (generic classes, then the gathering variable classes, then the streaming
    // factor classes), so that the tail of the launch is made of short blocks.
    {
        auto prio = [&](int c) {
            switch (L.classes[c].kind) {
                case K_V_HUB: return -1;  // serial chains of thousands of additions: the first workgroups of the grid
                case K_V_GEN: return 0;
                case K_F_GEN: return 1;
                case K_V_PACK: return 2;
                default: return 4;
            }
        };
        std::stable_sort(L.sweep_order.begin(), L.sweep_order.end(),
                         [&](int x, int y) { return prio(x) < prio(y); });
        if (p.layout_flags & (32 | 64)) {  // timing experiments only (results are wrong):
            std::vector<int32_t> keep;     // bit5 = factor side only, bit6 = variable side only
            for (int c : L.sweep_order) {
                const int k = L.classes[c].kind;
                const bool is_var = k == K_V_GEN || k == K_V_PACK || k == K_V_HUB;
                if (((p.layout_flags & 32) && !is_var) || ((p.layout_flags & 64) && is_var)) keep.push_back(c);
            }
            L.sweep_order = keep;
            if (p.layout_flags & 64) L.sweep_order2.clear();
        }
        if ((int)L.sweep_order.size() > MAX_CLASSES || (int)L.sweep_order2.size() > MAX_CLASSES)
            return "too many kernel classes";
        int nb = 0;
        for (int c : L.sweep_order) {
            ClassInfo& ci = L.classes[c];
            ci.block_base = nb;
            nb += (ci.count + ci.per_block - 1) / ci.per_block;
        }
        L.n_blocks_sweep = nb;
        nb = 0;
        for (int c : L.sweep_order2) {  // the cut factor classes: their own launch
            ClassInfo& ci = L.classes[c];
            ci.block_base = nb;
            nb += (ci.count + ci.per_block - 1) / ci.per_block;
        }
        L.n_blocks_sweep2 = nb;
        for (int c : L.sweep_order2) L.classes[c].wait_halo = 1;
        // the same classes as ONE grid: phase-1 classes, then the cut factor classes
        if (!L.sweep_order2.empty() && L.sweep_order.size() + L.sweep_order2.size() <= (size_t)MAX_CLASSES) {
            for (int c : L.sweep_order) L.fused_block_base.push_back(L.classes[c].block_base);
            for (int c : L.sweep_order2) L.fused_block_base.push_back(L.n_blocks_sweep + L.classes[c].block_base);
            L.n_blocks_fused = L.n_blocks_sweep + L.n_blocks_sweep2;
        }
        for (int c : L.sweep_order)
            if (!L.classes[c].start_only) L.sweep_regular = true;
        // ---- block schedule of launch 0 (see Layout::sched) ------------------------------
        if (L.opt.schedule && L.n_blocks_sweep >= 2 && !(p.layout_flags & (32 | 64))) {
            struct Blk { double key; int prio; uint32_t code; };
            std::vector<Blk> blks;
            std::vector<uint32_t> hub_blks;
            std::vector<Blk> long_blks;           // packed-class blocks of degree >= LONG_PACK_DEG
            constexpr int LONG_PACK_DEG = 24;
            blks.reserve(L.n_blocks_sweep);
            auto first_var_of_factor = [&](int fi) { return (double)L.edge_var_int[L.frowptr[fi]]; };
            for (size_t slot = 0; slot < L.sweep_order.size(); ++slot) {
                const ClassInfo& ci = L.classes[L.sweep_order[slot]];
                const int nbc = (ci.count + ci.per_block - 1) / ci.per_block;
                if (nbc >= (1 << 24)) { blks.clear(); break; }
                for (int j = 0; j < nbc; ++j) {
                    const int64_t i0 = (int64_t)j * ci.per_block;
                    const int64_t i1 = std::min<int64_t>(ci.count, i0 + ci.per_block) - 1;
                    double k0 = 0, k1 = 0;
                    int prio = 1;
                    if (ci.kind == K_V_HUB) {  // not part of the locality order: the first workgroups of the grid (below)
                        hub_blks.push_back((uint32_t)(slot << 24) | (uint32_t)j);
                        continue;
                    }
                    switch (ci.kind) {
                        case K_V_PACK: {  // lanes -> the wave's first variable
                            const WaveMeta& w0 = L.vwave[(ci.ell_base + i0) >> 6];
                            const WaveMeta& w1 = L.vwave[(ci.ell_base + i1) >> 6];
                            // A wave of high-degree variables walks chains of 3 * deg dependent steps (16 us at degree 64 where a
                            // block of degree-4 variables takes 5): such blocks -- the class is sorted by degree, they are its last --
                            // would start last in the locality order and be the launch's tail (scale-free graphs: 45 us of a
                            // 37-us launch's timeline).  They go to the front, behind the hub workgroups, longest first.
                            if ((int)((uint32_t)w1.deg_nv & 255u) >= LONG_PACK_DEG) {
                                long_blks.push_back(Blk{-(double)((uint32_t)w1.deg_nv & 255u), 0, (uint32_t)(slot << 24) | (uint32_t)j});
                                continue;
                            }
                            k0 = w0.first_var;
                            k1 = w1.first_var + (int)(((uint32_t)w1.deg_nv >> 8) & 255u) - 1;
                            prio = 0;
                            break;
                        }
                        case K_V_GEN:
                            k0 = (double)(ci.first + i0);
                            k1 = (double)(ci.first + i1);
                            prio = 0;
                            break;
                        case K_F_UNARY:
                        case K_F_BIN:
                            k0 = first_var_of_factor((int)(ci.first + i0));
                            k1 = first_var_of_factor((int)(ci.first + i1));
                            break;
                        default:  // K_F_GEN: thread per edge
                            k0 = (double)L.edge_var_int[L.fgen[L.edge_gen_factor[ci.edge_base + i0]].edge_base];
                            k1 = (double)L.edge_var_int[L.fgen[L.edge_gen_factor[ci.edge_base + i1]].edge_base];
                            break;
                    }
                    blks.push_back(Blk{0.5 * (k0 + k1), prio, (uint32_t)(slot << 24) | (uint32_t)j});
                }
            }
            std::stable_sort(long_blks.begin(), long_blks.end(), [](const Blk& a, const Blk& b) { return a.key < b.key; });
            for (const Blk& b : long_blks) hub_blks.push_back(b.code);
            if ((int)(blks.size() + hub_blks.size()) == L.n_blocks_sweep) {
                std::stable_sort(blks.begin(), blks.end(), [](const Blk& a, const Blk& b) {
                    return a.key != b.key ? a.key < b.key : a.prio < b.prio;
                });
                // the hub workgroups (K_V_HUB: the longest blocks by far) are the first of the grid, dealt round the XCDs;
                // behind them XCD x runs the workgroups b = x, x + 8, x + 16, ...: hand it the x-th contiguous piece of
                // the order
                L.sched.assign(L.n_blocks_sweep, 0);
                const int64_t nh = (int64_t)hub_blks.size();
                for (int64_t b = 0; b < nh; ++b) L.sched[b] = hub_blks[b];
                int64_t m = 0;
                for (int x = 0; x < NUM_XCD; ++x)
                    for (int64_t b = nh + ((x - nh) % NUM_XCD + NUM_XCD) % NUM_XCD; b < L.n_blocks_sweep; b += NUM_XCD)
                        L.sched[b] = blks[m++].code;
            }
        }
        // one compile-time D for every register / wave class -> leaner kernel
        int dsel = -1;
        std::vector<int32_t> both = L.sweep_order;
        both.insert(both.end(), L.sweep_order2.begin(), L.sweep_order2.end());
        for (int c : both) {
            const ClassInfo& ci = L.classes[c];
            if (ci.D == 0) continue;
            if (dsel == -1) dsel = ci.D;
            else if (dsel != ci.D) dsel = 0;
        }
        L.dsel = dsel < 0 ? 0 : dsel;
    }

    // ---- algorithmic bytes per cycle --------------------
    {
        const int64_t w = L.opt.word;
        int64_t b = 0;
        for (int e = 0; e < nE; ++e) b += 6 * (int64_t)L.edge_dom[e] * w + 8;
        b += L.eval_tab_off[nF] * w;
        for (int v = 0; v < nV; ++v) b += (int64_t)g.dom_size[v] * w + 8 + w;
        L.algorithmic_bytes = b;
    }
    return "";
}

}  // namespace mxs
