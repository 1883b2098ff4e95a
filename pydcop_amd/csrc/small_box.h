// small_box.h -- factor_costs_for_var (pydcop/algorithms/maxsum.py:382-447) for factors of ARITY 3..5 whose
// every domain has at most SMALL_P = 5 values and whose table a narrow type holds exactly: what the reference's
// `generate secp` emits (pydcop/commands/generators/secp.py: every variable `range(0, 5)` (:138); model
// constraints over 2..max_model_size lights + the model variable with values in {0, 10000} (:201-236); rules
// `10 * (abs(v - t) + ..)` over up to three variables (:239-299)).  Tables of 125 / 625 / 3 125 entries: too small
// for the workgroup-per-factor kernel (kernels.h k_factor_nary*: one wave with 25 of 64 lanes busy at arity 3,
// LDS atomics on 5 addresses per dimension, 22 f64 operations per entry at arity 5 -- 100 us per cycle on
// secp_100k, 313 with arity-5 models: profiles/r06_secp_bench_v2.txt), too many values for the register classes.
//
// G lanes work on one factor (8 at arity 3, 32 at arity 4 and 5: 8 resp. 2 factors share a wave).  The LEADING
// L = 1 (arity 3) or 2 digits of an entry's index pick the lane -- 5 resp. 25 of the G lanes are in use --, the
// remaining U = 2 or 3 TRAILING digits run through a fully unrolled loop: the lane's 25 (125) entries are ONE
// record of the image, in registers before anything else, every digit of every entry is a compile-time
// constant, and so the running minima towards the trailing variables are registers (U x 5 of them), the ones
// towards the leading variables one scalar each.  Nothing crosses lanes before the last entry: then every lane's
// partial minima go through LDS once, and one lane per outgoing message ELEMENT merges the partials of its element
// and runs apply_damping + the send rule (maxsum.py:346-377).
//
// Arithmetic: the reference's expression, op for op -- for output i, sum_cost = ((0 + m_a[d_a]) + m_b[d_b]) + ..
// over the OTHER variables in dimensions order, then `f_val + sum_cost` (maxsum.py:425-438).  The sums are
// written out per (entry, output); common prefixes (everything that does not depend on the last digits) are the
// same expression in many entries and are computed once.  Digits past a domain (a variable of fewer than 5
// values) are staged as +inf, the identity of the min-plus semiring; their (zero-filled) entries never win.
// Minima are exact and order-independent: bit for bit what k_factor_nary / factor_generic compute.
#pragma once
#include "kernels.h"

namespace mxs {

// Launch of one small-domain group (engine.hip, launch_nary).  Returns false when no instantiation exists.
// Defined in small_box.hip (a translation unit of its own).
template <typename T>
bool launch_factor_small(const NaryLaunch& nl, const SweepArgs<T>& a, const NaryDesc* d, hipStream_t stream);

#ifdef MXS_SMALL_IMPL

template <typename T, typename TT>
__device__ __forceinline__ T small_entry(const uint32_t* w, int e) {
    if constexpr (sizeof(TT) == 1) return (T)(int)(int8_t)(uint8_t)(w[e >> 2] >> (8 * (e & 3)));
    else if constexpr (sizeof(TT) == 2) return (T)(int)(int16_t)(uint16_t)(w[e >> 1] >> (16 * (e & 1)));
    else {
        float f;
        __builtin_memcpy(&f, &w[e], 4);
        return (T)f;
    }
}

template <typename T, typename TT, bool NEG, int A>
__global__ void __launch_bounds__(SMALL_WAVES * 64) k_factor_small(SweepArgs<T> a, const NaryDesc* descs, int n_factors) {
    constexpr int P = SMALL_P, L = small_lead(A), U = A - L, G = small_group(A), FPW = 64 / G;
    constexpr int GA = small_pow(L), NE = small_pow(U), NP = A * P;   // lanes in use, entries per lane, message element slots
    constexpr int EPL = (NP + G - 1) / G;                             // element slots per lane
    constexpr int RW = small_rec_bytes(A, (int)sizeof(TT)) / 4;       // dwords of a lane's record
    __shared__ T s_in[SMALL_WAVES][FPW][NP + 1];      // incoming V->F messages, +inf past a domain; [NP] = +inf (lanes not in use)
    __shared__ T s_lead[SMALL_WAVES][FPW][L][G];      // minima towards the leading variables, per lane
    // minima towards the trailing variables, per lane: row = message element, column = lane; G + 1 columns, so that the element
    // lanes -- one row each -- read different banks
    __shared__ T s_trail[SMALL_WAVES][FPW][U * P][G + 1];
    const int wv = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int fw = lane / G, g = lane % G;
    const int f_wave = ((int)blockIdx.x * SMALL_WAVES + wv) * FPW;
    if (f_wave >= n_factors) return;                 // (wave-uniform; only wave-level barriers below)
    const bool active = f_wave + fw < n_factors;     // the last wave: idle groups shadow the last factor, store nothing
    const NaryDesc* fd = descs + (active ? f_wave + fw : n_factors - 1);
    const int e0 = fd->edge_base;
    // one lane per message ELEMENT slot p = g + G * k of the padded scope (dimension p / P, value p % P): request the incoming
    // element and what the epilogue needs of the outgoing one (the message sent last, its send counter) ...
    int el_i[EPL], el_d[EPL], el_cnt[EPL], el_fo[EPL];
    T el_in[EPL], el_prev[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int p = g + G * k;
        const int i = p / P, d = p - i * P;
        const bool valid = p < NP && d < fd->dom[p < NP ? i : 0];
        el_i[k] = valid ? i : -1;
        el_d[k] = d;
        el_in[k] = el_prev[k] = (T)0;
        el_cnt[k] = el_fo[k] = 0;
        if (valid) {
            el_in[k] = a.v2f_old[fd->v2f_off[i] + d];
            el_fo[k] = fd->f2v_off[i];
            if (!a.start) {
                el_prev[k] = a.f2v_old[el_fo[k] + d];
                el_cnt[k] = a.cF[e0 + i];
            }
        }
    }
    // ... then the lane's record (the loads return in order: the messages are staged while the table is on its way)
    uint32_t w[RW];
    {
        const uint32_t* rec = (const uint32_t*)__builtin_assume_aligned(a.ctables + fd->tab_off + (int64_t)(g < GA ? g : 0) * (RW * 4), 4);
#pragma unroll
        for (int x = 0; x < RW; ++x) w[x] = rec[x];
    }
    T* in = s_in[wv][fw];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int p = g + G * k;
        if (p < NP) in[p] = el_i[k] >= 0 ? el_in[k] : pos_inf<T>();
    }
    if (g == 0) in[NP] = pos_inf<T>();
    __builtin_amdgcn_wave_barrier();
    // the lane's leading digits and their messages; the trailing variables' messages, whole
    T ml[L], mt[U][P], bl[L], acc[U][P];
    if constexpr (L == 1) {
        ml[0] = in[g < GA ? g : NP];
    } else {
        const int x0 = g / P, x1 = g - x0 * P;
        ml[0] = in[g < GA ? x0 : NP];
        ml[1] = in[g < GA ? P + x1 : NP];
    }
#pragma unroll
    for (int i = 0; i < L; ++i) bl[i] = pos_inf<T>();
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int x = 0; x < P; ++x) {
            mt[u][x] = in[(L + u) * P + x];
            acc[u][x] = pos_inf<T>();
        }
    // every entry of the record: digits known at compile time
    static_for<NE>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        constexpr int xt[3] = {U == 3 ? e / (P * P) : (U == 2 ? e / P : 0), U == 3 ? (e / P) % P : e % P, e % P};  // (U = 2: [0], [1])
        const T v = small_entry<T, TT>(w, e);
        const T t = NEG ? -v : v;
        static_for<A>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            T s = (T)0;  // sum_cost over the others, in dimensions order (maxsum.py:425-438)
            static_for<A>([&](auto oc) __attribute__((always_inline)) {
                constexpr int o = decltype(oc)::value;
                if constexpr (o != i) {
                    if constexpr (o < L) s += ml[o];
                    else s += mt[o - L][xt[o - L]];
                }
            });
            const T cand = t + s;
            if constexpr (i < L) bl[i] = min2(bl[i], cand);
            else acc[i - L][xt[i - L]] = min2(acc[i - L][xt[i - L]], cand);
        });
    });
    // The trailing minima of the factor: every lane's U * P partials go through LDS once and the element lane of a value merges
    // the GA partials of its row.  (Until this commit: a DPP butterfly over the G lanes -- quad swaps, row half mirror, row
    // mirror, permlane16 swap.  A 64-bit minimum takes no DPP operand: three VALU instructions per value and step, 228 of the
    // 749 of the arity-4 kernel, which is bound by VALU issue.)  Lanes not in use store their +inf; nobody reads them.
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int x = 0; x < P; ++x) s_trail[wv][fw][u * P + x][g] = acc[u][x];
#pragma unroll
    for (int i = 0; i < L; ++i) s_lead[wv][fw][i][g] = bl[i];
    __builtin_amdgcn_wave_barrier();
    // element lanes: the minimum over the lanes that share the digit, apply_damping, approx_match
    T el_m[EPL];
    bool bad[A];
#pragma unroll
    for (int i = 0; i < A; ++i) bad[i] = false;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = el_i[k], d = el_d[k];
        T m = pos_inf<T>();
        if (i >= L) {
            const T* row = s_trail[wv][fw][(i - L) * P + d];
            T m4[4] = {pos_inf<T>(), pos_inf<T>(), pos_inf<T>(), pos_inf<T>()};  // (four chains: the minimum is order-independent)
#pragma unroll
            for (int j = 0; j < GA; ++j) m4[j & 3] = min2(m4[j & 3], row[j]);
            m = min2(min2(m4[0], m4[1]), min2(m4[2], m4[3]));
        } else if (i >= 0) {
            if constexpr (L == 1) {
                m = s_lead[wv][fw][0][d];
            } else {  // lane = x0 * P + x1: the lanes of one x0 are neighbours, those of one x1 P apart
                const T* run = i == 0 ? &s_lead[wv][fw][0][d * P] : &s_lead[wv][fw][1][d];
                const int step = i == 0 ? 1 : P;
#pragma unroll
                for (int j = 0; j < P; ++j) m = min2(m, run[j * step]);
            }
        }
        if (i >= 0 && !a.start) {
            const T p = el_prev[k];
            const int cnt = el_cnt[k];
            if (cnt > 0 && a.damp_f) m = a.damping * p + ((T)1 - a.damping) * m;
            const bool b = cnt > 0 && !comp_match(m, p, a.stability);
#pragma unroll
            for (int q = 0; q < A; ++q) bad[q] = bad[q] || (i == q && b);
        }
        el_m[k] = m;
    }
    // the elements of a message agree on "changed": one ballot per edge, read by the factor's lanes
    const unsigned long long gm = ((1ull << G) - 1ull) << (fw * G);
    bool nomatch[A];
#pragma unroll
    for (int q = 0; q < A; ++q) nomatch[q] = (__ballot(bad[q]) & gm) != 0ull;
    if (!active) return;
    // send / send again / stay silent (the receiver keeps the old message)
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = el_i[k], d = el_d[k];
        if (i < 0) continue;
        const int fo = el_fo[k], e = e0 + i;
        if (a.start) {  // only start_messages == all makes a non-unary factor send (maxsum.py:311-328)
            a.f2v_new[fo + d] = a.start_mode == MXS_START_ALL ? el_m[k] : (T)0;
            if (d == 0) a.cF[e] = 0;
            continue;
        }
        const int cnt = el_cnt[k];
        bool nm = false;
#pragma unroll
        for (int q = 0; q < A; ++q) nm = nm || (i == q && nomatch[q]);
        const bool match = cnt > 0 && !nm;
        int out = 1;
        T val = el_m[k];
        if (match) {
            if (cnt < SAME_COUNT) {
                out = cnt + 1;
            } else {
                out = cnt;
                val = el_prev[k];
            }
        }
        a.f2v_new[fo + d] = val;
        if (d == 0) a.cF[e] = (uint8_t)out;
    }
}

template <typename T, typename TT, int A>
inline void launch_small_one(const SweepArgs<T>& a, const NaryDesc* d, int count, hipStream_t stream) {
    constexpr int FPB = SMALL_WAVES * (64 / small_group(A));
    const dim3 grid((unsigned)((count + FPB - 1) / FPB)), block((unsigned)(SMALL_WAVES * 64));
    if (a.tab_neg) MXS_LAUNCH((k_factor_small<T, TT, true, A>), grid, block, 0, stream, a, d, count);
    else MXS_LAUNCH((k_factor_small<T, TT, false, A>), grid, block, 0, stream, a, d, count);
}

template <typename T, typename TT>
inline bool launch_small_arity(int arity, const SweepArgs<T>& a, const NaryDesc* d, int count, hipStream_t stream) {
    switch (arity) {
        case 3: launch_small_one<T, TT, 3>(a, d, count, stream); return true;
        case 4: launch_small_one<T, TT, 4>(a, d, count, stream); return true;
        case 5: launch_small_one<T, TT, 5>(a, d, count, stream); return true;
        default: return false;
    }
}

template <typename T>
bool launch_factor_small(const NaryLaunch& nl, const SweepArgs<T>& a, const NaryDesc* d, hipStream_t stream) {
    switch (nl.tab_type) {
        case TAB_I8: return launch_small_arity<T, int8_t>(nl.arity, a, d, nl.count, stream);
        case TAB_I16: return launch_small_arity<T, int16_t>(nl.arity, a, d, nl.count, stream);
        case TAB_F32:
            if constexpr (sizeof(T) == 8) return launch_small_arity<T, float>(nl.arity, a, d, nl.count, stream);
            return false;
        default: return false;
    }
}

#endif  // MXS_SMALL_IMPL

}  // namespace mxs
