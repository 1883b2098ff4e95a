// bin_box.h -- factor_costs_for_var (pydcop/algorithms/maxsum.py:382-447) for BINARY and UNARY factors
// beyond the register classes: a domain of more than 4 values, or two different domain sizes, up to
// 64 x 64 -- the tables the reference's own generators emit for meeting scheduling (PEAV,
// pydcop/commands/generators/meetingscheduling.py:450-454, 497-510, 540-599: binary utility / conflict
// and equality tables over 18..24 time slots, a unary table for a resource with one event) and for
// colourings with 5..8 colours (graphcoloring.py:271).  Until round 5 these ran in factor_generic: a
// thread per EDGE, D0 * D1 scalar iterations with strided table reads each (6.05 ms per cycle on the
// 50k-variable PEAV instance, 0.77 ms on the 100k-variable 8-colouring: profiles/r05_before_generic_v1.json).
//
// Here G = L0 x L1 lanes (4, 16 or 64: layout.h Bin2Shape) work on one factor and 64 / G factors share a
// wave.  Lane (l0, l1) owns the rows d0 = l0 + r * L0 (r < B0) and B1 columns (narrow entries: d1 = l1 * B1 + i1;
// entries of 4 / 8 bytes: chunks of up to 16 bytes interleaved across the L1 lanes of a row, see the kernel)
// of the table [D0][D1]: its B0 row pieces are read into registers before anything else (the image is the
// row-major table cut into lane pieces, layout.h: the G lanes of a factor read L0 whole rows per
// instruction), every entry feeds BOTH outputs
//     to variable 0, value d0:  t + (0 + m1[d1])      to variable 1, value d1:  t + (0 + m0[d0])
// (the reference's expression: sum_cost over the one other variable, maxsum.py:425-438; a unary factor:
// t + 0), the lane keeps B0 row minima and B1 column minima, and nothing crosses lanes until its last entry
// is used; then the partial minima go through LDS once and one lane per outgoing message ELEMENT merges
// the L1 (or L0) partials of its element and runs apply_damping + the send rule (maxsum.py:346-377).
// Digits past a domain (the lane grid is larger than the table) are staged as +inf, the identity of the
// min-plus semiring: such an entry never wins a minimum, whatever the lane read for it.
// Minima are exact and order-independent: bit for bit what factor_generic computes.
#pragma once
#include "kernels.h"

namespace mxs {

// Launch of one lane-grid group (engine.hip, launch_nary).  Returns false when no instantiation exists.
// Defined in bin_box.hip (a translation unit of its own: 14 shapes x 4 storage types x 2 signs x 2 words x with / without
// the variable class riding in the grid).
// `cls8` / `n_blocks8`: the K_V_PACK8 variable class rides in this launch as its first n_blocks8 workgroups (0: it does not).
template <typename T>
bool launch_factor_bin2(const NaryLaunch& nl, const SweepArgs<T>& a, const NaryDesc* d, hipStream_t stream,
                        const ClassInfo* cls8 = nullptr, int n_blocks8 = 0);

#ifdef MXS_BIN2_IMPL

// entry e of a lane's row piece, widened exactly (narrow integers / f32) or as stored (full width)
template <typename T, typename TT>
__device__ __forceinline__ T bin2_entry(const uint32_t* w, int e) {
    if constexpr (std::is_same<TT, int8_t>::value) return (T)(int)(int8_t)(uint8_t)(w[e >> 2] >> (8 * (e & 3)));
    else if constexpr (std::is_same<TT, int16_t>::value) return (T)(int)(int16_t)(uint16_t)(w[e >> 1] >> (16 * (e & 1)));
    else if constexpr (sizeof(TT) == 4) {
        float f;
        __builtin_memcpy(&f, &w[e], 4);
        return (T)f;
    } else {
        double f;
        __builtin_memcpy(&f, &w[2 * e], 8);
        return (T)f;
    }
}

// the minimum of N partials stored side by side (N * sizeof(T) bytes, aligned to min(16, that))
template <typename T, int N>
__device__ __forceinline__ T bin2_min_run(const T* p, T m) {
    constexpr int BYTES = N * (int)sizeof(T), AL = BYTES >= 16 ? 16 : BYTES;
    const T* q = (const T*)__builtin_assume_aligned(p, AL);
    T v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = q[i];
#pragma unroll
    for (int i = 0; i < N; ++i) m = min2(m, v[i]);
    return m;
}

// PACK8: the instantiation that hosts the variable class (its registers are the larger of the two paths': a template
// parameter so that a launch without the class does not pay for it)
template <typename T, typename TT, bool NEG, int L0, int L1, int B0, int B1, bool PACK8>
__global__ void __launch_bounds__(BIN2_WAVES * 64) k_factor_bin(SweepArgs<T> a, const NaryDesc* descs, int n_factors,
                                                                const ClassInfo* cls8, int n_blocks8) {
    static_assert(BIN2_WAVES * 64 == BLOCK, "the variable class's workgroups ride in this grid");
    if (PACK8 && (int)blockIdx.x < n_blocks8) {  // (block-uniform) the lane-per-edge variable class of 5..8 values: the grid's first workgroups
        const ClassInfo ci = cls8[0];
        variable_pack8_block<T>(a, ci, (int)blockIdx.x);
        return;
    }
    constexpr int G = L0 * L1, FPW = 64 / G;            // lanes per factor, factors per wave
    constexpr int P0 = L0 * B0, P1 = L1 * B1, NP = P0 + P1;
    constexpr int EPL = (NP + G - 1) / G;               // message elements per lane
    constexpr int PIECE = bin2_piece_bytes(B1, (int)sizeof(TT)), PW = PIECE / 4, ROWB = L1 * PIECE;
    // A lane's B1 columns of a row are NCH chunks of CE entries (CB bytes): entries of 4 / 8 bytes in chunks of 16, 8 or 4 bytes
    // INTERLEAVED across the L1 lanes of the row -- chunk j of lane l1 is chunk l1 + L1 * j of the row, so that one load
    // instruction of the group reads L1 consecutive chunks of every row it touches (a contiguous piece per lane made every
    // instruction touch PIECE / CB times the lines it used: TCP_TOTAL_CACHE_ACCESSES 3 x the data's lines on 48-byte pieces,
    // profiles/r05_pmc_peav_50k_f64_v1.txt); narrow entries: the piece itself (at most 16 bytes) is the one chunk.
    // The image is the row-major table either way (layout.h): only the lane <-> column assignment differs.
    constexpr int EL = (int)sizeof(TT);
    constexpr int CE = EL < 4 ? B1 : (B1 * EL) % 16 == 0 ? 16 / EL : (B1 * EL) % 8 == 0 ? 8 / EL : 4 / EL;
    constexpr int NCH = B1 / CE, CB = EL < 4 ? PIECE : CE * EL, CW = CB / 4;
    static_assert(NCH * CE == B1 && NCH * CW == PW, "chunks tile the piece");
    constexpr int PAL = CB % 16 == 0 ? 16 : CB % 8 == 0 ? 8 : 4;
    constexpr int NPART = P0 * L1 + P1 * L0;
    static_assert(64 % G == 0 && G * (B0 + B1) * (int)sizeof(T) % 16 == 0, "lane grid");
    __shared__ T s_in[BIN2_WAVES][FPW][NP];                       // incoming V->F messages, `0 + m`; +inf past a domain
    __shared__ __attribute__((aligned(16))) T s_part[BIN2_WAVES][FPW][NPART];  // partial minima: [d0][l1], then [d1][l0]
    const int wv = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int fw = lane / G, g = lane % G, l1 = g % L1, l0 = g / L1;
    auto col = [&](int i1) { return (l1 + L1 * (i1 / CE)) * CE + i1 % CE; };  // the column of the lane's i1-th entry of a row
    const int f_wave = (((int)blockIdx.x - (PACK8 ? n_blocks8 : 0)) * BIN2_WAVES + wv) * FPW;
    if (f_wave >= n_factors) return;                    // (wave-uniform; only wave-level barriers below)
    const bool active = f_wave + fw < n_factors;        // the last wave: idle groups shadow the last factor, store nothing
    const NaryDesc* fd = descs + (active ? f_wave + fw : n_factors - 1);
    const int D0 = fd->dom[0], D1 = fd->dom[1];         // (a unary factor: dom[1] == 1)
    const bool unary = fd->arity == 1;
    const int e0 = fd->edge_base;
    const int vo0 = fd->v2f_off[0], vo1 = fd->v2f_off[1], fo0 = fd->f2v_off[0], fo1 = fd->f2v_off[1];
    const uint8_t* img = a.ctables + fd->tab_off + l1 * CB;
    // one lane per message ELEMENT slot p = g + G * k of the padded scope [0, P0) + [P0, P0 + P1): request the
    // incoming element and what the epilogue needs of the outgoing one (the message sent last, its counter) ...
    int el_i[EPL], el_d[EPL], el_cnt[EPL];
    T el_in[EPL], el_prev[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int p = g + G * k;
        const int i = p >= P0 ? 1 : 0, d = p - (i ? P0 : 0);
        const bool valid = p < NP && d < (i ? (unary ? 0 : D1) : D0);
        el_i[k] = valid ? i : -1;
        el_d[k] = d;
        el_in[k] = el_prev[k] = (T)0;
        el_cnt[k] = 0;
        if (valid) {
            el_in[k] = a.v2f_old[(i ? vo1 : vo0) + d];
            if (!a.start) {
                el_prev[k] = a.f2v_old[(i ? fo1 : fo0) + d];
                el_cnt[k] = a.cF[e0 + i];
            }
        }
    }
    // ... then the lane's B0 row pieces (the loads return in order: the messages are staged while the table is on its way)
    uint32_t w[B0][PW];
    // (non-temporal loads of the image on instances beyond the Infinity Cache -- it is read once per cycle -- measured
    // SLOWER: peav_50k f64 214-219 us against 196, f32 unchanged, profiles/r05_bin2_nt_tables_ab_v1.txt)
#pragma unroll
    for (int r = 0; r < B0; ++r) {
        const int d0 = l0 + r * L0;
        const uint8_t* row = img + (int64_t)(d0 < D0 ? d0 : D0 - 1) * ROWB;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const uint32_t* q = (const uint32_t*)__builtin_assume_aligned(row + j * (L1 * CB), PAL);
#pragma unroll
            for (int x = 0; x < CW; ++x) w[r][j * CW + x] = q[x];
        }
    }
    T* in = s_in[wv][fw];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int p = g + G * k;
        // (`0 + m`: the first step of the reference's sum_cost, maxsum.py:430-441; a unary factor adds the 0 alone)
        if (p < NP) in[p] = el_i[k] >= 0 ? (T)0 + el_in[k] : ((unary && p == P0) ? (T)0 : pos_inf<T>());
    }
    __builtin_amdgcn_wave_barrier();
    T z0[B0], z1[B1], acc1[B1];
#pragma unroll
    for (int r = 0; r < B0; ++r) z0[r] = in[l0 + r * L0];
#pragma unroll
    for (int i1 = 0; i1 < B1; ++i1) {
        z1[i1] = in[P0 + col(i1)];
        acc1[i1] = pos_inf<T>();
    }
    T* part = s_part[wv][fw];
#pragma unroll
    for (int r = 0; r < B0; ++r) {
        T b0 = pos_inf<T>();
#pragma unroll
        for (int i1 = 0; i1 < B1; ++i1) {
            const T v = bin2_entry<T, TT>(w[r], i1);
            const T t = NEG ? -v : v;
            b0 = min2(b0, t + z1[i1]);
            acc1[i1] = min2(acc1[i1], t + z0[r]);
        }
        part[(l0 + r * L0) * L1 + l1] = b0;             // (= g + r * G: the wave's lanes write one contiguous run)
    }
#pragma unroll
    for (int i1 = 0; i1 < B1; ++i1) part[P0 * L1 + col(i1) * L0 + l0] = acc1[i1];
    __builtin_amdgcn_wave_barrier();
    // element lanes: the minimum over the lanes that share the digit, apply_damping, approx_match
    T el_m[EPL];
    bool bad0 = false, bad1 = false;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = el_i[k], d = el_d[k];
        T m = pos_inf<T>();
        if (i == 0) m = bin2_min_run<T, L1>(part + d * L1, m);
        else if (i == 1) m = bin2_min_run<T, L0>(part + P0 * L1 + d * L0, m);
        if (i >= 0 && !a.start) {
            const T p = el_prev[k];
            const int cnt = el_cnt[k];
            if (cnt > 0 && a.damp_f) m = a.damping * p + ((T)1 - a.damping) * m;
            const bool bad = cnt > 0 && !comp_match(m, p, a.stability);
            bad0 = bad0 || (i == 0 && bad);
            bad1 = bad1 || (i == 1 && bad);
        }
        el_m[k] = m;
    }
    // the elements of a message agree on "changed": one ballot per edge, read by the factor's lanes
    const unsigned long long gm = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull) << (fw * G);
    const bool nomatch0 = (__ballot(bad0) & gm) != 0ull, nomatch1 = (__ballot(bad1) & gm) != 0ull;
    if (!active) return;
    // send / send again / stay silent (the receiver keeps the old message)
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int i = el_i[k], d = el_d[k];
        if (i < 0) continue;
        const int fo = i ? fo1 : fo0, e = e0 + i;
        if (a.start) {  // on_start (maxsum.py:311-328): a unary factor sends in every mode, the others with start_messages == all
            a.f2v_new[fo + d] = (unary || a.start_mode == MXS_START_ALL) ? el_m[k] : (T)0;
            if (d == 0) a.cF[e] = 0;
            continue;
        }
        const int cnt = el_cnt[k];
        const bool match = cnt > 0 && !(i ? nomatch1 : nomatch0);
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

template <typename T, typename TT, int S>
inline void launch_bin2_shape(const SweepArgs<T>& a, const NaryDesc* d, int count, hipStream_t stream, const ClassInfo* cls8, int nb8) {
    constexpr Bin2Shape sh = BIN2_SHAPES[S];
    constexpr int FPB = BIN2_WAVES * (64 / (sh.L0 * sh.L1));
    const dim3 grid((unsigned)(nb8 + (count + FPB - 1) / FPB)), block((unsigned)(BIN2_WAVES * 64));
    if (nb8 > 0) {
        if (a.tab_neg) MXS_LAUNCH((k_factor_bin<T, TT, true, sh.L0, sh.L1, sh.B0, sh.B1, true>), grid, block, 0, stream, a, d, count, cls8, nb8);
        else MXS_LAUNCH((k_factor_bin<T, TT, false, sh.L0, sh.L1, sh.B0, sh.B1, true>), grid, block, 0, stream, a, d, count, cls8, nb8);
    } else if (a.tab_neg) MXS_LAUNCH((k_factor_bin<T, TT, true, sh.L0, sh.L1, sh.B0, sh.B1, false>), grid, block, 0, stream, a, d, count, cls8, 0);
    else MXS_LAUNCH((k_factor_bin<T, TT, false, sh.L0, sh.L1, sh.B0, sh.B1, false>), grid, block, 0, stream, a, d, count, cls8, 0);
}

template <typename T, typename TT, int... S>
inline bool launch_bin2_any(int s, const SweepArgs<T>& a, const NaryDesc* d, int count, hipStream_t stream, const ClassInfo* cls8,
                            int nb8, std::integer_sequence<int, S...>) {
    return ((s == S ? (launch_bin2_shape<T, TT, S>(a, d, count, stream, cls8, nb8), true) : false) || ...);
}

template <typename T>
bool launch_factor_bin2(const NaryLaunch& nl, const SweepArgs<T>& a, const NaryDesc* d, hipStream_t stream,
                        const ClassInfo* cls8, int n_blocks8) {
    const int s = nl.box - BIN2_BASE;
    const auto all = std::make_integer_sequence<int, BIN2_N_SHAPES>{};
    switch (nl.tab_type) {
        case TAB_I8: return launch_bin2_any<T, int8_t>(s, a, d, nl.count, stream, cls8, n_blocks8, all);
        case TAB_I16: return launch_bin2_any<T, int16_t>(s, a, d, nl.count, stream, cls8, n_blocks8, all);
        case TAB_F32: return launch_bin2_any<T, float>(s, a, d, nl.count, stream, cls8, n_blocks8, all);
        default: return launch_bin2_any<T, T>(s, a, d, nl.count, stream, cls8, n_blocks8, all);
    }
}

#endif  // MXS_BIN2_IMPL

}  // namespace mxs
