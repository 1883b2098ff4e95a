// nary_box.h -- factor_costs_for_var (pydcop/algorithms/maxsum.py:382-447) for arity-3 factors whose
// table is stored in a narrow integer type and whose dimensions a box shape divides (layout.h,
// nary_box_shape): ONE WAVE PER FACTOR, every running minimum in registers.
//
// Why: the workgroup-per-factor kernel (kernels.h, k_factor_nary_packed) gives a lane fixed values of
// (d1, d2) and walks d0 -- the minima towards variables 1 and 2 are register accumulators, but the one
// towards variable 0 is a reduction ACROSS lanes for every d0: DPP / permlane moves, selects and an LDS
// atomic per four values of d0, 18 VALU instructions per table entry where the arithmetic needs 8
// (profiles/r03_meeting50k_pmc_kernels_v1.txt; tools/valu_bench.hip: every VALU instruction of that
// kernel, 32- or 64-bit, costs a SIMD the same ~4.5 cycles, and an LDS atomic whose lanes share
// addresses 100-500).  Here lane (l0, l1, l2) of an L0 x L1 x L2 lane grid owns a B0 x B1 x B2 sub-box
// of the table -- its entries are ONE record of the image, read with 16-byte loads into registers before
// anything else -- and keeps B0 + B1 + B2 partial minima, one per value of each of its three digits.
// Nothing crosses lanes until the last entry has been used; then the partials go through LDS once
// (plain stores, the 64 / L_p lanes that share a digit side by side) and the lanes of the wave -- one
// per outgoing message element -- reduce them with 16-byte reads and run apply_damping + the send rule
// (maxsum.py:346-377).
//
// Arithmetic: the reference's own expressions, op for op -- for the entry at (d0, d1, d2)
//     to variable 0:  t + ((0 + m1[d1]) + m2[d2])
//     to variable 1:  t + ((0 + m0[d0]) + m2[d2])
//     to variable 2:  t + ((0 + m0[d0]) + m1[d1])         (sum_cost over the others in dimensions order,
// maxsum.py:425-438), minima exact and order-independent: bit for bit what k_factor_nary computes.
#pragma once
#include "kernels.h"

namespace mxs {

#ifndef MXS_BOX_NT
#define MXS_BOX_NT 1  // the table records (read once per cycle, 0.7 GB on meeting_50k) with non-temporal loads: cycle 249.6 -> 245.1 us (f32 164.2 -> 159.4), profiles/r04_box_nt_ab_v1.jsonl
#endif
// one 16-byte piece of a lane's record
template <bool NT>
__device__ __forceinline__ Piece16 load_piece16(const uint8_t* p) {
    if constexpr (NT) {
        const v4u v = __builtin_nontemporal_load((const v4u*)__builtin_assume_aligned(p, 16));
        return Piece16{{v[0], v[1], v[2], v[3]}};
    }
    return *(const Piece16*)__builtin_assume_aligned(p, 16);
}

template <typename T>
struct alignas(4 * sizeof(T)) BoxQuad {
    T v[4];
};

// entry e of a lane's record (narrow integers, widened exactly)
template <typename T, typename TT>
__device__ __forceinline__ T box_entry(const uint32_t* w, int e) {
    if constexpr (sizeof(TT) == 1) return (T)(int)(int8_t)(uint8_t)(w[e >> 2] >> (8 * (e & 3)));
    else return (T)(int)(int16_t)(uint16_t)(w[e >> 1] >> (16 * (e & 1)));
}

#ifndef MXS_BOX_EU
#define MXS_BOX_EU 0  // > 0: register budget for that many waves per SIMD (experiments: profiles/r06_box3_isa.txt)
#endif
template <typename T, typename TT, bool NEG, int B0, int B1, int B2>
__global__ void __launch_bounds__(BOX_WAVES * 64)
#if MXS_BOX_EU > 0
__attribute__((amdgpu_waves_per_eu(MXS_BOX_EU, MXS_BOX_EU)))
#endif
k_factor_box3(SweepArgs<T> a, const NaryDesc* descs, int n_factors) {
    constexpr int E = B0 * B1 * B2, NV = B0 + B1 + B2;
    constexpr int NW = box_rec_words(E, (int)sizeof(TT)), FULL = NW / 4, REST = NW % 4;
    // A record of more than BOX_MAX_WORDS dwords (int16 entries on 6 x 6 x 6 boxes: 108) is worked through in TWO passes over
    // the leading box digit: the pieces that hold the entries of i0 < B0 / 2, then the ones of the rest (the piece that
    // straddles the middle is read twice) -- the same registers hold either half.
    constexpr int PASSES = NW <= BOX_MAX_WORDS ? 1 : 2;
    static_assert(box_record_fits(B0, B1, B2, (int)sizeof(TT)), "a lane's record (or half of it) lives in registers");
    constexpr int EH = E / PASSES;                                    // entries of a pass
    constexpr int P_LAST0 = (EH * (int)sizeof(TT) - 1) / 16;          // last piece of pass 0
    constexpr int P_FIRST1 = (EH * (int)sizeof(TT)) / 16;             // first piece of pass 1
    constexpr int NPW = PASSES == 1 ? NW : 4 * ((P_LAST0 + 1) > (FULL - P_FIRST1) ? (P_LAST0 + 1) : (FULL - P_FIRST1));  // dwords in registers
    static_assert(NPW <= BOX_MAX_WORDS, "a pass's pieces live in registers");
    __shared__ T s_in[BOX_WAVES][BOX_MAX_SUMD];      // the incoming V->F messages of the wave's factor
    __shared__ BoxQuad<T> s_part[BOX_WAVES][NV][16]; // partial minima: [value of a digit][lanes sharing it]
    const int wv = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int f = __builtin_amdgcn_readfirstlane((int)blockIdx.x * BOX_WAVES + wv);
    if (f >= n_factors) return;
    const NaryDesc fd = descs[f];  // wave-uniform: scalar loads
    const int D0 = fd.dom[0], D1 = fd.dom[1], D2 = fd.dom[2];
    // the lane grid (L1, L2 powers of two, L0 * L1 * L2 = 64: layout.h); it may overhang the table: P_i = L_i * B_i >= D_i
    const int L1 = box_lanes(D1, B1), L2 = box_lanes(D2, B2);
    const int sh2 = __builtin_ctz((unsigned)L2), sh1 = __builtin_ctz((unsigned)L1);
    const int P0 = (64 >> (sh1 + sh2)) * B0, P1 = L1 * B1;
    // one lane per message ELEMENT SLOT of the padded scope (two passes cover P0 + P1 + P2 <= 128): request the incoming
    // message element and what the epilogue needs of the outgoing one (the message sent last, its send counter) ...
    const int off1 = P0, off2 = P0 + P1, sumd = off2 + L2 * B2;
    int el_i[2], el_d[2], el_cnt[2];
    T el_prev[2], el_in[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int idx = lane + 64 * ps;
        const int i = (idx >= off1 ? 1 : 0) + (idx >= off2 ? 1 : 0);
        const int d = idx - (i == 0 ? 0 : i == 1 ? off1 : off2);
        const bool valid = idx < sumd && d < (i == 0 ? D0 : i == 1 ? D1 : D2);
        el_i[ps] = valid ? i : -1;
        el_d[ps] = d;
        el_prev[ps] = el_in[ps] = (T)0;
        el_cnt[ps] = 0;
        if (valid) {
            const int vo = i == 0 ? fd.v2f_off[0] : i == 1 ? fd.v2f_off[1] : fd.v2f_off[2];
            const int fo = i == 0 ? fd.f2v_off[0] : i == 1 ? fd.f2v_off[1] : fd.f2v_off[2];
            el_in[ps] = a.v2f_old[vo + d];
            if (!a.start) {
                el_prev[ps] = a.f2v_old[fo + d];
                el_cnt[ps] = a.cF[fd.edge_base + i];
            }
        }
    }
    // ... then the lane's record (the loads return in order: the wave can stage the messages while the
    // table is still on its way)
    uint32_t w[NPW];
    const uint8_t* img = a.ctables + fd.tab_off;
    {
        constexpr int K1 = PASSES == 1 ? FULL : P_LAST0 + 1;  // pieces of the first (or only) pass
#pragma unroll
        for (int k = 0; k < K1; ++k) {
            const Piece16 pc = load_piece16<MXS_BOX_NT != 0>(img + ((int64_t)k * 64 + lane) * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[4 * k + j] = pc.w[j];
        }
        if constexpr (PASSES == 1 && REST > 0) {
            const uint32_t* r = (const uint32_t*)__builtin_assume_aligned(img + (int64_t)FULL * 1024 + lane * (REST * 4), 4);
#pragma unroll
            for (int j = 0; j < REST; ++j) w[4 * FULL + j] = r[j];
        }
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        // (dimension 0's message is staged as `0 + m`: the first step of the reference's sum_cost for the
        // outputs to variables 1 and 2, maxsum.py:430-441)
        // a digit past its domain: +inf, the identity of the min-plus semiring -- its (zero-filled) entries never win
        if (lane + 64 * ps < sumd)
            s_in[wv][lane + 64 * ps] = el_i[ps] < 0 ? pos_inf<T>() : el_i[ps] == 0 ? (T)0 + el_in[ps] : el_in[ps];
    }
    __builtin_amdgcn_wave_barrier();
    // the lane's place in the lane grid
    const int l2 = lane & (L2 - 1), l1 = (lane >> sh2) & (L1 - 1), l0 = lane >> (sh1 + sh2);
    T a0[B0], m1[B1], z1[B1], m2[B2], acc1[B1], acc2[B2];
#pragma unroll
    for (int i = 0; i < B0; ++i) a0[i] = s_in[wv][l0 * B0 + i];
#pragma unroll
    for (int i = 0; i < B1; ++i) {
        m1[i] = s_in[wv][off1 + l1 * B1 + i];
        z1[i] = (T)0 + m1[i];
        acc1[i] = pos_inf<T>();
    }
#pragma unroll
    for (int i = 0; i < B2; ++i) {
        m2[i] = s_in[wv][off2 + l2 * B2 + i];
        acc2[i] = pos_inf<T>();
    }
    // where this lane's partial minima go: the lanes that share a digit are neighbours in the row of that
    // digit's value -- slot = digit * (64 / L) + rank among them
    T* part = (T*)&s_part[wv][0][0];
    const int slot0 = lane;                                              // l0 is the leading digit
    const int slot1 = l1 * (64 >> sh1) + (l0 << sh2) + l2;
    const int slot2 = l2 * (64 >> sh2) + (lane >> sh2);
#pragma unroll
    for (int i0 = 0; i0 < B0; ++i0) {
        if constexpr (PASSES == 2) {
            if (i0 == B0 / 2) {  // the second half of the record into the same registers
#pragma unroll
                for (int k = P_FIRST1; k < FULL; ++k) {
                    const Piece16 pc = load_piece16<MXS_BOX_NT != 0>(img + ((int64_t)k * 64 + lane) * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[4 * (k - P_FIRST1) + j] = pc.w[j];
                }
            }
        }
        // (entry index relative to the first entry the registers hold)
        const int e_base = (PASSES == 2 && i0 >= B0 / 2) ? P_FIRST1 * (16 / (int)sizeof(TT)) : 0;
        T sp1[B2];
#pragma unroll
        for (int i2 = 0; i2 < B2; ++i2) sp1[i2] = a0[i0] + m2[i2];
        T b0 = pos_inf<T>();
#pragma unroll
        for (int i1 = 0; i1 < B1; ++i1) {
            const T sp2 = a0[i0] + m1[i1];
#pragma unroll
            for (int i2 = 0; i2 < B2; ++i2) {
                const T v = box_entry<T, TT>(w, (i0 * B1 + i1) * B2 + i2 - e_base);
                const T t = NEG ? -v : v;
                b0 = min2(b0, t + (z1[i1] + m2[i2]));
                acc1[i1] = min2(acc1[i1], t + sp1[i2]);
                acc2[i2] = min2(acc2[i2], t + sp2);
            }
        }
        part[i0 * 64 + slot0] = b0;
    }
#pragma unroll
    for (int i = 0; i < B1; ++i) part[(B0 + i) * 64 + slot1] = acc1[i];
#pragma unroll
    for (int i = 0; i < B2; ++i) part[(B0 + B1 + i) * 64 + slot2] = acc2[i];
    __builtin_amdgcn_wave_barrier();
    // element lanes: the minimum over the lanes that share the digit, apply_damping, approx_match
    T el_m[2];
    bool el_bad[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int i = el_i[ps], d = el_d[ps];
        T m = pos_inf<T>();
        el_bad[ps] = false;
        if (i >= 0) {
            const int lp = i == 0 ? d / B0 : i == 1 ? d / B1 : d / B2;
            const int v = i == 0 ? d - lp * B0 : i == 1 ? B0 + (d - lp * B1) : B0 + B1 + (d - lp * B2);
            const int shp = i == 0 ? sh1 + sh2 : i == 1 ? 6 - sh1 : 6 - sh2;  // log2(64 / L_i); L0 = 64 / (L1 * L2)
            const int n = 1 << shp;
            const BoxQuad<T>* row = &s_part[wv][v][(lp << shp) >> 2];
            for (int k = 0; k < (n >> 2); ++k) {
                const BoxQuad<T> q = row[k];
                m = min2(min2(min2(min2(m, q.v[0]), q.v[1]), q.v[2]), q.v[3]);
            }
            if (!a.start) {
                const T p = el_prev[ps];
                const int cnt = el_cnt[ps];
                if (cnt > 0 && a.damp_f) m = a.damping * p + ((T)1 - a.damping) * m;
                el_bad[ps] = cnt > 0 && !comp_match(m, p, a.stability);
            }
        }
        el_m[ps] = m;
    }
    // the elements of a message agree on "changed": one ballot per edge
    bool nomatch[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        nomatch[i] = (__ballot((el_i[0] == i && el_bad[0]) || (el_i[1] == i && el_bad[1])) != 0ull);
    // send / send again / stay silent (the receiver keeps the old message)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int i = el_i[ps], d = el_d[ps];
        if (i < 0) continue;
        const int fo = i == 0 ? fd.f2v_off[0] : i == 1 ? fd.f2v_off[1] : fd.f2v_off[2];
        const int e = fd.edge_base + i;
        if (a.start) {  // only start_messages == all makes a non-unary factor send
            a.f2v_new[fo + d] = a.start_mode == MXS_START_ALL ? el_m[ps] : (T)0;
            if (d == 0) a.cF[e] = 0;
            continue;
        }
        const int cnt = el_cnt[ps];
        const bool match = cnt > 0 && !(i == 0 ? nomatch[0] : i == 1 ? nomatch[1] : nomatch[2]);
        int out = 1;
        T val = el_m[ps];
        if (match) {
            if (cnt < SAME_COUNT) {
                out = cnt + 1;
            } else {
                out = cnt;
                val = el_prev[ps];
            }
        }
        a.f2v_new[fo + d] = val;
        if (d == 0) a.cF[e] = (uint8_t)out;
    }
}

// Launch of one box group (engine.hip, launch_nary).  Returns false when no instantiation exists.
template <typename T>
inline bool launch_factor_box3(const NaryLaunch& nl, const SweepArgs<T>& a, const NaryDesc* d, hipStream_t stream) {
    const dim3 grid((unsigned)((nl.count + BOX_WAVES - 1) / BOX_WAVES)), block((unsigned)(BOX_WAVES * 64));
#define MXS_BOX_LAUNCH(TT, B0, B1, B2)                                                                              \
    do {                                                                                                             \
        if (a.tab_neg) MXS_LAUNCH((k_factor_box3<T, TT, true, B0, B1, B2>), grid, block, 0, stream, a, d, (int)nl.count);  \
        else MXS_LAUNCH((k_factor_box3<T, TT, false, B0, B1, B2>), grid, block, 0, stream, a, d, (int)nl.count);           \
        return true;                                                                                                 \
    } while (0)
    if (nl.tab_type == TAB_I8) {
        switch (nl.box) {
            case 1: MXS_BOX_LAUNCH(int8_t, 2, 2, 2);
            case 2: MXS_BOX_LAUNCH(int8_t, 3, 3, 3);
            case 3: MXS_BOX_LAUNCH(int8_t, 4, 4, 4);
            case 4: MXS_BOX_LAUNCH(int8_t, 6, 6, 6);
        }
    } else if (nl.tab_type == TAB_I16) {
        switch (nl.box) {
            case 1: MXS_BOX_LAUNCH(int16_t, 2, 2, 2);
            case 2: MXS_BOX_LAUNCH(int16_t, 3, 3, 3);
            case 3: MXS_BOX_LAUNCH(int16_t, 4, 4, 4);
            case 4: MXS_BOX_LAUNCH(int16_t, 6, 6, 6);  // (two passes over the record: 108 dwords per lane)
        }
    }
#undef MXS_BOX_LAUNCH
    return false;
}

}  // namespace mxs
