// kernels.h -- gfx950 (CDNA4) device code of the synchronous Max-Sum sweep.
//
// One launch of `k_sweep` is one synchronous cycle: every factor's and every
// variable's on_new_cycle (pydcop/algorithms/maxsum.py:339-379, 525-565).  Both
// sides read only the message buffers of cycle t-1 and write those of cycle t
// (the reference's BSP barrier, pydcop/infrastructure/computations.py:684-788,
// becomes the kernel boundary), so factor blocks and variable blocks run side by
// side in the same grid.
//
// This is a min-plus semiring bounded by memory traffic -- no MFMA.  What matters
// (cdna_hip_programming.md section 6, and the measurements of
// tools/pattern_bench.hip): every store is a coalesced full-line stream (F2V is
// factor-major and written by the factor side, V2F variable-major and written by
// the variable side), each side finds its own previous output in the old buffer
// of its own array, the only irregular accesses are two message-sized gathers
// per edge whose addresses come from coalesced index tables, tables are stored
// entry-major so a wave reads them as one coalesced segment per entry, counters
// are never gathered, and all per-item state lives in registers (compile-time
// D / degree bounds, no scratch).
//
// Max mode runs as min mode on negated costs (exact in IEEE arithmetic; the host
// negates on upload/download), so only `<` appears below.
//
// Arithmetic follows the reference's evaluation order expression by expression
// (see oracle/maxsum_oracle.c, which restates it on the CPU); build with
// -ffp-contract=off so no FMA contraction changes a rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <utility>

#include "layout.h"

namespace mxs {

// The launches of ONE cycle (or of one phase of a sharded cycle) read the buffers of cycle t-1 and write disjoint
// records of cycle t: they do not depend on each other.  On one stream HIP still sets the AQL barrier bit of every
// dispatch, so launch k + 1 starts only when launch k has drained -- a cycle of seven 5..20-us launches (SECP) pays
// seven ramps and seven tails.  With `chain` set the engine's FIRST launch of a set keeps the barrier (it waits for
// everything of the cycle before), every later one goes out with hipExtAnyOrderLaunch (no barrier bit): the command
// processor hands out its workgroups as soon as the launch before has handed out its own, the way the workgroups of
// one grid follow each other.  No events, no second stream.  The next set's first launch waits for all of them.
struct LaunchChain {
    unsigned flags = 0;   // of the next launch
    bool chain = false;   // later launches of the set go out without the barrier bit
};
extern thread_local LaunchChain g_launch;
#define MXS_LAUNCH(kernel, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                         \
        if (mxs::g_launch.flags)                                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), stream, nullptr, nullptr,           \
                                  mxs::g_launch.flags, __VA_ARGS__);                                             \
        else                                                                                                     \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                   \
        if (mxs::g_launch.chain) mxs::g_launch.flags = hipExtAnyOrderLaunch;                                     \
    } while (0)

constexpr int SAME_COUNT = 4;  // maxsum.py:106
#ifndef MXS_ASM_MIN
#define MXS_ASM_MIN 1
#endif
#ifndef SWEEP_MIN_WAVES
#define SWEEP_MIN_WAVES 4
#endif

// Cache policy experiments (build variants only; the default build is MXS_NT = 0).  Non-temporal
// accesses on  bit0 (1): the single-use index / table streams,  bit1 (2): a side's own previous
// records,  bit2 (4): the gathers,  bit3 (8): the stores.  tools/bw_sweep.hip: a streaming read
// of 1 GB runs at 7.2 TB/s with nt loads against 6.2 with the default policy.
#ifndef MXS_NT
#define MXS_NT 0
#endif
// The policy is a template parameter NT of the sweep: the engine runs k_sweep<.., NT_STREAMING>
// on instances whose cycle does not fit the 256-MB Infinity Cache (measured, profiles/
// r03_cache_policy_ab_v1.txt: nt stores 1M-variable colouring 279 -> 264 us, Ising 113 -> 106;
// on the cache-resident 100k instance every nt variant is SLOWER, 19.9 -> 21.5 us with nt stores:
// the next cycle finds its input in the cache only if the stores left it there).
constexpr int NT_STREAMING = 9;  // nt on the index / table streams and on the stores
#define MXS_NT_FLAGS(NT)                                                                        \
    constexpr bool NT_IDX = ((NT) & 1) != 0, NT_PREV = ((NT) & 2) != 0, NT_GATHER = ((NT) & 4) != 0; \
    (void)NT_IDX, (void)NT_PREV, (void)NT_GATHER
template <bool NT, typename U>
__device__ __forceinline__ U ldp(const U* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a
// compile-time constant in every copy of the body, so that per-thread arrays indexed by it are
// registers from the start (a `#pragma unroll` loop around bodies with inner run-time loops left
// them in scratch memory).
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// A wave hands its outgoing messages over through LDS so that every store instruction of
// the wave writes 1 KB of contiguous bytes (16 per lane) instead of 16-byte pieces at the
// stride of a message record: measured 1.5 % (cache-resident) to 4 % (HBM-resident) faster.
// One staging area per block, shared by the variable and the factor paths.
constexpr int STAGE_BYTES_PER_WAVE = 64 * 2 * MAX_REG_D * 8;  // two f64 messages per lane
__shared__ double g_stage[BLOCK / 64][STAGE_BYTES_PER_WAVE / 8];

struct Piece16 {  // 16 bytes, moved with one instruction
    unsigned int w[4];
};
// one 16-byte piece from the staging area to global memory, non-temporal when the launch streams (NT bit 3)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <bool NT_STORE>
__device__ __forceinline__ void store_piece16(char* dst, const char* src) {
    if constexpr (NT_STORE) {
        __builtin_nontemporal_store(*(const v4u*)__builtin_assume_aligned(src, 16), (v4u*)__builtin_assume_aligned(dst, 16));
        return;
    }
    *(Piece16*)__builtin_assume_aligned(dst, 16) = *(const Piece16*)__builtin_assume_aligned(src, 16);
}
#ifndef MXS_NT_TIGHT
#define MXS_NT_TIGHT 1  // the streaming policy's non-temporal stores also on records that are no multiple of 16 bytes
#endif

// Every lane of the wave holds ELEMS contiguous elements of `arr`, lane l at element
// wave_base + l * ELEMS.  ELEMS * sizeof(T) is a multiple of 16.  All 64 lanes take part.
template <typename T, int ELEMS, int NT = MXS_NT>
__device__ __forceinline__ void wave_store_linear(T* arr, int64_t wave_base, const T (&vals)[ELEMS]) {
    constexpr bool NT_STORE = (NT & 8) != 0;
    static_assert(64 * ELEMS * sizeof(T) % 16 == 0 && 64 * ELEMS * sizeof(T) <= STAGE_BYTES_PER_WAVE, "");
    if constexpr (ELEMS * sizeof(T) % 16 != 0) {
        // records that are no multiple of 16 bytes (unpadded D = 3: 24 / 12 bytes): the wave's 64
        // records are still ONE contiguous run of 64 * ELEMS * sizeof(T) bytes (the wave base is
        // 16-byte aligned: a class starts on 32 bytes and 64 records are a multiple of 16); 16-byte
        // pieces, the last instruction with part of the lanes
        const int w = (int)threadIdx.x >> 6, l = (int)threadIdx.x & 63;
        char* so = (char*)g_stage[w];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int d = 0; d < ELEMS; ++d) ((T*)so)[l * ELEMS + d] = vals[d];
        __builtin_amdgcn_wave_barrier();
        char* out = (char*)(arr + wave_base);
        constexpr int BYTES = 64 * ELEMS * (int)sizeof(T), FULL = BYTES / 1024, REST = (BYTES % 1024) / 16;
#pragma unroll
        for (int k = 0; k < FULL; ++k) store_piece16<NT_STORE && MXS_NT_TIGHT>(out + k * 1024 + l * 16, so + k * 1024 + l * 16);
        if (REST > 0 && l < REST) store_piece16<NT_STORE && MXS_NT_TIGHT>(out + FULL * 1024 + l * 16, so + FULL * 1024 + l * 16);
        return;
    }
    const int w = (int)threadIdx.x >> 6, l = (int)threadIdx.x & 63;
    char* so = (char*)g_stage[w];
    __builtin_amdgcn_wave_barrier();  // the previous use of the staging area is over
#pragma unroll
    for (int d = 0; d < ELEMS; ++d) ((T*)so)[l * ELEMS + d] = vals[d];
    __builtin_amdgcn_wave_barrier();
    char* out = (char*)(arr + wave_base);
    constexpr int PIECES = ELEMS * (int)sizeof(T) / 16;
#pragma unroll
    for (int k = 0; k < PIECES; ++k) store_piece16<NT_STORE>(out + k * 1024 + l * 16, so + k * 1024 + l * 16);
}

template <typename T>
struct SweepArgs {
    const T* v2f_old;   // V->F messages of cycle t-1 (variable-major)
    T* v2f_new;         // ... of cycle t
    const T* f2v_old;   // F->V messages of cycle t-1 (factor-major)
    T* f2v_new;
    const T* tables;
    const uint8_t* ctables;  // compact table records of the classes with ClassInfo::tab_type != TAB_FULL
    int32_t tab_neg;         // 1 (max mode): compact records hold un-negated values, negate on load
    const T* var_cost;
    uint8_t* cF;        // [n_edges] factor-major send counters
    uint8_t* cV;        // [n_cv] variable-side send counters (CSR slots + class tables)
    const int32_t* edge_v2f;  // [n_edges] V2F offset of the edge's message (factor-major)
    const int32_t* f2v_off;   // [n_edges] F2V offset of the edge's message
    const int32_t* vrowptr;
    const int32_t* vslot_f2v;  // [n_edges] CSR slot -> F2V offset (generic variables)
    const int32_t* vslot_v2f;  // [n_edges] CSR slot -> V2F offset
    const int32_t* vell;       // per lane of the packed variable classes: F2V offset / -1
    const WaveMeta* vwave;     // per wave of the packed variable classes
    const HubBlock* hub_blocks;  // per workgroup of the K_V_HUB class
    const int32_t* vdom;
    const int64_t* vcost_off;
    const int32_t* init_idx;
    const FactorGen* fgen;
    const int32_t* edge_gen_factor;
    const int32_t* edge_dom;
    int32_t* sel;
    T* belief;
    T damping;
    T stability;
    int32_t damp_f;      // damping_nodes in {factors, both}
    int32_t damp_v;      // damping_nodes in {vars, both}
    int32_t start;       // 1: cycle 0 (on_start), 0: regular cycle
    int32_t start_mode;  // MXS_START_*
    int32_t null_f2v;    // offset of the all-zero F2V block (padding slots)
    int64_t* timeline;   // profiling only (mxs_debug_timeline): per block {start, end} in
                         // wall_clock64 ticks and its class kind; NULL in normal runs
    // Fused sharded launch (engine.hip, step_compute): the blocks of a cut factor class
    // (ClassInfo::wait_halo) wait until halo_flags[0] -- the number of halo exchanges
    // unpacked so far -- reaches need_epoch.  NULL in every other launch.
    uint32_t* halo_flags;  // epoch word(s), error bits at [HALO_ERR_WORD]
    uint32_t need_epoch;
    // Sharded operation, direct exchange: the lane of a boundary edge also writes its record
    // into the send buffer the engine hands to RCCL (no pack kernel).  send_slot[pos] = record
    // index in send_out or -1, per lane of the packed variable classes; NULL otherwise.
    T* send_out;
    const int32_t* send_slot;
    // Peer-store exchange (engine.hip, "p2p"): no collective at all.  The lane of a cut edge
    // stores its record straight into the ghost region of the shard that holds the factor's
    // replica (peer memory mapped through hipIpc, xGMI): slots [peer_first[q], peer_first[q+1])
    // of the send order belong to peer q, whose copy of my block starts at peer_dst[q].  The
    // cut factors of a launch read the ghost region ghost_old (V2F offsets >= ghost_lo address
    // it) once every peer has published the epoch need_epoch in halo_flags[q].
    int32_t n_peers;                        // 0: not in peer-store mode
    int32_t me;                             // this shard's rank
    int32_t ghost_lo;                       // INT32_MAX when unused
    const T* ghost_old;
    int32_t peer_first[MXS_MAX_PEERS];      // unused entries: INT32_MAX
    T* peer_dst[MXS_MAX_PEERS];
    int32_t n_classes;
    // First block of every class of the launch (in launch order; unused entries
    // hold INT32_MAX): a block finds its class with compares on kernel arguments,
    // then one scalar load of its ClassInfo -- no chain of dependent global loads
    // before the block can start.
    int32_t block_base[MAX_CLASSES];
    const ClassInfo* classes;  // [n_classes] in launch order
    // Block schedule (layout.h, Layout::sched): workgroup b works on block (sched[b] & 0xffffff)
    // of class (sched[b] >> 24) -- one scalar load that depends on nothing but blockIdx, in place
    // of the compares on block_base[].  NULL: blocks in class order.
    const uint32_t* sched;
};

template <typename T>
__device__ __forceinline__ T pos_inf() {
    return (T)INFINITY;
}
__device__ __forceinline__ double absT(double x) { return fabs(x); }
__device__ __forceinline__ float absT(float x) { return fabsf(x); }

// approx_match, maxsum.py:688-710, one component (prev is not None).
template <typename T>
__device__ __forceinline__ bool comp_match(T c, T prev_c, T stability) {
    if (prev_c != c) {
        const T delta = absT(prev_c - c);
        if (prev_c + c != (T)0) {
            if (!(((T)2 * delta / absT(prev_c + c)) < stability)) return false;
        } else {
            return false;
        }
    }
    return true;
}

// apply_damping (maxsum.py:679-685) + the send rule of on_new_cycle
// (maxsum.py:349-377 / 540-564) on a register-resident message.  On return `m`
// is what the receiver holds after this cycle; returns the new send counter.
template <typename T, int D>
__device__ __forceinline__ uint8_t damp_and_filter(T (&m)[D], const T (&prev)[D], uint8_t cnt,
                                                   bool damp_on, T damping, T stability) {
    if (cnt > 0 && damp_on) {
#pragma unroll
        for (int d = 0; d < D; ++d) m[d] = damping * prev[d] + ((T)1 - damping) * m[d];
    }
    bool match = cnt > 0;
#pragma unroll
    for (int d = 0; d < D; ++d) match = match && comp_match(m[d], prev[d], stability);
    if (!match) return 1;                                   // sent (first time)
    if (cnt < SAME_COUNT) return (uint8_t)(cnt + 1);        // sent again
#pragma unroll
    for (int d = 0; d < D; ++d) m[d] = prev[d];             // not sent: receiver keeps
    return cnt;
}

// A message of a uniform class: D values padded to H = half_stride(D) elements,
// aligned to its own (power-of-two, <= 32 bytes) size.
template <typename T, int D>
struct Msg {
    static constexpr int H = half_stride(D, (int)sizeof(T));
    static constexpr int ALIGN = (H * (int)sizeof(T)) % 16 == 0 ? 16 : ((H * (int)sizeof(T)) % 8 == 0 ? 8 : 4);
    template <bool NT = false>
    static __device__ __forceinline__ void load(const T* p, T (&m)[D]) {
        const T* q = (const T*)__builtin_assume_aligned(p, ALIGN);
#pragma unroll
        for (int d = 0; d < D; ++d) m[d] = ldp<NT>(q + d);
    }
    // stores the padding too: full-sector, fully coalesced writes
    static __device__ __forceinline__ void store(T* p, const T (&m)[D]) {
        T* q = (T*)__builtin_assume_aligned(p, ALIGN);
#pragma unroll
        for (int d = 0; d < H; ++d) q[d] = d < D ? m[d < D ? d : 0] : (T)0;
    }
    // Element-wise loads that are coherent with stores of OTHER GPUs / other kernels in flight
    // (system scope: not served from a stale line of this XCD's L2): the ghost records of the
    // peer-store exchange.
    static __device__ __forceinline__ void load_sys(const T* p, T (&m)[D]) {
#pragma unroll
        for (int d = 0; d < D; ++d) m[d] = __hip_atomic_load(p + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // A record with padding (H > D, e.g. D = 3) keeps the SEND COUNTER of its message in the
    // first padding element (a small integer, exact in T): the owner of the record reads
    // and writes it with the message itself -- no separate one-byte load and store per
    // edge.  The other side gathers only the D values.  Records without padding use the
    // counter arrays cF / cV.
    static constexpr bool CNT_IN_MSG = H > D;
    template <bool NT = false>
    static __device__ __forceinline__ uint8_t load_c(const T* p, T (&m)[D]) {
        const T* q = (const T*)__builtin_assume_aligned(p, ALIGN);
#pragma unroll
        for (int d = 0; d < D; ++d) m[d] = ldp<NT>(q + d);
        return (uint8_t)(int)ldp<NT>(q + (D < H ? D : 0));
    }
    static __device__ __forceinline__ T padded(const T (&m)[D], uint8_t cnt, int d) {
        return d < D ? m[d < D ? d : 0] : (d == D ? (T)(int)cnt : (T)0);
    }
    static __device__ __forceinline__ void store_c(T* p, const T (&m)[D], uint8_t cnt) {
        T* q = (T*)__builtin_assume_aligned(p, ALIGN);
#pragma unroll
        for (int d = 0; d < H; ++d) q[d] = padded(m, cnt, d);
    }
};

// The N table entries of factor j of a register class.  Full width: entry-major (SoA), a wave
// reads each entry as one coalesced segment.  Compact types (layout.h, TabType): one record of
// back-to-back narrow entries per factor, read with whole-dword vector loads (consecutive lanes
// read consecutive records) and widened here -- every value is exactly what the full-width
// image holds, including the sign of a negated zero.
template <typename T, int N, int NT = MXS_NT>
__device__ __forceinline__ void load_table(const SweepArgs<T>& a, const ClassInfo& ci, int j, T (&tab)[N]) {
    MXS_NT_FLAGS(NT);
    if (ci.tab_type == TAB_FULL) {
#pragma unroll
        for (int k = 0; k < N; ++k) tab[k] = ldp<NT_IDX>(a.tables + ci.tab_base + (int64_t)k * ci.count + j);
        return;
    }
    if (ci.tab_type == TAB_I8) {
        constexpr int REC = tab_record_bytes(N, 1);
        uint32_t w[REC / 4];
        const uint32_t* p = (const uint32_t*)__builtin_assume_aligned(
            a.ctables + ci.ctab_base + (int64_t)j * REC, REC >= 16 ? 16 : REC);
#pragma unroll
        for (int i = 0; i < REC / 4; ++i) w[i] = ldp<NT_IDX>(p + i);
#pragma unroll
        for (int k = 0; k < N; ++k) tab[k] = (T)(int)(int8_t)(uint8_t)(w[k >> 2] >> (8 * (k & 3)));
    } else if (ci.tab_type == TAB_I16) {
        constexpr int REC = tab_record_bytes(N, 2);
        uint32_t w[REC / 4];
        const uint32_t* p = (const uint32_t*)__builtin_assume_aligned(
            a.ctables + ci.ctab_base + (int64_t)j * REC, REC >= 16 ? 16 : REC);
#pragma unroll
        for (int i = 0; i < REC / 4; ++i) w[i] = ldp<NT_IDX>(p + i);
#pragma unroll
        for (int k = 0; k < N; ++k) tab[k] = (T)(int)(int16_t)(uint16_t)(w[k >> 1] >> (16 * (k & 1)));
    } else {  // TAB_F32
        constexpr int REC = tab_record_bytes(N, 4);
        uint32_t w[REC / 4];
        const uint32_t* p = (const uint32_t*)__builtin_assume_aligned(
            a.ctables + ci.ctab_base + (int64_t)j * REC, REC >= 16 ? 16 : REC);
#pragma unroll
        for (int i = 0; i < REC / 4; ++i) w[i] = ldp<NT_IDX>(p + i);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float f;
            __builtin_memcpy(&f, &w[k], 4);
            tab[k] = (T)f;
        }
    }
    if (a.tab_neg) {
#pragma unroll
        for (int k = 0; k < N; ++k) tab[k] = -tab[k];
    }
}

// ---------------------------------------------------------------------------
// Factor side, register classes (thread per factor).
// factor_costs_for_var, maxsum.py:382-447:  out_i[d] = min over the other
// variables' values of  table[..] + sum of their messages.
// ---------------------------------------------------------------------------
template <typename T, int D, int NT = MXS_NT>
__device__ __forceinline__ void factor_unary(const SweepArgs<T>& a, const ClassInfo& ci, int j) {
    MXS_NT_FLAGS(NT);
    constexpr int H = Msg<T, D>::H;
    const int64_t fo = ci.f2v_base + (int64_t)j * H;
    const int e = ci.edge_base + j;
    constexpr bool CIM = Msg<T, D>::CNT_IN_MSG;
    T out[D], prev[D], tab[D];
    uint8_t cn;
    if constexpr (CIM) cn = Msg<T, D>::template load_c<NT_PREV>(a.f2v_old + fo, prev);
    else {
        Msg<T, D>::template load<NT_PREV>(a.f2v_old + fo, prev);
        cn = a.cF[e];
    }
    load_table<T, D, NT>(a, ci, j, tab);
#pragma unroll
    for (int d = 0; d < D; ++d)  // a single assignment of "the others": f_val + sum_cost with sum_cost = 0
        out[d] = tab[d] + (T)0;
    uint8_t c = 0;
    if (!a.start)  // on_start (maxsum.py:311-319): unary factors send in every mode, counter stays 0
        c = damp_and_filter<T, D>(out, prev, cn, a.damp_f != 0, a.damping, a.stability);
    if constexpr (CIM) {
        Msg<T, D>::store_c(a.f2v_new + fo, out, c);
    } else {
        Msg<T, D>::store(a.f2v_new + fo, out);
        a.cF[e] = c;
    }
}

template <typename T, int D, bool P2P = false, int NT = MXS_NT, bool CUT_HALF = true>
__device__ __forceinline__ void factor_binary(const SweepArgs<T>& a, const ClassInfo& ci, int j) {
    MXS_NT_FLAGS(NT);
    constexpr int H = Msg<T, D>::H;
    const int e = ci.edge_base + 2 * j;
    // the class's records are split by scope position: two dense streams
    const int64_t fo0 = ci.f2v_base + (int64_t)j * H, fo1 = ci.f2v_base1 + (int64_t)j * H;
    // everything addressed by j: coalesced
    constexpr bool CIM = Msg<T, D>::CNT_IN_MSG;
    // A shard's cut class computes only the message to its OWN variable (ClassInfo::own_pos, block-uniform): the
    // other one would go to a ghost variable nobody sweeps here -- its record, the owned variable's V->F message
    // it would be computed from and its counter are neither read nor written.
    // (CUT_HALF = false: the scheduled launch, which never holds a cut class -- no branch in the metric's instantiation)
    const bool do0 = !CUT_HALF || ci.own_pos != 2, do1 = !CUT_HALF || ci.own_pos != 1;
    const int v0 = do1 ? ldp<NT_IDX>(a.edge_v2f + e) : 0, v1 = do0 ? ldp<NT_IDX>(a.edge_v2f + e + 1) : 0;
    uint8_t cn0 = 0, cn1 = 0;
    T m0[D], p0[D], m1[D], p1[D], tab[D * D];
#pragma unroll
    for (int d = 0; d < D; ++d) m0[d] = p0[d] = m1[d] = p1[d] = (T)0;
    if constexpr (CIM) {
        if (do0) cn0 = Msg<T, D>::template load_c<NT_PREV>(a.f2v_old + fo0, p0);  // F->V message last sent to variable 0
        if (do1) cn1 = Msg<T, D>::template load_c<NT_PREV>(a.f2v_old + fo1, p1);  // (+ its send counter)
    } else {
        if (do0) {
            cn0 = a.cF[e];
            Msg<T, D>::template load<NT_PREV>(a.f2v_old + fo0, p0);
        }
        if (do1) {
            cn1 = a.cF[e + 1];
            Msg<T, D>::template load<NT_PREV>(a.f2v_old + fo1, p1);
        }
    }
    load_table<T, D * D, NT>(a, ci, j, tab);
    // the two gathers
    // (peer-store mode: the message of a ghost variable lives in the ghost region)
    if constexpr (P2P) {
        // a ghost record was stored by another GPU while this launch was already running
        if (do1) {
            if (v0 >= a.ghost_lo) Msg<T, D>::load_sys(a.ghost_old + (v0 - a.ghost_lo), m0);
            else Msg<T, D>::load(a.v2f_old + v0, m0);
        }
        if (do0) {
            if (v1 >= a.ghost_lo) Msg<T, D>::load_sys(a.ghost_old + (v1 - a.ghost_lo), m1);
            else Msg<T, D>::load(a.v2f_old + v1, m1);
        }
    } else {
        if (do1) Msg<T, D>::template load<NT_GATHER>(a.v2f_old + v0, m0);      // V->F message of scope variable 0
        if (do0) Msg<T, D>::template load<NT_GATHER>(a.v2f_old + v1, m1);
    }
    T o0[D], o1[D];
#pragma unroll
    for (int x = 0; x < D; ++x) {
        T best0 = pos_inf<T>(), best1 = pos_inf<T>();
#pragma unroll
        for (int y = 0; y < D; ++y) {
            const T c0 = tab[x * D + y] + ((T)0 + m1[y]);  // to var 0, value x; other = var 1
            if (best0 > c0) best0 = c0;
            const T c1 = tab[y * D + x] + ((T)0 + m0[y]);  // to var 1, value x; other = var 0
            if (best1 > c1) best1 = c1;
        }
        o0[x] = best0;
        o1[x] = best1;
    }
    uint8_t c0 = 0, c1 = 0;
    if (a.start) {  // only start_messages == all makes a binary factor send (maxsum.py:320-328)
        const bool sends = a.start_mode == MXS_START_ALL;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            o0[d] = sends ? o0[d] : (T)0;
            o1[d] = sends ? o1[d] : (T)0;
        }
    } else {
        c0 = damp_and_filter<T, D>(o0, p0, cn0, a.damp_f != 0, a.damping, a.stability);
        c1 = damp_and_filter<T, D>(o1, p1, cn1, a.damp_f != 0, a.damping, a.stability);
    }
    const int lw = (int)threadIdx.x & 63;
    if constexpr ((64 * H * sizeof(T)) % 16 == 0) {
        if ((j - lw + 63) < ci.count) {  // wave-uniform: a whole wave of factors
            T full[H];
            if (do0) {
#pragma unroll
                for (int d = 0; d < H; ++d) full[d] = Msg<T, D>::padded(o0, CIM ? c0 : 0, d);
                wave_store_linear<T, H, NT>(a.f2v_new, fo0 - (int64_t)lw * H, full);
            }
            if (do1) {
#pragma unroll
                for (int d = 0; d < H; ++d) full[d] = Msg<T, D>::padded(o1, CIM ? c1 : 0, d);
                wave_store_linear<T, H, NT>(a.f2v_new, fo1 - (int64_t)lw * H, full);
            }
        } else {
            if (do0) Msg<T, D>::store_c(a.f2v_new + fo0, o0, CIM ? c0 : 0);
            if (do1) Msg<T, D>::store_c(a.f2v_new + fo1, o1, CIM ? c1 : 0);
        }
    } else {
        if (do0) Msg<T, D>::store_c(a.f2v_new + fo0, o0, CIM ? c0 : 0);
        if (do1) Msg<T, D>::store_c(a.f2v_new + fo1, o1, CIM ? c1 : 0);
    }
    if constexpr (!CIM) {
        if (do0) a.cF[e] = c0;
        if (do1) a.cF[e + 1] = c1;
    }
}

// ---------------------------------------------------------------------------
// Factor side, generic class: thread per edge, any arity / domain sizes, scalar
// loops only (no local arrays -> no scratch).
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T factor_gen_value(const SweepArgs<T>& a, const FactorGen& fg, int pos,
                                              int d, int64_t others) {
    T best = pos_inf<T>();
    for (int64_t lin = 0; lin < others; ++lin) {
        int64_t rem = others, l = lin, t = 0;
        T sum_cost = (T)0;
        for (int i = 0; i < fg.arity; ++i) {  // scope in dimensions order
            const int e = fg.edge_base + i;
            const int Di = a.edge_dom[e];
            int digit;
            if (i == pos) {
                digit = d;
            } else {
                rem /= Di;
                digit = (int)(l / rem);
                l -= (int64_t)digit * rem;
                sum_cost += a.v2f_old[a.edge_v2f[e] + digit];
            }
            t = t * Di + digit;
        }
        const T cur = a.tables[fg.tab_off + t] + sum_cost;
        if (best > cur) best = cur;
    }
    return best;
}

template <typename T>
__device__ __forceinline__ void factor_generic(const SweepArgs<T>& a, const ClassInfo& ci, int j) {
    const int e = ci.edge_base + j;
    const FactorGen fg = a.fgen[a.edge_gen_factor[e]];
    const int pos = e - fg.edge_base;
    const int D = a.edge_dom[e];
    const T* prev = a.f2v_old + a.f2v_off[e];
    T* w = a.f2v_new + a.f2v_off[e];
    int64_t others = 1;
    for (int i = 0; i < fg.arity; ++i)
        if (i != pos) others *= a.edge_dom[fg.edge_base + i];
    if (a.start) {
        const bool sends = (fg.arity == 1 && a.start_mode != MXS_START_ALL) ||
                           a.start_mode == MXS_START_ALL;
        for (int d = 0; d < D; ++d) w[d] = sends ? factor_gen_value(a, fg, pos, d, others) : (T)0;
        a.cF[e] = 0;
        return;
    }
    const uint8_t cnt = a.cF[e];
    const bool damp = cnt > 0 && a.damp_f;
    bool match = cnt > 0;
    for (int d = 0; d < D; ++d) {
        T m = factor_gen_value(a, fg, pos, d, others);
        const T p = prev[d];
        if (damp) m = a.damping * p + ((T)1 - a.damping) * m;
        if (match) match = comp_match(m, p, a.stability);
        w[d] = m;
    }
    uint8_t out = 1;
    if (match) {
        if (cnt < SAME_COUNT) {
            out = (uint8_t)(cnt + 1);
        } else {
            out = cnt;
            for (int d = 0; d < D; ++d) w[d] = prev[d];
        }
    }
    a.cF[e] = out;
}

// ---------------------------------------------------------------------------
// Variable side, packed class (1 <= deg <= 64): ONE LANE PER EDGE.  The variables
// of a wave have the same degree and sit side by side (lane = var*deg + k); a
// lane's slot, counter, variable id and previous V->F message are contiguous
// across lanes (coalesced), then every lane gathers exactly one F->V message --
// all gathers of the wave are in flight together.  The sums walk the variable's
// lanes in edge order with cross-lane reads, so the arithmetic order is still the
// reference's:
//   select_value      maxsum.py:584-620
//   costs_for_factor  maxsum.py:623-676  (the mean excludes the own cost)
// ---------------------------------------------------------------------------
template <typename T, int D, bool P2P = false, int NT = MXS_NT>
__device__ __forceinline__ void variable_pack(const SweepArgs<T>& a, const ClassInfo& ci, int item) {
    MXS_NT_FLAGS(NT);
    constexpr int H = Msg<T, D>::H;
    const int lane_id = item + (int)threadIdx.x;
    if (lane_id >= ci.count) return;  // whole waves (count is a multiple of 64)
    const int64_t pos = ci.ell_base + lane_id;
    // what the wave works on: one scalar load (the wave index is wave-uniform)
    const WaveMeta wm = a.vwave[__builtin_amdgcn_readfirstlane((int)(pos >> 6))];
    const uint32_t dn = (uint32_t)wm.deg_nv;
    const int deg = (int)(dn & 255u), nv = (int)((dn >> 8) & 255u);
    const int l = (int)threadIdx.x & 63;
    const int var = (int)(((uint32_t)l * (dn >> 16)) >> 15);  // l / deg (exact for l < 64)
    const int k = l - var * deg;                              // edge position in the variable
    const bool has = var < nv;
    const int v = wm.first_var + (has ? var : 0);
    const int32_t slot = ldp<NT_IDX>(a.vell + pos);
    const int32_t send_at = a.send_slot != nullptr ? a.send_slot[pos] : -1;
    constexpr bool CIM = Msg<T, D>::CNT_IN_MSG;
    T pv[D], in[D], c[D], b[D], m[D];
    const int64_t vo = ci.v2f_base + (int64_t)lane_id * H;
    uint8_t cnt;
    if constexpr (CIM) cnt = Msg<T, D>::template load_c<NT_PREV>(a.v2f_old + vo, pv);  // V->F message last sent on
    else {                                                            // this edge (+ its counter)
        cnt = a.cV[ci.cv_base + lane_id];
        Msg<T, D>::template load<NT_PREV>(a.v2f_old + vo, pv);
    }
    Msg<T, D>::template load<NT_GATHER>(a.f2v_old + (has ? slot : a.null_f2v), in);  // F->V held from this factor
#pragma unroll
    for (int d = 0; d < D; ++d) c[d] = a.var_cost[ci.cost_base + (int64_t)(v - ci.first) * D + d];
    int init = -1;
    if (a.start) init = a.init_idx[v];
    const int seg = l - (has ? k : 0);  // first lane of the variable
    // d outer / factors inner, as the reference sums (maxsum.py:607-610, 651-665):
    // sum_cost is ONE accumulator running through all of it
    T sum_cost = (T)0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        T bd = c[d], md = c[d];
        for (int kk = 0; kk < deg; ++kk) {
            const T x = __shfl(in[d], seg + kk, 64);
            bd += x;                   // select_value: every factor
            if (kk != k) {             // costs_for_factor: every factor but the target
                sum_cost += x;
                md += x;
            }
        }
        b[d] = bd;
        m[d] = md;
    }
    int best = 0;
    T best_c = b[0];
#pragma unroll
    for (int d = 1; d < D; ++d)
        if (b[d] < best_c) {
            best = d;
            best_c = b[d];
        }
    if (init >= 0) {  // value_selection(initial_value), maxsum.py:497-498
        best = init;
        best_c = (T)0;
    }
    if (has && k == 0) {
        a.sel[v] = best;
        a.belief[v] = best_c;
    }
    const T avg = sum_cost / (T)D;
#pragma unroll
    for (int d = 0; d < D; ++d) m[d] = m[d] - avg;
    uint8_t co = 0;
    if (a.start) {
        const bool sends = (deg == 1 && a.start_mode == MXS_START_LEAFS) ||
                           a.start_mode != MXS_START_LEAFS;
#pragma unroll
        for (int d = 0; d < D; ++d) m[d] = sends ? m[d] : (T)0;
    } else {
        co = damp_and_filter<T, D>(m, pv, cnt, a.damp_v != 0, a.damping, a.stability);
    }
    if (!has) {  // padding lane: keep its slot zero
        co = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) m[d] = (T)0;
    }
    if (send_at >= 0) {  // a cut edge: the record crosses to the shard that holds its factor's replica
        T* dst = a.send_out + (int64_t)send_at * H;
        if constexpr (P2P) {  // peer-store mode: straight into that shard's ghost region
            T* base = a.peer_dst[0];
            int first = a.peer_first[0];
#pragma unroll
            for (int q = 1; q < MXS_MAX_PEERS; ++q) {
                const bool ge = send_at >= a.peer_first[q];
                base = ge ? a.peer_dst[q] : base;
                first = ge ? a.peer_first[q] : first;
            }
            dst = base + (int64_t)(send_at - first) * H;
        }
        Msg<T, D>::store_c(dst, m, CIM ? co : 0);
    }
    if constexpr ((64 * H * sizeof(T)) % 16 == 0) {  // the whole wave is here (count is a multiple of 64)
        T full[H];
#pragma unroll
        for (int d = 0; d < H; ++d) full[d] = Msg<T, D>::padded(m, CIM ? co : 0, d);
        wave_store_linear<T, H, NT>(a.v2f_new, vo - (int64_t)l * H, full);
    } else {
        Msg<T, D>::store(a.v2f_new + vo, m);
    }
    if constexpr (!CIM) a.cV[ci.cv_base + lane_id] = co;
}

// ---------------------------------------------------------------------------
// Variable side, packed class for domains of 5..8 values (round 5; K_V_PACK8, own launch): the lane-per-edge
// scheme of variable_pack with the record length of these domains -- 8 elements, whatever D (half_stride: 5..8
// values pad to 64 / 32 bytes) -- and the variable's own D at run time: elements past the domain are zeros on
// the way in (the `sum_cost` chain adds +0: exact, the accumulator starts at +0 and never becomes -0), are
// skipped by the selection and the send rule, and are stored as zeros.  What graph colouring with 5..8 colours
// (graphcoloring.py:271) puts on the variable side: until round 5 k_variable_wide's workgroup-per-run staging,
// 32 us per cycle for the 77 MB of coloring_100k_d8 (profiles/r05_kernel_stats_serial_coloring_100k_d8_f64_v1.csv).
// Arithmetic as variable_pack: select_value maxsum.py:584-620, costs_for_factor :623-676.
// ---------------------------------------------------------------------------
constexpr int PACK8_D = 8;
// (a device function: the class's workgroups run as a launch of their own, k_variable_pack8, or as the first workgroups of a
// lane-grid factor launch, bin_box.h -- two latency-bound launches of a cache-resident cycle overlap only inside ONE grid)
template <typename T>
__device__ __forceinline__ void variable_pack8_block(const SweepArgs<T>& a, const ClassInfo& ci, int block) {
    constexpr int H = PACK8_D;
    static_assert(half_stride(5, (int)sizeof(T)) == H && half_stride(8, (int)sizeof(T)) == H, "records of 5..8 values are 8 elements long");
    const int lane_id = block * BLOCK + (int)threadIdx.x;
    if (lane_id >= ci.count) return;  // whole waves (count is a multiple of 64)
    const int64_t pos = ci.ell_base + lane_id;
    const WaveMeta wm = a.vwave[__builtin_amdgcn_readfirstlane((int)(pos >> 6))];
    const uint32_t dn = (uint32_t)wm.deg_nv;
    const int deg = (int)(dn & 255u), nv = (int)((dn >> 8) & 255u);
    const int l = (int)threadIdx.x & 63;
    const int var = (int)(((uint32_t)l * (dn >> 16)) >> 15);  // l / deg (exact for l < 64)
    const int k = l - var * deg;                              // edge position in the variable
    const bool has = var < nv;
    const int v = wm.first_var + (has ? var : 0);
    const int32_t slot = a.vell[pos];
    const int D = ci.uni_D ? ci.uni_D : a.vdom[v];  // (one domain size in the class: no load, and the costs by arithmetic)
    const int64_t vo = ci.v2f_base + (int64_t)lane_id * H;
    const uint8_t cnt = a.cV[ci.cv_base + lane_id];
    T pv[H], in[H], c[H], m[H];
    Msg<T, H>::load(a.v2f_old + vo, pv);                               // V->F message last sent on this edge
    Msg<T, H>::load(a.f2v_old + (has ? slot : a.null_f2v), in);        // F->V held from this factor
    const T* cp = a.var_cost + (ci.uni_D ? ci.cost_base + (int64_t)(v - ci.first) * ci.uni_D : a.vcost_off[v]);
#pragma unroll
    for (int d = 0; d < H; ++d) {
        c[d] = d < D ? cp[d < D ? d : 0] : (T)0;
        in[d] = d < D ? in[d] : (T)0;
    }
    int init = -1;
    if (a.start) init = a.init_idx[v];
    const int seg = l - (has ? k : 0);  // first lane of the variable
    // d outer / factors inner, as the reference sums (maxsum.py:607-610, 651-665): ONE accumulator runs through all of it
    T sum_cost = (T)0, best_c = (T)0;
    int best = 0;
    // (a class of ONE domain size -- every variable of a SECP instance has five values -- stops at it: the elements past the
    // domain are zeros, `sum_cost + 0` leaves sum_cost as it is -- it starts at +0 and so never is -0 --, their m[] is zeroed
    // below and the selection skips them.  The bound is the class's, not the lane's: the cross-lane reads need whole waves.)
    const int DU = ci.uni_D ? ci.uni_D : H;
    static_for<H>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        T bd = c[d], md = c[d];
        if (d < DU) {
            for (int kk = 0; kk < deg; ++kk) {
                const T x = __shfl(in[d], seg + kk, 64);
                bd += x;                   // select_value: every factor
                if (kk != k) {             // costs_for_factor: every factor but the target
                    sum_cost += x;
                    md += x;
                }
            }
        }
        m[d] = md;
        if (d == 0 || (d < D && bd < best_c)) {  // first index attaining the minimum
            best = d;
            best_c = bd;
        }
    });
    if (init >= 0) {  // value_selection(initial_value), maxsum.py:497-498
        best = init;
        best_c = (T)0;
    }
    if (has && k == 0) {
        a.sel[v] = best;
        a.belief[v] = best_c;
    }
    const T avg = sum_cost / (T)D;
#pragma unroll
    for (int d = 0; d < H; ++d) m[d] = m[d] - avg;
    uint8_t co = 0;
    if (a.start) {
        const bool sends = (deg == 1 && a.start_mode == MXS_START_LEAFS) || a.start_mode != MXS_START_LEAFS;
#pragma unroll
        for (int d = 0; d < H; ++d) m[d] = sends ? m[d] : (T)0;
    } else {  // apply_damping + the send rule on the D live elements (damp_and_filter with a run-time length)
        if (cnt > 0 && a.damp_v) {
#pragma unroll
            for (int d = 0; d < H; ++d) m[d] = a.damping * pv[d] + ((T)1 - a.damping) * m[d];
        }
        bool match = cnt > 0;
#pragma unroll
        for (int d = 0; d < H; ++d) match = match && (d >= D || comp_match(m[d], pv[d], a.stability));
        if (!match) {
            co = 1;
        } else if (cnt < SAME_COUNT) {
            co = (uint8_t)(cnt + 1);
        } else {
            co = cnt;
#pragma unroll
            for (int d = 0; d < H; ++d) m[d] = pv[d];
        }
    }
#pragma unroll
    for (int d = 0; d < H; ++d) m[d] = (has && d < D) ? m[d] : (T)0;  // padding lanes and elements past the domain: zeros
    if (!has) co = 0;
    wave_store_linear<T, H>(a.v2f_new, vo - (int64_t)l * H, m);
    a.cV[ci.cv_base + lane_id] = co;
}
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_variable_pack8(SweepArgs<T> a, const ClassInfo* __restrict__ cls) {
    const ClassInfo ci = cls[0];
    variable_pack8_block<T>(a, ci, (int)blockIdx.x);
}

// Variable side, generic class: thread per variable, any domain size / degree,
// scalar loops only.  Sums run in the reference's order (d outer, factors inner).
template <typename T>
__device__ __forceinline__ void variable_generic(const SweepArgs<T>& a, const ClassInfo& ci, int j) {
    if (ci.start_only && !a.start) return;  // a variable without factor never cycles
    const int v = ci.first + j;
    const int D = a.vdom[v];
    const int k0 = a.vrowptr[v], k1 = a.vrowptr[v + 1];
    const int deg = k1 - k0;
    const T* c = a.var_cost + a.vcost_off[v];
    int best = 0;
    T best_c = (T)0;
    for (int d = 0; d < D; ++d) {
        T b = c[d];
        for (int k = k0; k < k1; ++k) b += a.f2v_old[a.vslot_f2v[k] + d];
        if (d == 0 || b < best_c) {
            best = d;
            best_c = b;
        }
    }
    if (a.start && a.init_idx[v] >= 0) {
        best = a.init_idx[v];
        best_c = (T)0;
    }
    a.sel[v] = best;
    a.belief[v] = best_c;
    const bool start_sends = (deg == 1 && a.start_mode == MXS_START_LEAFS) ||
                             a.start_mode != MXS_START_LEAFS;
    for (int ko = k0; ko < k1; ++ko) {
        const T* prev = a.v2f_old + a.vslot_v2f[ko];
        T* w = a.v2f_new + a.vslot_v2f[ko];
        T sum_cost = (T)0;
        for (int d = 0; d < D; ++d)
            for (int k = k0; k < k1; ++k)
                if (k != ko) sum_cost += a.f2v_old[a.vslot_f2v[k] + d];
        const T avg = sum_cost / (T)D;
        const uint8_t cnt = a.start ? 0 : a.cV[ko];
        const bool damp = cnt > 0 && a.damp_v;
        bool match = cnt > 0;
        for (int d = 0; d < D; ++d) {
            T m = c[d];
            for (int k = k0; k < k1; ++k)
                if (k != ko) m += a.f2v_old[a.vslot_f2v[k] + d];
            m = m - avg;
            if (a.start) {
                w[d] = start_sends ? m : (T)0;
                continue;
            }
            const T p = prev[d];
            if (damp) m = a.damping * p + ((T)1 - a.damping) * m;
            if (match) match = comp_match(m, p, a.stability);
            w[d] = m;
        }
        uint8_t out = 1;
        if (a.start) {
            out = 0;
        } else if (match) {
            if (cnt < SAME_COUNT) {
                out = (uint8_t)(cnt + 1);
            } else {
                out = cnt;
                for (int d = 0; d < D; ++d) w[d] = prev[d];
            }
        }
        a.cV[ko] = out;
    }
}

// ---------------------------------------------------------------------------
// Variable side, hub class (round 6): the variables no other class takes -- degree above 64 on a small domain, above
// 256 on any, deg * D > 1024 -- i.e. the hubs of a scale-free graph (graphcoloring.py:322-340, `--graph scalefree`:
// 100k variables, m = 2 -> degrees up to ~1 000).  The reference's costs_for_factor (maxsum.py:651-665) is, per
// OUTGOING edge, ONE serial accumulator `sum_cost` through all (d, f != target) in d-major order plus D serial sums
// msg_costs[d] over the same terms: O(deg * D) dependent additions per edge, O(deg^2 * D) per variable, none of which
// may be reassociated.  variable_generic walked all of a variable's edges in ONE thread (a 700-edge hub: 3e6
// dependent loads + additions, 555 ms per cycle: profiles/r06_before_coloring_100k_scalefree_f64.json).  Here a
// WORKGROUP takes HUB_EDGES outgoing edges of one variable (layout.h HubBlock), a lane per edge: the deg chains are
// independent, so the depth drops from O(deg^2 * D) to O(deg * D), the reference's order inside each chain untouched.
//   * steps of (a few values of d) x (up to HUB_TILE edges): the workgroup stages the F->V elements in_k[d] in LDS
//     ONCE for its four waves (every lane reads the SAME element next: broadcast reads, no conflicts); degree 700 on
//     three values is one step -- a step is two dependent round trips to memory (slot offsets, elements), and what
//     a block's life is made of is such round trips (~1.2 us each in a busy launch) beside the chains themselves
//     (~5 ns per element): the block's record (HubBlock) holds every per-variable quantity so that nothing else is
//     chained in front;
//   * per element ONE addition in the lane: two waves share 64 edges, one walks `sum_cost`, the other `msg_costs[d]`
//     (sum_cost crosses to its message's lane through LDS at the end).  The lane's own edge (k == ko) and the
//     padding of the last tile contribute -0.0, the exact additive identity of IEEE addition (y + -0.0 == y bit
//     for bit, for y = +-0, inf and NaN as well): no branch on the chains;
//   * msg[d] waits in registers (D <= 4; wider domains: parked in the lane's own record of v2f_new, which nobody
//     reads in this cycle) until sum_cost is complete, then it is normalised, damped and filtered like every
//     other message (maxsum.py:540-564);
//   * the lane one past the last edge leaves nothing out: its sums are the beliefs of select_value
//     (maxsum.py:607-610), the selection falls out of the same loop.
// Rides in the sweep launch (k_sweep_hub), its workgroups first in the grid.
// ---------------------------------------------------------------------------
// Workgroup barrier that orders the block's LDS traffic only: the global loads a wave has in flight stay
// in flight (__syncthreads() is a fence over every address space: it drains them, vmcnt(0)).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename T>
__device__ __forceinline__ void variable_hub(const SweepArgs<T>& a, const ClassInfo& ci, int item) {
    constexpr int R = HUB_LDS / BLOCK;
    static_assert(BLOCK == 128 * HUB_CW && (HUB_CW & (HUB_CW - 1)) == 0, "two waves (sum_cost, msg_costs) per 64 edges");
    __shared__ __attribute__((aligned(16))) T s_hub[HUB_LDS + 64];  // (+ slack: the read-ahead of the last row)
    __shared__ T s_sum[HUB_EDGES];                                  // sum_cost of every edge of the block
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const HubBlock hb = a.hub_blocks[ci.first + item];  // block-uniform: scalar loads, everything the block needs
    const int D = hb.D, k0 = hb.slot0, deg = hb.deg;
    // Two waves per 64 edges: wave g walks the `sum_cost` chains of edges ko0 + 64 g .., wave HUB_CW + g the `msg_costs[d]`
    // chains of the same edges -- ONE dependent addition per element and wave.  (A lone wave issues an instruction every
    // 4-8 cycles: with both chains in one lane the block was bound by its own issue rate, 9 ns per element where the
    // dependent addition costs 4: profiles/r06_hub_steps_v1.txt.)
    const int grp = wave & (HUB_CW - 1);
    const bool msg_role = wave >= HUB_CW;
    const int ko_w = hb.ko0 + (grp << 6);        // first edge of this wave
    const int ko = ko_w + lane;
    const bool wave_on = ko_w <= deg;            // (wave-uniform) some lane has an edge, or is the belief lane
    const bool real = msg_role && wave_on && ko < deg, bel = msg_role && wave_on && ko == deg;
    const T* c = a.var_cost + hb.cost_off;
    const T nzero = (T)-0.0;
    if (wave_on) __builtin_amdgcn_s_setprio(3);  // chains of dependent additions: whenever they can issue, they should
    // A step = ND values of d x NK edges: rows of ROW = 8 + NK elements -- the own cost c[d] (msg_costs[d] starts from it) and
    // seven fillers, then in_kt[d] .. in_{kt+NK-1}[d] (NK a multiple of 8; past the degree: -0.0).  A degree above HUB_TILE takes
    // several steps per d (ND = 1); a smaller one several d per step -- degree 700 on three values: ONE step.
    const int NK = deg < HUB_TILE ? ((deg + 7) & ~7) : HUB_TILE, ROW = NK + 8;
    const int ND = HUB_LDS / ROW;
    const uint32_t magic = hb.magic;  // ceil(2^32 / ROW): e / ROW for e < 2^16
    // what the epilogue needs of the lane's own edge, requested before anything else
    const int vo = a.vslot_v2f[k0 + (real ? ko : 0)];
    const uint8_t cnt = a.start ? 0 : a.cV[k0 + (real ? ko : 0)];
    // Every load is unconditional -- ONE index load and ONE value load per element, from a selected address (clamped indices,
    // no branch: the R requests of a step leave the wave back to back; with two candidate loads per element the compiler
    // built a branch and a full wait around each, 20 serial round trips per step).
    T pre[R];
    auto request = [&](int d0, int kt) __attribute__((always_inline)) {
        int off[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = tid + BLOCK * r;
            const int dd = (int)(((uint64_t)(uint32_t)e * magic) >> 32), p = e - dd * ROW;
            const int k = kt + p - 8;
            const bool is_x = dd < ND && d0 + dd < D && p >= 8 && k < deg;
            off[r] = a.vslot_f2v[k0 + (is_x ? k : 0)];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = tid + BLOCK * r;
            const int dd = (int)(((uint64_t)(uint32_t)e * magic) >> 32), p = e - dd * ROW;
            const int dc = d0 + dd, k = kt + p - 8;
            const bool row_ok = dd < ND && dc < D;
            const bool is_x = row_ok && p >= 8 && k < deg, is_c = row_ok && p == 0;
            const T* src = is_x ? a.f2v_old + (off[r] + dc) : c + (is_c ? dc : 0);
            pre[r] = *src;
        }
    };
    // msg_costs[d] until sum_cost is complete: in registers for D <= 4, else parked in the lane's own record of v2f_new
    T mreg[4] = {(T)0, (T)0, (T)0, (T)0};
    const bool park = D > 4;
    T acc = (T)0, best_c = (T)0;  // the wave's chain: sum_cost (one accumulator through every d) or msg_costs[d]
    int best = 0;
    int d0 = 0, kt = 0;
    request(0, 0);
    while (d0 < D) {
        lds_barrier();  // the previous step's reads are over
#pragma unroll
        for (int r = 0; r < R; ++r) {  // own cost | message element | filler (-0.0)
            const int e = tid + BLOCK * r;
            const int dd = (int)(((uint64_t)(uint32_t)e * magic) >> 32), p = e - dd * ROW;
            const bool row_ok = dd < ND && d0 + dd < D;
            s_hub[e] = (row_ok && (p == 0 || (p >= 8 && kt + p - 8 < deg))) ? pre[r] : nzero;
        }
        lds_barrier();
        // the step after this one: the next edges of this d, or the next values of d
        const bool last_k = kt + NK >= deg;
        const int d0n = last_k ? d0 + ND : d0, ktn = last_k ? 0 : kt + NK;
        if (d0n < D) request(d0n, ktn);
        const int n = deg - kt < NK ? deg - kt : NK;
        auto walk = [&](auto msg_c) __attribute__((always_inline)) {
            constexpr bool MSG = decltype(msg_c)::value;
            for (int dd = 0; dd < ND && d0 + dd < D; ++dd) {
                const T* row = s_hub + dd * ROW;
                if (MSG && kt == 0) acc = row[0];  // msg_costs[d] = cost_for_val(d), maxsum.py:648
                // Three runs of the row: before, inside and behind the 64 edges this wave's lanes own -- only the middle one
                // pays the selects (k == ko contributes -0.0).  A run walks blocks of eight elements, the reads of the
                // blocks ahead requested from LDS before the additions of the current one (three register sets by hand:
                // the chain waits for nothing but itself); no branch but the loop's own.
                auto fetch = [&](T (&x)[8], int kk) __attribute__((always_inline)) {
#if defined(MXS_HUB_EXP) && MXS_HUB_EXP == 3   // timing experiment (results wrong): the chain without its LDS reads
                    if (kk > 16) return;
#endif
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = row[8 + kk + u];
                };
                auto consume = [&](T (&x)[8], int kk, auto sel) __attribute__((always_inline)) {
                    if constexpr (decltype(sel)::value) {
                        const int kabs = kt + kk;
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = (kabs + u == ko) ? nzero : x[u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += x[u];
                };
                auto run = [&](int k_lo, int k_hi, auto sel) __attribute__((always_inline)) {  // [k_lo, k_hi): multiples of 8
                    if (k_lo >= k_hi) return;
                    T xa[8], xb[8], xc[8];
                    fetch(xa, k_lo);
                    fetch(xb, k_lo + 8);
                    int kk = k_lo;
                    for (; kk + 24 <= k_hi; kk += 24) {  // (reads past the run's end: the row's next blocks or the slack, never used)
                        fetch(xc, kk + 16);
                        consume(xa, kk, sel);
                        fetch(xa, kk + 24);
                        consume(xb, kk + 8, sel);
                        fetch(xb, kk + 32);
                        consume(xc, kk + 16, sel);
                    }
                    if (kk < k_hi) consume(xa, kk, sel);
                    if (kk + 8 < k_hi) consume(xb, kk + 8, sel);
                };
                const int n8 = (n + 7) & ~7;  // (the row is padded with -0.0 to whole blocks)
                const int w_lo = ko_w - kt < 0 ? 0 : (ko_w - kt > n8 ? n8 : ((ko_w - kt) & ~7));
                const int w_hi = ko_w + 64 - kt < 0 ? 0 : (ko_w + 64 - kt > n8 ? n8 : ((ko_w + 64 - kt + 7) & ~7));
                run(0, w_lo, std::false_type{});
                run(w_lo, w_hi, std::true_type{});
                run(w_hi, n8, std::false_type{});
                if (MSG && last_k) {  // msg_costs[d] is complete
                    const int d = d0 + dd;
                    if (park) {
                        if (real) a.v2f_new[vo + d] = acc;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) mreg[q] = d == q ? acc : mreg[q];
                    }
                    if (d == 0 || acc < best_c) {  // (the belief lane's: first index attaining the minimum)
                        best = d;
                        best_c = acc;
                    }
                }
            }
        };
        if (wave_on) {
            if (msg_role) walk(std::true_type{});
            else walk(std::false_type{});
        }
        d0 = d0n;
        kt = ktn;
    }
    if (!msg_role && wave_on) s_sum[(grp << 6) + lane] = acc;  // sum_cost of edge ko: to the lane that holds its msg_costs
    lds_barrier();
    if (bel) {
        const int v = hb.var;
        if (a.start && a.init_idx[v] >= 0) {  // value_selection(initial_value), maxsum.py:497-498
            best = a.init_idx[v];
            best_c = (T)0;
        }
        a.sel[v] = best;
        a.belief[v] = best_c;
    }
    if (!real) return;
    // ---- normalise, damp, filter (the lane's own edge) ---------------------------------------
    const T avg = s_sum[(grp << 6) + lane] / (T)D;
    const T* prev = a.v2f_old + vo;
    T* w = a.v2f_new + vo;
    const bool start_sends = (deg == 1 && a.start_mode == MXS_START_LEAFS) || a.start_mode != MXS_START_LEAFS;
    const bool damp = cnt > 0 && a.damp_v;
    bool match = cnt > 0;
    for (int dd = 0; dd < D; ++dd) {
        T mm = park ? w[dd] : (dd == 0 ? mreg[0] : dd == 1 ? mreg[1] : dd == 2 ? mreg[2] : mreg[3]);
        mm = mm - avg;
        if (a.start) {
            w[dd] = start_sends ? mm : (T)0;
            continue;
        }
        const T p = prev[dd];
        if (damp) mm = a.damping * p + ((T)1 - a.damping) * mm;
        if (match) match = comp_match(mm, p, a.stability);
        w[dd] = mm;
    }
    uint8_t out = 1;
    if (a.start) {
        out = 0;
    } else if (match) {
        if (cnt < SAME_COUNT) {
            out = (uint8_t)(cnt + 1);
        } else {  // not sent: the receiver keeps the old message
            out = cnt;
            for (int dd = 0; dd < D; ++dd) w[dd] = prev[dd];
        }
    }
    a.cV[k0 + ko] = out;
}

// ---------------------------------------------------------------------------
// The sweep: one block = up to blockDim.x items of one class.  The class and the
// first item of a block follow from blockIdx and the class table in the kernel
// arguments.  DSEL != 0 instantiates the register / wave paths for that domain
// size only (the engine picks it when the graph has a single D), which keeps the
// kernel's register allocation -- the maximum over all paths -- small.
// ---------------------------------------------------------------------------
template <typename T, int D, bool P2P = false, int NT = MXS_NT, bool CUT_HALF = true>
__device__ __forceinline__ void sweep_d(const SweepArgs<T>& a, const ClassInfo& ci, int item) {
    if (ci.kind == K_V_PACK) {  // one lane per edge
        variable_pack<T, D, P2P, NT>(a, ci, item);
        return;
    }
    const int j = item + (int)threadIdx.x;
    if (j >= ci.count) return;
    if (ci.kind == K_F_BIN) factor_binary<T, D, P2P, NT, CUT_HALF>(a, ci, j);
    else if (ci.kind == K_F_UNARY) factor_unary<T, D, NT>(a, ci, j);
}

// A block of a cut factor class in the fused sharded launch: the ghost V->F messages it
// gathers are delivered by the halo exchange of the previous cycle (comm stream), whose
// unpack kernel publishes its number in halo_flags[0].  These blocks are the LAST of the
// grid and the exchange had the whole launch to finish, so normally nothing waits; when
// it does, lane 0 polls with a sleep.  A wait that exceeds ~5 s (100 MHz ticks) sets an
// error bit the host reports at the next sync instead of hanging the GPU.
constexpr unsigned long long HALO_WAIT_TICKS = 500000000ull;
constexpr int HALO_ERR_WORD = 32;  // halo_flags[32]: error bits (flags[0..] are epochs)
// n_peers == 0: one epoch word, written by this GPU's publish kernel (fused launch);
// n_peers  > 0: one word per rank, written by that rank's publish kernel over xGMI.
__device__ __forceinline__ void wait_for_halo(uint32_t* flags, uint32_t need, int n_peers, int me) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        const int n = n_peers > 0 ? n_peers : 1;
        for (int q = 0; q < n; ++q) {
            if (n_peers > 0 && q == me) continue;
            while (__hip_atomic_load(flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need &&
                   __hip_atomic_load(flags + HALO_ERR_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(64);
                if (wall_clock64() - t0 > HALO_WAIT_TICKS) {
                    atomicOr(flags + HALO_ERR_WORD, 1u);  // the others fail fast
                    break;
                }
            }
        }
        // The polling lane ends with an ACQUIRE at the scope of the writers (other GPUs: system)
        // -- the documented pairing with the publisher's release; workgroup scope is not an
        // acquire for data written by other CUs or devices.  The barrier below then orders every
        // wave of the block behind it.
        if (n_peers > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    if (n_peers > 0) {
        // peer-store mode: the ghost records themselves are read with system-scope loads
        // (Msg::load_sys), which no cache of this GPU serves stale.  (A per-WAVE agent-scope
        // fence here, i.e. an L2 invalidation per wave of every cut block, measured 37.5
        // instead of 28 us per launch with 535 cut blocks; the per-block system-scope acquire
        // of the polling lane above is the price of correctness over xGMI and is paid only in
        // this opt-in mode.)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        // fused launch over RCCL: the unpack kernel wrote the ghost slots through another XCD's
        // L2; they must not be served from a stale line of this one
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

// SCHED: the launch has a block schedule (launch 0 of every cycle: a.sched != NULL) -- the instantiation
// then carries neither the compare chain on block_base[] nor the halo wait of the cut classes (which run
// in launches without schedule), and its prologue is two dependent scalar loads (schedule word, class
// record) instead of a chain of argument pieces.
// HUB: the launch may hold the K_V_HUB class (its LDS tile and its registers are in that instantiation only).
template <typename T, int DSEL, bool P2P = false, int NT = MXS_NT, bool SCHED = false, bool HUB = false>
__device__ __forceinline__ int sweep_block(const SweepArgs<T>& a) {
    int c = 0, blk = -1;
    if (SCHED || a.sched != nullptr) {
        const uint32_t s = a.sched[blockIdx.x];
        c = (int)(s >> 24);
        blk = (int)(s & 0xffffffu);
    } else {
#pragma unroll
        for (int i = 1; i < MAX_CLASSES; ++i) c += ((int)blockIdx.x >= a.block_base[i]) ? 1 : 0;
    }
    const ClassInfo ci = a.classes[c];
    // (Forcing the whole class record into SGPRs in one round of scalar loads -- two dependent rounds fewer
    // before the first vector load -- measured no gain: profiles/r03_class_preload_ab_v1.txt.)
    const int item = (blk >= 0 ? blk : (int)blockIdx.x - ci.block_base) * ci.per_block;
    if (!SCHED && ci.wait_halo && a.halo_flags != nullptr) wait_for_halo(a.halo_flags, a.need_epoch, a.n_peers, a.me);
    if (HUB && ci.kind == K_V_HUB) {
        variable_hub<T>(a, ci, item);
        return ci.kind;
    }
    if (ci.kind == K_F_GEN || ci.kind == K_V_GEN) {
        const int j = item + (int)threadIdx.x;
        if (j >= ci.count) return ci.kind;
        if (ci.kind == K_F_GEN) factor_generic<T>(a, ci, j);
        else variable_generic<T>(a, ci, j);
        return ci.kind;
    }
    // (Several consecutive tiles per workgroup -- fewer, longer blocks, one generation of them on
    // the 100k instance -- measured slower everywhere: 2 / 3 / 4 tiles 24.4 / 23.9 / 26.1 us against
    // 20.5 on coloring_100k, 274 / 277 / 284 against 259 on the 1M instance, 10.3 / 13.6 / 16.5 against
    // 6.4 on the 10k one: profiles/r03_tiles_ab_v1.txt.)
    if (DSEL != 0) {
        sweep_d<T, (DSEL != 0 ? DSEL : 2), P2P, NT, !SCHED>(a, ci, item);
    } else {
        switch (ci.D) {
            case 2: sweep_d<T, 2, P2P, NT, !SCHED>(a, ci, item); break;
            case 3: sweep_d<T, 3, P2P, NT, !SCHED>(a, ci, item); break;
            case 4: sweep_d<T, 4, P2P, NT, !SCHED>(a, ci, item); break;
            default: break;
        }
    }
    return ci.kind;
}

// At most 80 SGPRs: a CU admits 8 workgroups of 256 threads only up to that count (6 at the
// 98 the compiler would otherwise use -- MI355X_MICROARCH.md, "Residency"; seen as 1536
// instead of 2048 resident blocks in the per-block timeline).
// (Forcing the D = 2 / 3 instantiations to 64 VGPRs -- 8 waves per SIMD, 2 048 resident workgroups instead of
// 1 792 -- changes nothing measurable: profiles/r03_sweep_waves_ab_v1.txt.)
// SGPR budget per instantiation: what keeps the occupancy the VGPRs allow.  D = 2 (53-56 VGPRs, 8 waves per
// SIMD) must stay within 80 SGPRs -- 8 workgroups per CU; D = 3 runs 7 waves per SIMD on its 69 VGPRs
// anyway, and with 104 SGPRs its prologue (kernel arguments, class record) holds what it loaded instead of
// spilling and reloading it: coloring_100k 18.75 -> 18.15 us, 10k 6.5 -> 5.9, the 1M instance 244 -> 234;
// Ising (D = 2) with 104: 99 -> 101 us, f32 71.5 -> 78.5 (`profiles/r03_sweep_sgpr_ab_v1.txt`).
#ifndef SWEEP_SGPRS
#define SWEEP_SGPRS 104
#endif
template <typename T, int DSEL, int NT = MXS_NT, bool SCHED = false>
__global__ void __launch_bounds__(BLOCK, SWEEP_MIN_WAVES) __attribute__((amdgpu_num_sgpr(SWEEP_SGPRS)))
k_sweep(SweepArgs<T> a) {
    sweep_block<T, DSEL, false, NT, SCHED>(a);
}
// (the attribute takes no template-dependent value: the D = 2 instantiations are a kernel of their own)
template <typename T, int NT = MXS_NT, bool SCHED = false>
__global__ void __launch_bounds__(BLOCK, SWEEP_MIN_WAVES) __attribute__((amdgpu_num_sgpr(80)))
k_sweep_d2(SweepArgs<T> a) {
    sweep_block<T, 2, false, NT, SCHED>(a);
}

// The sweep of a graph with hub variables (K_V_HUB): a kernel of its own, so that the hub path's LDS tile and registers cost
// the plain sweep nothing.  DSEL 3 (colourings) or 0.
template <typename T, int DSEL, int NT = MXS_NT, bool SCHED = false>
__global__ void __launch_bounds__(BLOCK, SWEEP_MIN_WAVES) k_sweep_hub(SweepArgs<T> a) {
    sweep_block<T, DSEL, false, NT, SCHED, true>(a);
}

// The sweep of a shard in peer-store mode (engine.hip, p2p): same blocks, plus the stores of
// cut-edge records into the peers' ghost regions and the ghost addressing of the cut factors.
// A kernel of its own so that these cost the plain sweep no register.
template <typename T, int DSEL>
__global__ void __launch_bounds__(BLOCK, SWEEP_MIN_WAVES) k_sweep_p2p(SweepArgs<T> a) {
    sweep_block<T, DSEL, true, MXS_NT, false, true>(a);
}

// Profiling twin (mxs_debug_timeline): when did each block start, when were its stores done.
// A kernel of its own so that the instrumentation costs the real one no register.
template <typename T, int DSEL, bool HUB>
__device__ __forceinline__ void sweep_timeline(const SweepArgs<T>& a) {
    const int64_t t0 = (int64_t)wall_clock64();
    const int kind = sweep_block<T, DSEL, false, MXS_NT, false, HUB>(a);
    __builtin_amdgcn_s_waitcnt(0);  // loads back, stores acknowledged
    const int64_t t1 = (int64_t)wall_clock64();
    if (threadIdx.x == 0) {
        a.timeline[3 * (int64_t)blockIdx.x + 0] = t0;
        a.timeline[3 * (int64_t)blockIdx.x + 1] = t1;
        a.timeline[3 * (int64_t)blockIdx.x + 2] = kind;
    }
}
template <typename T, int DSEL>
__global__ void __launch_bounds__(BLOCK, SWEEP_MIN_WAVES) k_sweep_timeline(SweepArgs<T> a) {
    sweep_timeline<T, DSEL, false>(a);
}
// (with the hub class: the register budget of k_sweep_hub, which carries no clock reads -- no occupancy bound here)
template <typename T, int DSEL>
__global__ void __launch_bounds__(BLOCK) k_sweep_timeline_hub(SweepArgs<T> a) {
    sweep_timeline<T, DSEL, true>(a);
}

// ---------------------------------------------------------------------------
// Factor side, workgroup-per-factor class (arity 2..5, 64 <= R <= 1024 where R is
// the product of the dimensions after the first): factor_costs_for_var
// (maxsum.py:382-447) for tables too large for one thread.
//
// The table [D0][R] is read from HBM exactly once, coalesced (lane <-> q in [0,R),
// loop over d0); every entry feeds all `arity` outputs at once:
//   * outputs to variables 1.. : the digits of q are fixed while d0 runs, so the
//     running minima live in registers and reach LDS with ONE atomic per (q, p);
//   * output to variable 0     : for each d0 the minimum over all q is a wavefront
//     reduction (cross-lane min) followed by one LDS atomic per wave.
// Minima are exact and order-independent, so reducing in any order reproduces the
// reference's sequential `optimal > current` scan (maxsum.py:439-443) bit for bit;
// each candidate is evaluated with the reference's own expression
//   table[...] + (((0 + m_a) + m_b) + ...)   others in dimensions order (:425-438).
// The incoming V->F messages are staged in LDS; minima are combined in LDS as
// order-preserving integer keys (ds_min_u64 / ds_min_u32).
// ---------------------------------------------------------------------------
constexpr int NARY_MAX_SUMD = 1024;  // sum of the scope's domain sizes
constexpr int NARY_MAX_R = 1024;     // BLOCK * NARY_NJ
constexpr int NARY_MAX_NJ = NARY_MAX_R / BLOCK;
constexpr int NARY_MAX_ARITY = 6;

template <typename T>
struct OrdKey;
template <>
struct OrdKey<double> {
    typedef unsigned long long U;
    static __device__ __forceinline__ U enc(double x) {
        U b;
        memcpy(&b, &x, 8);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    }
    static __device__ __forceinline__ double dec(U k) {
        const U b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        double x;
        memcpy(&x, &b, 8);
        return x;
    }
};
template <>
struct OrdKey<float> {
    typedef unsigned int U;
    static __device__ __forceinline__ U enc(float x) {
        U b;
        memcpy(&b, &x, 4);
        return (b >> 31) ? ~b : (b | 0x80000000u);
    }
    static __device__ __forceinline__ float dec(U k) {
        const U b = (k >> 31) ? (k & 0x7fffffffu) : ~k;
        float x;
        memcpy(&x, &b, 4);
        return x;
    }
};

template <typename T>
__device__ __forceinline__ T wave_min(T x) {  // all 64 lanes get the minimum
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const T y = __shfl(x, (int)((threadIdx.x & 63) ^ s), 64);
        x = y < x ? y : x;
    }
    return x;
}

// min(a, b) as one instruction.  Equal to the reference's `if (best > cur) best = cur`
// scan for every value but the sign of a zero minimum, which nothing downstream observes.
// (f64: the instruction itself.  Through __builtin_fmin the compiler first canonicalises every operand it
// cannot prove canonical -- a `v_max_f64 x, x, x` per value that came out of a lane exchange or a conversion,
// 120 of the ~1 000 f64 instructions of the n-ary kernel -- to quiet signalling NaNs, which no cost is.)
__device__ __forceinline__ double min2(double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__) && MXS_ASM_MIN
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
#else
    return __builtin_fmin(x, y);
#endif
}
__device__ __forceinline__ float min2(float x, float y) { return __builtin_fminf(x, y); }

// Wavefront minimum with DPP moves instead of ds_bpermute shuffles: quad swaps, row half mirror,
// row mirror (every lane of a 16-lane row then holds the row minimum), row_bcast15 / row_bcast31
// (rows 1,3 <- lane 15 of rows 0,2; rows 2,3 <- lane 31): LANE 63 ends up with the minimum of the
// wave.  Plain VALU moves: nothing goes through the LDS crossbar.
#ifndef MXS_NARY_DPP
#define MXS_NARY_DPP 1  // measured: meeting_50k 1168 -> 1099 us (f32 681 -> 655), parity green
#endif
#if MXS_NARY_DPP  // (the emulated build of the CPU tests provides the same lane-selection rules: tests/emu/hip)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float x) {
    int b = __float_as_int(x);
    b = __builtin_amdgcn_update_dpp(b, b, CTRL, ROW_MASK, 0xf, false);
    return __int_as_float(b);
}
template <typename T>
__device__ __forceinline__ T wave_min_to_lane63(T x) {
    x = min2(x, dpp_mov<0xB1, 0xf>(x));   // quad_perm [1,0,3,2]
    x = min2(x, dpp_mov<0x4E, 0xf>(x));   // quad_perm [2,3,0,1]
    x = min2(x, dpp_mov<0x141, 0xf>(x));  // row_half_mirror
    x = min2(x, dpp_mov<0x140, 0xf>(x));  // row_mirror
    x = min2(x, dpp_mov<0x142, 0xa>(x));  // row_bcast15 -> rows 1, 3
    x = min2(x, dpp_mov<0x143, 0xc>(x));  // row_bcast31 -> rows 2, 3
    return x;
}
// The two registers a permlane swap of (x, x) leaves behind, per 32-bit half of T.
typedef unsigned int swap2u __attribute__((ext_vector_type(2)));
template <bool W32, bool SECOND>
__device__ __forceinline__ unsigned int swap_word(unsigned int w) {
    const swap2u r = W32 ? __builtin_amdgcn_permlane32_swap(w, w, false, false)
                         : __builtin_amdgcn_permlane16_swap(w, w, false, false);
    return SECOND ? r[1] : r[0];
}
template <bool W32, bool SECOND>
__device__ __forceinline__ double swap_val(double x) {
    const unsigned int lo = swap_word<W32, SECOND>((unsigned int)__double2loint(x));
    const unsigned int hi = swap_word<W32, SECOND>((unsigned int)__double2hiint(x));
    return __hiloint2double((int)hi, (int)lo);
}
template <bool W32, bool SECOND>
__device__ __forceinline__ float swap_val(float x) {
    return __int_as_float((int)swap_word<W32, SECOND>((unsigned int)__float_as_int(x)));
}
template <typename T> __device__ __forceinline__ T swap16_lo(T x) { return swap_val<false, false>(x); }
template <typename T> __device__ __forceinline__ T swap16_hi(T x) { return swap_val<false, true>(x); }
template <typename T> __device__ __forceinline__ T swap32_lo(T x) { return swap_val<true, false>(x); }
template <typename T> __device__ __forceinline__ T swap32_hi(T x) { return swap_val<true, true>(x); }
// FOUR wavefront minima at once, for less than the price of two: lanes trade values before they
// reduce them.  Step A (partner l^1): even lanes keep b[0], b[1], odd lanes b[2], b[3], each gets
// the partner's copies of what it keeps; step B (partner l^2): one value per lane is left --
// value index 2*(l&1) + ((l>>1)&1); then lanes l+4, l+8, l+12 of the row (row_ror keeps l & 3) and
// the other rows (l^16, l^32).  Every lane returns the wave minimum of ITS value index.
// 7 minima and 14 moves instead of 24 and 48.
#ifndef MXS_NARY_REDUCE4
#define MXS_NARY_REDUCE4 1  // measured: meeting_50k 774 -> 601 us (f32 483 -> 386), parity green
#endif
template <typename T>
__device__ __forceinline__ T wave_min4(const T (&b)[4]) {
    const int l = (int)threadIdx.x & 63;
    const bool odd = (l & 1) != 0, hi = (l & 2) != 0;
    T k0 = odd ? b[2] : b[0], k1 = odd ? b[3] : b[1];
    const T s0 = odd ? b[0] : b[2], s1 = odd ? b[1] : b[3];
    k0 = min2(k0, dpp_mov<0xB1, 0xf>(s0));
    k1 = min2(k1, dpp_mov<0xB1, 0xf>(s1));
    T k = hi ? k1 : k0;
    const T s = hi ? k0 : k1;
    k = min2(k, dpp_mov<0x4E, 0xf>(s));
    k = min2(k, dpp_mov<0x124, 0xf>(k));  // row_ror:4
    k = min2(k, dpp_mov<0x128, 0xf>(k));  // row_ror:8
    // lanes l ^ 16 and l ^ 32: gfx950's permlane swaps (VALU, no trip through the LDS crossbar that
    // ds_bpermute takes).  With the same value in both operands, v_permlane16_swap leaves rows {0,0,2,2}
    // in one register and {1,1,3,3} in the other, v_permlane32_swap the low half in both halves of one
    // and the high half in the other: their minimum is min(k[l], k[l ^ 16]) resp. min(k[l], k[l ^ 32]).
    k = min2(swap16_lo(k), swap16_hi(k));
    k = min2(swap32_lo(k), swap32_hi(k));
    return k;
}
#endif

constexpr int NARY_UNR = 4;  // values of d0 per batch: UNR * NJ table loads per lane in flight,
                             // and the next batch is requested before this one is reduced

// One batch of d0 values: `tv[u][j]` = table[d0 + u][q_j].  Every entry feeds all A outputs.
// MASKED: some (u, j) are out of range (tail batch / q >= R) and must not count.
// LS ("last digit the same", arity 3): the block size is a multiple of the last dimension, so a lane's NJ
// entries of a table row differ in the digit of dimension 1 only -- the message of dimension 2 they add is
// ONE value, and the sum (0 + m0) + m2 for the output to variable 1 is computed once per d0, not once per
// entry (24^3 tables on 192 lanes: 8 of the 96 f64 operations of a batch).  Same values, same bits.
template <typename T, int A, int NJ, bool MASKED, bool LS = false>
__device__ __forceinline__ void nary_batch(const T (&tv)[NARY_UNR][NJ], int d0, int D0, const T* s_m0,
                                           const T (&ms)[NJ][A], const T (&s0)[NJ], const bool (&live)[NJ],
                                           T (&acc)[NJ][A], typename OrdKey<T>::U* s_key0) {
    T best0[NARY_UNR];
#pragma unroll
    for (int u = 0; u < NARY_UNR; ++u) {
        // (a row past the table -- the tail batch of a first dimension that is no multiple of NARY_UNR: three of the eight rows a
        // five-value first variable makes the block walk -- is skipped, block-uniformly, instead of computed and masked)
        if (MASKED && d0 + u >= D0) {
            best0[u] = pos_inf<T>();
            continue;
        }
        const T m0 = s_m0[(!MASKED || d0 + u < D0) ? d0 + u : 0];
        const T a0 = m0;  // = 0 + the message, added when it was staged
        T b0 = pos_inf<T>();
        const T sp1_shared = (LS && A == 3) ? a0 + ms[0][A - 1] : (T)0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            T t = tv[u][j];
            if (MASKED) t = (live[j] && d0 + u < D0) ? t : pos_inf<T>();
            b0 = min2(b0, t + s0[j]);  // to variable 0: the others are 1..A-1 in dimensions order
            // to variable p >= 1: the others are 0 and the remaining ones, in order
#pragma unroll
            for (int p = 1; p < A; ++p) {
                T sp = a0;
                if (LS && A == 3 && p == 1) {
                    sp = sp1_shared;
                } else {
#pragma unroll
                    for (int i = 1; i < A; ++i)
                        if (i != p) sp += ms[j][i];
                }
                acc[j][p] = min2(acc[j][p], t + sp);
            }
        }
        best0[u] = b0;
    }
#if MXS_NARY_DPP && MXS_NARY_REDUCE4
    static_assert(NARY_UNR == 4, "wave_min4 reduces four values");
    {
        const T m = wave_min4(best0);
        const int l = (int)threadIdx.x & 63;
        const int u = 2 * (l & 1) + ((l >> 1) & 1);
        if (l < 4 && (!MASKED || d0 + u < D0)) atomicMin(&s_key0[d0 + u], OrdKey<T>::enc(m));
    }
    if (false) {
#elif MXS_NARY_DPP
#pragma unroll
    for (int u = 0; u < NARY_UNR; ++u) best0[u] = wave_min_to_lane63(best0[u]);
    if ((threadIdx.x & 63) == 63) {
#else
    // UNR independent wavefront reductions, interleaved step by step
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
#pragma unroll
        for (int u = 0; u < NARY_UNR; ++u)
            best0[u] = min2(best0[u], __shfl(best0[u], (int)((threadIdx.x & 63) ^ sft), 64));
    }
    if ((threadIdx.x & 63) == 0) {
#endif
#pragma unroll
        for (int u = 0; u < NARY_UNR; ++u)
            if (!MASKED || d0 + u < D0) atomicMin(&s_key0[d0 + u], OrdKey<T>::enc(best0[u]));
    }
}

// x / D and x % D for a block-uniform D with the host's magic number (NaryDesc::magic): a multiply-high
// and a multiply instead of the ~25 instructions of an emulated integer division (x < 2^16).
__device__ __forceinline__ void nary_divmod(int x, int D, uint32_t magic, int& q, int& r) {
    q = D == 1 ? x : (int)(((uint64_t)(uint32_t)x * magic) >> 32);
    r = x - q * D;
}

// dig[j][1..A-1] = the mixed-radix digits (dimensions 1..A-1, last fastest) of q_j = tid + j * NT (`tid`: the lane's first q),
// those of R - 1 where q_j is past the table.  Divisions for q_0 and for the block-uniform stride
// NT only; q_j = q_(j-1) + NT is a digit-wise addition with carry.
template <int A, int NJ>
__device__ __forceinline__ void nary_digits(int tid, int NT, const int (&Dm)[A], const uint32_t (&mg)[A],
                                            const bool (&live)[NJ], int (&dig)[NJ][A]) {
    int step[A];
    int rem = tid, rs = NT;
#pragma unroll
    for (int i = A - 1; i >= 1; --i) {
        int q;
        nary_divmod(rem, Dm[i], mg[i], q, dig[0][i]);
        rem = q;
        nary_divmod(rs, Dm[i], mg[i], q, step[i]);
        rs = q;
    }
#pragma unroll
    for (int j = 1; j < NJ; ++j) {
        int carry = 0;
#pragma unroll
        for (int i = A - 1; i >= 1; --i) {
            const int x = dig[j - 1][i] + step[i] + carry;
            carry = x >= Dm[i] ? 1 : 0;
            dig[j][i] = x - (carry ? Dm[i] : 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 1; i < A; ++i) dig[j][i] = live[j] ? dig[j][i] : Dm[i] - 1;
}

// entry j of a run of narrow entries held in dwords (little endian), widened exactly
template <typename T, typename TT>
__device__ __forceinline__ T nary_slot_entry_fwd(const uint32_t* w, int j) {
    if constexpr (sizeof(TT) == 1) return (T)(int)(int8_t)(uint8_t)(w[j >> 2] >> (8 * (j & 3)));
    else return (T)(int)(int16_t)(uint16_t)(w[j >> 1] >> (16 * (j & 1)));
}

// blockDim.x = NT threads (a multiple of 64, <= BLOCK) with R <= NJ * NT: the launch groups of
// layout.cpp pick NT so that, whenever R allows it, every lane owns exactly NJ live q's.
// MULTI (round 6): R beyond NJ * NT -- arity 3 over more than 32 values, arity 4 over more than 10, arity 5 over more than 5,
// arity 6 (a SECP instance generated with --max_model_size 5: 5^6 entries) -- in PASSES of NJ * NT q's: a pass walks every d0
// for its q's exactly as the single pass does; the minima of all passes meet in the same LDS keys (they are order-independent).
// Up to round 5's end such factors took factor_generic: a thread per edge walking the whole table.
// TT / NEG (MULTI only): the table read from a NARROW ROW-MAJOR image in ctables (int8 / int16 entries at their row-major index: a
// wave reads 64 consecutive entries per load) instead of the full-width array -- a factor of this class has at least 1 024 entries
// per value of its first variable, eight bytes each at full width; every entry is widened (and negated in max mode: narrow
// images hold un-negated values) before anything is computed with it, as in k_factor_nary_packed.
template <typename T, int A, int NJ, bool MULTI = false, typename TT = T, bool NEG = false>
__global__ void __launch_bounds__(BLOCK) k_factor_nary(SweepArgs<T> a, const NaryDesc* descs, int cap) {
    static_assert(MULTI || (std::is_same<TT, T>::value && !NEG), "narrow row-major images: the multi-pass groups only");
    typedef typename OrdKey<T>::U U;
    constexpr int UNR = NARY_UNR;
    // Dynamic LDS, sized by the launch for its group's largest scope (3 arrays of `cap` elements: a block of 5 x 5 x 5 factors
    // holds 0.4 KB where the fixed arrays of 1 024 elements held 24 KB -- and six blocks a CU: profiles/r06_secp_*):
    HIP_DYNAMIC_SHARED(unsigned long long, s_dyn)
    T* s_msg = (T*)s_dyn;                // incoming V->F messages; in the epilogue the new ones
    U* s_key = (U*)(s_msg + cap);        // running minima of the outgoing messages (ordered keys)
    T* s_prev = (T*)(s_key + cap);       // epilogue: the messages sent last
    __shared__ int s_nomatch[NARY_MAX_ARITY];
    __shared__ int s_cnt[NARY_MAX_ARITY];
    const NaryDesc fd = descs[blockIdx.x];  // block-uniform: one scalar load
    const int tid = (int)threadIdx.x, NT = (int)blockDim.x;
    int Dm[A], off[A];
    uint32_t mg[A];
    int sumd = 0;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        Dm[i] = fd.dom[i];
        mg[i] = fd.magic[i];
        off[i] = sumd;
        sumd += Dm[i];
    }
    int R = 1;
#pragma unroll
    for (int i = 1; i < A; ++i) R *= Dm[i];
    const int D0 = Dm[0];
    const TT* tab = std::is_same<TT, T>::value ? (const TT*)(a.tables + fd.tab_off) : (const TT*)(a.ctables + fd.tab_off);
    // RUN (narrow images): a lane owns NJ CONSECUTIVE q's of a pass -- its NJ entries of a table row are one 4- / 8-byte load (four
    // byte loads with the strided assignment below: meeting_5k_d40 341 -> 330 us, profiles/r06_multi_pass_nary_v3.txt); full-width tables keep
    // q = tid + j * NT, where a wave's load instruction reads 64 consecutive entries.
    constexpr bool RUN = !std::is_same<TT, T>::value;
    auto entry = [&](int64_t k) __attribute__((always_inline)) {
        if constexpr (!RUN) return tab[k];
        else {
            const T v = (T)(int)tab[k];
            return NEG ? -v : v;
        }
    };
    // A table row as a lane holds it between its load and its use: NJ values -- or, RUN, the raw dwords of its NJ narrow entries,
    // widened (and negated) only where they are used: the prefetched batch is 4 registers to carry, not 32, and the negation
    // folds into the additions as an operand modifier (26 VALU instructions per table entry before, SQ_INSTS_VALU of
    // profiles/r06_multi_pass_pmc_v1.txt: 2 of them the copies of the double buffer, 1 the sign).
    constexpr int RWD = RUN ? (NJ * (int)sizeof(TT) + 3) / 4 : 1;
    struct RowT { T v[NJ]; };
    struct RowW { uint32_t w[RWD]; };
    typedef typename std::conditional<RUN, RowW, RowT>::type Row;
    // the lane's NJ entries of the row that starts at entry `row`.  (RUN, no branch: a run that crosses the end of its row reads
    // on into the next row -- behind the last row into the 16 bytes of slack every such image ends with, layout.h
    // nary_place_bytes -- and those entries belong to q's that are not live: masked)
    auto load_row = [&](int64_t row, const int (&qcl)[NJ], Row& out) __attribute__((always_inline)) {
        if constexpr (RUN) {
            __builtin_memcpy(out.w, (const uint8_t*)tab + (row + qcl[0]) * (int64_t)sizeof(TT), NJ * sizeof(TT));  // (any byte offset: gfx950 reads unaligned dwords)
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) out.v[j] = entry(row + qcl[j]);
        }
    };
    auto unpack = [&](const Row (&r)[UNR], T (&tv)[UNR][NJ]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (RUN) {
                    const T v = nary_slot_entry_fwd<T, TT>(r[u].w, j);
                    tv[u][j] = NEG ? -v : v;
                } else {
                    tv[u][j] = r[u].v[j];
                }
            }
    };
    const int n_full = D0 / UNR;  // batches without a masked d0
    const int n_pass = MULTI ? (R + NJ * NT - 1) / (NJ * NT) : 1;
    for (int ps = 0; ps < n_pass; ++ps) {
    const int qb = ps * (NJ * NT);
    // own q's (clamped into the table so that every load is in range)
    int qc[NJ];
    bool live[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = RUN ? qb + tid * NJ + j : qb + tid + j * NT;
        live[j] = q < R;
        qc[j] = live[j] ? q : R - 1;
    }
    // (RUN: a wave none of whose lanes has a live q in this pass sits it out -- no workgroup barrier below the first pass)
    if (RUN && ps > 0 && qb + (tid & ~63) * NJ >= R) continue;
    Row cur[UNR];
    if (n_full > 0) {  // first batch: requested before anything else, independent of the messages
#pragma unroll
        for (int u = 0; u < UNR; ++u) load_row((int64_t)u * R, qc, cur[u]);
    }
    if (ps == 0) {
    // stage the incoming messages, arm the minima (the table loads above are in flight)
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const int vo = fd.v2f_off[i];
        for (int d = tid; d < Dm[i]; d += NT) {
            // (dimension 0's message is staged as `0 + m`: the first step of the reference's sum_cost for the
            // outputs p >= 1, maxsum.py:430-441 -- once per factor here instead of once per lane and d0)
            const T x = a.v2f_old[vo + d];
            s_msg[off[i] + d] = i == 0 ? (T)0 + x : x;
            s_key[off[i] + d] = OrdKey<T>::enc(pos_inf<T>());
        }
    }
    if (tid < NARY_MAX_ARITY) s_nomatch[tid] = 0;
    __syncthreads();
    }
    // per owned q: its digits' messages and the running minima for p >= 1
    T ms[NJ][A], acc[NJ][A], s0[NJ];
    int dig[NJ][A];
    nary_digits<A, NJ>(RUN ? qb + tid * NJ : qb + tid, RUN ? 1 : NT, Dm, mg, live, dig);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = A - 1; i >= 1; --i) {
            ms[j][i] = s_msg[off[i] + dig[j][i]];
            acc[j][i] = pos_inf<T>();
        }
        // sum of the others' messages for the output to variable 0: fixed per q
        T s = (T)0;
#pragma unroll
        for (int i = 1; i < A; ++i) s += ms[j][i];
        s0[j] = s;
    }
    const bool all_live = MULTI ? qb + NJ * NT <= R : R == NJ * NT;  // block-uniform
    for (int b = 0; b < n_full; ++b) {
        const int d0 = b * UNR;
        Row nxt[UNR];
        if (b + 1 < n_full) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) load_row((int64_t)(d0 + UNR + u) * R, qc, nxt[u]);
        } else {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if constexpr (RUN) {
#pragma unroll
                    for (int x = 0; x < RWD; ++x) nxt[u].w[x] = 0u;
                } else {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) nxt[u].v[j] = pos_inf<T>();
                }
            }
        }
        T tv[UNR][NJ];
        unpack(cur, tv);
        if (all_live) nary_batch<T, A, NJ, false>(tv, d0, D0, s_msg + off[0], ms, s0, live, acc, s_key + off[0]);
        else nary_batch<T, A, NJ, true>(tv, d0, D0, s_msg + off[0], ms, s0, live, acc, s_key + off[0]);
#pragma unroll
        for (int u = 0; u < UNR; ++u) cur[u] = nxt[u];
    }
    if (n_full * UNR < D0) {  // tail batch
        const int d0 = n_full * UNR;
#pragma unroll
        for (int u = 0; u < UNR; ++u) load_row((int64_t)(d0 + u < D0 ? d0 + u : D0 - 1) * R, qc, cur[u]);
        T tv[UNR][NJ];
        unpack(cur, tv);
        nary_batch<T, A, NJ, true>(tv, d0, D0, s_msg + off[0], ms, s0, live, acc, s_key + off[0]);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (live[j]) {
#pragma unroll
            for (int p = 1; p < A; ++p) atomicMin(&s_key[off[p] + dig[j][p]], OrdKey<T>::enc(acc[j][p]));
        }
    }  // passes
    __syncthreads();
    // apply_damping + the send rule (maxsum.py:346-377), one thread per message ELEMENT so
    // that the previous messages arrive with one round of parallel loads.
    // phase 1: damp, compare with the message sent last
    for (int idx = tid; idx < sumd; idx += NT) {
        int i = 0, off_i = 0, fo_i = fd.f2v_off[0];  // selects, not indexed loads: the
#pragma unroll                                       // descriptor stays in registers
        for (int ii = 1; ii < A; ++ii)
            if (idx >= off[ii]) {
                i = ii;
                off_i = off[ii];
                fo_i = fd.f2v_off[ii];
            }
        const int d = idx - off_i;
        T m = OrdKey<T>::dec(s_key[idx]);
        if (a.start) {  // only start_messages == all makes a non-unary factor send
            s_msg[idx] = a.start_mode == MXS_START_ALL ? m : (T)0;
            continue;
        }
        const T p = a.f2v_old[fo_i + d];
        const int cnt = a.cF[fd.edge_base + i];
        if (cnt > 0 && a.damp_f) m = a.damping * p + ((T)1 - a.damping) * m;
        if (cnt > 0 && !comp_match(m, p, a.stability)) s_nomatch[i] = 1;  // same value from all writers
        if (d == 0) s_cnt[i] = cnt;
        s_msg[idx] = m;
        s_prev[idx] = p;
    }
    __syncthreads();
    // phase 2: send / send again / stay silent (the receiver keeps the old message)
    for (int idx = tid; idx < sumd; idx += NT) {
        int i = 0, off_i = 0, fo_i = fd.f2v_off[0];
#pragma unroll
        for (int ii = 1; ii < A; ++ii)
            if (idx >= off[ii]) {
                i = ii;
                off_i = off[ii];
                fo_i = fd.f2v_off[ii];
            }
        const int d = idx - off_i;
        const int e = fd.edge_base + i;
        T* w = a.f2v_new + fo_i;
        if (a.start) {
            w[d] = s_msg[idx];
            if (d == 0) a.cF[e] = 0;
            continue;
        }
        const int cnt = s_cnt[i];
        const bool match = cnt > 0 && !s_nomatch[i];
        int out = 1;
        T val = s_msg[idx];
        if (match) {
            if (cnt < SAME_COUNT) {
                out = cnt + 1;
            } else {
                out = cnt;
                val = s_prev[idx];
            }
        }
        w[d] = val;
        if (d == 0) a.cF[e] = (uint8_t)out;
    }
}

// The same factor block for a table stored in a NARROW type (layout.h TabType), LANE-PACKED
// (nary_packed_pos): per d0 and lane one slot of 4 / 8 / 16 bytes with the lane's NJ entries, read
// with one aligned vector load.  k_factor_nary above streams full-width tables at the HBM ceiling;
// with fewer bytes per entry the limit becomes how many loads are in flight, so here PF batches
// of d0 are requested ahead (a slot is 1-4 registers where NJ full-width entries were 2 * NJ).
// Every entry is widened to T (and negated in max mode: narrow images hold un-negated values)
// before anything is computed with it: the arithmetic is that of the full-width kernel, bit for bit.
#ifndef MXS_NARY_PF
#define MXS_NARY_PF 4
#endif
template <int SW>
struct alignas(SW * 4) NarySlot {
    uint32_t w[SW];
};
template <typename T, typename TT>
__device__ __forceinline__ T nary_slot_entry(const uint32_t* w, int j) {
    if constexpr (sizeof(TT) == 1) return (T)(int)(int8_t)(uint8_t)(w[j >> 2] >> (8 * (j & 3)));
    else if constexpr (sizeof(TT) == 2) return (T)(int)(int16_t)(uint16_t)(w[j >> 1] >> (16 * (j & 1)));
    else {
        float f;
        __builtin_memcpy(&f, &w[j], 4);
        return (T)f;
    }
}

// NEG (max mode: narrow images hold un-negated values) is a template parameter: the negation then
// folds into the first use of the entry as an operand modifier instead of a 64-bit select per entry.
template <typename T, int A, int NJ, typename TT, bool NEG, bool LS = false>
__global__ void __launch_bounds__(BLOCK) k_factor_nary_packed(SweepArgs<T> a, const NaryDesc* descs, int cap) {
    typedef typename OrdKey<T>::U U;
    constexpr int UNR = NARY_UNR;
    constexpr int PF = MXS_NARY_PF;                         // batches requested ahead
    constexpr int SW = nary_slot_bytes(NJ, (int)sizeof(TT)) / 4;
    HIP_DYNAMIC_SHARED(unsigned long long, s_dyn)  // (as k_factor_nary: 3 arrays of `cap` elements)
    T* s_msg = (T*)s_dyn;
    U* s_key = (U*)(s_msg + cap);
    T* s_prev = (T*)(s_key + cap);
    __shared__ int s_nomatch[NARY_MAX_ARITY];
    __shared__ int s_cnt[NARY_MAX_ARITY];
    const NaryDesc fd = descs[blockIdx.x];
    const int tid = (int)threadIdx.x, NT = (int)blockDim.x;
    int Dm[A], off[A];
    uint32_t mg[A];
    int sumd = 0;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        Dm[i] = fd.dom[i];
        mg[i] = fd.magic[i];
        off[i] = sumd;
        sumd += Dm[i];
    }
    int R = 1;
#pragma unroll
    for (int i = 1; i < A; ++i) R *= Dm[i];
    const int D0 = Dm[0];
    const NarySlot<SW>* slots = (const NarySlot<SW>*)(a.ctables + fd.tab_off);  // [D0][NT]
    int qc[NJ];
    bool live[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = tid + j * NT;
        live[j] = q < R;
        qc[j] = live[j] ? q : R - 1;
    }
    const int n_batches = (D0 + UNR - 1) / UNR;
    NarySlot<SW> buf[PF][UNR];
    // the first PF batches: requested before anything else (rows past D0 clamp to the last one)
#pragma unroll
    for (int pb = 0; pb < PF; ++pb)
        if (pb < n_batches) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int row = pb * UNR + u < D0 ? pb * UNR + u : D0 - 1;
                buf[pb][u] = slots[(int64_t)row * NT + tid];
            }
        }
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const int vo = fd.v2f_off[i];
        for (int d = tid; d < Dm[i]; d += NT) {
            // (dimension 0's message is staged as `0 + m`: the first step of the reference's sum_cost for the
            // outputs p >= 1, maxsum.py:430-441 -- once per factor here instead of once per lane and d0)
            const T x = a.v2f_old[vo + d];
            s_msg[off[i] + d] = i == 0 ? (T)0 + x : x;
            s_key[off[i] + d] = OrdKey<T>::enc(pos_inf<T>());
        }
    }
    if (tid < NARY_MAX_ARITY) s_nomatch[tid] = 0;
    __syncthreads();
    T ms[NJ][A], acc[NJ][A], s0[NJ];
    int dig[NJ][A];
    nary_digits<A, NJ>(tid, NT, Dm, mg, live, dig);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = A - 1; i >= 1; --i) {
            ms[j][i] = s_msg[off[i] + dig[j][i]];
            acc[j][i] = pos_inf<T>();
        }
        T s = (T)0;
#pragma unroll
        for (int i = 1; i < A; ++i) s += ms[j][i];
        s0[j] = s;
    }
    const bool all_live = R == NJ * NT;  // block-uniform
    for (int b0 = 0; b0 < n_batches; b0 += PF) {
#pragma unroll
        for (int pb = 0; pb < PF; ++pb) {
            const int b = b0 + pb;
            if (b < n_batches) {  // block-uniform
                T tv[UNR][NJ];
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const T v = nary_slot_entry<T, TT>(buf[pb][u].w, j);
                        tv[u][j] = NEG ? -v : v;
                    }
                if (b + PF < n_batches) {  // this buffer's next tenant
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int row = (b + PF) * UNR + u < D0 ? (b + PF) * UNR + u : D0 - 1;
                        buf[pb][u] = slots[(int64_t)row * NT + tid];
                    }
                }
                const int d0 = b * UNR;
                if (all_live && d0 + UNR <= D0)
                    nary_batch<T, A, NJ, false, LS>(tv, d0, D0, s_msg + off[0], ms, s0, live, acc, s_key + off[0]);
                else
                    nary_batch<T, A, NJ, true, LS>(tv, d0, D0, s_msg + off[0], ms, s0, live, acc, s_key + off[0]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (live[j]) {
#pragma unroll
            for (int p = 1; p < A; ++p) atomicMin(&s_key[off[p] + dig[j][p]], OrdKey<T>::enc(acc[j][p]));
        }
    __syncthreads();
    // apply_damping + the send rule: as in k_factor_nary
    for (int idx = tid; idx < sumd; idx += NT) {
        int i = 0, off_i = 0, fo_i = fd.f2v_off[0];
#pragma unroll
        for (int ii = 1; ii < A; ++ii)
            if (idx >= off[ii]) {
                i = ii;
                off_i = off[ii];
                fo_i = fd.f2v_off[ii];
            }
        const int d = idx - off_i;
        T m = OrdKey<T>::dec(s_key[idx]);
        if (a.start) {
            s_msg[idx] = a.start_mode == MXS_START_ALL ? m : (T)0;
            continue;
        }
        const T p = a.f2v_old[fo_i + d];
        const int cnt = a.cF[fd.edge_base + i];
        if (cnt > 0 && a.damp_f) m = a.damping * p + ((T)1 - a.damping) * m;
        if (cnt > 0 && !comp_match(m, p, a.stability)) s_nomatch[i] = 1;
        if (d == 0) s_cnt[i] = cnt;
        s_msg[idx] = m;
        s_prev[idx] = p;
    }
    __syncthreads();
    for (int idx = tid; idx < sumd; idx += NT) {
        int i = 0, off_i = 0, fo_i = fd.f2v_off[0];
#pragma unroll
        for (int ii = 1; ii < A; ++ii)
            if (idx >= off[ii]) {
                i = ii;
                off_i = off[ii];
                fo_i = fd.f2v_off[ii];
            }
        const int d = idx - off_i;
        const int e = fd.edge_base + i;
        T* w = a.f2v_new + fo_i;
        if (a.start) {
            w[d] = s_msg[idx];
            if (d == 0) a.cF[e] = 0;
            continue;
        }
        const int cnt = s_cnt[i];
        const bool match = cnt > 0 && !s_nomatch[i];
        int out = 1;
        T val = s_msg[idx];
        if (match) {
            if (cnt < SAME_COUNT) {
                out = cnt + 1;
            } else {
                out = cnt;
                val = s_prev[idx];
            }
        }
        w[d] = val;
        if (d == 0) a.cF[e] = (uint8_t)out;
    }
}

// ---------------------------------------------------------------------------
// Variable side, wide class: domains too large for the packed classes (9 <= D <= 256; 5..8: k_variable_pack8) or degrees
// above 64, as long as deg * D <= 1024, deg <= 256.  ONE WORKGROUP PER RUN OF VARIABLES of one
// domain size (layout.h WideBlock; e.g. 32 variables of D = 24 and degree 3), every phase with
// all lanes busy:
//   1. stage: lane <-> (edge, d) element of the incoming F->V messages, all of a thread's
//      gathers in flight together; own costs and send counters likewise;
//   2. lane <-> outgoing edge: the serial `sum_cost` chain of costs_for_factor (ONE accumulator
//      runs through all (d, f != target) in d-major order, maxsum.py:651-665) walked in LDS --
//      the chains of ALL the block's edges side by side; lane <-> variable: belief + selection;
//   3. lane <-> (outgoing edge, d) element: the new message, damped, against the one sent last;
//   4. the send rule per edge (the elements of an edge agree through an LDS flag), stores as
//      one contiguous stream (the V->F records of consecutive variables are adjacent).
// (Round 2 ran a wave per variable with lanes over d and a serial loop over the outgoing edges:
// D = 24 used 24 of 64 lanes, the chains 3 -- 171 us per cycle on meeting_50k, 7 % of the HBM
// peak.)  Arithmetic order is the reference's, op for op:
//   select_value      maxsum.py:584-620   b[d] = c[d] + in_0[d] + in_1[d] + ...
//   costs_for_factor  maxsum.py:623-676
// Own launch (its LDS must not cap the occupancy of the register classes).
// ---------------------------------------------------------------------------

// c + in_0[d] + in_1[d] + ... in order, skipping edge `skip` (-1: none); reads four at a time,
// the last one to three together as well
template <typename T>
__device__ __forceinline__ T wide_sum_edges(T c, const T* in, int D, int deg, int d, int skip) {
    T acc = c;
    int k = 0;
    for (; k + 4 <= deg; k += 4) {
        T x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = in[(k + u) * D + d];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k + u != skip) acc += x[u];
    }
    const int rem = deg - k;
    if (rem > 0) {
        const T x0 = in[k * D + d];
        const T x1 = in[(rem > 1 ? k + 1 : k) * D + d];
        const T x2 = in[(rem > 2 ? k + 2 : k) * D + d];
        if (k != skip) acc += x0;
        if (rem > 1 && k + 1 != skip) acc += x1;
        if (rem > 2 && k + 2 != skip) acc += x2;
    }
    return acc;
}

// sum over (d, k != ko) of in[k][d], d major / k minor: the serial `sum_cost` chain of costs_for_factor (ONE
// accumulator, maxsum.py:651-665).  A skipped term (k == ko, k >= deg) is read from a row of zeros: the
// accumulator starts at +0 and can never become -0, so `sc + 0` is `sc` bit for bit -- no select on the
// dependent chain, one add per term.  The reads of a pass (two values of d of four rows) are requested
// before the additions of the previous pass.
// NR row pointers in registers (deg - 1 <= NR: the target's own row is left out), ND values of d per pass; the reads of a pass are requested before
// the additions of the pass before.
template <typename T, int NR, int ND>
__device__ __forceinline__ T wide_chain_rows(const T* in, const T* zero, int D, int deg, int ko) {
    // the rows of the OTHER edges, in order (row i = edge i, or i + 1 from the skipped one on); the rest: zeros
    const T* r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int k = i < ko ? i : i + 1;
        r[i] = k < deg ? in + k * D : zero;
    }
    T q[ND][NR], sc = (T)0;
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int k = 0; k < NR; ++k) q[i][k] = r[k][i < D ? i : 0];
    int d = 0;
    for (; d + ND <= D; d += ND) {
        T n[ND][NR];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int dn = d + ND + i < D ? d + ND + i : 0;  // (past the end: a read nobody uses)
#pragma unroll
            for (int k = 0; k < NR; ++k) n[i][k] = r[k][dn];
        }
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int k = 0; k < NR; ++k) sc += q[i][k];
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int k = 0; k < NR; ++k) q[i][k] = n[i][k];
    }
#pragma unroll
    for (int i = 0; i < ND - 1; ++i)  // D % ND values of d are left, already read
        if (d + i < D) {
#pragma unroll
            for (int k = 0; k < NR; ++k) sc += q[i][k];
        }
    return sc;
}
template <typename T>
__device__ __forceinline__ T wide_chain(const T* in, const T* zero, int D, int deg, int ko) {
    // (the variables of the class are sorted by domain size, then by degree in steps of four: a wave rarely
    // mixes the paths)
#ifndef MXS_WIDE_ND
#define MXS_WIDE_ND 2  // values of d per pass of the chain at degree <= 4 (experiments: 4 = half the LDS round trips, twice the registers)
#endif
    if (deg <= 4) return wide_chain_rows<T, 3, MXS_WIDE_ND>(in, zero, D, deg, ko);
    if (deg <= 8) return wide_chain_rows<T, 7, 1>(in, zero, D, deg, ko);
    T sc = (T)0;
    for (int d = 0; d < D; ++d)
        for (int k = 0; k < deg; k += 8) {
            T x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = (k + u < deg && k + u != ko) ? in[(k + u) * D + d] : zero[d];
#pragma unroll
            for (int u = 0; u < 8; ++u) sc += x[u];
        }
    return sc;
}

#ifndef MXS_WIDE_SKIP
#define MXS_WIDE_SKIP 0  // timing experiments only (results wrong): 1 no chains, 2 no message arithmetic,
#endif                   // 4 no beliefs, 8 no gathers, 16 no stores
constexpr int WIDE_R = WIDE_CAPB / WIDE_TPB;                          // staged elements per thread
constexpr int WIDE_CR = (WIDE_MAX_COSTS + WIDE_TPB - 1) / WIDE_TPB;   // own costs per thread
// (a thread holds ONE variable's slot range and ONE slot's counter, WideRec: the block size is no free build knob)
static_assert(WIDE_TPB >= WIDE_MAX_VARS && WIDE_TPB >= WIDE_MAX_SLOTS && WIDE_CAPB % WIDE_TPB == 0,
              "k_variable_wide: one variable and one slot per thread");

// What a thread holds of a block between the request and the staging (registers; the loads are in flight
// while the workgroup works on the block before).
template <typename T>
struct WideIdx {
    int fo[WIDE_R], vo[WIDE_R];  // F2V / V2F offsets of the SLOTS of its elements idx = tid + r * WIDE_TPB
};
template <typename T>
struct WideRec {
    T x[WIDE_R], p[WIDE_R];      // the incoming F->V element, the V->F element sent last
    T c[WIDE_CR];                // own costs i = tid + q * WIDE_TPB
    int lo, hi;                  // variable j = tid: its CSR slot range
    int cnt;                     // slot s = tid: send counter
};
// Every load below is unconditional (clamped indices instead of branches) and nothing is computed from a
// loaded value: the requests of a block leave the wave back to back and nothing waits for them here.
template <typename T>
__device__ __forceinline__ void wide_request_idx(const SweepArgs<T>& a, const WideBlock& wb, int tid, WideIdx<T>& ix) {
    const int ne = wb.n_slots * wb.D;
    static_for<WIDE_R>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const int idx = tid + r * WIDE_TPB;
        const int s = wb.D == 1 ? idx : (int)(((uint64_t)(uint32_t)idx * wb.magic) >> 32);  // idx / D
        const int so = wb.slot0 + (idx < ne ? s : 0);
        ix.fo[r] = a.vslot_f2v[so];
        ix.vo[r] = a.vslot_v2f[so];
    });
}
template <typename T>
__device__ __forceinline__ void wide_request_rec(const SweepArgs<T>& a, const WideBlock& wb, int tid, const WideIdx<T>& ix,
                                                 WideRec<T>& rc_) {
    const int ne = wb.n_slots * wb.D;
    static_for<WIDE_R>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const int idx = tid + r * WIDE_TPB;
        const int s = wb.D == 1 ? idx : (int)(((uint64_t)(uint32_t)idx * wb.magic) >> 32);
        const int dd = idx < ne ? idx - s * wb.D : 0;
        rc_.x[r] = (MXS_WIDE_SKIP & 8) ? (T)0 : a.f2v_old[ix.fo[r] + dd];
        rc_.p[r] = a.start ? (T)0 : a.v2f_old[ix.vo[r] + dd];  // the message sent last on this edge
    });
    static_for<WIDE_CR>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        const int i = tid + q * WIDE_TPB;
        rc_.c[q] = a.var_cost[wb.cost_off + (i < wb.n_vars * wb.D ? i : 0)];
    });
    const int j = tid < wb.n_vars ? tid : 0;
    rc_.lo = a.vrowptr[wb.first_var + j];
    rc_.hi = a.vrowptr[wb.first_var + j + 1];
    rc_.cnt = a.start ? 0 : (int)a.cV[wb.slot0 + (tid < wb.n_slots ? tid : 0)];
}

// One workgroup per block (layout.h WideBlock).  (A PERSISTENT variant -- a few hundred workgroups walking the blocks with
// the next block's records and the indices of the one after in flight -- was measured and dropped: 64.5 us with 768 workgroups
// against 57.4 then, profiles/r04_wide_phases_v1.txt: a block's LDS phases are one dependency chain, and eight co-resident
// workgroups overlap them better than three with a prefetch pipeline.)
#ifndef MXS_WIDE_WAVES
#define MXS_WIDE_WAVES 8  // register budget for that many waves per SIMD (0: the compiler's choice, 80 VGPRs in f64: 45.6 us against 42.3)
#endif
template <typename T>
__global__ void __launch_bounds__(WIDE_TPB)
#if MXS_WIDE_WAVES > 0
__attribute__((amdgpu_waves_per_eu(MXS_WIDE_WAVES, MXS_WIDE_WAVES)))
#endif
k_variable_wide(SweepArgs<T> a, const WideBlock* __restrict__ blocks) {
    constexpr int R = WIDE_R;
    __shared__ T s_in[WIDE_CAPB];               // staged F->V messages: [slot][d]
    __shared__ T s_c[WIDE_MAX_COSTS];           // own costs: [variable][d]
    __shared__ T s_b[WIDE_MAX_COSTS];           // beliefs: [variable][d]
    __shared__ T s_avg[WIDE_MAX_SLOTS];         // per outgoing edge: sum_cost / D
    __shared__ T s_zero[WIDE_MAX_D];            // a row of zeros (wide_chain)
    __shared__ uint8_t s_svar[WIDE_MAX_SLOTS];  // ... its variable (local index)
    __shared__ uint8_t s_cnt[WIDE_MAX_SLOTS];   // ... its send counter
    __shared__ uint8_t s_nom[WIDE_MAX_SLOTS];   // ... 1: some element differs from the message sent last
    __shared__ int s_vk0[WIDE_MAX_VARS];        // per variable: its first (local) slot
    __shared__ int s_vdeg[WIDE_MAX_VARS];
    const int tid = (int)threadIdx.x;
    const WideBlock wb = blocks[blockIdx.x];    // block-uniform: scalar loads
    for (int i = tid; i < WIDE_MAX_D; i += WIDE_TPB) s_zero[i] = (T)0;
    WideIdx<T> iA;
    WideRec<T> rA;
    wide_request_idx(a, wb, tid, iA);
    wide_request_rec(a, wb, tid, iA, rA);
#ifdef MXS_WIDE_PROFILE
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WIDE_TICK(k) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); prof[k] += t_ - t_last; t_last = t_; } while (0)
#else
#define WIDE_TICK(k) ((void)0)
#endif
    {
#ifdef MXS_WIDE_PROFILE
        long long t_last = (long long)__builtin_amdgcn_s_memtime();
        const long long t_top = t_last;
#endif
        const int D = wb.D, ns = wb.n_slots, ne = ns * D;
        // ---- 1. stage ------------------------------------------------------------------------
        int sl[R], dd[R];
        static_for<R>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            const int idx = tid + r * WIDE_TPB;
            sl[r] = D == 1 ? idx : (int)(((uint64_t)(uint32_t)idx * wb.magic) >> 32);
            dd[r] = idx - sl[r] * D;
            if (idx < ne) s_in[idx] = rA.x[r];
        });
        static_for<WIDE_CR>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            if (tid + q * WIDE_TPB < wb.n_vars * D) s_c[tid + q * WIDE_TPB] = rA.c[q];
        });
        if (tid < wb.n_vars) {
            const int k0 = rA.lo - wb.slot0, deg = rA.hi - rA.lo;
            s_vk0[tid] = k0;
            s_vdeg[tid] = deg;
            for (int k = 0; k < deg; ++k) s_svar[k0 + k] = (uint8_t)tid;
        }
        if (tid < ns) {
            s_cnt[tid] = (uint8_t)rA.cnt;
            s_nom[tid] = 0;
        }
        lds_barrier();
        WIDE_TICK(0);
        // ---- 2. chains and beliefs --------------------------------------------------------------
        for (int t = tid; t < ns; t += WIDE_TPB) {  // the mean of an outgoing message: its serial chain
            const int j = s_svar[t], k0 = s_vk0[j];
            s_avg[t] = (MXS_WIDE_SKIP & 1) ? (T)0 : wide_chain<T>(s_in + k0 * D, s_zero, D, s_vdeg[j], t - k0) / (T)D;
        }
        for (int i = tid; i < wb.n_vars * D; i += WIDE_TPB) {  // b[d] = c[d] + in_0[d] + in_1[d] + ...
            const int j = D == 1 ? i : (int)(((uint64_t)(uint32_t)i * wb.magic) >> 32);
            s_b[i] = (MXS_WIDE_SKIP & 4) ? s_c[i] : wide_sum_edges<T>(s_c[i], s_in + s_vk0[j] * D, D, s_vdeg[j], i - j * D, -1);
        }
        lds_barrier();
        WIDE_TICK(1);
        for (int j = tid; j < wb.n_vars; j += WIDE_TPB) {  // selection: first index attaining the minimum
            const int v = wb.first_var + j;
            T bb = s_b[j * D];
            int bi = 0;
            int d = 1;
            for (; d + 4 <= D; d += 4) {  // four reads requested together, compared in order
                T bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) bv[u] = s_b[j * D + d + u];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (bv[u] < bb) {
                        bb = bv[u];
                        bi = d + u;
                    }
            }
            for (; d < D; ++d) {
                const T bv = s_b[j * D + d];
                if (bv < bb) {
                    bb = bv;
                    bi = d;
                }
            }
            if (a.start && a.init_idx[v] >= 0) {  // value_selection(initial_value), maxsum.py:497-498
                bi = a.init_idx[v];
                bb = (T)0;
            }
            a.sel[v] = bi;
            a.belief[v] = bb;
        }
        WIDE_TICK(2);
        // ---- 3. the new messages, damped, against the ones sent last ---------------------------------
        T m[R];
        static_for<R>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            m[r] = (T)0;
            if (tid + r * WIDE_TPB < ne) {
                const int s = sl[r], j = s_svar[s], k0 = s_vk0[j], deg = s_vdeg[j];
                T mm = (MXS_WIDE_SKIP & 2) ? s_avg[s] : wide_sum_edges<T>(s_c[j * D + dd[r]], s_in + k0 * D, D, deg, dd[r], s - k0) - s_avg[s];
                if (a.start) {
                    const bool start_sends = (deg == 1 && a.start_mode == MXS_START_LEAFS) ||
                                             a.start_mode != MXS_START_LEAFS;
                    mm = start_sends ? mm : (T)0;
                } else {
                    if (s_cnt[s] > 0 && a.damp_v) mm = a.damping * rA.p[r] + ((T)1 - a.damping) * mm;
                    if (!comp_match(mm, rA.p[r], a.stability)) s_nom[s] = 1;  // same value from all writers
                }
                m[r] = mm;
            }
        });
        lds_barrier();
        WIDE_TICK(3);
        // ---- 4. send / send again / stay silent (the receiver keeps the old message) -------------------
        static_for<R>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            if (tid + r * WIDE_TPB < ne) {
                const int s = sl[r];
                const int cnt = s_cnt[s];
                int out = 1;
                bool keep_old = false;
                if (a.start) {
                    out = 0;
                } else if (cnt > 0 && !s_nom[s]) {
                    if (cnt < SAME_COUNT) {
                        out = cnt + 1;
                    } else {
                        out = cnt;
                        keep_old = true;
                    }
                }
                // (the old message from the registers: a second read of the record would have to wait for
                // every load in flight, the next block's records among them)
                const T val = keep_old ? rA.p[r] : m[r];
                if (!(MXS_WIDE_SKIP & 16)) a.v2f_new[iA.vo[r] + dd[r]] = val;
                if (dd[r] == 0) a.cV[wb.slot0 + s] = (uint8_t)out;
            }
        });
        WIDE_TICK(4);
#ifdef MXS_WIDE_PROFILE
        prof[7] += 1;
        prof[6] += (long long)__builtin_amdgcn_s_memtime() - t_top;
        if (tid == 0 && a.timeline)
            for (int k = 0; k < 8; ++k) a.timeline[8 * blockIdx.x + k] += prof[k];
#endif
    }
}

// (Round 5: ONE WAVE per run of variables -- the same phases and helpers, a run cut for 64 lanes x 12 staged elements, every
// barrier a wave barrier so that the chain phase has a lane per edge of a full wave and no wave waits for another -- was built,
// parity-tested and measured SLOWER than the workgroup above: peav_50k 128.4 us against 77.4, coloring_100k_d8 51.1 against
// 32.4, meeting_50k cycle 269 against 249 (profiles/r05_variable_wave_*): 145 VGPRs and 11 KB of LDS per wave leave 3 waves per
// SIMD where the workgroup version runs 8, and a run's life is the same chain of dependent global loads either way.  Removed.)

// ---------------------------------------------------------------------------
// solution_cost (pydcop/dcop/dcop.py:319-367): per-block partial sums of the
// factor and variable costs of an assignment; a term equal to `infinity` is a
// violation.  Always f64 on row-major un-negated tables.
// ---------------------------------------------------------------------------
struct EvalArgs {
    const int32_t* frowptr;       // [n_factors+1] internal
    const int32_t* edge_var_int;  // [n_edges]
    const int32_t* edge_dom;
    const int64_t* tab_off;
    const double* tables;
    const int32_t* vdom;
    const int64_t* vcost_off;
    const double* var_cost;
    const uint8_t* owned;
    const uint8_t* fowned;        // [n_factors] 1 = counted on this shard
    const int32_t* idx;           // [n_vars] internal order
    double* part_cost;            // [gridDim.x]
    unsigned long long* part_viol;
    int32_t n_factors, n_vars;
    double infinity;
};

// (the kernels that are no templates are `static`: this header is compiled into more than one translation unit)
static __global__ void __launch_bounds__(BLOCK) k_eval(EvalArgs a) {
    __shared__ double s_cost[BLOCK];
    __shared__ unsigned long long s_viol[BLOCK];
    double cost = 0.0;
    unsigned long long viol = 0;
    const int stride = (int)(gridDim.x * blockDim.x);
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < a.n_factors + a.n_vars; i += stride) {
        double r;
        if (i < a.n_factors) {
            if (!a.fowned[i]) continue;
            int64_t lin = 0;
            for (int e = a.frowptr[i]; e < a.frowptr[i + 1]; ++e)
                lin = lin * a.edge_dom[e] + a.idx[a.edge_var_int[e]];
            r = a.tables[a.tab_off[i] + lin];
        } else {
            const int v = i - a.n_factors;
            if (!a.owned[v]) continue;
            r = a.var_cost[a.vcost_off[v] + a.idx[v]];
        }
        if (r != a.infinity) cost += r;
        else viol += 1;
    }
    s_cost[threadIdx.x] = cost;
    s_viol[threadIdx.x] = viol;
    __syncthreads();
    for (int s = BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_cost[threadIdx.x] += s_cost[threadIdx.x + s];
            s_viol[threadIdx.x] += s_viol[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.part_cost[blockIdx.x] = s_cost[0];
        a.part_viol[blockIdx.x] = s_viol[0];
    }
}

// ---------------------------------------------------------------------------
// Halo (multi-GPU): pack the V->F halves of the cut edges this shard owns into a
// dense send buffer / scatter the received ones into the ghost records.
// One thread per (edge, d) element; `elem_edge`/`elem_d` are precomputed.
// ---------------------------------------------------------------------------
// mxs_update_factor_table: entry k of the new table -> tables[base + k * stride]
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_table_update(T* tables, int64_t base, int64_t stride,
                                                        const T* src, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) tables[base + k * stride] = src[k];
}

// mxs_slice_factor: the active table of a factor = its parent relation sliced at the current
// values of the external (read-only) dimensions (maxsum_dynamic.py:113-186, relation.slice).
// Entry k of the active table (row-major over the writable dimensions) comes from
// parent[base + sum_i digit_i(k) * stride[i]].
struct SliceDims {
    int32_t n;                 // writable dimensions
    int32_t dom[MAX_ARITY];    // their sizes, in scope order
    int64_t stride[MAX_ARITY]; // their strides in the parent table
    int64_t base;              // offset contributed by the external dimensions' values
};
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_table_slice(T* tables, int64_t tab_base, int64_t tab_stride,
                                                       double* eval_tables, const double* parent,
                                                       SliceDims sd, double sign, int64_t n,
                                                       uint8_t* crec, int ctype, NaryPlace place) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int64_t rem = k, lin = sd.base;
    for (int i = sd.n - 1; i >= 0; --i) {
        const int64_t digit = rem % sd.dom[i];
        rem /= sd.dom[i];
        lin += digit * sd.stride[i];
    }
    const double v = parent[lin];
    tables[tab_base + k * tab_stride] = (T)(sign * v);
    eval_tables[k] = v;
    if (crec != nullptr) {  // the factor's narrow image too (the parent was checked to fit):
        // entry k of a register class's record, or the lane-packed place of a workgroup-per-factor table
        uint8_t* at = crec + ((place.nt > 0 || place.box > 0) ? nary_place_pos(place, k) : k * place.elem);
        if (ctype == TAB_I8) *(int8_t*)at = (int8_t)v;
        else if (ctype == TAB_I16) *(int16_t*)at = (int16_t)v;
        else if (ctype == TAB_F32) *(float*)at = (float)v;
        else *(T*)at = (T)v;  // (the lane-grid image of a binary table at full width: un-negated, like every image)
    }
}

template <typename T>
__global__ void __launch_bounds__(BLOCK) k_halo_pack(const T* rec, const int64_t* elem_off, T* out,
                                                     int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rec[elem_off[i]];
}
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_halo_unpack(T* rec, const int64_t* elem_off, const T* in,
                                                       int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rec[elem_off[i]] = in[i];
}
// Publishes the number of the exchange just unpacked for the waiting cut-factor blocks of the
// fused sharded launch (wait_for_halo).  A kernel of its own behind the unpack kernel: the
// kernel boundary makes the ghost messages visible device-wide ONCE -- a release fence inside
// the unpack kernel writes the L2 back per block, with the sweep's dirty lines in it
// (measured: 249 us instead of 3).
static __global__ void k_halo_publish(uint32_t* flags, uint32_t epoch) {
    __hip_atomic_store(flags, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Peer-store mode: tell every other rank that the records of launch `epoch` are in its ghost
// region (the launch that stored them is complete: this kernel runs behind it in the stream).
struct PeerFlags {
    uint32_t* p[MXS_MAX_PEERS];
};
static __global__ void k_p2p_publish(PeerFlags peers, int me, int world, uint32_t epoch) {
    const int q = (int)threadIdx.x;
    if (q < world && q != me)
        __hip_atomic_store(peers.p[q] + me, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Peer-store mode, outside the sweep (after mxs_peer_connect / mxs_reset): copy the current
// records of the cut edges to the peers.  slot i of the send order -> peer's ghost region.
template <typename T>
struct PeerDst {
    T* p[MXS_MAX_PEERS];
    int32_t first[MXS_MAX_PEERS];
};
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_p2p_push(const T* rec, const int64_t* elem_off, PeerDst<T> dst,
                                                    int H, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int slot = (int)(i / H);
    T* base = dst.p[0];
    int first = dst.first[0];
#pragma unroll
    for (int q = 1; q < MXS_MAX_PEERS; ++q) {
        const bool ge = slot >= dst.first[q];
        base = ge ? dst.p[q] : base;
        first = ge ? dst.first[q] : first;
    }
    base[(int64_t)(slot - first) * H + (i - (int64_t)slot * H)] = rec[elem_off[i]];
}

}  // namespace mxs
