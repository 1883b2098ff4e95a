// amaxsum.hip -- the reference's ASYNCHRONOUS Max-Sum (pydcop/algorithms/amaxsum.py) on gfx950,
// under first-in-first-out delivery, one GENERATION of messages per step.
//
// The reference runs one handler per delivered message (`_on_maxsum_msg`, amaxsum.py:191-250 for
// factors, :366-424 for variables); what it computes depends on the order messages arrive in.
// The one order that is defined without a thread scheduler is a single FIFO queue with the
// computations started in graph order (variables, then factors) -- what oracle/ref_harness.py
// runs the reference's own objects under, and what oracle/amaxsum_oracle.c restates.  A FIFO
// handles every message of generation g (generation 0 = the start messages) before any of
// generation g + 1 (= those sent while handling generation g), and two messages of one
// generation interact only when they go to the SAME computation.  So a generation is processed
// as a batch:
//
//   k_dest      per message: its destination computation and how many messages its handler
//               can send at most (the destination's other neighbours)
//   scan        exclusive sum of those capacities = the message's block of output slots, in
//               FIFO order; k_stamp writes it into the message's record
//   sort        stable radix sort of (destination, FIFO index) by destination (hipCUB)
//   k_permute   the records themselves gathered into destination order (round 4): a queue is a
//               contiguous run, a delivery one sequential 32-byte read
//   k_process   a lane / lane group per destination, a kernel and a stream per destination class
//               (they run side by side): handles ITS messages one after the other in FIFO
//               order, exactly like the reference's handler (same expressions, same order of
//               additions -- select_value in first-arrival order of the factors, maxsum.py:609),
//               writing what it sends as one record into the trigger's output slots
//   compact     the slots that hold a message, in slot order = the FIFO order of generation g + 1
//
// A message is ONE record (8-byte header + payload, 32 bytes for three f64 values): the generations
// are bound by the number of random cache lines they touch (~32 G lines/s measured), and three
// parallel arrays (code, slot, payload) cost three lines per delivery and two per produced message.
//
// Nothing here is a dense contraction: integer bookkeeping + a few adds per message element.
// The run ends by itself when the send rule (approx_match + SAME_COUNT) has silenced every edge.
//
// Built into libmaxsum_hip.so by hipcc.  (The host emulation of the CPU tests compiles this very
// file against serial stand-ins for the two hipCUB primitives, tests/emu/hipcub/.)
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/maxsum_gpu.h"

extern "C" __attribute__((visibility("hidden"))) void mxs_set_last_error(const char* msg);  // engine.hip

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace amx {

constexpr int SAME_COUNT = 4;  // maxsum.py:106
constexpr int TPB = 256;

template <typename U>
struct Buf {
    U* p = nullptr;
    size_t n = 0;
    hipError_t reserve(size_t count) {  // contents are NOT kept
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        // a generation is about three times the previous one while the message count explodes:
        // head-room of 3x (at most 2^30 elements) so that the buffers are re-allocated every other
        // generation at most (hipFree + hipMalloc of GB-sized buffers dominated the host side)
        const size_t room = count * 3 < ((size_t)1 << 30) ? count * 3 : ((size_t)1 << 30);
        n = count + room + 1024;
        return hipMalloc((void**)&p, n * sizeof(U));
    }
    hipError_t upload(const std::vector<U>& h) {
        hipError_t e = reserve(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice);
    }
    ~Buf() {
        if (p) (void)hipFree(p);
    }
};

template <typename T>
struct Dev {  // what the kernels see
    int32_t n_vars, n_factors, n_edges, dmax, is_max, start_mode, damp_f, damp_v;
    int32_t rs;  // elements of T per queue record (rec_stride)
    T damping, stability;
    const int32_t *dom_size, *factor_rowptr, *edge_var, *edge_factor, *var_rowptr, *var_edges, *init_idx;
    const int64_t *table_off, *cost_off, *msg_off;
    const T *tables, *var_cost;
    T *f_cost, *f_prev, *v_cost, *v_prev;
    uint8_t *f_has, *f_cnt, *v_has, *v_cnt;
    int32_t *f_nhas, *v_narr, *v_order, *sel;
    T* belief;
};

template <typename T>
__device__ __forceinline__ T absT(T x) { return x < (T)0 ? -x : x; }

// approx_match, maxsum.py:688-710, one component
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

// A message of the queues is ONE record: an 8-byte header (code = edge * 2 + direction, the first output slot of its
// handler) followed by the payload, padded to a multiple of 16 bytes (D = 3 in f64: 32 bytes).  Producing or
// delivering a message then touches one cache line at a random place, not three arrays' worth -- the generations are
// bound by how many random lines they touch (profiles/r04_amaxsum_dispatches_v2.txt: 32 G lines/s).
template <typename T>
struct RecHead {
    static constexpr int W = 8 / (int)sizeof(T);  // header length in elements of T
};
inline int rec_stride(int dmax, int word) { return (8 + dmax * word + 15) / 16 * 16 / word; }
template <typename T>
__host__ __device__ __forceinline__ int32_t rec_code(const T* r) { return ((const int32_t*)r)[0]; }
template <typename T>
__device__ __forceinline__ int32_t rec_base(const T* r) { return ((const int32_t*)r)[1]; }
template <typename T>
__device__ __forceinline__ void rec_set_head(T* r, int32_t code, int32_t base) {
    ((int32_t*)r)[0] = code;
    ((int32_t*)r)[1] = base;
}
template <typename T>
__device__ __forceinline__ T* rec_pay(T* r) { return r + RecHead<T>::W; }
template <typename T>
__device__ __forceinline__ const T* rec_pay(const T* r) { return r + RecHead<T>::W; }
// payload elements past the message's D values up to the end of the record: zeros
template <typename T>
__device__ __forceinline__ void rec_pad(const Dev<T>& g, T* r, int D) {
    for (int d = RecHead<T>::W + D; d < g.rs; ++d) r[d] = (T)0;
}

// factor_costs_for_var (maxsum.py:382-447), value d of the variable at scope position pos:
// opt over the other variables' assignments of  f_val + sum_cost,  a variable not heard from
// contributing nothing (:430-436).  Scalar loops, no local arrays.
template <typename T>
__device__ T factor_value(const Dev<T>& g, int f, int pos, int d, int64_t others) {
    const int e0 = g.factor_rowptr[f], arity = g.factor_rowptr[f + 1] - e0;
    T best = g.is_max ? -(T)INFINITY : (T)INFINITY;
    for (int64_t lin = 0; lin < others; ++lin) {
        int64_t rem = others, l = lin, t = 0;
        T sum_cost = (T)0;
        for (int i = 0; i < arity; ++i) {
            const int e = e0 + i;
            const int Di = g.dom_size[g.edge_var[e]];
            int digit;
            if (i == pos) {
                digit = d;
            } else {
                rem /= Di;
                digit = (int)(l / rem);
                l -= (int64_t)digit * rem;
                if (g.f_has[e]) sum_cost += g.f_cost[g.msg_off[e] + digit];
            }
            t = t * Di + digit;
        }
        const T cur = g.tables[g.table_off[f] + t] + sum_cost;
        if (g.is_max ? best < cur : best > cur) best = cur;
    }
    return best;
}

// The whole message of factor f (arity A as a template: positions, strides and "heard from" flags in
// registers, static indexing only) to its scope position p: out[d] = factor_value(g, f, p, d, .) for
// every d, the other variables' assignments walked like an odometer (last position fastest = the order
// of the reference's generator) instead of one 64-bit division per position and entry.
template <typename T, int A>
__device__ void factor_message(const Dev<T>& g, int f, int p, T* out) {
    const int e0 = g.factor_rowptr[f];
    int Dm[A], stride[A];
    int64_t moff[A];
    bool has[A];
    int st = 1;
#pragma unroll
    for (int i = A - 1; i >= 0; --i) {
        Dm[i] = g.dom_size[g.edge_var[e0 + i]];
        stride[i] = st;
        st *= Dm[i];
        moff[i] = g.msg_off[e0 + i];
        has[i] = g.f_has[e0 + i] != 0;
    }
    int Dp = 1, sp = 0;
    int64_t others = 1;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        if (i == p) {
            Dp = Dm[i];
            sp = stride[i];
        } else {
            others *= Dm[i];
        }
    }
    const T* tab = g.tables + g.table_off[f];
    for (int d = 0; d < Dp; ++d) {
        T best = g.is_max ? -(T)INFINITY : (T)INFINITY;
        int dig[A];
#pragma unroll
        for (int i = 0; i < A; ++i) dig[i] = 0;
        for (int64_t lin = 0; lin < others; ++lin) {
            int t = d * sp;
            T sum_cost = (T)0;
#pragma unroll
            for (int i = 0; i < A; ++i)
                if (i != p) {
                    t += dig[i] * stride[i];
                    if (has[i]) sum_cost += g.f_cost[moff[i] + dig[i]];
                }
            const T cur = tab[t] + sum_cost;
            if (g.is_max ? best < cur : best > cur) best = cur;
            bool carry = true;  // next assignment of the others
#pragma unroll
            for (int i = A - 1; i >= 0; --i)
                if (i != p && carry) {
                    dig[i] += 1;
                    carry = dig[i] == Dm[i];
                    if (carry) dig[i] = 0;
                }
        }
        out[d] = best;
    }
}

// out[0 .. D_p) for any arity
template <typename T>
__device__ void factor_message_any(const Dev<T>& g, int f, int p, T* out) {
    const int e0 = g.factor_rowptr[f], ar = g.factor_rowptr[f + 1] - e0;
    switch (ar) {
        case 1: factor_message<T, 1>(g, f, p, out); return;
        case 2: factor_message<T, 2>(g, f, p, out); return;
        case 3: factor_message<T, 3>(g, f, p, out); return;
        case 4: factor_message<T, 4>(g, f, p, out); return;
        default: break;
    }
    int64_t others = 1;
    for (int q = 0; q < ar; ++q)
        if (q != p) others *= g.dom_size[g.edge_var[e0 + q]];
    const int D = g.dom_size[g.edge_var[e0 + p]];
    for (int d = 0; d < D; ++d) out[d] = factor_value(g, f, p, d, others);
}

// apply_damping + the send rule (amaxsum.py:213-244 / 386-424) on the message sitting in `msg`
// (D values, global memory).  Returns true if it is sent (prev / count updated).
template <typename T>
__device__ bool damp_and_decide(const Dev<T>& g, T* msg, T* prev, uint8_t* cnt, int D, bool damp_on) {
    const uint8_t c = *cnt;
    bool match = c > 0;
    for (int d = 0; d < D; ++d) {
        T m = msg[d];
        if (c > 0 && damp_on) m = g.damping * prev[d] + ((T)1 - g.damping) * m;  // apply_damping: identity when prev is None
        msg[d] = m;
        if (match) match = comp_match(m, prev[d], g.stability);
    }
    if (match && c >= SAME_COUNT) return false;  // same and already sent SAME_COUNT times
    for (int d = 0; d < D; ++d) prev[d] = msg[d];
    *cnt = match ? (uint8_t)(c + 1) : (uint8_t)1;
    return true;
}

// select_value (maxsum.py:584-620): held costs summed in first-arrival order, first index wins ties
template <typename T>
__device__ void select_value(const Dev<T>& g, int v) {
    const int D = g.dom_size[v], k0 = g.var_rowptr[v], na = g.v_narr[v];
    const T* c = g.var_cost + g.cost_off[v];
    int best = 0;
    T best_c = (T)0;
    for (int d = 0; d < D; ++d) {
        T b = c[d];
        for (int r = 0; r < na; ++r) b += g.v_cost[g.msg_off[g.v_order[k0 + r]] + d];
        if (d == 0 || (g.is_max ? b > best_c : b < best_c)) {
            best = d;
            best_c = b;
        }
    }
    g.sel[v] = best;
    g.belief[v] = best_c;
}

// costs_for_factor (maxsum.py:623-676) for the slot k of variable v, written to out[0..D)
template <typename T>
__device__ void costs_for_factor(const Dev<T>& g, int v, int kout, T* out) {
    const int D = g.dom_size[v], k0 = g.var_rowptr[v], k1 = g.var_rowptr[v + 1];
    const T* c = g.var_cost + g.cost_off[v];
    T sum_cost = (T)0;
    for (int d = 0; d < D; ++d) {
        T m = c[d];
        for (int k = k0; k < k1; ++k) {
            const int e = g.var_edges[k];
            if (k == kout || !g.v_has[e]) continue;
            const T x = g.v_cost[g.msg_off[e] + d];
            sum_cost += x;
            m += x;
        }
        out[d] = m;
    }
    const T avg = sum_cost / (T)D;
    for (int d = 0; d < D; ++d) out[d] = out[d] - avg;
}

// ---- start(): variables then factors, messages straight into the generation-0 queue -------
// q_code = edge * 2 + dir (dir 0: variable -> factor, 1: factor -> variable)
template <typename T>
__global__ void k_start_count(Dev<T> g, int32_t* cnt) {  // cnt[node] = start messages of the node
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < g.n_vars) {
        const int deg = g.var_rowptr[i + 1] - g.var_rowptr[i];
        const bool sends = (deg == 1 && g.start_mode == MXS_START_LEAFS) || g.start_mode != MXS_START_LEAFS;
        cnt[i] = sends ? deg : 0;
    } else if (i < g.n_vars + g.n_factors) {
        const int f = i - g.n_vars;
        const int ar = g.factor_rowptr[f + 1] - g.factor_rowptr[f];
        const bool sends = (ar == 1 && g.start_mode != MXS_START_ALL) || g.start_mode == MXS_START_ALL;
        cnt[i] = sends ? ar : 0;
    }
}

template <typename T>
__global__ void k_start_emit(Dev<T> g, const int32_t* base, T* q_rec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < g.n_vars) {
        const int v = i;
        if (g.init_idx && g.init_idx[v] >= 0) {  // value_selection(initial_value, None)
            g.sel[v] = g.init_idx[v];
            g.belief[v] = (T)0;
        } else {
            select_value(g, v);
        }
        const int k0 = g.var_rowptr[v], deg = g.var_rowptr[v + 1] - k0;
        const bool sends = (deg == 1 && g.start_mode == MXS_START_LEAFS) || g.start_mode != MXS_START_LEAFS;
        if (!sends) return;
        for (int k = 0; k < deg; ++k) {
            const int64_t at = (int64_t)base[i] + k;
            T* r = q_rec + at * g.rs;  // (the queue was zero-filled: header base, payload padding)
            costs_for_factor(g, v, k0 + k, rec_pay(r));
            rec_set_head(r, g.var_edges[k0 + k] * 2, 0);
        }
    } else if (i < g.n_vars + g.n_factors) {
        const int f = i - g.n_vars;
        const int e0 = g.factor_rowptr[f], ar = g.factor_rowptr[f + 1] - e0;
        const bool sends = (ar == 1 && g.start_mode != MXS_START_ALL) || g.start_mode == MXS_START_ALL;
        if (!sends) return;
        for (int p = 0; p < ar; ++p) {
            const int64_t at = (int64_t)base[i] + p;
            T* r = q_rec + at * g.rs;
            factor_message_any(g, f, p, rec_pay(r));
            rec_set_head(r, (e0 + p) * 2 + 1, 0);
        }
    }
}

// ---- one generation --------------------------------------------------------------------------
template <typename T>
__global__ void k_dest(Dev<T> g, const T* q_rec, int64_t n, int32_t* dest, int32_t* cap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t code = rec_code(q_rec + i * g.rs);
    const int e = code >> 1, dir = code & 1;
    if (dir == 0) {
        const int f = g.edge_factor[e];
        dest[i] = g.n_vars + f;
        cap[i] = g.factor_rowptr[f + 1] - g.factor_rowptr[f] - 1;
    } else {
        const int v = g.edge_var[e];
        dest[i] = v;
        cap[i] = g.var_rowptr[v + 1] - g.var_rowptr[v] - 1;
    }
}

template <typename T>
__device__ void handle(const Dev<T>& g, const T* rec, T* s_rec, bool last) {
    const int32_t code = rec_code(rec);
    const T* pay = rec_pay(rec);
    T* s_out = s_rec + (int64_t)rec_base(rec) * g.rs;  // the handler's output slots
    const int e = code >> 1;
    const int v = g.edge_var[e], f = g.edge_factor[e];
    const int D = g.dom_size[v];
    if ((code & 1) == 0) {  // variable -> factor: amaxsum.py:191-250
        for (int d = 0; d < D; ++d) g.f_cost[g.msg_off[e] + d] = pay[d];
        if (!g.f_has[e]) {
            g.f_has[e] = 1;
            g.f_nhas[f] += 1;
        }
        const int e0 = g.factor_rowptr[f], ar = g.factor_rowptr[f + 1] - e0;
        if (g.f_nhas[f] != ar) return;  // still waiting for some variable (:206)
        int slot = 0;
        for (int p = 0; p < ar; ++p) {
            const int e2 = e0 + p;
            if (e2 == e) continue;  // not back to the sender
            const int D2 = g.dom_size[g.edge_var[e2]];
            T* orec = s_out + (int64_t)slot * g.rs;
            T* out = rec_pay(orec);
            factor_message_any(g, f, p, out);
            if (damp_and_decide(g, out, g.f_prev + g.msg_off[e2], &g.f_cnt[e2], D2, g.damp_f != 0)) {
                rec_pad(g, orec, D2);
                rec_set_head(orec, e2 * 2 + 1, 0);
            }
            ++slot;
        }
    } else {  // factor -> variable: amaxsum.py:366-424
        for (int d = 0; d < D; ++d) g.v_cost[g.msg_off[e] + d] = pay[d];
        const int k0 = g.var_rowptr[v], k1 = g.var_rowptr[v + 1];
        if (!g.v_has[e]) {
            g.v_has[e] = 1;
            g.v_order[k0 + g.v_narr[v]] = e;
            g.v_narr[v] += 1;
        }
        // select_value only leaves sel / belief behind: of a run of messages to the same variable,
        // the one after the last delivery is what the generation ends with
        if (last) select_value(g, v);
        int slot = 0;
        for (int k = k0; k < k1; ++k) {
            const int e2 = g.var_edges[k];
            if (e2 == e) continue;
            T* orec = s_out + (int64_t)slot * g.rs;
            T* out = rec_pay(orec);
            costs_for_factor(g, v, k, out);
            if (damp_and_decide(g, out, g.v_prev + g.msg_off[e2], &g.v_cnt[e2], D, g.damp_v != 0)) {
                rec_pad(g, orec, D);
                rec_set_head(orec, e2 * 2, 0);
            }
            ++slot;
        }
    }
}

// ---- a destination's whole queue of one generation, state in registers ------------------------
// The handler chain of a destination is sequential, and the destinations with the most mail (the
// hubs, and the factors next to them) decide how long a generation lasts.  `handle` above costs a
// couple of hundred DEPENDENT global loads per message; the two chains below load the
// destination's state once, keep it in registers while its queue is delivered, and write it back.

// damp_and_decide on a message held in registers (D <= N values)
template <typename T, int N>
__device__ __forceinline__ bool damp_and_decide_reg(const Dev<T>& g, T (&m)[N], T (&prev)[N], uint8_t& cnt, int D,
                                                    bool damp_on) {
    const uint8_t c = cnt;
    bool match = c > 0;
#pragma unroll
    for (int d = 0; d < N; ++d)
        if (d < D) {
            T x = m[d];
            if (c > 0 && damp_on) x = g.damping * prev[d] + ((T)1 - g.damping) * x;
            m[d] = x;
            if (match) match = comp_match(x, prev[d], g.stability);
        }
    if (match && c >= SAME_COUNT) return false;
#pragma unroll
    for (int d = 0; d < N; ++d) prev[d] = m[d];
    cnt = match ? (uint8_t)(c + 1) : (uint8_t)1;
    return true;
}

// The messages of a generation in DESTINATION-SORTED order (k_permute: one streaming pass gathers what the stable
// sort's index array points at): a destination's queue is then a contiguous run -- code, first output slot, payload
// at consecutive addresses -- and the chains below read it sequentially, RING deliveries ahead of the one they
// handle.  (Round 3 followed order[r] -> q_code / q_pay / slot_base per delivery, one ahead: three dependent random
// loads under full load, 3-5 us per step of a chain that is sequential anyway; the longest queue of a generation is
// what the generation lasts -- profiles/r04_amaxsum_dispatches_v1.txt.)
template <typename T>
struct Sorted {
    const int32_t* dst;        // [n] destination computation (variables first, then n_vars + factor)
    const T* rec;              // [n * rs] the records (header: code, first output slot of the delivery's handler)
    const int32_t* seg_first;  // [segments] position of the first message of the t-th destination to run
    const uint64_t* seg_key;   // [segments] class << 32 | ~(queue length)
    int64_t n;
};
constexpr int RING = 4;   // deliveries in flight per chain of a binary factor (a lane each: a small step body)
constexpr int VRING = 2;  // ... of a variable's lane group (the step body is unrolled once per ring entry: with 4 the
                          // kernels needed 248 VGPRs + scratch and 47-75 KB of code, and ran at half the speed)
template <typename T, int N>
struct Mail {
    int32_t code;
    int32_t base;
    T pay[N];
};
template <typename T, int N>
__device__ __forceinline__ void fetch_mail(const Dev<T>& g, const Sorted<T>& sq, int64_t pos, Mail<T, N>& m) {
    const T* r = (const T*)__builtin_assume_aligned(sq.rec + pos * g.rs, 16);
    m.code = rec_code(r);
    m.base = rec_base(r);
#pragma unroll
    for (int d = 0; d < N; ++d) m.pay[d] = rec_pay(r)[d < g.dmax ? d : 0];
}
template <typename T>
__device__ __forceinline__ int64_t queue_length(const Sorted<T>& sq, int64_t t) {
    return (int64_t)(uint32_t)~(uint32_t)sq.seg_key[t];
}

// Lane K of every 16-lane row to all lanes of the row: one DPP move per 32 bits (row_newbcast) -- VALU, where
// __shfl goes through the LDS crossbar (ds_bpermute).  The 16-lane groups of chain_variable are rows.
template <int K>
__device__ __forceinline__ double row_bcast(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + K, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ float row_bcast(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x150 + K, 0xf, 0xf, false));
}
template <typename F, int... I>
__device__ __forceinline__ void amx_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void amx_static_for(F&& f) {
    amx_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// A variable of domain size D (template) and degree <= GROUP: a group of GROUP lanes (8, 16 or the
// whole wave), lane k of the group = the variable's k-th factor (var_edges order) holding that
// factor's last message, the message last sent to it and its send counter.  Per delivered message:
// the sender's lane takes the costs, every other lane builds its factor's message from the held costs
// of the group (D * deg cross-lane reads, the reference's order of additions: d outer, factors inner,
// maxsum.py:651-665), damps, applies the send rule and writes its output slot.  The groups of a
// wave walk their own queues in lock step (a group whose queue is done idles: the destinations are
// sorted by queue length, so the queues of a wave are about equally long); `t` = the group's
// destination in seg_first, groups past `seg_end` have none.
template <typename T, int D, int GROUP>
__device__ void chain_variable(const Dev<T>& g, const Sorted<T>& sq, int64_t t, int64_t seg_end, T* s_rec) {
    constexpr bool WHOLE = GROUP == 64;
    const int lane = (int)threadIdx.x & 63;
    const int gl = lane % GROUP, gbase = lane - gl;
    const unsigned long long gmask = WHOLE ? ~0ull : (((1ull << (GROUP % 64)) - 1ull) << gbase);
    const bool valid = t < seg_end;
    const int64_t p = valid ? sq.seg_first[t] : 0;
    const int64_t len = valid ? queue_length(sq, t) : 0;
    const int32_t dst = sq.dst[p];
    const int v = valid ? dst : 0;
    const int k0 = g.var_rowptr[v], deg = valid ? g.var_rowptr[v + 1] - k0 : 0;
    const bool active = gl < deg;
    const int ek = active ? g.var_edges[k0 + gl] : -1;
    const int64_t mo = active ? g.msg_off[ek] : 0;
    T held[D], prev[D], c[D];
    uint8_t cnt = 0;
    bool has = false;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        held[d] = active ? g.v_cost[mo + d] : (T)0;
        prev[d] = active ? g.v_prev[mo + d] : (T)0;
        c[d] = g.var_cost[g.cost_off[v] + d];
    }
    if (active) {
        cnt = g.v_cnt[ek];
        has = g.v_has[ek] != 0;
    }
    int narr = valid ? g.v_narr[v] : 0;
    int my_rank = -1;  // first-arrival rank of this lane's factor (select_value sums in that order)
    for (int r = 0; r < narr; ++r)
        if (g.v_order[k0 + r] == ek) my_rank = r;
    // the queue, VRING deliveries ahead; the groups of a wave walk their queues in lock step (every lane executes
    // every step -- the ballots and cross-lane reads need the whole wave -- a group past its queue's end idles)
    Mail<T, D> ring[VRING];
#pragma unroll
    for (int j = 0; j < VRING; ++j) fetch_mail<T, D>(g, sq, p + (j < len ? j : (len > 0 ? len - 1 : 0)), ring[j]);
    auto step = [&](const Mail<T, D>& ml, bool alive) __attribute__((always_inline)) {
            const int e = ml.code >> 1;
            const unsigned long long from = __ballot(alive && active && ek == e) & gmask;
            const int j = from ? __builtin_ctzll(from) - gbase : -1;  // the sender's lane of the group
            const bool mine = alive && gl == j;
            if (mine) {
#pragma unroll
                for (int d = 0; d < D; ++d) held[d] = ml.pay[d];
            }
            const bool is_new = (__ballot(mine && !has) & gmask) != 0ull;
            if (mine && !has) {
                has = true;
                my_rank = narr;
                g.v_order[k0 + narr] = e;
            }
            narr += is_new ? 1 : 0;
            const unsigned long long hasmask = (__ballot(has) & gmask) >> gbase;
            // costs_for_factor for this lane's factor
            T m[D];
            T sum_cost = (T)0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                T md = c[d];
                if constexpr (WHOLE) {
                    for (int k2 = 0; k2 < deg; ++k2) {  // deg is wave-uniform here
                        const T x = __shfl(held[d], k2, 64);
                        if (k2 != gl && ((hasmask >> k2) & 1ull)) {
                            sum_cost += x;
                            md += x;
                        }
                    }
                } else if constexpr (GROUP == 16) {  // the group is a DPP row
                    amx_static_for<16>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int k2 = decltype(kc)::value;
                        const T x = row_bcast<k2>(held[d]);
                        if (k2 != gl && ((hasmask >> k2) & 1ull)) {
                            sum_cost += x;
                            md += x;
                        }
                    });
                } else {
#pragma unroll
                    for (int k2 = 0; k2 < GROUP; ++k2) {
                        const T x = __shfl(held[d], k2, GROUP);
                        if (k2 != gl && ((hasmask >> k2) & 1ull)) {  // lanes past the degree never "have"
                            sum_cost += x;
                            md += x;
                        }
                    }
                }
                m[d] = md;
            }
            const T avg = sum_cost / (T)D;
#pragma unroll
            for (int d = 0; d < D; ++d) m[d] = m[d] - avg;
            if (alive && active && gl != j) {
                if (damp_and_decide_reg<T, D>(g, m, prev, cnt, D, g.damp_v != 0)) {
                    T* o = (T*)__builtin_assume_aligned(s_rec + (int64_t)(ml.base + (gl < j ? gl : gl - 1)) * g.rs, 16);
#pragma unroll
                    for (int d = 0; d < D; ++d) rec_pay(o)[d] = m[d];
                    rec_pad(g, o, D);
                    rec_set_head(o, ek * 2, 0);
                }
            }
    };
    for (int64_t r0 = 0; __ballot(r0 < len) != 0ull; r0 += VRING) {
#pragma unroll
        for (int j = 0; j < VRING; ++j) {
            const int64_t r = r0 + j;
            step(ring[j], r < len);
            if (r + VRING < len) fetch_mail<T, D>(g, sq, p + r + VRING, ring[j]);  // this register set's next tenant
        }
    }
    // select_value on what is held now (maxsum.py:584-620): factors in first-arrival order
    {
        int best = 0;
        T best_c = (T)0;
        const int rmax = WHOLE ? narr : GROUP;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            T b = c[d];
            for (int r = 0; r < rmax; ++r) {
                const unsigned long long who = __ballot(my_rank == r) & gmask;
                const int src = who ? __builtin_ctzll(who) - gbase : 0;
                const T x = __shfl(held[d], src, GROUP);
                if (r < narr) b += x;
            }
            if (d == 0 || (g.is_max ? b > best_c : b < best_c)) {
                best = d;
                best_c = b;
            }
        }
        if (valid && gl == 0) {
            g.sel[v] = best;
            g.belief[v] = best_c;
            g.v_narr[v] = narr;
        }
    }
    if (active) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            g.v_cost[mo + d] = held[d];
            g.v_prev[mo + d] = prev[d];
        }
        g.v_cnt[ek] = cnt;
        g.v_has[ek] = has ? 1 : 0;
    }
}

// out[y] = opt over the sender's values x of  tab(x, y) + (0 + cost[x])  -- factor_costs_for_var of
// a binary factor; SENDER_FIRST: the sender is scope position 0 (rows of the table)
template <typename T, bool SENDER_FIRST>
__device__ __forceinline__ void factor2_message(const Dev<T>& g, const T (&tab)[16], const T (&cost)[4], int Ds, int Dt,
                                                T (&out)[4]) {
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        T best = g.is_max ? -(T)INFINITY : (T)INFINITY;
#pragma unroll
        for (int x = 0; x < 4; ++x)
            if (x < Ds && y < Dt) {
                const T cur = (SENDER_FIRST ? tab[x * 4 + y] : tab[y * 4 + x]) + ((T)0 + cost[x]);
                if (g.is_max ? best < cur : best > cur) best = cur;
            }
        out[y] = best;
    }
}

// A binary factor over domains of at most 4 values: one lane, table / held costs / last-sent
// messages in registers.
template <typename T>
__device__ void chain_factor2(const Dev<T>& g, int f, const Sorted<T>& sq, int64_t p, int64_t len, T* s_rec) {
    const int eA = g.factor_rowptr[f], eB = eA + 1;
    const int DA = g.dom_size[g.edge_var[eA]], DB = g.dom_size[g.edge_var[eB]];
    const int64_t moA = g.msg_off[eA], moB = g.msg_off[eB];
    T tab[16], cA[4], cB[4], pA[4], pB[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int y = 0; y < 4; ++y)
            tab[x * 4 + y] = g.tables[g.table_off[f] + (int64_t)(x < DA ? x : DA - 1) * DB + (y < DB ? y : DB - 1)];
        cA[x] = g.f_cost[moA + (x < DA ? x : DA - 1)];
        pA[x] = g.f_prev[moA + (x < DA ? x : DA - 1)];
        cB[x] = g.f_cost[moB + (x < DB ? x : DB - 1)];
        pB[x] = g.f_prev[moB + (x < DB ? x : DB - 1)];
    }
    bool hasA = g.f_has[eA] != 0, hasB = g.f_has[eB] != 0;
    uint8_t cntA = g.f_cnt[eA], cntB = g.f_cnt[eB];
    Mail<T, 4> ring[RING];
#pragma unroll
    for (int j = 0; j < RING; ++j) fetch_mail<T, 4>(g, sq, p + (j < len ? j : len - 1), ring[j]);
    auto step = [&](const Mail<T, 4>& ml) __attribute__((always_inline)) {
        const int e = ml.code >> 1;
        T out[4];
        if (e == eA) {  // from scope variable 0: the message goes to variable 1
#pragma unroll
            for (int x = 0; x < 4; ++x) cA[x] = ml.pay[x];  // past the domain: the zero padding of the slot
            hasA = true;
            if (hasB) {  // else: still waiting for the other variable (amaxsum.py:206)
                factor2_message<T, true>(g, tab, cA, DA, DB, out);
                if (damp_and_decide_reg<T, 4>(g, out, pB, cntB, DB, g.damp_f != 0)) {
                    T* o = (T*)__builtin_assume_aligned(s_rec + (int64_t)ml.base * g.rs, 16);
#pragma unroll
                    for (int y = 0; y < 4; ++y)
                        if (y < DB) rec_pay(o)[y] = out[y];
                    rec_pad(g, o, DB);
                    rec_set_head(o, eB * 2 + 1, 0);
                }
            }
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) cB[x] = ml.pay[x];
            hasB = true;
            if (hasA) {
                factor2_message<T, false>(g, tab, cB, DB, DA, out);
                if (damp_and_decide_reg<T, 4>(g, out, pA, cntA, DA, g.damp_f != 0)) {
                    T* o = (T*)__builtin_assume_aligned(s_rec + (int64_t)ml.base * g.rs, 16);
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if (x < DA) rec_pay(o)[x] = out[x];
                    rec_pad(g, o, DA);
                    rec_set_head(o, eA * 2 + 1, 0);
                }
            }
        }
    };
    for (int64_t r0 = 0; r0 < len; r0 += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            const int64_t r = r0 + j;
            if (r < len) step(ring[j]);
            if (r + RING < len) fetch_mail<T, 4>(g, sq, p + r + RING, ring[j]);  // this register set's next tenant
        }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        if (x < DA) {
            g.f_cost[moA + x] = cA[x];
            g.f_prev[moA + x] = pA[x];
        }
        if (x < DB) {
            g.f_cost[moB + x] = cB[x];
            g.f_prev[moB + x] = pB[x];
        }
    }
    g.f_has[eA] = hasA ? 1 : 0;
    g.f_has[eB] = hasB ? 1 : 0;
    g.f_cnt[eA] = cntA;
    g.f_cnt[eB] = cntB;
    g.f_nhas[f] = (hasA ? 1 : 0) + (hasB ? 1 : 0);
}

// Destination classes of a generation (what runs its queue):
constexpr int CLS_FACTOR2 = 0;  // binary factor, domains <= 4: a lane (chain_factor2), 64 per wave
constexpr int CLS_VAR8 = 1;     // variable, domain 2..4, degree <= 8: 8 lanes, 8 per wave
constexpr int CLS_VAR16 = 2;    //                        degree <= 16: 16 lanes, 4 per wave
constexpr int CLS_VAR64 = 3;    //                        degree <= 64: the wave
constexpr int CLS_GENERIC = 4;  // everything else: a lane on the per-message handler, 64 per wave
constexpr int N_CLS = 5;

template <typename T>
__device__ __forceinline__ int class_of(const Dev<T>& g, int32_t dst) {
    if (dst < g.n_vars) {
        const int D = g.dom_size[dst], deg = g.var_rowptr[dst + 1] - g.var_rowptr[dst];
        if (D < 2 || D > 4 || deg > 64) return CLS_GENERIC;
        return deg <= 8 ? CLS_VAR8 : (deg <= 16 ? CLS_VAR16 : CLS_VAR64);
    }
    const int f = dst - g.n_vars, e0 = g.factor_rowptr[f];
    return (g.factor_rowptr[f + 1] - e0 == 2 && g.dom_size[g.edge_var[e0]] <= 4 && g.dom_size[g.edge_var[e0 + 1]] <= 4)
               ? CLS_FACTOR2 : CLS_GENERIC;
}

template <typename T, int GROUP>
__device__ __forceinline__ void variables_of_wave(const Dev<T>& g, const Sorted<T>& sq, int64_t seg_begin, int64_t seg_end,
                                                  T* s_rec) {
    constexpr int PER_WAVE = 64 / GROUP;
    const int lane = (int)threadIdx.x & 63;
    const int64_t t = seg_begin + (int64_t)blockIdx.x * PER_WAVE + lane / GROUP;
    // the domain sizes of the wave's variables: one pass per size present (wave-uniform branches)
    const int myD = t < seg_end ? g.dom_size[sq.dst[sq.seg_first[t]]] : 0;
    for (int D = 2; D <= 4; ++D) {
        if (__ballot(myD == D) == 0ull) continue;
        const int64_t tt = myD == D ? t : seg_end;  // the other groups sit this pass out
        if (D == 2) chain_variable<T, 2, GROUP>(g, sq, tt, seg_end, s_rec);
        else if (D == 3) chain_variable<T, 3, GROUP>(g, sq, tt, seg_end, s_rec);
        else chain_variable<T, 4, GROUP>(g, sq, tt, seg_end, s_rec);
    }
}

// sq.seg_first[t]: position of the first message of the t-th destination to run -- by class, longest queues
// first (step()); a launch runs the destinations [seg_begin, seg_end) of one class.  Blocks of one wave; a kernel
// per class, so that each has the registers of its own path only.
template <typename T, int GROUP>
__global__ void __launch_bounds__(64) k_process_vars(Dev<T> g, Sorted<T> sq, int64_t seg_begin, int64_t seg_end, T* s_rec) {
    variables_of_wave<T, GROUP>(g, sq, seg_begin, seg_end, s_rec);
}

template <typename T, bool FACTOR2>  // (two kernels: the binary-factor chains do not pay for the generic handler's registers)
__global__ void __launch_bounds__(64) k_process_lanes(Dev<T> g, Sorted<T> sq, int64_t seg_begin, int64_t seg_end, T* s_rec) {
    const int64_t t = seg_begin + (int64_t)blockIdx.x * 64 + ((int)threadIdx.x & 63);
    if (t >= seg_end) return;
    const int64_t p = sq.seg_first[t], len = queue_length(sq, t);
    const int32_t dst = sq.dst[p];
    if constexpr (FACTOR2) {
        chain_factor2<T>(g, dst - g.n_vars, sq, p, len, s_rec);
        return;
    }
    for (int64_t r = p; r < p + len; ++r)  // its messages, in FIFO order
        handle(g, sq.rec + r * g.rs, s_rec, r + 1 == p + len);
}

// the generation's records gathered into destination-sorted order (Sorted): order[p] = FIFO index of the p-th message
// after the stable sort by destination.  One thread per 16-byte piece: the writes are one contiguous stream, the
// reads one random line per message.
template <typename T>
__global__ void k_permute(const T* q_rec, const int32_t* order, int64_t n, int rs, T* m_rec) {
    const int pieces = rs * (int)sizeof(T) / 16;
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = x / pieces;
    const int k = (int)(x - p * pieces);
    if (p >= n) return;
    struct alignas(16) P16 { uint32_t w[4]; };
    ((P16*)(m_rec + p * rs))[k] = ((const P16*)(q_rec + (int64_t)order[p] * rs))[k];
}
// the first output slot of every message's handler into its record (after the scan of the capacities)
template <typename T>
__global__ void k_stamp(T* q_rec, const int64_t* slot_base, int64_t n, int rs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ((int32_t*)(q_rec + i * rs))[1] = (int32_t)slot_base[i];  // (fewer than 2^31 slots per generation: step())
}

__global__ void k_iota(int32_t* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}

// head[p] = 1 where a destination's run starts in the sorted order
__global__ void k_heads(const int32_t* dest_sorted, int64_t n, int32_t* head) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) head[p] = (p == 0 || dest_sorted[p] != dest_sorted[p - 1]) ? 1 : 0;
}

// seg_pos[idx[p]] = p for every head p
__global__ void k_seg_starts(const int32_t* head, const int64_t* idx, int64_t n, int32_t* seg_pos) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && head[p]) seg_pos[idx[p]] = (int32_t)p;
}

// key = class << 32 | ~length: an ascending sort groups the classes and runs the longest queues of
// each class first
template <typename T>
__global__ void k_seg_keys(Dev<T> g, const int32_t* dest_sorted, const int32_t* seg_pos, int64_t n_seg, int64_t n,
                           int generic_only, uint64_t* key) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seg) return;
    const int64_t end = t + 1 < n_seg ? seg_pos[t + 1] : n;
    const int cls = generic_only ? CLS_GENERIC : class_of(g, dest_sorted[seg_pos[t]]);
    key[t] = ((uint64_t)cls << 32) | (uint32_t)~(uint32_t)(end - seg_pos[t]);
}

// first[c] = number of sorted keys below class c (c = 0 .. N_CLS)
__global__ void k_class_bounds(const uint64_t* key_sorted, int64_t n_seg, int64_t* first) {
    const int c = threadIdx.x;
    if (c > N_CLS) return;
    int64_t lo = 0, hi = n_seg;
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if ((key_sorted[mid] >> 32) < (uint64_t)c) lo = mid + 1;
        else hi = mid;
    }
    first[c] = lo;
}

template <typename T>
__global__ void k_gather(const T* s_rec, const int64_t* pos, int64_t n_slots, int rs, T* q_rec) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots || rec_code(s_rec + s * rs) < 0) return;
    const int64_t at = pos[s];
    for (int d = 0; d < rs; ++d) q_rec[at * rs + d] = s_rec[s * rs + d];
}

// 1 where output slot s holds a message: what the compaction scan sums, read straight from the records
template <typename T>
struct SlotFilled {
    const T* rec;
    int rs;
    __host__ __device__ __forceinline__ int64_t operator()(int64_t slot) const { return rec_code(rec + slot * rs) >= 0 ? 1 : 0; }
};

struct Base {
    virtual ~Base() {}
    virtual int init(const mxs_graph& g, const mxs_params& p, int device) = 0;
    virtual int reset() = 0;
    virtual int run(int32_t max_generations, int64_t* delivered) = 0;
    virtual int get_assignment(int32_t* idx, double* belief) = 0;
    virtual int get_messages(double* fc, double* vc, double* fp, double* vp, uint8_t* fh, uint8_t* vh, uint8_t* fn,
                             uint8_t* vn) = 0;
    virtual int eval_cost(const int32_t* idx, double infinity, double* cost, int64_t* viol) = 0;
    virtual int update_table(int32_t factor, const double* table, int64_t n) = 0;
    int32_t next_generation = 0;
    int64_t pending = 0, delivered_total = 0;
    std::vector<int64_t> gen_sizes;
};

static int fail(int code, const std::string& msg) {
    mxs_set_last_error(msg.c_str());
    return code;
}
#define AMX_TRY(call)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail(MXS_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__));       \
    } while (0)

template <typename T>
struct Engine : Base {
    int device = 0;
    Dev<T> g{};
    // One stream per destination class: the class kernels of a generation work on disjoint destinations and run
    // SIDE BY SIDE (the few hundred waves of the high-degree variables are a latency chain of their own: alone
    // they took as long as the 12 000 waves of the low-degree ones before them -- profiles/r04_amaxsum_dispatches_v1.txt).
    // Blocking streams: they wait for the null stream's earlier work, the null stream's later work waits for them.
    hipStream_t cls_stream[N_CLS] = {};
    ~Engine() override {
        for (hipStream_t st : cls_stream)
            if (st) (void)hipStreamDestroy(st);
    }
    std::vector<int32_t> h_dom, h_frow, h_evar, h_vrow, h_vedges;
    std::vector<int64_t> h_toff, h_coff, h_moff;
    std::vector<double> h_tables, h_eval_cost;
    Buf<int32_t> dom_size, factor_rowptr, edge_var, edge_factor, var_rowptr, var_edges, init_idx;
    Buf<int64_t> table_off, cost_off, msg_off;
    Buf<T> tables, var_cost, f_cost, f_prev, v_cost, v_prev, belief;
    Buf<uint8_t> f_has, f_cnt, v_has, v_cnt;
    Buf<int32_t> f_nhas, v_narr, v_order, sel;
    // queue of the current generation, work arrays of a step
    Buf<int32_t> dest, dest_sorted, order, order_in, cap, start_cnt, start_base;
    Buf<T> q_rec, q_rec2, s_rec, m_rec;  // queue of the generation, of the next one, the output slots, the sorted copy
    Buf<int64_t> slot_base, pos, cap64, head_idx;
    Buf<int32_t> head, seg_pos, seg_first;
    Buf<uint64_t> seg_key, seg_key_sorted;
    Buf<int64_t> cls_first;
    Buf<uint8_t> temp;
    int64_t nm = 0;

    int grid(int64_t n) const { return (int)((n + TPB - 1) / TPB); }

    int init(const mxs_graph& G, const mxs_params& p, int dev) override {
        device = dev;
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            return fail(MXS_E_NODEVICE, "no HIP device visible: the Max-Sum engine has no CPU fallback");
        if (dev < 0 || dev >= count) return fail(MXS_E_INVALID, "device index out of range");
        AMX_TRY(hipSetDevice(dev));
        const int nV = G.n_vars, nF = G.n_factors, nE = G.n_edges;
        if (nV < 0 || nF < 0 || nE < 0) return fail(MXS_E_INVALID, "negative size");
        h_dom.assign(G.dom_size, G.dom_size + nV);
        h_frow.assign(G.factor_rowptr, G.factor_rowptr + nF + 1);
        h_evar.assign(G.edge_var, G.edge_var + nE);
        h_vrow.assign(G.var_rowptr, G.var_rowptr + nV + 1);
        h_vedges.assign(G.var_edges, G.var_edges + nE);
        h_toff.assign(G.table_off, G.table_off + nF + 1);
        int dmax = 1;
        h_coff.assign(nV + 1, 0);
        for (int v = 0; v < nV; ++v) {
            if (h_dom[v] < 1 || h_dom[v] > 4096) return fail(MXS_E_INVALID, "domain size not in 1..4096");
            dmax = std::max(dmax, h_dom[v]);
            h_coff[v + 1] = h_coff[v] + h_dom[v];
        }
        h_moff.assign(nE + 1, 0);
        std::vector<int32_t> h_efac(nE);
        for (int f = 0; f < nF; ++f) {
            if (h_frow[f + 1] <= h_frow[f] || h_frow[f + 1] - h_frow[f] > MXS_MAX_ARITY) return fail(MXS_E_INVALID, "bad factor arity");
            for (int e = h_frow[f]; e < h_frow[f + 1]; ++e) h_efac[e] = f;
        }
        for (int e = 0; e < nE; ++e) {
            if (h_evar[e] < 0 || h_evar[e] >= nV) return fail(MXS_E_INVALID, "edge_var out of range");
            h_moff[e + 1] = h_moff[e] + h_dom[h_evar[e]];
        }
        nm = h_moff[nE];
        h_tables.assign(G.tables, G.tables + h_toff[nF]);
        const double* ev = G.eval_var_cost ? G.eval_var_cost : G.var_cost;
        h_eval_cost.assign(ev, ev + h_coff[nV]);
        std::vector<T> tt(h_tables.size()), vc((size_t)h_coff[nV]);
        for (size_t i = 0; i < tt.size(); ++i) tt[i] = (T)h_tables[i];
        for (size_t i = 0; i < vc.size(); ++i) vc[i] = (T)G.var_cost[i];
        AMX_TRY(dom_size.upload(h_dom));
        AMX_TRY(factor_rowptr.upload(h_frow));
        AMX_TRY(edge_var.upload(h_evar));
        AMX_TRY(edge_factor.upload(h_efac));
        AMX_TRY(var_rowptr.upload(h_vrow));
        AMX_TRY(var_edges.upload(h_vedges));
        AMX_TRY(table_off.upload(h_toff));
        AMX_TRY(cost_off.upload(h_coff));
        AMX_TRY(msg_off.upload(h_moff));
        AMX_TRY(tables.upload(tt));
        AMX_TRY(var_cost.upload(vc));
        if (G.init_idx) {
            std::vector<int32_t> ii(G.init_idx, G.init_idx + nV);
            AMX_TRY(init_idx.upload(ii));
        }
        AMX_TRY(f_cost.reserve(nm + 1));
        AMX_TRY(f_prev.reserve(nm + 1));
        AMX_TRY(v_cost.reserve(nm + 1));
        AMX_TRY(v_prev.reserve(nm + 1));
        AMX_TRY(f_has.reserve(nE + 1));
        AMX_TRY(f_cnt.reserve(nE + 1));
        AMX_TRY(v_has.reserve(nE + 1));
        AMX_TRY(v_cnt.reserve(nE + 1));
        AMX_TRY(v_order.reserve(nE + 1));
        AMX_TRY(f_nhas.reserve(nF + 1));
        AMX_TRY(v_narr.reserve(nV + 1));
        AMX_TRY(sel.reserve(nV + 1));
        AMX_TRY(belief.reserve(nV + 1));
        g.n_vars = nV; g.n_factors = nF; g.n_edges = nE; g.dmax = dmax;
        g.rs = rec_stride(dmax, (int)sizeof(T));
        g.is_max = p.mode == MXS_MODE_MAX;
        g.start_mode = p.start_messages;
        g.damp_f = (p.damping_nodes & MXS_DAMP_FACTORS) ? 1 : 0;
        g.damp_v = (p.damping_nodes & MXS_DAMP_VARS) ? 1 : 0;
        g.damping = (T)p.damping;
        g.stability = (T)p.stability;
        g.dom_size = dom_size.p; g.factor_rowptr = factor_rowptr.p; g.edge_var = edge_var.p;
        g.edge_factor = edge_factor.p; g.var_rowptr = var_rowptr.p; g.var_edges = var_edges.p;
        g.init_idx = G.init_idx ? init_idx.p : nullptr;
        g.table_off = table_off.p; g.cost_off = cost_off.p; g.msg_off = msg_off.p;
        g.tables = tables.p; g.var_cost = var_cost.p;
        g.f_cost = f_cost.p; g.f_prev = f_prev.p; g.v_cost = v_cost.p; g.v_prev = v_prev.p;
        g.f_has = f_has.p; g.f_cnt = f_cnt.p; g.v_has = v_has.p; g.v_cnt = v_cnt.p;
        g.f_nhas = f_nhas.p; g.v_narr = v_narr.p; g.v_order = v_order.p; g.sel = sel.p; g.belief = belief.p;
        return reset();
    }

    // A few bytes back to the host between the phases of a step.  The device is drained FIRST with
    // hipDeviceSynchronize (an active wait): a blocking hipMemcpy behind a long kernel was measured
    // to return milliseconds late (16 generations at 100k variables: 0.40 s of wall time for 0.05 s of
    // kernels; 0.12 s with the explicit synchronisation).
    hipError_t read_back(void* dst, const void* src, size_t bytes) {
        hipError_t e = hipDeviceSynchronize();
        return e != hipSuccess ? e : hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
    }

    // exclusive prefix sum of int32 counts into int64 (hipCUB), total read back
    int scan32(const int32_t* in, int64_t* out, int64_t n, int64_t* total) {
        *total = 0;
        if (n == 0) return MXS_OK;
        size_t bytes = 0;
        hipcub::TransformInputIterator<int64_t, hipcub::CastOp<int64_t>, const int32_t*> it(in, hipcub::CastOp<int64_t>());
        AMX_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, it, out, (int)n));
        AMX_TRY(temp.reserve(bytes));
        AMX_TRY(hipcub::DeviceScan::ExclusiveSum(temp.p, bytes, it, out, (int)n));
        int64_t last_off = 0;
        int32_t last_cnt = 0;
        AMX_TRY(read_back(&last_off, out + n - 1, 8));
        AMX_TRY(read_back(&last_cnt, in + n - 1, 4));
        *total = last_off + last_cnt;
        return MXS_OK;
    }

    int reset() override {
        AMX_TRY(hipSetDevice(device));
        AMX_TRY(hipMemset(f_cost.p, 0, sizeof(T) * (nm + 1)));
        AMX_TRY(hipMemset(f_prev.p, 0, sizeof(T) * (nm + 1)));
        AMX_TRY(hipMemset(v_cost.p, 0, sizeof(T) * (nm + 1)));
        AMX_TRY(hipMemset(v_prev.p, 0, sizeof(T) * (nm + 1)));
        AMX_TRY(hipMemset(f_has.p, 0, g.n_edges + 1));
        AMX_TRY(hipMemset(f_cnt.p, 0, g.n_edges + 1));
        AMX_TRY(hipMemset(v_has.p, 0, g.n_edges + 1));
        AMX_TRY(hipMemset(v_cnt.p, 0, g.n_edges + 1));
        AMX_TRY(hipMemset(f_nhas.p, 0, sizeof(int32_t) * (g.n_factors + 1)));
        AMX_TRY(hipMemset(v_narr.p, 0, sizeof(int32_t) * (g.n_vars + 1)));
        AMX_TRY(hipMemset(sel.p, 0, sizeof(int32_t) * (g.n_vars + 1)));
        AMX_TRY(hipMemset(belief.p, 0, sizeof(T) * (g.n_vars + 1)));
        next_generation = 0;
        delivered_total = 0;
        gen_sizes.clear();
        // start(): every computation in graph order; its messages are generation 0
        const int64_t nodes = (int64_t)g.n_vars + g.n_factors;
        pending = 0;
        if (nodes > 0) {
            AMX_TRY(start_cnt.reserve(nodes));
            AMX_TRY(slot_base.reserve(nodes));
            hipLaunchKernelGGL((k_start_count<T>), dim3(grid(nodes)), dim3(TPB), 0, 0, g, start_cnt.p);
            AMX_TRY(hipGetLastError());
            int64_t total = 0;
            { int rc = scan32(start_cnt.p, slot_base.p, nodes, &total); if (rc) return rc; }
            // (node bases fit 32 bits: at most one message per directed edge)
            AMX_TRY(start_base.reserve(nodes));
            {
                std::vector<int64_t> hb(nodes);
                AMX_TRY(hipMemcpy(hb.data(), slot_base.p, 8 * nodes, hipMemcpyDeviceToHost));
                std::vector<int32_t> hb32(nodes);
                for (int64_t i = 0; i < nodes; ++i) hb32[i] = (int32_t)hb[i];
                AMX_TRY(hipMemcpy(start_base.p, hb32.data(), 4 * nodes, hipMemcpyHostToDevice));
            }
            AMX_TRY(q_rec.reserve((total + 1) * g.rs));
            AMX_TRY(hipMemset(q_rec.p, 0, sizeof(T) * (total + 1) * g.rs));
            hipLaunchKernelGGL((k_start_emit<T>), dim3(grid(nodes)), dim3(TPB), 0, 0, g, start_base.p, q_rec.p);
            AMX_TRY(hipGetLastError());
            AMX_TRY(hipDeviceSynchronize());
            pending = total;
        }
        if (pending) gen_sizes.push_back(pending);
        return MXS_OK;
    }

    int step() {  // deliver the whole pending generation
        const int64_t n = pending;
        if (n > (int64_t)1 << 30) return fail(MXS_E_NOMEM, "amaxsum: more than 2^30 messages in one generation");
        AMX_TRY(dest.reserve(n));
        AMX_TRY(dest_sorted.reserve(n));
        AMX_TRY(order.reserve(n));
        AMX_TRY(order_in.reserve(n));
        AMX_TRY(cap.reserve(n));
        AMX_TRY(slot_base.reserve(n));
        hipLaunchKernelGGL((k_dest<T>), dim3(grid(n)), dim3(TPB), 0, 0, g, (const T*)q_rec.p, n, dest.p, cap.p);
        AMX_TRY(hipGetLastError());
        int64_t n_slots = 0;
        { int rc = scan32(cap.p, slot_base.p, n, &n_slots); if (rc) return rc; }
        if (n_slots > (int64_t)INT32_MAX)  // the compaction scans the slots with 32-bit counts
            return fail(MXS_E_NOMEM, "amaxsum: more than 2^31 output slots in one generation");
        hipLaunchKernelGGL((k_stamp<T>), dim3(grid(n)), dim3(TPB), 0, 0, q_rec.p, (const int64_t*)slot_base.p, n, g.rs);
        AMX_TRY(hipGetLastError());
        {   // FIFO indices 0..n-1, then the stable sort by destination
            hipLaunchKernelGGL(k_iota, dim3(grid(n)), dim3(TPB), 0, 0, order_in.p, n);
            AMX_TRY(hipGetLastError());
            size_t bytes = 0;
            int bits = 1;
            while (((int64_t)1 << bits) < (int64_t)g.n_vars + g.n_factors + 1) ++bits;
            AMX_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dest.p, dest_sorted.p, order_in.p, order.p, (int)n, 0, bits));
            AMX_TRY(temp.reserve(bytes));
            AMX_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, bytes, dest.p, dest_sorted.p, order_in.p, order.p, (int)n, 0, bits));
        }
        // One thread per DESTINATION that has mail (not per message), the destinations with the
        // longest queues first: a handler chain is sequential, so the generation lasts as long as
        // its longest chain or as the total work over the machine, whichever is larger -- full
        // waves of chains of similar length instead of one or two busy lanes per wave.
        int64_t n_seg = 0;
        {
            AMX_TRY(head.reserve(n));
            AMX_TRY(head_idx.reserve(n));
            hipLaunchKernelGGL(k_heads, dim3(grid(n)), dim3(TPB), 0, 0, dest_sorted.p, n, head.p);
            AMX_TRY(hipGetLastError());
            { int rc = scan32(head.p, head_idx.p, n, &n_seg); if (rc) return rc; }
            AMX_TRY(seg_pos.reserve(n_seg));
            AMX_TRY(seg_first.reserve(n_seg));
            AMX_TRY(seg_key.reserve(n_seg));
            AMX_TRY(seg_key_sorted.reserve(n_seg));
            hipLaunchKernelGGL(k_seg_starts, dim3(grid(n)), dim3(TPB), 0, 0, head.p, head_idx.p, n, seg_pos.p);
            AMX_TRY(hipGetLastError());
            const char* env = std::getenv("MAXSUM_AMAXSUM_GENERIC");  // =1: the per-message handler only (A/B, tests)
            hipLaunchKernelGGL((k_seg_keys<T>), dim3(grid(n_seg)), dim3(TPB), 0, 0, g, dest_sorted.p, seg_pos.p, n_seg, n,
                               (env && env[0] == '1') ? 1 : 0, seg_key.p);
            AMX_TRY(hipGetLastError());
            size_t bytes = 0;
            AMX_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, seg_key.p, seg_key_sorted.p, seg_pos.p, seg_first.p, (int)n_seg, 0, 35));
            AMX_TRY(temp.reserve(bytes));
            AMX_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, bytes, seg_key.p, seg_key_sorted.p, seg_pos.p, seg_first.p, (int)n_seg, 0, 35));
            AMX_TRY(cls_first.reserve(N_CLS + 1));
            hipLaunchKernelGGL(k_class_bounds, dim3(1), dim3(64), 0, 0, seg_key_sorted.p, n_seg, cls_first.p);
            AMX_TRY(hipGetLastError());
        }
        // the records themselves into sorted order: the chains read their queues as contiguous runs
        AMX_TRY(m_rec.reserve(n * g.rs));
        {
            const int64_t pieces = n * (g.rs * (int64_t)sizeof(T) / 16);
            hipLaunchKernelGGL((k_permute<T>), dim3(grid(pieces)), dim3(TPB), 0, 0, (const T*)q_rec.p, (const int32_t*)order.p, n, g.rs, m_rec.p);
            AMX_TRY(hipGetLastError());
        }
        const Sorted<T> sq{dest_sorted.p, m_rec.p, seg_first.p, seg_key_sorted.p, n};
        int64_t h_first[N_CLS + 1];
        AMX_TRY(read_back(h_first, cls_first.p, sizeof(h_first)));
        AMX_TRY(s_rec.reserve((n_slots + 1) * g.rs));
        AMX_TRY(hipMemset(s_rec.p, 0xFF, sizeof(T) * (n_slots + 1) * g.rs));  // code -1: empty slot
        for (int cls = 0; cls < N_CLS; ++cls) {
            const int64_t b = h_first[cls], e = h_first[cls + 1];
            if (e <= b) continue;
            const int per_wave = cls == CLS_VAR8 ? 8 : (cls == CLS_VAR16 ? 4 : (cls == CLS_VAR64 ? 1 : 64));
            const dim3 gr((unsigned)((e - b + per_wave - 1) / per_wave)), bl(64);
#define AMX_ARGS g, sq, b, e, s_rec.p
            if (!cls_stream[cls]) AMX_TRY(hipStreamCreateWithFlags(&cls_stream[cls], 0));
            hipStream_t st = cls_stream[cls];
            if (cls == CLS_VAR8) hipLaunchKernelGGL((k_process_vars<T, 8>), gr, bl, 0, st, AMX_ARGS);
            else if (cls == CLS_VAR16) hipLaunchKernelGGL((k_process_vars<T, 16>), gr, bl, 0, st, AMX_ARGS);
            else if (cls == CLS_VAR64) hipLaunchKernelGGL((k_process_vars<T, 64>), gr, bl, 0, st, AMX_ARGS);
            else if (cls == CLS_FACTOR2) hipLaunchKernelGGL((k_process_lanes<T, true>), gr, bl, 0, st, AMX_ARGS);
            else hipLaunchKernelGGL((k_process_lanes<T, false>), gr, bl, 0, st, AMX_ARGS);
#undef AMX_ARGS
            AMX_TRY(hipGetLastError());
        }
        // compaction of the filled slots, slot order = FIFO order of the next generation
        int64_t n_next = 0;
        if (n_slots > 0) {
            AMX_TRY(pos.reserve(n_slots));
            size_t bytes = 0;
            hipcub::CountingInputIterator<int64_t> slots(0);
            hipcub::TransformInputIterator<int64_t, SlotFilled<T>, hipcub::CountingInputIterator<int64_t>> filled(
                slots, SlotFilled<T>{s_rec.p, g.rs});
            AMX_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, filled, pos.p, (int)n_slots));
            AMX_TRY(temp.reserve(bytes));
            AMX_TRY(hipcub::DeviceScan::ExclusiveSum(temp.p, bytes, filled, pos.p, (int)n_slots));
            int64_t last_pos = 0;
            int32_t last_code = -1;
            AMX_TRY(read_back(&last_pos, pos.p + n_slots - 1, 8));
            AMX_TRY(hipMemcpy(&last_code, s_rec.p + (n_slots - 1) * g.rs, 4, hipMemcpyDeviceToHost));
            const int64_t last_flag = last_code >= 0 ? 1 : 0;
            n_next = last_pos + last_flag;
            AMX_TRY(q_rec2.reserve((n_next + 1) * g.rs));
            hipLaunchKernelGGL((k_gather<T>), dim3(grid(n_slots)), dim3(TPB), 0, 0, (const T*)s_rec.p, (const int64_t*)pos.p, n_slots,
                               g.rs, q_rec2.p);
            AMX_TRY(hipGetLastError());
        }
        AMX_TRY(hipDeviceSynchronize());
        std::swap(q_rec.p, q_rec2.p);
        std::swap(q_rec.n, q_rec2.n);
        delivered_total += n;
        next_generation += 1;
        pending = n_next;
        if (n_next) gen_sizes.push_back(n_next);
        return MXS_OK;
    }

    int run(int32_t max_generations, int64_t* delivered) override {
        AMX_TRY(hipSetDevice(device));
        int64_t done = 0;
        while (pending > 0 && (max_generations < 0 || next_generation < max_generations)) {
            const int64_t n = pending;
            int rc = step();
            if (rc) return rc;
            done += n;
        }
        if (delivered) *delivered = done;
        return MXS_OK;
    }

    int get_assignment(int32_t* idx, double* bel) override {
        AMX_TRY(hipSetDevice(device));
        const int nV = g.n_vars;
        std::vector<int32_t> hs(nV);
        std::vector<T> hb(nV);
        if (nV) {
            AMX_TRY(hipMemcpy(hs.data(), sel.p, 4 * nV, hipMemcpyDeviceToHost));
            AMX_TRY(hipMemcpy(hb.data(), belief.p, sizeof(T) * nV, hipMemcpyDeviceToHost));
        }
        for (int v = 0; v < nV; ++v) {
            if (idx) idx[v] = hs[v];
            if (bel) bel[v] = (double)hb[v];
        }
        return MXS_OK;
    }

    int get_messages(double* fc, double* vc, double* fp, double* vp, uint8_t* fh, uint8_t* vh, uint8_t* fn,
                     uint8_t* vn) override {
        AMX_TRY(hipSetDevice(device));
        std::vector<T> h((size_t)nm);
        auto pull = [&](const T* src, double* dst) -> hipError_t {
            if (!dst || !nm) return hipSuccess;
            hipError_t e = hipMemcpy(h.data(), src, sizeof(T) * nm, hipMemcpyDeviceToHost);
            for (int64_t i = 0; i < nm; ++i) dst[i] = (double)h[i];
            return e;
        };
        AMX_TRY(pull(f_cost.p, fc));
        AMX_TRY(pull(v_cost.p, vc));
        AMX_TRY(pull(f_prev.p, fp));
        AMX_TRY(pull(v_prev.p, vp));
        const int nE = g.n_edges;
        if (nE) {
            if (fh) AMX_TRY(hipMemcpy(fh, f_has.p, nE, hipMemcpyDeviceToHost));
            if (vh) AMX_TRY(hipMemcpy(vh, v_has.p, nE, hipMemcpyDeviceToHost));
            if (fn) AMX_TRY(hipMemcpy(fn, f_cnt.p, nE, hipMemcpyDeviceToHost));
            if (vn) AMX_TRY(hipMemcpy(vn, v_cnt.p, nE, hipMemcpyDeviceToHost));
        }
        return MXS_OK;
    }

    // change_factor_function, same scope (maxsum_dynamic.py:80-104): the factor's table replaced in place
    int update_table(int32_t factor, const double* table, int64_t n) override {
        const int nF = (int)h_toff.size() - 1;
        if (factor < 0 || factor >= nF) return fail(MXS_E_INVALID, "factor out of range");
        const int64_t lo = h_toff[factor], hi = h_toff[factor + 1];
        if (!table || n != hi - lo) return fail(MXS_E_INVALID, "table size differs from the factor's");
        std::vector<T> tt((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            h_tables[lo + i] = table[i];
            tt[i] = (T)table[i];
        }
        AMX_TRY(hipSetDevice(device));
        AMX_TRY(hipDeviceSynchronize());  // no delivery is in flight between two run() calls; be sure
        AMX_TRY(hipMemcpy(tables.p + lo, tt.data(), sizeof(T) * (size_t)n, hipMemcpyHostToDevice));
        return MXS_OK;
    }

    // DCOP.solution_cost (dcop.py:308-367) of the selection: reporting, evaluated on the host
    int eval_cost(const int32_t* idx, double infinity, double* cost, int64_t* viol) override {
        std::vector<int32_t> cur;
        if (!idx) {
            cur.resize(g.n_vars);
            int rc = get_assignment(cur.data(), nullptr);
            if (rc) return rc;
            idx = cur.data();
        }
        double soft = 0;
        int64_t hard = 0;
        for (int f = 0; f < g.n_factors; ++f) {
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
            const double c = h_eval_cost[h_coff[v] + idx[v]];
            if (c != infinity) soft += c; else hard += 1;
        }
        if (cost) *cost = soft;
        if (viol) *viol = hard;
        return MXS_OK;
    }
};

}  // namespace amx

struct mxs_amaxsum {
    amx::Base* impl;
};

extern "C" {

int mxs_amaxsum_create(const mxs_graph* g, const mxs_params* p, int32_t device, mxs_amaxsum** out) {
    if (!g || !p || !out) return amx::fail(MXS_E_INVALID, "null argument");
    *out = nullptr;
    if (p->mode != MXS_MODE_MIN && p->mode != MXS_MODE_MAX) return amx::fail(MXS_E_INVALID, "invalid mode");
    if (p->damping_nodes < 0 || p->damping_nodes > 3) return amx::fail(MXS_E_INVALID, "invalid damping_nodes");
    if (p->start_messages < 0 || p->start_messages > 2) return amx::fail(MXS_E_INVALID, "invalid start_messages");
    try {
        amx::Base* impl = p->dtype == MXS_DTYPE_F32 ? (amx::Base*)new amx::Engine<float>() : (amx::Base*)new amx::Engine<double>();
        int rc = impl->init(*g, *p, device);
        if (rc) {
            delete impl;
            return rc;
        }
        *out = new mxs_amaxsum{impl};
        return MXS_OK;
    } catch (const std::exception& ex) {
        return amx::fail(MXS_E_NOMEM, ex.what());
    }
}
int mxs_amaxsum_reset(mxs_amaxsum* e) { return e ? e->impl->reset() : amx::fail(MXS_E_INVALID, "null handle"); }
int mxs_amaxsum_update_factor_table(mxs_amaxsum* e, int32_t factor, const double* table, int64_t n_entries) {
    if (!e) return amx::fail(MXS_E_INVALID, "null handle");
    try {
        return e->impl->update_table(factor, table, n_entries);
    } catch (const std::exception& ex) {
        return amx::fail(MXS_E_NOMEM, ex.what());
    }
}
int mxs_amaxsum_run(mxs_amaxsum* e, int32_t max_generations, int64_t* delivered) {
    if (!e) return amx::fail(MXS_E_INVALID, "null handle");
    try {
        return e->impl->run(max_generations, delivered);
    } catch (const std::exception& ex) {
        return amx::fail(MXS_E_NOMEM, ex.what());
    }
}
int mxs_amaxsum_status(const mxs_amaxsum* e, int32_t* next_generation, int64_t* pending, int64_t* delivered) {
    if (!e) return amx::fail(MXS_E_INVALID, "null handle");
    if (next_generation) *next_generation = e->impl->next_generation;
    if (pending) *pending = e->impl->pending;
    if (delivered) *delivered = e->impl->delivered_total;
    return MXS_OK;
}
int mxs_amaxsum_generation_sizes(const mxs_amaxsum* e, int64_t* out, int32_t cap, int32_t* n) {
    if (!e) return amx::fail(MXS_E_INVALID, "null handle");
    const std::vector<int64_t>& s = e->impl->gen_sizes;
    for (size_t i = 0; i < s.size() && (int32_t)i < cap; ++i) out[i] = s[i];
    if (n) *n = (int32_t)s.size();
    return MXS_OK;
}
int mxs_amaxsum_get_assignment(mxs_amaxsum* e, int32_t* idx, double* belief) {
    return e ? e->impl->get_assignment(idx, belief) : amx::fail(MXS_E_INVALID, "null handle");
}
int mxs_amaxsum_get_messages(mxs_amaxsum* e, double* f_cost, double* v_cost, double* f_prev, double* v_prev,
                             uint8_t* f_has, uint8_t* v_has, uint8_t* f_cnt, uint8_t* v_cnt) {
    return e ? e->impl->get_messages(f_cost, v_cost, f_prev, v_prev, f_has, v_has, f_cnt, v_cnt)
             : amx::fail(MXS_E_INVALID, "null handle");
}
int mxs_amaxsum_eval_cost(mxs_amaxsum* e, const int32_t* idx, double infinity, double* cost, int64_t* violations) {
    return e ? e->impl->eval_cost(idx, infinity, cost, violations) : amx::fail(MXS_E_INVALID, "null handle");
}
int mxs_amaxsum_destroy(mxs_amaxsum* e) {
    if (e) {
        delete e->impl;
        delete e;
    }
    return MXS_OK;
}

}  // extern "C"

