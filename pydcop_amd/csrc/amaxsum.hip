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
//   k_process   (the previous step) every handler writes what it sends as one record into its own OUTPUT SLOTS --
//               numbered in FIFO order of the deliveries, one per message the handler can send -- and, beside
//               it, one 4-byte word per slot into a dense array: destination, the slots the message's own handler
//               will need; 0 = nothing sent.  Slot order = the FIFO order of the next generation.
//   scan        ONE exclusive scan over the slot words: messages in earlier slots (the FIFO index) and output slots
//               their handlers take (the message's first output slot), packed in one 64-bit sum; the totals are
//               the one thing the host waits for per generation
//   k_compact   (destination, slot) of every message, in slot order
//   sort        stable radix sort of those pairs by destination (hipCUB): a queue = a run, in FIFO order
//   k_permute   the records gathered from their slots into destination order (one random 32-byte read per
//               message) and stamped with their first output slot: a delivery is one sequential read
//   k_process   a lane / lane group per destination, a kernel and a stream per destination class (they run side
//               by side): handles ITS messages one after the other in FIFO order, exactly like the reference's
//               handler (same expressions, same order of additions -- select_value in first-arrival order of
//               the factors, maxsum.py:609).  A step of a chain is straight-line code: direction, "heard from",
//               "sent" are selects (round 4: the branchy version executed both directions for every message).
//
// Round 3/4 compacted the sent messages into a queue (gather of every slot's record), computed destination and
// capacity per message, scanned, stamped, sorted, and sorted the destinations by queue length every generation:
// half of a generation was bookkeeping over 32-byte records.  Now the bookkeeping reads 4 bytes per slot, the only
// pass over the records is k_permute's gather, and the destinations run in a static order (re-ordered by actual
// queue length only in generations of a million messages and more).
//
// A message is ONE record (8-byte header + payload, 32 bytes for three f64 values): the generations are bound by
// the number of random cache lines they touch (~32 G lines/s measured) -- per message one in k_permute, one for the
// record and one for the slot word in k_process.
//
// Nothing here is a dense contraction: integer bookkeeping + a few adds per message element.
// The run ends by itself when the send rule (approx_match + SAME_COUNT) has silenced every edge.
//
// Built into libmaxsum_hip.so by hipcc.  (The host emulation of the CPU tests compiles this very
// file against serial stand-ins for the two hipCUB primitives, tests/emu/hipcub/.)
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <algorithm>
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
    int32_t cap_bits;         // slot words: low bits holding the handler's output slot count (0: looked up in node_cap)
    const int32_t* node_cap;  // [computation] output slots the handler of a message to it needs (its other neighbours)
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

// approx_match's component test without a branch: prev == c, or (prev + c != 0 and 2 |prev - c| / |prev + c| < stability).
// The same expressions as comp_match in the same order; the quotient is computed whether it is needed or not.
template <typename T>
__device__ __forceinline__ bool comp_match_flat(T c, T prev_c, T stability) {
    const T delta = absT(prev_c - c), sum = prev_c + c;
    const bool close = ((T)2 * delta / absT(sum)) < stability;
    return !(prev_c != c) | ((sum != (T)0) & close);
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

// Beside its record, every produced message leaves one word in the dense array s_hdr[slot]: (its destination
// computation + 1) << cap_bits | the output slots its handler needs (the destination's other neighbours) -- 0 = the
// slot holds no message.  The bookkeeping between two generations (compaction, sort keys, the next generation's slot
// numbers) reads these 4 bytes per slot and never the 32-byte records.  (cap_bits = 0 when destination and count do
// not fit 31 bits together: the count is then looked up in node_cap by the destination.)
__host__ __device__ __forceinline__ int32_t slot_dest(int32_t h, int cap_bits) { return (int32_t)((uint32_t)h >> cap_bits) - 1; }
__host__ __device__ __forceinline__ int64_t slot_cap(int32_t h, int cap_bits, const int32_t* node_cap) {
    return cap_bits ? (int64_t)(h & ((1 << cap_bits) - 1)) : (int64_t)node_cap[h - 1];
}
template <typename T>
__device__ __forceinline__ int32_t slot_word(const Dev<T>& g, int dest) {
    return ((dest + 1) << g.cap_bits) | (g.cap_bits ? g.node_cap[dest] : 0);
}
template <typename T>
__device__ __forceinline__ int32_t hdr_to_factor(const Dev<T>& g, int e) {  // a variable's message on edge e
    return slot_word(g, g.n_vars + g.edge_factor[e]);
}
template <typename T>
__device__ __forceinline__ int32_t hdr_to_var(const Dev<T>& g, int e) {  // a factor's message on edge e
    return slot_word(g, g.edge_var[e]);
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
__global__ void k_start_emit(Dev<T> g, const int32_t* base, T* s_rec, int32_t* s_hdr) {
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
            T* r = s_rec + at * g.rs;  // (the slots were zero-filled: header base, payload padding)
            costs_for_factor(g, v, k0 + k, rec_pay(r));
            rec_set_head(r, g.var_edges[k0 + k] * 2, 0);
            s_hdr[at] = hdr_to_factor(g, g.var_edges[k0 + k]);
        }
    } else if (i < g.n_vars + g.n_factors) {
        const int f = i - g.n_vars;
        const int e0 = g.factor_rowptr[f], ar = g.factor_rowptr[f + 1] - e0;
        const bool sends = (ar == 1 && g.start_mode != MXS_START_ALL) || g.start_mode == MXS_START_ALL;
        if (!sends) return;
        for (int p = 0; p < ar; ++p) {
            const int64_t at = (int64_t)base[i] + p;
            T* r = s_rec + at * g.rs;
            factor_message_any(g, f, p, rec_pay(r));
            rec_set_head(r, (e0 + p) * 2 + 1, 0);
            s_hdr[at] = hdr_to_var(g, e0 + p);
        }
    }
}

// ---- one generation --------------------------------------------------------------------------
template <typename T>
__device__ void handle(const Dev<T>& g, const T* rec, T* s_rec, int32_t* s_hdr, bool last) {
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
        if (g.f_nhas[f] != ar) {  // still waiting for some variable (:206): nothing in its output slots
            for (int k = 0; k < ar - 1; ++k) s_hdr[(int64_t)rec_base(rec) + k] = 0;
            return;
        }
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
                s_hdr[(int64_t)rec_base(rec) + slot] = hdr_to_var(g, e2);
            } else {
                s_hdr[(int64_t)rec_base(rec) + slot] = 0;
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
                s_hdr[(int64_t)rec_base(rec) + slot] = hdr_to_factor(g, e2);
            } else {
                s_hdr[(int64_t)rec_base(rec) + slot] = 0;
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

// The messages of a generation in DESTINATION-SORTED order (k_permute: one pass gathers the records the stable
// sort's slot array points at and stamps each with its handler's first output slot): a destination's queue is then a
// contiguous run -- code, first output slot, payload at consecutive addresses -- and the chains below read it
// sequentially, RING deliveries ahead of the one they handle.  (Round 3 followed an index array into three arrays per
// delivery, one ahead: three dependent random loads under full load, 3-5 us per step of a chain that is sequential
// anyway; the longest queue of a generation is what the generation lasts -- profiles/r04_amaxsum_dispatches_v1.txt.)
template <typename T>
struct Sorted {
    const T* rec;              // [n * rs] the records (header: code, first output slot of the delivery's handler)
    const int32_t* seg_node;   // [nodes] the t-th destination to run (variables first, then n_vars + factor): static,
                               // by class, inside a class by expected queue length (Engine::init)
    const int32_t* run_first;  // [nodes] a destination's queue: the records [run_first, run_last) (equal: no mail)
    const int32_t* run_last;
    const int64_t* cls_first;  // [N_CLS + 1] first t of every class
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
// lane K of every quad to the four lanes of the quad (DPP quad_perm: [K, K, K, K])
template <int K>
__device__ __forceinline__ double quad_bcast(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, K * 0x55, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, K * 0x55, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ float quad_bcast(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), K * 0x55, 0xf, 0xf, false));
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
// wave walk their own queues in lock step (a group whose queue is done idles: the destinations run in
// order of (expected) queue length, so the queues of a wave are about equally long); `t` = the group's
// destination in seg_node, groups past `seg_end` have none.
template <typename T, int D, int GROUP>
__device__ void chain_variable(const Dev<T>& g, const Sorted<T>& sq, int64_t t, int64_t seg_end, T* s_rec, int32_t* s_hdr) {
    constexpr bool WHOLE = GROUP == 64;
    const int lane = (int)threadIdx.x & 63;
    const int gl = lane % GROUP, gbase = lane - gl;
    const unsigned long long gmask = WHOLE ? ~0ull : (((1ull << (GROUP % 64)) - 1ull) << gbase);
    const bool valid = t < seg_end;
    const int v = valid ? sq.seg_node[t] : 0;
    const int64_t p = valid ? sq.run_first[v] : 0;
    const int64_t len = valid ? sq.run_last[v] - p : 0;
    const int k0 = g.var_rowptr[v], deg = valid ? g.var_rowptr[v + 1] - k0 : 0;
    const bool active = gl < deg;
    const int ek = active ? g.var_edges[k0 + gl] : -1;
    const int64_t mo = active ? g.msg_off[ek] : 0;
    const int32_t my_hdr = active ? hdr_to_factor(g, ek) : 0;  // what a message to this lane's factor leaves in s_hdr
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
    // A step is straight-line code (like chain_factor2's): "is it my factor's message", "has that factor been heard
    // from", "is it sent" are selects.  A factor that does not count contributes -0.0: x + (-0.0) == x for every x,
    // bit for bit, so the sums are the reference's sums over the factors that do count, in its order.
    const bool damp_on = g.damp_v != 0;
    const int64_t last = len > 0 ? len - 1 : 0;
    auto step = [&](const Mail<T, D>& ml, bool alive) __attribute__((always_inline)) {
            const int e = ml.code >> 1;
            const unsigned long long from = __ballot(alive & active & (ek == e)) & gmask;
            const int j = from ? __builtin_ctzll(from) - gbase : -1;  // the sender's lane of the group
            const bool mine = alive & (gl == j);
#pragma unroll
            for (int d = 0; d < D; ++d) held[d] = mine ? ml.pay[d] : held[d];
            const bool fresh = mine & !has;
            const bool is_new = (__ballot(fresh) & gmask) != 0ull;
            if (fresh) g.v_order[k0 + narr] = e;
            my_rank = fresh ? narr : my_rank;
            has |= mine;
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
                        const T xx = ((k2 != gl) & (((hasmask >> k2) & 1ull) != 0ull)) ? x : -(T)0;
                        sum_cost += xx;
                        md += xx;
                    }
                } else {  // the group is a DPP row (16 lanes), half of one (8) or a quad
                    amx_static_for<GROUP>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int k2 = decltype(kc)::value;
                        T x;
                        if constexpr (GROUP == 16) {
                            x = row_bcast<k2>(held[d]);
                        } else if constexpr (GROUP == 4) {
                            x = quad_bcast<k2>(held[d]);
                        } else {
                            const T lo = row_bcast<k2>(held[d]), hi = row_bcast<k2 + 8>(held[d]);
                            x = (lane & 8) ? hi : lo;
                        }
                        const T xx = ((k2 != gl) & (((hasmask >> k2) & 1ull) != 0ull)) ? x : -(T)0;  // lanes past the degree never "have"
                        sum_cost += xx;
                        md += xx;
                    });
                }
                m[d] = md;
            }
            const T avg = sum_cost / (T)D;
#pragma unroll
            for (int d = 0; d < D; ++d) m[d] = m[d] - avg;
            // apply_damping + the send rule (damp_and_decide_reg, without branches)
            const bool emit = alive & active & (gl != j);
            const bool c0 = cnt > 0;
            bool match = c0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                m[d] = (c0 & damp_on) ? g.damping * prev[d] + ((T)1 - g.damping) * m[d] : m[d];
                match &= comp_match_flat(m[d], prev[d], g.stability);
            }
            const bool sent = emit & !(match & (cnt >= SAME_COUNT));
            cnt = sent ? (uint8_t)(match ? cnt + 1 : 1) : cnt;
#pragma unroll
            for (int d = 0; d < D; ++d) prev[d] = sent ? m[d] : prev[d];
            // this lane's output slot of the delivery: the slot word always (0: not sent -- the slots are not cleared
            // between generations, every handler writes all of its own), the record when there is one
            const int64_t slot = (int64_t)ml.base + (gl < j ? gl : gl - 1);
            if (emit) s_hdr[slot] = sent ? my_hdr : 0;
            if (sent) {
                T* o = (T*)__builtin_assume_aligned(s_rec + slot * g.rs, 16);
                rec_set_head(o, ek * 2, 0);
#pragma unroll
                for (int d = 0; d < D; ++d) rec_pay(o)[d] = m[d];
            }
    };
    for (int64_t r0 = 0; __ballot(r0 < len) != 0ull; r0 += VRING) {
#pragma unroll
        for (int j = 0; j < VRING; ++j) {
            const int64_t r = r0 + j;
            step(ring[j], r < len);
            const int64_t nx = r + VRING;
            fetch_mail<T, D>(g, sq, p + (nx < last ? nx : last), ring[j]);  // this register set's next tenant (always a load)
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

// A binary factor over domains of at most N <= 4 values: one lane, table / held costs / last-sent messages in
// registers.  A step (one delivered message) is STRAIGHT-LINE code: the direction (from scope variable 0 or 1), "is
// the other variable heard from", "is it sent" are data -- selects -- not control flow.  (Round 3/4 branched on each of
// them per lane: 909 basic blocks, both directions executed by every wave for every message, 3.8 us per step of the
// longest queue -- and a generation lasts as long as its longest queue, profiles/r04_amaxsum_dispatches_v4.txt.)
//   out[j] = opt over the sender's values k of  t(k, j) + (0 + in[k])            factor_costs_for_var, maxsum.py:382-447
// with in[k] = the received cost, or the optimum's identity (+-inf) past the sender's domain: such a k never wins.
template <typename T, int N, bool MAX>
__device__ void chain_factor2(const Dev<T>& g, int f, const Sorted<T>& sq, int64_t p, int64_t len, T* s_rec, int32_t* s_hdr) {
    const int eA = g.factor_rowptr[f], eB = eA + 1;
    const int32_t hdrA = hdr_to_var(g, eA), hdrB = hdr_to_var(g, eB);
    const int DA = g.dom_size[g.edge_var[eA]], DB = g.dom_size[g.edge_var[eB]];
    const int64_t moA = g.msg_off[eA], moB = g.msg_off[eB];
    const bool damp_on = g.damp_f != 0;
    const T padv = MAX ? -(T)INFINITY : (T)INFINITY;
    T tab[N * N], cA[N], cB[N], pA[N], pB[N];
#pragma unroll
    for (int x = 0; x < N; ++x) {
#pragma unroll
        for (int y = 0; y < N; ++y)
            tab[x * N + y] = g.tables[g.table_off[f] + (int64_t)(x < DA ? x : DA - 1) * DB + (y < DB ? y : DB - 1)];
        cA[x] = g.f_cost[moA + (x < DA ? x : DA - 1)];
        pA[x] = g.f_prev[moA + (x < DA ? x : DA - 1)];
        cB[x] = g.f_cost[moB + (x < DB ? x : DB - 1)];
        pB[x] = g.f_prev[moB + (x < DB ? x : DB - 1)];
    }
    bool hasA = g.f_has[eA] != 0, hasB = g.f_has[eB] != 0;
    int cntA = g.f_cnt[eA], cntB = g.f_cnt[eB];
    Mail<T, N> ring[RING];
    const int64_t last = len > 0 ? len - 1 : 0;
#pragma unroll
    for (int j = 0; j < RING; ++j) fetch_mail<T, N>(g, sq, p + (j < last ? j : last), ring[j]);
    auto step = [&](const Mail<T, N>& ml, bool live) __attribute__((always_inline)) {
        const bool fromA = (ml.code >> 1) == eA;
        const int Ds = fromA ? DA : DB, Dt = fromA ? DB : DA;
        const bool ready = live & (fromA ? hasB : hasA);  // else: still waiting for the other variable (amaxsum.py:206)
        const int cnt = fromA ? cntB : cntA;              // of the message to the OTHER variable
        T in[N], pt[N], out[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            in[k] = k < Ds ? ml.pay[k] : padv;
            pt[k] = fromA ? pB[k] : pA[k];
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            T best = padv;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const T cur = (fromA ? tab[k * N + j] : tab[j * N + k]) + ((T)0 + in[k]);
                best = (MAX ? best < cur : best > cur) ? cur : best;
            }
            out[j] = best;
        }
        // apply_damping + the send rule (damp_and_decide_reg, without branches)
        const bool c0 = cnt > 0;
        bool match = c0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            out[j] = (c0 & damp_on) ? g.damping * pt[j] + ((T)1 - g.damping) * out[j] : out[j];
            match &= (j >= Dt) | comp_match_flat(out[j], pt[j], g.stability);
        }
        const bool sent = ready & !(match & (cnt >= SAME_COUNT));
        const int cnt2 = sent ? (match ? cnt + 1 : 1) : cnt;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            cA[k] = (live & fromA) ? ml.pay[k] : cA[k];
            cB[k] = (live & !fromA) ? ml.pay[k] : cB[k];
            pB[k] = (sent & fromA) ? out[k] : pB[k];
            pA[k] = (sent & !fromA) ? out[k] : pA[k];
        }
        hasA |= live & fromA;
        hasB |= live & !fromA;
        cntB = fromA ? cnt2 : cntB;
        cntA = fromA ? cntA : cnt2;
        // the delivery's one output slot: the slot word always (0: not sent), the record when there is one (past the
        // target's domain: never read)
        if (live) s_hdr[ml.base] = sent ? (fromA ? hdrB : hdrA) : 0;
        if (sent) {
            T* o = (T*)__builtin_assume_aligned(s_rec + (int64_t)ml.base * g.rs, 16);
            rec_set_head(o, (fromA ? eB : eA) * 2 + 1, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) rec_pay(o)[j] = out[j];
        }
    };
    for (int64_t r0 = 0; r0 < len; r0 += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            const int64_t r = r0 + j;
            step(ring[j], r < len);
            const int64_t nx = r + RING;
            fetch_mail<T, N>(g, sq, p + (nx < last ? nx : last), ring[j]);  // this register set's next tenant (always a load)
        }
    }
#pragma unroll
    for (int x = 0; x < N; ++x) {
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
    g.f_cnt[eA] = (uint8_t)cntA;
    g.f_cnt[eB] = (uint8_t)cntB;
    g.f_nhas[f] = (hasA ? 1 : 0) + (hasB ? 1 : 0);
}

// Destination classes of a generation (what runs its queue):
constexpr int CLS_FACTOR2 = 0;  // binary factor, domains <= 4: a lane (chain_factor2), 64 per wave
constexpr int CLS_VAR8 = 1;     // variable, domain 2..4, degree <= 8: 8 lanes, 8 per wave
constexpr int CLS_VAR16 = 2;    //                        degree <= 16: 16 lanes, 4 per wave
constexpr int CLS_VAR64 = 3;    //                        degree <= 64: the wave
constexpr int CLS_GENERIC = 4;  // everything else: a lane on the per-message handler, 64 per wave
constexpr int CLS_VAR4 = 5;     // variable, domain 2..4, degree <= 4: a quad, 16 per wave (the variable kernels are
                                // bound by instruction issue -- 530 per step of 8 deliveries in the 8-lane groups,
                                // profiles/r04_amaxsum_chain_pmc_v1.txt -- and half the variables of a degree-4 graph fit a quad)
constexpr int N_CLS = 6;        // (N_CLS itself: a destination without mail in this generation)
constexpr int64_t DYNAMIC_ORDER_FROM = 1 << 20;  // messages in a generation from which its destinations are re-ordered

template <typename T>
__device__ __forceinline__ int class_of(const Dev<T>& g, int32_t dst) {
    if (dst < g.n_vars) {
        const int D = g.dom_size[dst], deg = g.var_rowptr[dst + 1] - g.var_rowptr[dst];
        if (D < 2 || D > 4 || deg > 64) return CLS_GENERIC;
        return deg <= 4 ? CLS_VAR4 : (deg <= 8 ? CLS_VAR8 : (deg <= 16 ? CLS_VAR16 : CLS_VAR64));
    }
    const int f = dst - g.n_vars, e0 = g.factor_rowptr[f];
    return (g.factor_rowptr[f + 1] - e0 == 2 && g.dom_size[g.edge_var[e0]] <= 4 && g.dom_size[g.edge_var[e0 + 1]] <= 4)
               ? CLS_FACTOR2 : CLS_GENERIC;
}

// A generation lasts as long as its longest queue, and the destinations run longest (expected) queue first: the first
// waves of a class kernel are its critical path.  They get the issue priority over the waves they share a SIMD with.
__device__ __forceinline__ void chain_priority() {
#ifndef AMX_NO_PRIORITY
    if (blockIdx.x < 64) __builtin_amdgcn_s_setprio(3);
    else if (blockIdx.x < 512) __builtin_amdgcn_s_setprio(2);
    else if (blockIdx.x < 4096) __builtin_amdgcn_s_setprio(1);
#endif
}

template <typename T, int GROUP>
__device__ __forceinline__ void variables_of_wave(const Dev<T>& g, const Sorted<T>& sq, int cls, T* s_rec, int32_t* s_hdr) {
    constexpr int PER_WAVE = 64 / GROUP;
    const int lane = (int)threadIdx.x & 63;
    const int64_t seg_begin = sq.cls_first[cls], seg_end = sq.cls_first[cls + 1];
    if (seg_begin + (int64_t)blockIdx.x * PER_WAVE >= seg_end) return;  // (the grid covers every variable of the class)
    chain_priority();
    const int64_t t = seg_begin + (int64_t)blockIdx.x * PER_WAVE + lane / GROUP;
    // the domain sizes of the wave's variables: one pass per size present (wave-uniform branches)
    int myD = 0;  // (0: no destination for this group, or one without mail in this generation)
    if (t < seg_end) {
        const int v = sq.seg_node[t];
        if (sq.run_last[v] > sq.run_first[v]) myD = g.dom_size[v];
    }
    for (int D = 2; D <= 4; ++D) {
        if (__ballot(myD == D) == 0ull) continue;
        const int64_t tt = myD == D ? t : seg_end;  // the other groups sit this pass out
        if (D == 2) chain_variable<T, 2, GROUP>(g, sq, tt, seg_end, s_rec, s_hdr);
        else if (D == 3) chain_variable<T, 3, GROUP>(g, sq, tt, seg_end, s_rec, s_hdr);
        else chain_variable<T, 4, GROUP>(g, sq, tt, seg_end, s_rec, s_hdr);
    }
}

// sq.seg_node[t]: the t-th destination to run -- by class, longest (expected) queues first (Engine::init, step());
// a launch runs the destinations of one class, [cls_first[cls], cls_first[cls + 1]) -- read from device memory: the
// grid is sized for every computation of the class (static), the blocks past the count leave at once, a destination
// without mail sits the generation out, and the host waits for nothing.  Blocks of one wave; a kernel per class, so
// that each has the registers of its own path only.
template <typename T, int GROUP>
__global__ void __launch_bounds__(64) k_process_vars(Dev<T> g, Sorted<T> sq, int cls, T* s_rec, int32_t* s_hdr) {
    variables_of_wave<T, GROUP>(g, sq, cls, s_rec, s_hdr);
}

template <typename T, bool FACTOR2>  // (two kernels: the binary-factor chains do not pay for the generic handler's registers)
__global__ void __launch_bounds__(64) k_process_lanes(Dev<T> g, Sorted<T> sq, int cls, T* s_rec, int32_t* s_hdr) {
    chain_priority();
    const int64_t t = sq.cls_first[cls] + (int64_t)blockIdx.x * 64 + ((int)threadIdx.x & 63);
    const bool in_class = t < sq.cls_first[cls + 1];
    const int32_t dst = in_class ? sq.seg_node[t] : 0;
    const int64_t p = in_class ? sq.run_first[dst] : 0, len = in_class ? sq.run_last[dst] - p : 0;
    const bool valid = len > 0;  // (a destination without mail in this generation: nothing to do)
    if constexpr (FACTOR2) {
        // the wave's largest domain (wave-uniform): registers and work for 2, 3 or 4 values
        const int f = valid ? dst - g.n_vars : 0;
        int dm = 0;
        if (valid) {
            const int e0 = g.factor_rowptr[f];
            const int d0 = g.dom_size[g.edge_var[e0]], d1 = g.dom_size[g.edge_var[e0 + 1]];
            dm = d0 > d1 ? d0 : d1;
        }
        // (never more than the largest domain of the instance: a record has room for dmax values)
        const int n = __ballot(dm > 3) != 0ull ? 4 : (__ballot(dm > 2) != 0ull ? 3 : (__ballot(dm > 1) != 0ull ? 2 : 1));
        if (!valid) return;
        if (g.is_max) {
            if (n == 4) chain_factor2<T, 4, true>(g, f, sq, p, len, s_rec, s_hdr);
            else if (n == 3) chain_factor2<T, 3, true>(g, f, sq, p, len, s_rec, s_hdr);
            else if (n == 2) chain_factor2<T, 2, true>(g, f, sq, p, len, s_rec, s_hdr);
            else chain_factor2<T, 1, true>(g, f, sq, p, len, s_rec, s_hdr);
        } else {
            if (n == 4) chain_factor2<T, 4, false>(g, f, sq, p, len, s_rec, s_hdr);
            else if (n == 3) chain_factor2<T, 3, false>(g, f, sq, p, len, s_rec, s_hdr);
            else if (n == 2) chain_factor2<T, 2, false>(g, f, sq, p, len, s_rec, s_hdr);
            else chain_factor2<T, 1, false>(g, f, sq, p, len, s_rec, s_hdr);
        }
        return;
    }
    if (!valid) return;
    for (int64_t r = p; r < p + len; ++r)  // its messages, in FIFO order
        handle(g, sq.rec + r * g.rs, s_rec, s_hdr, r + 1 == p + len);
}

// ---- between two generations: 4 bytes per output slot, one random record read per message -------------------
// ONE scan over the slot words gives both numbers every filled slot needs: how many messages sit in earlier slots
// (its FIFO index) and how many output slots their handlers take (its handler's first output slot), packed as
// count << 33 | slots.  (The totals must stay below 2^30 messages and 2^31 slots -- finish() -- so neither field
// overflows when slots * (largest capacity + 1) < 2^33; past that, finish() scans the two separately.)
constexpr int PACK_SHIFT = 33;
constexpr int64_t PACK_MASK = ((int64_t)1 << PACK_SHIFT) - 1;
struct SlotFilled {  // 1 where the slot holds a message
    __host__ __device__ __forceinline__ int64_t operator()(int32_t h) const { return h ? 1 : 0; }
};
struct SlotCap {  // output slots the handler of the slot's message needs (0 for an empty slot)
    const int32_t* node_cap;
    int cap_bits;
    __host__ __device__ __forceinline__ int64_t operator()(int32_t h) const { return h ? slot_cap(h, cap_bits, node_cap) : 0; }
};
struct SlotPacked {
    const int32_t* node_cap;
    int cap_bits;
    __host__ __device__ __forceinline__ int64_t operator()(int32_t h) const {
        return h ? (((int64_t)1 << PACK_SHIFT) | slot_cap(h, cap_bits, node_cap)) : 0;
    }
};
__global__ void k_pack(const int64_t* pos, const int64_t* slot_base, int64_t n_slots, int64_t* packed) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_slots) packed[s] = (pos[s] << PACK_SHIFT) | (slot_base[s] & PACK_MASK);
}

// out[0] = messages of the generation, out[1] = output slots its handlers need (the scan's totals)
__global__ void k_totals(const int32_t* s_hdr, const int64_t* packed, const int32_t* node_cap, int cap_bits, int64_t n_slots,
                         int64_t* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t h = s_hdr[n_slots - 1];
    out[0] = (packed[n_slots - 1] >> PACK_SHIFT) + SlotFilled()(h);
    out[1] = (packed[n_slots - 1] & PACK_MASK) + SlotCap{node_cap, cap_bits}(h);
}
// the same from the two separate scans (before k_pack folds them: the totals are checked first)
__global__ void k_totals2(const int32_t* s_hdr, const int64_t* pos, const int64_t* slot_base, const int32_t* node_cap,
                          int cap_bits, int64_t n_slots, int64_t* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t h = s_hdr[n_slots - 1];
    out[0] = pos[n_slots - 1] + SlotFilled()(h);
    out[1] = slot_base[n_slots - 1] + SlotCap{node_cap, cap_bits}(h);
}

// the sort's input: destination and slot of every message, compacted in slot order = the FIFO order
__global__ void k_compact(const int32_t* s_hdr, const int64_t* packed, int64_t n_slots, int cap_bits, int32_t* dest,
                          int32_t* slot_in) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const int32_t h = s_hdr[s];
    if (!h) return;
    const int64_t at = packed[s] >> PACK_SHIFT;
    dest[at] = slot_dest(h, cap_bits);
    slot_in[at] = (int32_t)s;  // (fewer than 2^31 slots per generation: finish())
}

// The generation's records gathered into destination-sorted order (Sorted): slot[p] = output slot holding the p-th
// message after the stable sort by destination.  One thread per 16-byte piece: the writes are one contiguous stream,
// the reads one random line per message (+ its handler's first output slot, stamped into the header).
template <typename T>
__global__ void k_permute(const T* s_rec, const int32_t* slot, const int64_t* packed, int64_t n, int rs, T* m_rec) {
    const int pieces = rs * (int)sizeof(T) / 16;
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = x / pieces;
    const int k = (int)(x - p * pieces);
    if (p >= n) return;
    struct alignas(16) P16 { uint32_t w[4]; };
    const int64_t s = slot[p];
    P16 piece = ((const P16*)(s_rec + s * rs))[k];
    if (k == 0) piece.w[1] = (uint32_t)(packed[s] & PACK_MASK);
    ((P16*)(m_rec + p * rs))[k] = piece;
}

// run of every destination in the sorted order: first[d] .. last[d] (both 0 for a destination without mail)
__global__ void k_bounds(const int32_t* dest_sorted, int64_t n, int32_t* first, int32_t* last) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t b = dest_sorted[p];
    if (p == 0 || dest_sorted[p - 1] != b) {
        first[b] = (int32_t)p;
        if (p > 0) last[dest_sorted[p - 1]] = (int32_t)p;
    }
    if (p + 1 == n) last[b] = (int32_t)n;
}

// class of every computation (static)
template <typename T>
__global__ void k_node_class(Dev<T> g, int32_t* cls) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < g.n_vars + g.n_factors) cls[i] = class_of(g, i);
}

// A LARGE generation re-orders its destinations by their actual queue lengths (the static order of Engine::init ranks
// them by expected length: good enough while a generation is short, 30 % slower chains at 10 M messages):
// key = class << 13 | (8191 - min(length, 8191)), 16 bits; a destination without mail sorts behind them all.
constexpr int LEN_BITS = 13;
__global__ void k_node_keys(const int32_t* node_cls, const int32_t* first, const int32_t* last, int64_t nodes,
                            uint32_t* key, int32_t* node) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nodes) return;
    const int32_t len = last[d] - first[d];
    const int cls = len == 0 ? N_CLS : node_cls[d];
    const int32_t cl = len < (1 << LEN_BITS) - 1 ? len : (1 << LEN_BITS) - 1;
    key[d] = ((uint32_t)cls << LEN_BITS) | (uint32_t)((1 << LEN_BITS) - 1 - cl);
    node[d] = (int32_t)d;
}
// first[c] = number of sorted keys below class c (c = 0 .. N_CLS)
__global__ void k_class_bounds(const uint32_t* key_sorted, int64_t n_seg, int64_t* first) {
    const int c = threadIdx.x;
    if (c > N_CLS) return;
    int64_t lo = 0, hi = n_seg;
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if ((key_sorted[mid] >> LEN_BITS) < (uint32_t)c) lo = mid + 1;
        else hi = mid;
    }
    first[c] = lo;
}

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
    // (four streams: the runtime maps streams onto four hardware queues by default, and kernels of streams that share a
    // queue run one after the other -- with a stream per class the quads started when the 8-lane groups had finished;
    // the classes that share a stream here rarely have mail in the same generation)
    static constexpr int N_STREAMS = 4;
    hipStream_t cls_stream[N_STREAMS] = {};
    ~Engine() override {
        for (hipStream_t st : cls_stream)
            if (st) (void)hipStreamDestroy(st);
    }
    std::vector<int32_t> h_dom, h_frow, h_evar, h_vrow, h_vedges;
    std::vector<int64_t> h_toff, h_coff, h_moff;
    std::vector<double> h_tables, h_eval_cost;
    Buf<int32_t> dom_size, factor_rowptr, edge_var, edge_factor, var_rowptr, var_edges, init_idx, node_cap;
    Buf<int64_t> table_off, cost_off, msg_off;
    Buf<T> tables, var_cost, f_cost, f_prev, v_cost, v_prev, belief;
    Buf<uint8_t> f_has, f_cnt, v_has, v_cnt;
    Buf<int32_t> f_nhas, v_narr, v_order, sel;
    // The pending generation: its messages sit in the output slots their senders wrote them to -- s_rec (records),
    // s_hdr (8 bytes per slot: destination, capacity + 1, 0 = empty) -- `slots` of them, `pending` filled, and the
    // scan over s_hdr is done: packed (messages before the slot << 33 | output slots before its handler's).
    Buf<T> s_rec, m_rec;  // the output slots; the generation's records in destination-sorted order
    Buf<int32_t> s_hdr;
    Buf<int64_t> packed, pos, slot_base, totals;
    int64_t slots = 0, slots_next = 0;  // slots of the pending generation, slots its handlers need
    int64_t max_cap = 0;                // largest number of output slots a handler needs
    Buf<int32_t> dest, dest_sorted, slot_in, slot_sorted, start_cnt, start_base;
    Buf<int32_t> node_cls, run_first, run_last, seg_node;  // (seg_node, cls_first: the static running order)
    Buf<int32_t> node_in, seg_node_dyn;                    // the order by actual queue length of a large generation
    Buf<uint32_t> seg_key, seg_key_sorted;
    Buf<int64_t> cls_first_dyn;
    bool generic_only = false;
    Buf<int64_t> cls_first;
    Buf<uint8_t> temp;
    int64_t cls_nodes[N_CLS] = {};  // computations of every class (static): the class kernels' grids
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
        {   // output slots the handler of a message to a computation needs: its other neighbours
            std::vector<int32_t> nc((size_t)nV + nF + 1, 0);
            max_cap = 0;
            for (int v = 0; v < nV; ++v) nc[v] = std::max(0, h_vrow[v + 1] - h_vrow[v] - 1);
            for (int f = 0; f < nF; ++f) nc[(size_t)nV + f] = h_frow[f + 1] - h_frow[f] - 1;
            for (int32_t c : nc) max_cap = std::max<int64_t>(max_cap, c);
            AMX_TRY(node_cap.upload(nc));
            int cb = 0;
            while (((int64_t)1 << cb) <= max_cap) ++cb;
            g.cap_bits = ((int64_t)nV + nF + 2) < ((int64_t)1 << (31 - cb)) ? cb : 0;
            if (const char* env = std::getenv("MAXSUM_AMAXSUM_CAP_LOOKUP"); env && env[0] == '1') g.cap_bits = 0;  // (tests)
            g.node_cap = node_cap.p;
        }
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
        {   // The computations in running order, once: by class, and inside a class by how much mail they can expect --
            // a generation lasts as long as its longest queue, and the lanes of a wave walk their queues in lock
            // step, so a wave should hold queues of similar length.  The number of messages a computation gets in
            // generation g is the number of (non-backtracking, not yet silenced) walks of length g that end at it; the
            // plain walk count after a few steps ranks the computations nearly the same way, and it is static.
            // (Sorting the destinations by their actual queue length every generation cost 130 us of a small
            // generation's 500 -- profiles/r04_amaxsum_dispatches_v4.txt; any order is correct.)
            const int64_t nodes = (int64_t)nV + nF;
            AMX_TRY(node_cls.reserve(nodes + 1));
            std::vector<int32_t> hc((size_t)nodes);
            if (nodes) {
                hipLaunchKernelGGL((k_node_class<T>), dim3(grid(nodes)), dim3(TPB), 0, 0, g, node_cls.p);
                AMX_TRY(hipGetLastError());
                AMX_TRY(hipMemcpy(hc.data(), node_cls.p, 4 * nodes, hipMemcpyDeviceToHost));
            }
            const char* env = std::getenv("MAXSUM_AMAXSUM_GENERIC");  // =1: the per-message handler only (A/B, tests)
            generic_only = env && env[0] == '1';
            if (generic_only) {
                for (int32_t& c : hc) c = CLS_GENERIC;
                if (nodes) AMX_TRY(hipMemcpy(node_cls.p, hc.data(), 4 * nodes, hipMemcpyHostToDevice));
            }
            std::vector<double> w((size_t)nodes, 1.0), w2((size_t)nodes);
            for (int it = 0; it < 8; ++it) {
                double top = 0;
                for (int v = 0; v < nV; ++v) {
                    double a = 0;
                    for (int k = h_vrow[v]; k < h_vrow[v + 1]; ++k) a += w[(size_t)nV + h_efac[h_vedges[k]]];
                    w2[v] = a;
                    top = std::max(top, a);
                }
                for (int f = 0; f < nF; ++f) {
                    double a = 0;
                    for (int e = h_frow[f]; e < h_frow[f + 1]; ++e) a += w[h_evar[e]];
                    w2[(size_t)nV + f] = a;
                    top = std::max(top, a);
                }
                const double scale = top > 0 ? 1.0 / top : 1.0;
                for (int64_t i = 0; i < nodes; ++i) w[i] = w2[i] * scale;
            }
            std::vector<int32_t> order((size_t)nodes);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
                return hc[a] != hc[b] ? hc[a] < hc[b] : w[a] > w[b];
            });
            std::vector<int64_t> first(N_CLS + 1, 0);
            for (int c = 0; c < N_CLS; ++c) cls_nodes[c] = 0;
            for (int32_t c : hc) cls_nodes[c] += 1;
            for (int c = 0; c < N_CLS; ++c) first[c + 1] = first[c] + cls_nodes[c];
            AMX_TRY(seg_node.upload(order));
            AMX_TRY(cls_first.upload(first));
            AMX_TRY(cls_first_dyn.reserve(N_CLS + 1));
            AMX_TRY(totals.reserve(2));
        }
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
            if (total > (int64_t)INT32_MAX) return fail(MXS_E_NOMEM, "amaxsum: more than 2^31 start messages");
            AMX_TRY(s_rec.reserve((total + 1) * g.rs));
            AMX_TRY(s_hdr.reserve(total + 1));
            AMX_TRY(hipMemset(s_rec.p, 0, sizeof(T) * (total + 1) * g.rs));  // (header base, payload padding)
            AMX_TRY(hipMemset(s_hdr.p, 0, 4 * (total + 1)));
            hipLaunchKernelGGL((k_start_emit<T>), dim3(grid(nodes)), dim3(TPB), 0, 0, g, start_base.p, s_rec.p, s_hdr.p);
            AMX_TRY(hipGetLastError());
            int rc = finish(total);
            if (rc) return rc;
        }
        if (pending) gen_sizes.push_back(pending);
        return MXS_OK;
    }

    // The generation just produced sits in `n_slots` output slots: count it and number its handlers' output slots
    // (one scan over the 4-byte slot words), totals back to the host -- the one wait of a generation.
    int finish(int64_t n_slots) {
        slots = n_slots;
        pending = 0;
        slots_next = 0;
        if (n_slots == 0) return MXS_OK;
        AMX_TRY(packed.reserve(n_slots));
        const char* env = std::getenv("MAXSUM_AMAXSUM_TWO_SCANS");  // =1: the separate scans whatever the size (tests)
        // (the packed sum carries the message count above bit PACK_SHIFT of a SIGNED 64-bit word: at most 2^30 slots --
        // as many messages at most -- so that it cannot overflow before the guards below see the totals)
        const bool one_scan = !(env && env[0] == '1') && n_slots < ((int64_t)1 << PACK_SHIFT) / (max_cap + 1) &&
                              n_slots <= ((int64_t)1 << 30);
        size_t bytes = 0;
        if (one_scan) {
            hipcub::TransformInputIterator<int64_t, SlotPacked, const int32_t*> both(s_hdr.p, SlotPacked{node_cap.p, g.cap_bits});
            AMX_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, both, packed.p, (int)n_slots));
            AMX_TRY(temp.reserve(bytes));
            AMX_TRY(hipcub::DeviceScan::ExclusiveSum(temp.p, bytes, both, packed.p, (int)n_slots));
            hipLaunchKernelGGL(k_totals, dim3(1), dim3(64), 0, 0, (const int32_t*)s_hdr.p, (const int64_t*)packed.p,
                               (const int32_t*)node_cap.p, g.cap_bits, n_slots, totals.p);
        } else {
            AMX_TRY(pos.reserve(n_slots));
            AMX_TRY(slot_base.reserve(n_slots));
            hipcub::TransformInputIterator<int64_t, SlotFilled, const int32_t*> filled(s_hdr.p, SlotFilled());
            hipcub::TransformInputIterator<int64_t, SlotCap, const int32_t*> caps(s_hdr.p, SlotCap{node_cap.p, g.cap_bits});
            AMX_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, filled, pos.p, (int)n_slots));
            AMX_TRY(temp.reserve(bytes));
            AMX_TRY(hipcub::DeviceScan::ExclusiveSum(temp.p, bytes, filled, pos.p, (int)n_slots));
            AMX_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, caps, slot_base.p, (int)n_slots));
            AMX_TRY(temp.reserve(bytes));
            AMX_TRY(hipcub::DeviceScan::ExclusiveSum(temp.p, bytes, caps, slot_base.p, (int)n_slots));
            hipLaunchKernelGGL(k_totals2, dim3(1), dim3(64), 0, 0, (const int32_t*)s_hdr.p, (const int64_t*)pos.p,
                               (const int64_t*)slot_base.p, (const int32_t*)node_cap.p, g.cap_bits, n_slots, totals.p);
        }
        AMX_TRY(hipGetLastError());
        int64_t h[2] = {0, 0};
        AMX_TRY(read_back(h, totals.p, sizeof(h)));
        pending = h[0];
        slots_next = h[1];
        if (pending > (int64_t)1 << 30) return fail(MXS_E_NOMEM, "amaxsum: more than 2^30 messages in one generation");
        if (slots_next > (int64_t)INT32_MAX)  // slot numbers are 32-bit in the records and the sort
            return fail(MXS_E_NOMEM, "amaxsum: more than 2^31 output slots in one generation");
        if (!one_scan) {
            hipLaunchKernelGGL(k_pack, dim3(grid(n_slots)), dim3(TPB), 0, 0, (const int64_t*)pos.p, (const int64_t*)slot_base.p,
                               n_slots, packed.p);
            AMX_TRY(hipGetLastError());
        }
        return MXS_OK;
    }

    int step() {  // deliver the whole pending generation
        const int64_t n = pending, nodes = (int64_t)g.n_vars + g.n_factors;
        AMX_TRY(dest.reserve(n));
        AMX_TRY(dest_sorted.reserve(n));
        AMX_TRY(slot_in.reserve(n));
        AMX_TRY(slot_sorted.reserve(n));
        hipLaunchKernelGGL(k_compact, dim3(grid(slots)), dim3(TPB), 0, 0, (const int32_t*)s_hdr.p, (const int64_t*)packed.p, slots,
                           g.cap_bits, dest.p, slot_in.p);
        AMX_TRY(hipGetLastError());
        {   // the stable sort by destination: a queue = a run, in FIFO order
            size_t bytes = 0;
            int bits = 1;
            while (((int64_t)1 << bits) < nodes + 1) ++bits;
            AMX_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dest.p, dest_sorted.p, slot_in.p, slot_sorted.p, (int)n, 0, bits));
            AMX_TRY(temp.reserve(bytes));
            AMX_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, bytes, dest.p, dest_sorted.p, slot_in.p, slot_sorted.p, (int)n, 0, bits));
        }
        // One thread (or lane group) per DESTINATION, not per message: where its queue starts and ends
        AMX_TRY(run_first.reserve(nodes));
        AMX_TRY(run_last.reserve(nodes));
        AMX_TRY(hipMemset(run_first.p, 0, 4 * nodes));
        AMX_TRY(hipMemset(run_last.p, 0, 4 * nodes));
        hipLaunchKernelGGL(k_bounds, dim3(grid(n)), dim3(TPB), 0, 0, (const int32_t*)dest_sorted.p, n, run_first.p, run_last.p);
        AMX_TRY(hipGetLastError());
        const char* env_order = std::getenv("MAXSUM_AMAXSUM_ORDER");  // static / dynamic: force one (A/B, tests)
        const bool dynamic_order = env_order ? env_order[0] == 'd' : n >= DYNAMIC_ORDER_FROM;
        if (dynamic_order) {
            AMX_TRY(node_in.reserve(nodes));
            AMX_TRY(seg_node_dyn.reserve(nodes));
            AMX_TRY(seg_key.reserve(nodes));
            AMX_TRY(seg_key_sorted.reserve(nodes));
            hipLaunchKernelGGL(k_node_keys, dim3(grid(nodes)), dim3(TPB), 0, 0, (const int32_t*)node_cls.p,
                               (const int32_t*)run_first.p, (const int32_t*)run_last.p, nodes, seg_key.p, node_in.p);
            AMX_TRY(hipGetLastError());
            size_t bytes = 0;
            AMX_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, seg_key.p, seg_key_sorted.p, node_in.p, seg_node_dyn.p, (int)nodes, 0, LEN_BITS + 3));
            AMX_TRY(temp.reserve(bytes));
            AMX_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, bytes, seg_key.p, seg_key_sorted.p, node_in.p, seg_node_dyn.p, (int)nodes, 0, LEN_BITS + 3));
            hipLaunchKernelGGL(k_class_bounds, dim3(1), dim3(64), 0, 0, (const uint32_t*)seg_key_sorted.p, nodes, cls_first_dyn.p);
            AMX_TRY(hipGetLastError());
        }
        if (const char* dbg = std::getenv("MAXSUM_AMAXSUM_DEBUG"); dbg && dbg[0] == '1') {  // longest queue of every class
            std::vector<int32_t> hf((size_t)nodes), hl((size_t)nodes), hcls((size_t)nodes);
            AMX_TRY(read_back(hf.data(), run_first.p, 4 * nodes));
            AMX_TRY(hipMemcpy(hl.data(), run_last.p, 4 * nodes, hipMemcpyDeviceToHost));
            AMX_TRY(hipMemcpy(hcls.data(), node_cls.p, 4 * nodes, hipMemcpyDeviceToHost));
            int32_t longest[N_CLS] = {};
            int64_t with_mail[N_CLS] = {};
            for (int64_t d = 0; d < nodes; ++d) {
                longest[hcls[d]] = std::max(longest[hcls[d]], hl[d] - hf[d]);
                with_mail[hcls[d]] += hl[d] > hf[d];
            }
            std::fprintf(stderr, "amaxsum generation %d: %lld messages in %lld slots;", next_generation, (long long)n, (long long)slots);
            for (int c = 0; c < N_CLS; ++c)
                std::fprintf(stderr, " class %d: %lld destinations, longest queue %d;", c, (long long)with_mail[c], longest[c]);
            std::fprintf(stderr, "\n");
        }
        // the records themselves into sorted order: the chains read their queues as contiguous runs
        AMX_TRY(m_rec.reserve(n * g.rs));
        {
            const int64_t pieces = n * (g.rs * (int64_t)sizeof(T) / 16);
            hipLaunchKernelGGL((k_permute<T>), dim3(grid(pieces)), dim3(TPB), 0, 0, (const T*)s_rec.p, (const int32_t*)slot_sorted.p,
                               (const int64_t*)packed.p, n, g.rs, m_rec.p);
            AMX_TRY(hipGetLastError());
        }
        // the handlers' output slots (the records of this generation are in m_rec now)
        const int64_t n_out = slots_next;
        AMX_TRY(s_rec.reserve((n_out + 1) * g.rs));
        AMX_TRY(s_hdr.reserve(n_out + 1));
        if (const char* clr = std::getenv("MAXSUM_AMAXSUM_CLEAR_SLOTS"); clr && clr[0] == '1')  // (every handler writes the
            AMX_TRY(hipMemset(s_hdr.p, 0xFF, 4 * (n_out + 1)));  // words of all its slots; =1 poisons them first: tests)
        const Sorted<T> sq{m_rec.p, dynamic_order ? seg_node_dyn.p : seg_node.p, run_first.p, run_last.p,
                           dynamic_order ? cls_first_dyn.p : cls_first.p, n};
        static const int launch_order[N_CLS] = {CLS_VAR16, CLS_VAR8, CLS_VAR64, CLS_GENERIC, CLS_VAR4, CLS_FACTOR2};  // long chains first
        static const int stream_of[N_CLS] = {/*FACTOR2*/ 0, /*VAR8*/ 1, /*VAR16*/ 2, /*VAR64*/ 3, /*GENERIC*/ 3, /*VAR4*/ 0};
        for (int cls : launch_order) {
            const int64_t count = cls_nodes[cls];
            if (count <= 0) continue;
            const int per_wave = cls == CLS_VAR4 ? 16 : (cls == CLS_VAR8 ? 8 : (cls == CLS_VAR16 ? 4 : (cls == CLS_VAR64 ? 1 : 64)));
            const dim3 gr((unsigned)((count + per_wave - 1) / per_wave)), bl(64);
#define AMX_ARGS g, sq, cls, s_rec.p, s_hdr.p
            hipStream_t& st = cls_stream[stream_of[cls]];
            if (!st) AMX_TRY(hipStreamCreateWithFlags(&st, 0));
            if (cls == CLS_VAR4) hipLaunchKernelGGL((k_process_vars<T, 4>), gr, bl, 0, st, AMX_ARGS);
            else if (cls == CLS_VAR8) hipLaunchKernelGGL((k_process_vars<T, 8>), gr, bl, 0, st, AMX_ARGS);
            else if (cls == CLS_VAR16) hipLaunchKernelGGL((k_process_vars<T, 16>), gr, bl, 0, st, AMX_ARGS);
            else if (cls == CLS_VAR64) hipLaunchKernelGGL((k_process_vars<T, 64>), gr, bl, 0, st, AMX_ARGS);
            else if (cls == CLS_FACTOR2) hipLaunchKernelGGL((k_process_lanes<T, true>), gr, bl, 0, st, AMX_ARGS);
            else hipLaunchKernelGGL((k_process_lanes<T, false>), gr, bl, 0, st, AMX_ARGS);
#undef AMX_ARGS
            AMX_TRY(hipGetLastError());
        }
        delivered_total += n;
        next_generation += 1;
        if (const char* dbg = std::getenv("MAXSUM_AMAXSUM_DEBUG"); dbg && dbg[0] == '2') {  // the output slots as written
            AMX_TRY(hipDeviceSynchronize());
            std::vector<int32_t> hh((size_t)n_out);
            std::vector<T> hr((size_t)n_out * g.rs);
            AMX_TRY(hipMemcpy(hh.data(), s_hdr.p, 4 * n_out, hipMemcpyDeviceToHost));
            AMX_TRY(hipMemcpy(hr.data(), s_rec.p, sizeof(T) * n_out * g.rs, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n_out && i < 400; ++i)
                std::fprintf(stderr, "slot %lld: dest + 1 %d code %d\n", (long long)i, (int)hh[i], rec_code(hr.data() + i * g.rs));
        }
        int rc = finish(n_out);
        if (rc) return rc;
        if (pending) gen_sizes.push_back(pending);
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

