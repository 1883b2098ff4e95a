// pattern_bench.hip -- which HBM layout should the Max-Sum sweep use?
//
// Not product code.  Pure data-movement stand-ins for one synchronous cycle on a
// binary-factor graph with D=3 f64 messages (32-byte padded halves), in the
// candidate layouts, timed alone (factor side / variable side) and fused in one
// launch:
//   A  edge records [V->F | F->V] (64 B), factor-major: factor streams its two
//      records, variable gathers 64-B records and scatters 24 B       (v1 engine)
//   B  as A, the factor block stages its contiguous 32 KB of records through LDS
//      with coalesced 16-B-per-lane loads
//   C  split arrays V2F[e], F2V[e] (factor-major) + a variable-private prevV in
//      slot order: factor streams, variable gathers 32 B + scatters 32 B
//   G  gather-only: V2F variable-major, F2V factor-major, every write coalesced,
//      each side gathers the other's array; no private copies
//   S  scatter layout: V2F factor-major, F2V variable-major (slot order), private
//      prevF / prevV: both sides stream their inputs and scatter 32-B outputs
// Every variable has 4 slots (degree 4); slot -> edge is a random permutation.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                                    \
        }                                                               \
    } while (0)

typedef double2 d2;  // 16 bytes
constexpr int BLOCK = 256;

struct Args {
    const d2* rec_old;  d2* rec_new;          // A/B: 4 d2 per edge
    const d2* v2f_old;  d2* v2f_new;          // C/S: 2 d2 per edge, factor-major
    const d2* f2v_old;  d2* f2v_new;          // C: factor-major; S: slot-major
    d2* prevF;          d2* prevV;            // private copies (S both, C prevV)
    const double* tables;                     // [9][nF] entry-major
    uint8_t* cF;        uint8_t* cV;
    const int32_t* slot_edge;                 // [4][nV] slot -> edge id (ELL, k-major)
    const int32_t* edge_slot;                 // [E] edge -> slot position (k*nV + j)
    double* belief;
    int nF, nV, fblocks;
};

__device__ __forceinline__ d2 add2(d2 a, d2 b) { return d2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ d2 min2(d2 a, d2 b) { return d2{a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y}; }

// ---- factor side -----------------------------------------------------------
__device__ __forceinline__ void f_compute(const d2 (&m)[8], const double (&t)[9], d2 (&o)[4]) {
    // stand-in for the 2 x 3 mins over 3 sums
    d2 s0 = add2(m[0], m[4]), s1 = add2(m[1], m[5]), s2 = add2(m[2], m[6]), s3 = add2(m[3], m[7]);
    double tt = t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7] + t[8];
    o[0] = min2(s0, d2{tt, tt});
    o[1] = min2(s1, s0);
    o[2] = min2(s2, d2{tt, tt});
    o[3] = min2(s3, s2);
}

__device__ __forceinline__ void factor_A(const Args& a, int j) {
    if (j >= a.nF) return;
    const d2* r = a.rec_old + (size_t)j * 8;
    d2 m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = r[i];
    double t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = a.tables[(size_t)k * a.nF + j];
    const uint8_t c0 = a.cF[2 * j], c1 = a.cF[2 * j + 1];
    d2 o[4];
    f_compute(m, t, o);
    d2* w = a.rec_new + (size_t)j * 8;
    w[2] = o[0];
    ((double*)(w + 3))[0] = o[1].x;   // 24 bytes of the 32-byte half, as the engine does
    w[6] = o[2];
    ((double*)(w + 7))[0] = o[3].x;
    a.cF[2 * j] = c0 + 1;
    a.cF[2 * j + 1] = c1 + 1;
}

// LDS-staged: the block's 256 factors = 32 KB contiguous; rows padded to 9 d2
__device__ __forceinline__ void factor_B(const Args& a, int first) {
    __shared__ d2 tile[BLOCK * 9];
    const int t = threadIdx.x;
    const size_t base = (size_t)first * 8;
    const size_t limit = (size_t)a.nF * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = i * BLOCK + t;  // chunk of the tile
        if (base + c < limit) tile[(c >> 3) * 9 + (c & 7)] = a.rec_old[base + c];
    }
    const int j = first + t;
    double tb[9];
    uint8_t c0 = 0, c1 = 0;
    if (j < a.nF) {
#pragma unroll
        for (int k = 0; k < 9; ++k) tb[k] = a.tables[(size_t)k * a.nF + j];
        c0 = a.cF[2 * j];
        c1 = a.cF[2 * j + 1];
    }
    __syncthreads();
    if (j >= a.nF) return;
    d2 m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = tile[t * 9 + i];
    d2 o[4];
    f_compute(m, tb, o);
    d2* w = a.rec_new + (size_t)j * 8;
    w[2] = o[0];
    ((double*)(w + 3))[0] = o[1].x;
    w[6] = o[2];
    ((double*)(w + 7))[0] = o[3].x;
    a.cF[2 * j] = c0 + 1;
    a.cF[2 * j + 1] = c1 + 1;
}

// split arrays, direct 64 B per thread per stream, full 64-B stores
__device__ __forceinline__ void factor_C(const Args& a, int j) {
    if (j >= a.nF) return;
    const d2* r = a.v2f_old + (size_t)j * 4;
    const d2* p = a.f2v_old + (size_t)j * 4;
    d2 m[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[i] = r[i];
        m[4 + i] = p[i];
    }
    double t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = a.tables[(size_t)k * a.nF + j];
    const uint8_t c0 = a.cF[2 * j], c1 = a.cF[2 * j + 1];
    d2 o[4];
    f_compute(m, t, o);
    d2* w = a.f2v_new + (size_t)j * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = o[i];
    a.cF[2 * j] = c0 + 1;
    a.cF[2 * j + 1] = c1 + 1;
}

// split arrays, LDS staged loads and stores (everything 16 B per lane, coalesced)
__device__ __forceinline__ void factor_D(const Args& a, int first) {
    __shared__ d2 tin[BLOCK * 9];
    const int t = threadIdx.x;
    const size_t base = (size_t)first * 4, limit = (size_t)a.nF * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * BLOCK + t;
        if (base + c < limit) {
            tin[(c >> 2) * 9 + (c & 3)] = a.v2f_old[base + c];
            tin[(c >> 2) * 9 + 4 + (c & 3)] = a.f2v_old[base + c];
        }
    }
    const int j = first + t;
    double tb[9];
    uint8_t c0 = 0, c1 = 0;
    if (j < a.nF) {
#pragma unroll
        for (int k = 0; k < 9; ++k) tb[k] = a.tables[(size_t)k * a.nF + j];
        c0 = a.cF[2 * j];
        c1 = a.cF[2 * j + 1];
    }
    __syncthreads();
    d2 m[8], o[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = tin[t * 9 + i];
    f_compute(m, tb, o);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) tin[t * 9 + i] = o[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * BLOCK + t;
        if (base + c < limit) a.f2v_new[base + c] = tin[(c >> 2) * 9 + (c & 3)];
    }
    if (j < a.nF) {
        a.cF[2 * j] = c0 + 1;
        a.cF[2 * j + 1] = c1 + 1;
    }
}

// scatter layout: stream V2F + prevF, write prevF (coalesced), scatter 2 x 32 B
__device__ __forceinline__ void factor_S(const Args& a, int j) {
    if (j >= a.nF) return;
    const d2* r = a.v2f_old + (size_t)j * 4;
    d2* p = a.prevF + (size_t)j * 4;
    d2 m[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[i] = r[i];
        m[4 + i] = p[i];
    }
    double t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = a.tables[(size_t)k * a.nF + j];
    const uint8_t c0 = a.cF[2 * j], c1 = a.cF[2 * j + 1];
    const int s0 = a.edge_slot[2 * j], s1 = a.edge_slot[2 * j + 1];
    d2 o[4];
    f_compute(m, t, o);
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = o[i];
    d2* w0 = a.f2v_new + (size_t)s0 * 2;
    d2* w1 = a.f2v_new + (size_t)s1 * 2;
    w0[0] = o[0]; w0[1] = o[1];
    w1[0] = o[2]; w1[1] = o[3];
    a.cF[2 * j] = c0 + 1;
    a.cF[2 * j + 1] = c1 + 1;
}

template <bool NT>
__device__ __forceinline__ d2 ld(const d2* p) {
    if (NT) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d v = __builtin_nontemporal_load((const v2d*)p);
        return d2{v.x, v.y};
    }
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st(d2* p, d2 v) {
    if (NT) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d x = {v.x, v.y};
        __builtin_nontemporal_store(x, (v2d*)p);
    } else {
        *p = v;
    }
}

// gather-only layout: V2F slot-major (written coalesced by the variable side),
// F2V factor-major (written coalesced by the factor side); each side gathers the
// other's array (32-B random reads) and finds its previous output in its own one.
template <bool NT>
__device__ __forceinline__ void factor_G(const Args& a, int j) {
    if (j >= a.nF) return;
    const int s0 = a.edge_slot[2 * j], s1 = a.edge_slot[2 * j + 1];
    const uint8_t c0 = a.cF[2 * j], c1 = a.cF[2 * j + 1];
    const d2* p = a.f2v_old + (size_t)j * 4;
    d2 m[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) m[4 + i] = ld<NT>(p + i);
    double t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = a.tables[(size_t)k * a.nF + j];
    const d2* r0 = a.v2f_old + (size_t)s0 * 2;
    const d2* r1 = a.v2f_old + (size_t)s1 * 2;
    m[0] = ld<NT>(r0); m[1] = ld<NT>(r0 + 1); m[2] = ld<NT>(r1); m[3] = ld<NT>(r1 + 1);
    d2 o[4];
    f_compute(m, t, o);
    d2* w = a.f2v_new + (size_t)j * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) st<NT>(w + i, o[i]);
    a.cF[2 * j] = c0 + 1;
    a.cF[2 * j + 1] = c1 + 1;
}

// gather-only layout, factor side with ONE LANE PER EDGE: lane i of factor j reads its
// own previous output (coalesced), the partner edge's V2F message (gather) and the
// table, writes its own F2V message (coalesced)
__device__ __forceinline__ void factor_H(const Args& a, size_t e) {
    if (e >= (size_t)2 * a.nF) return;
    const size_t j = e >> 1;
    const int s = a.edge_slot[e ^ 1];
    const uint8_t c = a.cF[e];
    const d2* p = a.f2v_old + e * 2;
    const d2 p0 = p[0], p1 = p[1];
    double t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = a.tables[(size_t)((e & 1) ? (k % 3) * 3 + k / 3 : k) * a.nF + j];
    const d2* r = a.v2f_old + (size_t)s * 2;
    const d2 m0 = r[0], m1 = r[1];
    double tt = t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7] + t[8];
    d2* w = a.f2v_new + e * 2;
    w[0] = min2(add2(m0, p0), d2{tt, tt});
    w[1] = min2(add2(m1, p1), m0);
    a.cF[e] = c + 1;
}

template <bool NT>
__device__ __forceinline__ void var_G(const Args& a, int j) {
    if (j >= a.nV) return;
    int e[4];
    uint8_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e[k] = a.slot_edge[(size_t)k * a.nV + j];
        c[k] = a.cV[(size_t)k * a.nV + j];
    }
    d2 in[4][2], pv[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const d2* p = a.v2f_old + ((size_t)k * a.nV + j) * 2;
        pv[k][0] = ld<NT>(p);
        pv[k][1] = ld<NT>(p + 1);
        const d2* r = a.f2v_old + (size_t)e[k] * 2;
        in[k][0] = ld<NT>(r);
        in[k][1] = ld<NT>(r + 1);
    }
    d2 s0 = add2(add2(in[0][0], in[1][0]), add2(in[2][0], in[3][0]));
    d2 s1 = add2(add2(in[0][1], in[1][1]), add2(in[2][1], in[3][1]));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        d2* w = a.v2f_new + ((size_t)k * a.nV + j) * 2;
        st<NT>(w, add2(s0, pv[k][0]));
        st<NT>(w + 1, add2(s1, pv[k][1]));
        a.cV[(size_t)k * a.nV + j] = c[k] + 1;
    }
    a.belief[j] = s0.x + s1.y;
}

// ---- variable side -----------------------------------------------------------
__device__ __forceinline__ void var_A(const Args& a, int j) {  // thread per variable, 4 record gathers
    if (j >= a.nV) return;
    int e[4];
    uint8_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e[k] = a.slot_edge[(size_t)k * a.nV + j];
        c[k] = a.cV[(size_t)k * a.nV + j];
    }
    d2 m[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const d2* r = a.rec_old + (size_t)e[k] * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) m[k][i] = r[i];
    }
    d2 s0 = add2(add2(m[0][2], m[1][2]), add2(m[2][2], m[3][2]));
    d2 s1 = add2(add2(m[0][3], m[1][3]), add2(m[2][3], m[3][3]));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        d2* w = a.rec_new + (size_t)e[k] * 4;
        w[0] = add2(s0, m[k][0]);
        ((double*)(w + 1))[0] = s1.x + m[k][1].x;
        a.cV[(size_t)k * a.nV + j] = c[k] + 1;
    }
    a.belief[j] = s0.x + s1.y;
}

__device__ __forceinline__ void var_C(const Args& a, int j) {  // split arrays: 32-B gathers, prevV private
    if (j >= a.nV) return;
    int e[4];
    uint8_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e[k] = a.slot_edge[(size_t)k * a.nV + j];
        c[k] = a.cV[(size_t)k * a.nV + j];
    }
    d2 in[4][2], pv[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const d2* r = a.f2v_old + (size_t)e[k] * 2;
        in[k][0] = r[0];
        in[k][1] = r[1];
        const d2* p = a.prevV + ((size_t)k * a.nV + j) * 2;
        pv[k][0] = p[0];
        pv[k][1] = p[1];
    }
    d2 s0 = add2(add2(in[0][0], in[1][0]), add2(in[2][0], in[3][0]));
    d2 s1 = add2(add2(in[0][1], in[1][1]), add2(in[2][1], in[3][1]));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const d2 o0 = add2(s0, pv[k][0]), o1 = add2(s1, pv[k][1]);
        d2* w = a.v2f_new + (size_t)e[k] * 2;
        w[0] = o0;
        w[1] = o1;
        d2* p = a.prevV + ((size_t)k * a.nV + j) * 2;
        p[0] = o0;
        p[1] = o1;
        a.cV[(size_t)k * a.nV + j] = c[k] + 1;
    }
    a.belief[j] = s0.x + s1.y;
}

__device__ __forceinline__ void var_S(const Args& a, int j) {  // scatter layout: all reads coalesced
    if (j >= a.nV) return;
    int e[4];
    uint8_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e[k] = a.slot_edge[(size_t)k * a.nV + j];
        c[k] = a.cV[(size_t)k * a.nV + j];
    }
    d2 in[4][2], pv[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const d2* r = a.f2v_old + ((size_t)k * a.nV + j) * 2;
        in[k][0] = r[0];
        in[k][1] = r[1];
        const d2* p = a.prevV + ((size_t)k * a.nV + j) * 2;
        pv[k][0] = p[0];
        pv[k][1] = p[1];
    }
    d2 s0 = add2(add2(in[0][0], in[1][0]), add2(in[2][0], in[3][0]));
    d2 s1 = add2(add2(in[0][1], in[1][1]), add2(in[2][1], in[3][1]));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const d2 o0 = add2(s0, pv[k][0]), o1 = add2(s1, pv[k][1]);
        d2* w = a.v2f_new + (size_t)e[k] * 2;
        w[0] = o0;
        w[1] = o1;
        d2* p = a.prevV + ((size_t)k * a.nV + j) * 2;
        p[0] = o0;
        p[1] = o1;
        a.cV[(size_t)k * a.nV + j] = c[k] + 1;
    }
    a.belief[j] = s0.x + s1.y;
}

// lane per edge (slot-major order), record layout A
__device__ __forceinline__ void var_E(const Args& a, size_t i) {
    if (i >= (size_t)4 * a.nV) return;
    const int e = a.slot_edge[i];
    const uint8_t c = a.cV[i];
    const d2* r = a.rec_old + (size_t)e * 4;
    const d2 m0 = r[0], m1 = r[1], m2 = r[2], m3 = r[3];
    d2* w = a.rec_new + (size_t)e * 4;
    w[0] = add2(m0, m2);
    ((double*)(w + 1))[0] = m1.x + m3.x;
    a.cV[i] = c + 1;
}

// mode: 1 factor only, 2 variable only, 3 both in one launch
template <char L>
__global__ void __launch_bounds__(BLOCK) k_cycle(Args a, int mode) {
    int b = blockIdx.x;
    bool is_f = (mode == 1) || (mode == 3 && b < a.fblocks);
    int vb_i = -1;
    if (mode == 4) {  // factor and variable blocks interleaved in proportion
        const int n = gridDim.x, nf = a.fblocks;
        const long long f0 = (long long)b * nf / n, f1 = (long long)(b + 1) * nf / n;
        is_f = f1 > f0;
        vb_i = b - (int)f1;   // variable blocks before this one
        if (is_f) b = (int)f0;
    }
    if (is_f) {
        const int first = b * BLOCK, j = first + (int)threadIdx.x;
        if (L == 'A' || L == 'E') factor_A(a, j);
        else if (L == 'B') factor_B(a, first);
        else if (L == 'C') factor_C(a, j);
        else if (L == 'D') factor_D(a, first);
        else if (L == 'G') factor_G<false>(a, j);
        else if (L == 'J') factor_G<true>(a, j);
        else if (L == 'H') { factor_H(a, (size_t)b * BLOCK * 2 + threadIdx.x); factor_H(a, (size_t)b * BLOCK * 2 + BLOCK + threadIdx.x); }
        else if (L == 'I') factor_H(a, (size_t)b * BLOCK + threadIdx.x);
        else factor_S(a, j);
    } else {
        const int vb = (mode == 3) ? b - a.fblocks : (mode == 4) ? vb_i : b;
        const int j = vb * BLOCK + (int)threadIdx.x;
        if (L == 'A' || L == 'B') var_A(a, j);
        else if (L == 'E') var_E(a, (size_t)vb * BLOCK + threadIdx.x);
        else if (L == 'C' || L == 'D') var_C(a, j);
        else if (L == 'G' || L == 'H' || L == 'I') var_G<false>(a, j);
        else if (L == 'J') var_G<true>(a, j);
        else var_S(a, j);
    }
}

template <char L>
static void run(const char* name, Args a, hipStream_t st, hipEvent_t e0, hipEvent_t e1, int reps) {
    const int fb = L == 'I' ? (2 * a.nF + BLOCK - 1) / BLOCK : (a.nF + BLOCK - 1) / BLOCK;
    const int vb = L == 'E' ? (4 * a.nV + BLOCK - 1) / BLOCK : (a.nV + BLOCK - 1) / BLOCK;
    a.fblocks = fb;
    float ms[5] = {0, 0, 0, 0, 0};
    for (int mode = 1; mode <= 4; ++mode) {
        const int grid = mode == 1 ? fb : mode == 2 ? vb : fb + vb;
        Args x = a;
        for (int i = 0; i < reps + 6; ++i) {
            if (i == 6) CHECK(hipEventRecord(e0, st));
            hipLaunchKernelGGL((k_cycle<L>), dim3(grid), dim3(BLOCK), 0, st, x, mode);
            // ping-pong like the engine
            { d2* t = (d2*)x.rec_old; x.rec_old = x.rec_new; x.rec_new = t; }
            { d2* t = (d2*)x.v2f_old; x.v2f_old = x.v2f_new; x.v2f_new = t; }
            { d2* t = (d2*)x.f2v_old; x.f2v_old = x.f2v_new; x.f2v_new = t; }
        }
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
    }
    printf("{\"bench\": \"pattern\", \"layout\": \"%s\", \"factors\": %d, \"vars\": %d, "
           "\"factor_us\": %.2f, \"variable_us\": %.2f, \"fused_us\": %.2f, \"interleaved_us\": %.2f}\n",
           name, a.nF, a.nV, 1e3 * ms[1] / reps, 1e3 * ms[2] / reps, 1e3 * ms[3] / reps, 1e3 * ms[4] / reps);
    fflush(stdout);
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int nV : {100000, 1500000}) {
        const int nF = 2 * nV, E = 2 * nF;
        Args a{};
        a.nF = nF;
        a.nV = nV;
        d2 *rec[2], *v2f[2], *f2v[2];
        for (int b = 0; b < 2; ++b) {
            CHECK(hipMalloc((void**)&rec[b], (size_t)E * 64));
            CHECK(hipMalloc((void**)&v2f[b], (size_t)E * 32));
            CHECK(hipMalloc((void**)&f2v[b], (size_t)E * 32));
            CHECK(hipMemsetAsync(rec[b], 0, (size_t)E * 64, st));
            CHECK(hipMemsetAsync(v2f[b], 0, (size_t)E * 32, st));
            CHECK(hipMemsetAsync(f2v[b], 0, (size_t)E * 32, st));
        }
        a.rec_old = rec[0]; a.rec_new = rec[1];
        a.v2f_old = v2f[0]; a.v2f_new = v2f[1];
        a.f2v_old = f2v[0]; a.f2v_new = f2v[1];
        CHECK(hipMalloc((void**)&a.prevF, (size_t)E * 32));
        CHECK(hipMalloc((void**)&a.prevV, (size_t)E * 32));
        CHECK(hipMemsetAsync(a.prevF, 0, (size_t)E * 32, st));
        CHECK(hipMemsetAsync(a.prevV, 0, (size_t)E * 32, st));
        double* tab;
        CHECK(hipMalloc((void**)&tab, (size_t)nF * 72));
        CHECK(hipMemsetAsync(tab, 0, (size_t)nF * 72, st));
        a.tables = tab;
        CHECK(hipMalloc((void**)&a.cF, E));
        CHECK(hipMalloc((void**)&a.cV, E));
        CHECK(hipMemsetAsync(a.cF, 0, E, st));
        CHECK(hipMemsetAsync(a.cV, 0, E, st));
        CHECK(hipMalloc((void**)&a.belief, (size_t)nV * 8));
        std::vector<int32_t> se(E), es(E);
        for (int i = 0; i < E; ++i) se[i] = i;
        uint64_t s = 88172645463325252ull;
        for (int i = E - 1; i > 0; --i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            std::swap(se[i], se[s % (uint64_t)(i + 1)]);
        }
        for (int i = 0; i < E; ++i) es[se[i]] = i;
        int32_t *dse, *des;
        CHECK(hipMalloc((void**)&dse, (size_t)E * 4));
        CHECK(hipMalloc((void**)&des, (size_t)E * 4));
        CHECK(hipMemcpyAsync(dse, se.data(), (size_t)E * 4, hipMemcpyHostToDevice, st));
        CHECK(hipMemcpyAsync(des, es.data(), (size_t)E * 4, hipMemcpyHostToDevice, st));
        CHECK(hipStreamSynchronize(st));
        a.slot_edge = dse;
        a.edge_slot = des;
        const int reps = nV > 500000 ? 30 : 300;
        run<'A'>("A records, direct", a, st, e0, e1, reps);
        run<'B'>("B records, factor via LDS", a, st, e0, e1, reps);
        run<'E'>("E records, variable = lane per edge", a, st, e0, e1, reps);
        run<'C'>("C split arrays, direct", a, st, e0, e1, reps);
        run<'D'>("D split arrays, factor via LDS", a, st, e0, e1, reps);
        run<'S'>("S scatter layout", a, st, e0, e1, reps);
        run<'G'>("G gather-only layout", a, st, e0, e1, reps);
        run<'J'>("J gather-only, non-temporal message loads/stores", a, st, e0, e1, reps);
        run<'H'>("H gather-only, factor lane per edge (2 edges per thread, strided)", a, st, e0, e1, reps);
        run<'I'>("I gather-only, factor lane per edge", a, st, e0, e1, reps);
        for (int b = 0; b < 2; ++b) {
            CHECK(hipFree(rec[b])); CHECK(hipFree(v2f[b])); CHECK(hipFree(f2v[b]));
        }
        CHECK(hipFree(a.prevF)); CHECK(hipFree(a.prevV)); CHECK(hipFree(tab));
        CHECK(hipFree(a.cF)); CHECK(hipFree(a.cV)); CHECK(hipFree(a.belief));
        CHECK(hipFree(dse)); CHECK(hipFree(des));
    }
    return 0;
}
