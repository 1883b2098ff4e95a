// valu_bench.hip -- issue cost of the instructions the workgroup-per-factor kernels are made of
// (calibration tool, not product code).  The n-ary factor kernel of meeting_50k is VALU-bound
// (profiles/r03_meeting50k_pmc_kernels_v1.txt: VALU busy 83 %); to restructure it one has to know
// what an f64 add / min, an int8 -> f64 conversion, a DPP move, a permlane swap and an LDS atomic
// with same-address lanes cost per wave on a gfx950 SIMD.
//
// Every kernel runs ITER iterations of 32 instructions of ONE kind on 8 independent registers
// (no dependent-issue stalls), W waves per SIMD on every SIMD of the chip; a wave times its own
// loop with s_memtime.  Reported: cycles per wave-instruction seen by one wave, and that divided
// by W = cycles of SIMD time per wave-instruction when W waves share the SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_bench valu_bench.hip && ./valu_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define HIP_OK(x)                                                              \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP32(S) REP8(S) REP8(S) REP8(S) REP8(S)

// T: register type of the 8 accumulators; INIT: how they start; ASM(i): one instruction on x[i]
#define KERNEL(NAME, T, INIT, ASM)                                                           \
    __global__ void __launch_bounds__(256) NAME(int iters, long long* ticks, double* sink) {  \
        T x[8];                                                                              \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) x[i] = INIT;                           \
        const double c = 1.0 + (double)threadIdx.x * 1e-9;                                   \
        const float cf = 1.0f + (float)threadIdx.x * 1e-6f;                                  \
        const f2 cp = {cf, cf};                                                              \
        const int ci = (int)threadIdx.x * 0x01010101 + 0x7f3c8912;                           \
        (void)c, (void)cf, (void)cp, (void)ci;                                               \
        const long long t0 = (long long)__builtin_amdgcn_s_memtime();                        \
        for (int it = 0; it < iters; ++it) { REP32(ASM) }                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                   \
        const long long t1 = (long long)__builtin_amdgcn_s_memtime();                        \
        double s = 0;                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) s += tod(x[i]);                        \
        if (s == 123.456) sink[0] = s;                                                       \
        if ((threadIdx.x & 63) == 0)                                                         \
            ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                   \
    }

__device__ __forceinline__ double tod(double x) { return x; }
__device__ __forceinline__ double tod(float x) { return (double)x; }
__device__ __forceinline__ double tod(int x) { return (double)x; }
__device__ __forceinline__ double tod(f2 x) { return (double)x.x + (double)x.y; }

#define A_ADD_F64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define A_MIN_F64(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define A_CVT_F64_I32(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[i]) : "v"(ci));
#define A_CVT_F64_F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[i]) : "v"(cf));
#define A_CVT_F32_I8(i) \
    asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(x[i]) : "v"(ci));
#define A_CVT_F32_UB(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x[i]) : "v"(ci));
#define A_BFE_I32(i) asm volatile("v_bfe_i32 %0, %1, 8, 8" : "=v"(x[i]) : "v"(ci));
#define A_ADD_F32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(cf));
#define A_MIN_F32(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(cf));
#define A_MIN3_F32(i) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(cf));
#define A_PK_ADD_F32(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(cp));
#define A_DPP_QUAD(i) \
    asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(ci));
#define A_DPP_ROR(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(ci));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(ci) : "vcc");
#define A_SWAP32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[(i + 1) & 7]));
#define A_ADD_MIN_F64(i) \
    asm volatile("v_add_f64 %0, %1, %2\n\tv_min_f64 %1, %1, %0" : "=&v"(t64), "+v"(x[i]) : "v"(c));
#define A_ADD_F64_DEP(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[0]) : "v"(c));
#define A_ADD_F32_DEP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(cf));
#define A_LSHL_ADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(ci));
#define A_MUL_F64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define A_FMA_F64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c));
#define A_MAX_F64(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c));
#define A_CMP_F64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x[i]), "v"(c) : "vcc");

KERNEL(k_add_f64, double, 0.0, A_ADD_F64)
KERNEL(k_min_f64, double, 1e9, A_MIN_F64)
KERNEL(k_add_f64_dep, double, 0.0, A_ADD_F64_DEP)
KERNEL(k_add_f32_dep, float, 0.0f, A_ADD_F32_DEP)
KERNEL(k_max_f64, double, -1e9, A_MAX_F64)
KERNEL(k_mul_f64, double, 1.0, A_MUL_F64)
KERNEL(k_fma_f64, double, 1.0, A_FMA_F64)
KERNEL(k_cvt_f64_i32, double, 0.0, A_CVT_F64_I32)
KERNEL(k_cvt_f64_f32, double, 0.0, A_CVT_F64_F32)
KERNEL(k_cvt_f32_i8_sdwa, float, 0.0f, A_CVT_F32_I8)
KERNEL(k_cvt_f32_ubyte, float, 0.0f, A_CVT_F32_UB)
KERNEL(k_bfe_i32, int, 0, A_BFE_I32)
KERNEL(k_add_f32, float, 0.0f, A_ADD_F32)
KERNEL(k_min_f32, float, 1e9f, A_MIN_F32)
KERNEL(k_min3_f32, float, 1e9f, A_MIN3_F32)
KERNEL(k_pk_add_f32, f2, (f2{0.0f, 0.0f}), A_PK_ADD_F32)
KERNEL(k_dpp_quad, int, 1, A_DPP_QUAD)
KERNEL(k_dpp_ror, int, 1, A_DPP_ROR)
KERNEL(k_cndmask, int, 1, A_CNDMASK)
KERNEL(k_swap32, int, (int)threadIdx.x, A_SWAP32)
KERNEL(k_lshl_add, int, 1, A_LSHL_ADD)
KERNEL(k_cmp_f64, double, 0.0, A_CMP_F64)
// selects: the e32 form reads VCC; the e64 form any SGPR pair; a compare feeding the select
#define A_CNDMASK_E64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(ci), "s"(smask));
#define A_CMP_CND(i) \
    asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(ci) : "vcc");
#define A_CMP_CND_S(i) \
    asm volatile("v_cmp_lt_i32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(ci), "s"(smask));
#define KERNEL_S(NAME, ASM)                                                                  \
    __global__ void __launch_bounds__(256) NAME(int iters, long long* ticks, double* sink) {  \
        int x[8];                                                                            \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) x[i] = i;                              \
        const int ci = (int)threadIdx.x * 0x01010101 + 0x7f3c8912;                           \
        unsigned long long smask = 0x5555555555555555ull * (unsigned long long)(iters | 1);  \
        smask = __builtin_amdgcn_readfirstlane((int)smask) | ((unsigned long long)iters << 33); \
        const long long t0 = (long long)__builtin_amdgcn_s_memtime();                        \
        for (int it = 0; it < iters; ++it) { REP32(ASM) }                                    \
        const long long t1 = (long long)__builtin_amdgcn_s_memtime();                        \
        double s = 0;                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) s += (double)x[i];                     \
        if (s == 123.456) sink[0] = s + (double)smask;                                       \
        if ((threadIdx.x & 63) == 0)                                                         \
            ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                   \
    }
KERNEL_S(k_cndmask_e64, A_CNDMASK_E64)
KERNEL_S(k_cmp_cnd, A_CMP_CND)
// what the compiler makes of a select on doubles under a lane-dependent condition
__global__ void __launch_bounds__(256) k_select_f64(int iters, long long* ticks, double* sink) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (double)i;
    const bool odd = (threadIdx.x & 1) != 0;
    const double c = 1.0 + (double)threadIdx.x * 1e-9;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                double y = odd ? x[i] : x[(i + 1) & 7];
                asm volatile("" : "+v"(y));  // keep the select
                x[i] = y;
            }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    double s = c;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456) sink[0] = s;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// add + dependent min: the pair every table entry costs per output (2 instructions per ASM)
__global__ void __launch_bounds__(256) k_add_min_f64x2(int iters, long long* ticks, double* sink) {
    double x[8], t64;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1e9;
    const double c = 1.0 + (double)threadIdx.x * 1e-9;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) { REP32(A_ADD_MIN_F64) }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456) sink[0] = s;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// LDS: atomic minima with K lanes of the wave on the same address (K = 1: all 64 distinct), and
// plain 8-byte reads / writes.
template <int K, int MODE>  // MODE 0: ds_min_u64, 1: ds_read_b64, 2: ds_write_b64, 3: ds_min_u32
__global__ void __launch_bounds__(256) k_lds(int iters, long long* ticks, double* sink) {
    __shared__ unsigned long long s[4][64];
    const int w = (int)threadIdx.x >> 6, l = (int)threadIdx.x & 63;
    s[w][l] = ~0ull;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(&s[w][l / K]);  // LDS byte address (low 32 bits of the generic pointer)
    unsigned long long v = 0x7ff0000000000000ull - (unsigned long long)threadIdx.x;
    unsigned long long r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE == 0) asm volatile("ds_min_u64 %0, %1" ::"v"(addr), "v"(v) : "memory");
            else if (MODE == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(r[i & 7]) : "v"(addr) : "memory");
            else if (MODE == 2) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
            else asm volatile("ds_min_u32 %0, %1" ::"v"(addr), "v"((unsigned)v) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    unsigned long long acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += r[i];
    if (acc == 12345 || s[w][l] == 77) sink[0] = 1.0;
    if (l == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(int, long long*, double*);
struct Entry {
    const char* name;
    kern_t fn;
    int per_asm;  // instructions per ASM statement
};

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    long long* d_ticks;
    double* d_sink;
    const int max_waves = cus * 4 * 8 + 64;
    HIP_OK(hipMalloc(&d_ticks, sizeof(long long) * max_waves));
    HIP_OK(hipMalloc(&d_sink, 8));
    std::vector<Entry> es = {
        {"v_add_f64", k_add_f64, 1},
        {"v_min_f64", k_min_f64, 1},
        {"v_add_f64 ONE dependent chain", k_add_f64_dep, 1},
        {"v_add_f32 ONE dependent chain", k_add_f32_dep, 1},
        {"v_max_f64", k_max_f64, 1},
        {"v_mul_f64", k_mul_f64, 1},
        {"v_fma_f64", k_fma_f64, 1},
        {"v_add_f64+v_min_f64 (dependent pair)", k_add_min_f64x2, 2},
        {"v_cmp_lt_f64", k_cmp_f64, 1},
        {"v_cvt_f64_i32", k_cvt_f64_i32, 1},
        {"v_cvt_f64_f32", k_cvt_f64_f32, 1},
        {"v_cvt_f32_i32_sdwa(byte)", k_cvt_f32_i8_sdwa, 1},
        {"v_cvt_f32_ubyte1", k_cvt_f32_ubyte, 1},
        {"v_bfe_i32", k_bfe_i32, 1},
        {"v_lshl_add_u32", k_lshl_add, 1},
        {"v_add_f32", k_add_f32, 1},
        {"v_min_f32", k_min_f32, 1},
        {"v_min3_f32", k_min3_f32, 1},
        {"v_pk_add_f32", k_pk_add_f32, 1},
        {"v_mov_b32_dpp quad_perm", k_dpp_quad, 1},
        {"v_mov_b32_dpp row_ror", k_dpp_ror, 1},
        {"v_cndmask_b32", k_cndmask, 1},
        {"v_cndmask_b32_e64 (sgpr pair)", k_cndmask_e64, 1},
        {"v_cmp_lt_i32+v_cndmask_b32 (vcc)", k_cmp_cnd, 2},
        {"select on f64 (compiler: 2 x v_cndmask)", k_select_f64, 2},
        {"v_permlane32_swap_b32", k_swap32, 1},
        {"ds_min_u64 64 addresses", k_lds<1, 0>, 1},
        {"ds_min_u64 4 lanes/address", k_lds<4, 0>, 1},
        {"ds_min_u64 16 lanes/address", k_lds<16, 0>, 1},
        {"ds_min_u64 64 lanes/address", k_lds<64, 0>, 1},
        {"ds_min_u32 16 lanes/address", k_lds<16, 3>, 1},
        {"ds_read_b64", k_lds<1, 1>, 1},
        {"ds_write_b64", k_lds<1, 2>, 1},
    };
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"clock_khz\": %d}\n", prop.name, cus, iters, prop.clockRate);
    for (const Entry& e : es) {
        for (int W : {1, 2, 4, 8}) {
            const int blocks = cus * W;  // 256 threads = 4 waves = one per SIMD of a CU
            const int waves = blocks * 4;
            HIP_OK(hipMemset(d_ticks, 0, sizeof(long long) * max_waves));
            e.fn<<<blocks, 256>>>(iters / 10 + 1, d_ticks, d_sink);  // warm-up
            hipEvent_t a, b;
            HIP_OK(hipEventCreate(&a));
            HIP_OK(hipEventCreate(&b));
            HIP_OK(hipEventRecord(a));
            e.fn<<<blocks, 256>>>(iters, d_ticks, d_sink);
            HIP_OK(hipEventRecord(b));
            HIP_OK(hipEventSynchronize(b));
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, a, b));
            std::vector<long long> h(waves);
            HIP_OK(hipMemcpy(h.data(), d_ticks, sizeof(long long) * waves, hipMemcpyDeviceToHost));
            double sum = 0;
            for (long long t : h) sum += (double)t;
            const double n_inst = (double)iters * 32 * e.per_asm;
            // s_memtime counts at a fixed 100 MHz on gfx9: convert with the kernel's wall time instead
            const double per_wave_ticks = sum / waves / n_inst;
            const double ns_per_inst_simd = (double)ms * 1e6 / (n_inst * W);  // SIMD time per wave-instruction
            printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"ns_per_wave_inst_simd\": %.4f, "
                   "\"cycles_at_2p4ghz\": %.3f, \"memtime_ticks_per_inst\": %.4f, \"kernel_ms\": %.4f}\n",
                   e.name, W, ns_per_inst_simd, ns_per_inst_simd * 2.4, per_wave_ticks, ms);
            HIP_OK(hipEventDestroy(a));
            HIP_OK(hipEventDestroy(b));
        }
    }
    return 0;
}
