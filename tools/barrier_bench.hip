// barrier_bench.hip -- would ONE persistent kernel running K Max-Sum cycles beat K launches on the
// small instance (BASELINE configs[1]: 10k variables, 7.9 MB per cycle, 236 workgroups)?
// ("with the grid barrier cost measured rather than quoted".)
//
// A cycle is modelled as what k_sweep does to memory at that size: every workgroup reads 32 KB it
// did not write (written by ANOTHER workgroup -- on another XCD -- in the previous cycle) and writes
// 16 KB, ping-pong.  Between two cycles every workgroup must see every other workgroup's stores:
//   (a) kernel boundary, eager launches          (b) kernel boundary, hipGraph replay of 32
//   (c) one persistent launch, grid barrier = per-XCD counters + a top counter (release fence before
//       arriving, acquire fence after leaving: the stores of the other XCDs' L2s must be visible)
//   (d) as (c) with a single flat counter
// Prints one JSON line per variant.  Not product code.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                                    \
        }                                                               \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int BLOCK = 256;
constexpr int RD = 32 * 1024 / 16 / BLOCK;  // 16-byte loads per thread: 8
constexpr int WR = 16 * 1024 / 16 / BLOCK;  // stores per thread: 4

__device__ __forceinline__ void cycle_body(const v4f* __restrict__ in, v4f* __restrict__ out, int nb) {
    // read the 32 KB a "neighbour" workgroup wrote (block b reads what block (b * 37 + 11) % nb owns)
    const int src = (int)(((unsigned)blockIdx.x * 37u + 11u) % (unsigned)nb);
    const v4f* p = in + (size_t)src * (RD * BLOCK) + threadIdx.x;
    v4f acc = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < RD; ++r) acc += p[r * BLOCK];
    v4f* q = out + (size_t)blockIdx.x * (RD * BLOCK) + threadIdx.x;
#pragma unroll
    for (int w = 0; w < WR; ++w) q[w * BLOCK] = acc + (float)w;
}

__global__ void __launch_bounds__(BLOCK) k_cycle(const v4f* in, v4f* out, int nb) { cycle_body(in, out, nb); }

// bar[0]: top counter, bar[16 * (1 + x)]: counter of XCD x (workgroup b runs on XCD b % 8), bar[16 * 9]: flat
__device__ __forceinline__ void grid_barrier_xcd(unsigned* bar, int nb, unsigned epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int x = blockIdx.x & 7;
        const unsigned in_xcd = (unsigned)((nb - x + 7) / 8);
        unsigned* mine = bar + 16 * (1 + x);
        if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch * in_xcd - 1)
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = epoch * (unsigned)(nb < 8 ? nb : 8);
        for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin)
            __builtin_amdgcn_s_sleep(1);  // (bounded: a bug must not hang the GPU)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
__device__ __forceinline__ void grid_barrier_flat(unsigned* bar, int nb, unsigned epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        unsigned* c = bar + 16 * 9;
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * (unsigned)nb; ++spin)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <bool XCD, bool WORK>
__global__ void __launch_bounds__(BLOCK) k_persistent(v4f* a, v4f* b, int nb, int cycles, unsigned* bar) {
    for (int c = 0; c < cycles; ++c) {
        if (WORK) cycle_body((c & 1) ? b : a, (c & 1) ? a : b, nb);
        if (XCD) grid_barrier_xcd(bar, nb, (unsigned)(c + 1));
        else grid_barrier_flat(bar, nb, (unsigned)(c + 1));
    }
}

// (e) ONE XCD: a launch of 8 * W workgroups of which only those with blockIdx % 8 == 0 work
// (workgroup b runs on XCD b % 8: W workgroups on the 32 CUs of XCD 0), each walking the cycle's nb virtual
// blocks w, w + W, ...; the barrier is one counter in THAT XCD's L2 (atomics without system / agent scope bits are
// executed in the L2 every CU of the XCD shares), the data a workgroup did not write itself is read with
// non-temporal loads (they bypass the per-CU L1, which another CU's stores do not invalidate).
template <bool WORK, int HALF>  // HALF: 1 = the f32-sized working set (half the bytes per virtual block)
__global__ void __launch_bounds__(BLOCK) k_persistent_one_xcd(v4f* a, v4f* b, int nb, int cycles, unsigned* bar, int W) {
    if ((blockIdx.x & 7) != 0) return;
    const int w = (int)blockIdx.x >> 3;
    unsigned* ctr = bar + 16 * 10;
    for (int c = 0; c < cycles; ++c) {
        if (WORK) {
            const v4f* in = (c & 1) ? b : a;
            v4f* out = (c & 1) ? a : b;
            for (int vb = w; vb < nb; vb += W) {
                const int src = (int)(((unsigned)vb * 37u + 11u) % (unsigned)nb);
                const v4f* p = in + (size_t)src * (RD * BLOCK) + threadIdx.x;
                v4f acc = {0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < RD / (1 + HALF); ++r) acc += __builtin_nontemporal_load(p + r * BLOCK);
                v4f* q = out + (size_t)vb * (RD * BLOCK) + threadIdx.x;
#pragma unroll
                for (int k = 0; k < WR / (1 + HALF); ++k) q[k * BLOCK] = acc + (float)k;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned want = (unsigned)(c + 1) * (unsigned)W;
            // (polled with an agent-scope load: served by the L2, never by this CU's L1; bounded: a bug must not hang the GPU)
            for (int spin = 0; spin < (1 << 14) && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin)
                __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 236;
    const int K = 2048;
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t bytes = (size_t)nb * RD * BLOCK * 16;
    v4f *a, *b;
    unsigned* bar;
    CHECK(hipMalloc((void**)&a, bytes));
    CHECK(hipMalloc((void**)&b, bytes));
    CHECK(hipMalloc((void**)&bar, 4096));
    CHECK(hipMemsetAsync(a, 0, bytes, s));
    CHECK(hipMemsetAsync(b, 0, bytes, s));
    auto ms = [&]() {
        CHECK(hipEventSynchronize(e1));
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        return t;
    };
    // (a) eager launches
    for (int c = 0; c < 64; ++c) hipLaunchKernelGGL(k_cycle, dim3(nb), dim3(BLOCK), 0, s, (c & 1) ? b : a, (c & 1) ? a : b, nb);
    CHECK(hipEventRecord(e0, s));
    for (int c = 0; c < K; ++c) hipLaunchKernelGGL(k_cycle, dim3(nb), dim3(BLOCK), 0, s, (c & 1) ? b : a, (c & 1) ? a : b, nb);
    CHECK(hipEventRecord(e1, s));
    printf("{\"bench\": \"cycle_per_launch_eager\", \"workgroups\": %d, \"us_per_cycle\": %.3f}\n", nb, 1e3 * ms() / K);
    // (b) hipGraph replay of 32 launches
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int c = 0; c < 32; ++c) hipLaunchKernelGGL(k_cycle, dim3(nb), dim3(BLOCK), 0, s, (c & 1) ? b : a, (c & 1) ? a : b, nb);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 4; ++i) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < K / 32; ++i) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e1, s));
    printf("{\"bench\": \"cycle_per_launch_graph32\", \"workgroups\": %d, \"us_per_cycle\": %.3f}\n", nb, 1e3 * ms() / K);
    // (c), (d) persistent kernels: with the cycle's memory work and with the barrier alone
    struct V { const char* name; void (*k)(v4f*, v4f*, int, int, unsigned*); };
    const V vs[4] = {{"persistent_xcd_barrier", k_persistent<true, true>}, {"persistent_flat_barrier", k_persistent<false, true>},
                     {"xcd_barrier_alone", k_persistent<true, false>}, {"flat_barrier_alone", k_persistent<false, false>}};
    for (const V& v : vs) {
        for (int rep = 0; rep < 2; ++rep) {  // first run warms up
            CHECK(hipMemsetAsync(bar, 0, 4096, s));
            CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(v.k, dim3(nb), dim3(BLOCK), 0, s, a, b, nb, K, bar);
            CHECK(hipEventRecord(e1, s));
            const float t = ms();
            if (rep) printf("{\"bench\": \"%s\", \"workgroups\": %d, \"us_per_cycle\": %.3f}\n", v.name, nb, 1e3 * t / K);
        }
    }
    // (e) one XCD: W workgroups on the 32 CUs of XCD 0
    for (int W : {32, 64, 128}) {
        struct V1 { const char* name; void (*k)(v4f*, v4f*, int, int, unsigned*, int); };
        const V1 v1[3] = {{"one_xcd_persistent", k_persistent_one_xcd<true, 0>}, {"one_xcd_persistent_f32_bytes", k_persistent_one_xcd<true, 1>},
                          {"one_xcd_barrier_alone", k_persistent_one_xcd<false, 0>}};
        for (const V1& v : v1) {
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemsetAsync(bar, 0, 4096, s));
                CHECK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(v.k, dim3(8 * W), dim3(BLOCK), 0, s, a, b, nb, K / 4, bar, W);
                CHECK(hipEventRecord(e1, s));
                const float t = ms();
                if (rep) printf("{\"bench\": \"%s\", \"virtual_blocks\": %d, \"workgroups_on_the_xcd\": %d, \"us_per_cycle\": %.3f}\n", v.name, nb, W, 1e3 * t / (K / 4));
                fflush(stdout);
            }
        }
    }
    return 0;
}
