// microbench.hip -- calibration of the MI355X numbers DESIGN.md reasons with.
//
// Not part of the product: a stand-alone gfx950 binary that measures, with HIP
// events on one stream,
//   1. the cadence of dependent kernel launches (eager and hipGraph replay), i.e.
//      the floor of "one synchronous Max-Sum cycle = one launch";
//   2. streaming copy bandwidth for a working set that fits the 256 MB Infinity
//      Cache and one that does not;
//   3. the access pattern of the variable side of the sweep: every lane gathers
//      one random 64-byte record and scatters 32 bytes to another random record,
//      again in-cache and out-of-cache;
//   4. dependent-load latency (pointer chase) at the same two sizes.
// Prints one JSON object per measurement.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

__global__ void __launch_bounds__(256) k_empty(int* p) {
    if (p && threadIdx.x == 1024) p[0] = 1;
}

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ in, float4* __restrict__ out,
                                              size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 acc = {0, 0, 0, 0};
    for (; i + 3 * stride < n; i += 4 * stride) {  // 4 independent 16-B loads in flight per lane
        const float4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y;
        acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
    }
    for (; i < n; i += stride) { const float4 a = in[i]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;  // keep the loads alive
}

__global__ void __launch_bounds__(256) k_fill(float4* __restrict__ out, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float4 x = {v, v, v, v};
    for (; i < n; i += stride) out[i] = x;
}

// one 64-B record gathered, 32 B scattered per lane (records of 4 x float4)
__global__ void __launch_bounds__(256) k_gather(const float4* __restrict__ rec, float4* __restrict__ out,
                                                const uint32_t* __restrict__ idx, size_t n_lanes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lanes) return;
    const uint32_t r = idx[i];
    const float4 a = rec[(size_t)r * 4 + 0], b = rec[(size_t)r * 4 + 1];
    const float4 c = rec[(size_t)r * 4 + 2], d = rec[(size_t)r * 4 + 3];
    float4 s0, s1;
    s0.x = a.x + c.x; s0.y = a.y + c.y; s0.z = a.z + c.z; s0.w = a.w + c.w;
    s1.x = b.x + d.x; s1.y = b.y + d.y; s1.z = b.z + d.z; s1.w = b.w + d.w;
    out[(size_t)r * 4 + 0] = s0;
    out[(size_t)r * 4 + 1] = s1;
}

// one 32-B record (a padded D=3 f64 message) gathered per lane, 16 B written linearly:
// the shape of a message gather of the sweep.  Index patterns: random, every record in
// order (dense), every other record in order (half dense) -- calibrates what FETCH_SIZE
// reports for near-sequential gathers.
__global__ void __launch_bounds__(256) k_gather32(const float4* __restrict__ rec, float4* __restrict__ out,
                                                  const uint32_t* __restrict__ idx, size_t n_lanes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lanes) return;
    const uint32_t r = idx[i];
    const float4 a = rec[(size_t)r * 2 + 0], b = rec[(size_t)r * 2 + 1];
    float4 s0;
    s0.x = a.x + b.x; s0.y = a.y + b.y; s0.z = a.z + b.z; s0.w = a.w + b.w;
    out[i] = s0;
}

__global__ void k_chase(const uint32_t* __restrict__ next, uint32_t start, int steps, uint32_t* out) {
    uint32_t p = start;
    for (int i = 0; i < steps; ++i) p = next[(size_t)p * 16];  // one 64-B line per hop
    out[0] = p;
}

static float time_ms(hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"bench\": \"device\", \"name\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n",
           prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);

    // 1. launch cadence ------------------------------------------------------
    for (int blocks : {256, 1200, 2400}) {
        const int n = 2000;
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, st, (int*)nullptr);
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, st, (int*)nullptr);
        CHECK(hipEventRecord(e1, st));
        const float eager = time_ms(st, e0, e1);
        hipGraph_t g;
        hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, st, (int*)nullptr);
        CHECK(hipStreamEndCapture(st, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 4; ++i) CHECK(hipGraphLaunch(ge, st));
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < n / 32; ++i) CHECK(hipGraphLaunch(ge, st));
        CHECK(hipEventRecord(e1, st));
        const float graph = time_ms(st, e0, e1);
        printf("{\"bench\": \"launch_cadence\", \"blocks\": %d, \"eager_us\": %.3f, \"graph32_us\": %.3f}\n",
               blocks, 1e3 * eager / n, 1e3 * graph / (n / 32 * 32));
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
    }

    // 2. streaming copy --------------------------------------------------------
    for (size_t mb : {32, 64, 1024}) {
        const size_t bytes = mb << 20, n = bytes / sizeof(float4);
        float4 *a, *b;
        CHECK(hipMalloc((void**)&a, bytes));
        CHECK(hipMalloc((void**)&b, bytes));
        CHECK(hipMemsetAsync(a, 1, bytes, st));
        const int blocks = 256 * 8, reps = mb >= 1024 ? 20 : 200;
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, a, b, n);
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, a, b, n);
        CHECK(hipEventRecord(e1, st));
        const float ms = time_ms(st, e0, e1);
        printf("{\"bench\": \"copy\", \"mb_each_way\": %zu, \"us_per_launch\": %.2f, \"read_plus_write_GBps\": %.1f}\n",
               mb, 1e3 * ms / reps, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
        float* sink;
        CHECK(hipMalloc((void**)&sink, 64));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, st, a, sink, n);
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, st, a, sink, n);
        CHECK(hipEventRecord(e1, st));
        const float msr = time_ms(st, e0, e1);
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, st, b, n, 1.0f);
        CHECK(hipEventRecord(e1, st));
        const float msw = time_ms(st, e0, e1);
        printf("{\"bench\": \"read_only\", \"mb\": %zu, \"us_per_launch\": %.2f, \"GBps\": %.1f}\n", mb,
               1e3 * msr / reps, (double)bytes * reps / (msr * 1e-3) / 1e9);
        printf("{\"bench\": \"write_only\", \"mb\": %zu, \"us_per_launch\": %.2f, \"GBps\": %.1f}\n", mb,
               1e3 * msw / reps, (double)bytes * reps / (msw * 1e-3) / 1e9);
        CHECK(hipFree(sink));
        CHECK(hipFree(a));
        CHECK(hipFree(b));
    }

    // 3. random 64-B gather + 32-B scatter -------------------------------------------
    for (size_t n_rec : {(size_t)400000, (size_t)6000000, (size_t)16000000}) {
        const size_t bytes = n_rec * 64;
        float4 *rec, *out;
        uint32_t* idx;
        CHECK(hipMalloc((void**)&rec, bytes));
        CHECK(hipMalloc((void**)&out, bytes));
        CHECK(hipMalloc((void**)&idx, n_rec * 4));
        CHECK(hipMemsetAsync(rec, 0, bytes, st));
        std::vector<uint32_t> h(n_rec);
        for (size_t i = 0; i < n_rec; ++i) h[i] = (uint32_t)i;
        uint64_t s = 88172645463325252ull;
        for (size_t i = n_rec - 1; i > 0; --i) {  // Fisher-Yates with xorshift
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            std::swap(h[i], h[s % (i + 1)]);
        }
        CHECK(hipMemcpyAsync(idx, h.data(), n_rec * 4, hipMemcpyHostToDevice, st));
        CHECK(hipStreamSynchronize(st));
        const int blocks = (int)((n_rec + 255) / 256), reps = n_rec > 1000000 ? 20 : 200;
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, st, rec, out, idx, n_rec);
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, st, rec, out, idx, n_rec);
        CHECK(hipEventRecord(e1, st));
        const float ms = time_ms(st, e0, e1);
        printf("{\"bench\": \"gather64_scatter32\", \"records\": %zu, \"mb\": %.1f, \"us_per_launch\": %.2f, "
               "\"useful_GBps\": %.1f}\n",
               n_rec, bytes / 1048576.0, 1e3 * ms / reps, (double)n_rec * (64 + 32 + 4) * reps / (ms * 1e-3) / 1e9);
        CHECK(hipFree(rec));
        CHECK(hipFree(out));
        CHECK(hipFree(idx));
    }

    // 3b. 32-B record gathers: random / dense / half dense ----------------------------
    {
        const size_t n_lanes = 6000000, n_rec = 2 * n_lanes;  // 384 MB of records: beyond the cache
        float4 *rec, *out;
        uint32_t* idx;
        CHECK(hipMalloc((void**)&rec, n_rec * 32));
        CHECK(hipMalloc((void**)&out, n_lanes * 16));
        CHECK(hipMalloc((void**)&idx, n_lanes * 4));
        CHECK(hipMemsetAsync(rec, 0, n_rec * 32, st));
        std::vector<uint32_t> h(n_lanes);
        const char* names[3] = {"random", "dense", "half_dense"};
        for (int pat = 0; pat < 3; ++pat) {
            for (size_t i = 0; i < n_lanes; ++i) h[i] = pat == 1 ? (uint32_t)i : (uint32_t)(2 * i);
            if (pat == 0) {
                uint64_t s = 88172645463325252ull;
                for (size_t i = n_lanes - 1; i > 0; --i) {
                    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                    std::swap(h[i], h[s % (i + 1)]);
                }
            }
            CHECK(hipMemcpyAsync(idx, h.data(), n_lanes * 4, hipMemcpyHostToDevice, st));
            CHECK(hipStreamSynchronize(st));
            // the grid size tells the patterns apart in a counter trace: +0 / +1 / +2 blocks
            const int blocks = (int)((n_lanes + 255) / 256) + pat, reps = 10;
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_gather32, dim3(blocks), dim3(256), 0, st, rec, out, idx, n_lanes);
            CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gather32, dim3(blocks), dim3(256), 0, st, rec, out, idx, n_lanes);
            CHECK(hipEventRecord(e1, st));
            const float ms = time_ms(st, e0, e1);
            printf("{\"bench\": \"gather32_%s\", \"lanes\": %zu, \"grid_blocks\": %d, \"us_per_launch\": %.2f, "
                   "\"payload_GBps\": %.1f}\n",
                   names[pat], n_lanes, blocks, 1e3 * ms / reps, (double)n_lanes * (32 + 16 + 4) * reps / (ms * 1e-3) / 1e9);
        }
        CHECK(hipFree(rec));
        CHECK(hipFree(out));
        CHECK(hipFree(idx));
    }

    // 4. pointer chase -----------------------------------------------------------------
    for (size_t mb : {2, 64, 2048}) {
        const size_t lines = (mb << 20) / 64;
        uint32_t *next, *out;
        CHECK(hipMalloc((void**)&next, lines * 64));
        CHECK(hipMalloc((void**)&out, 64));
        std::vector<uint32_t> perm(lines), h(lines * 16, 0);
        for (size_t i = 0; i < lines; ++i) perm[i] = (uint32_t)i;
        uint64_t s = 1234567ull;
        for (size_t i = lines - 1; i > 0; --i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            std::swap(perm[i], perm[s % (i + 1)]);
        }
        for (size_t i = 0; i < lines; ++i) h[(size_t)perm[i] * 16] = perm[(i + 1) % lines];  // one cycle
        CHECK(hipMemcpyAsync(next, h.data(), lines * 64, hipMemcpyHostToDevice, st));
        CHECK(hipStreamSynchronize(st));
        const int steps = 20000;
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, st, next, perm[0], 2000, out);
        CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, st, next, perm[0], steps, out);
        CHECK(hipEventRecord(e1, st));
        const float ms = time_ms(st, e0, e1);
        printf("{\"bench\": \"pointer_chase\", \"mb\": %zu, \"ns_per_hop\": %.1f}\n", mb, 1e6 * ms / steps);
        CHECK(hipFree(next));
        CHECK(hipFree(out));
    }
    return 0;
}
