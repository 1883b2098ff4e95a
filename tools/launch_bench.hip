// launch_bench.hip -- what does a launch cost as a function of its grid?  (calibration, not product code)
// The sweep of the 100k-variable instance is 2 401 workgroups of 256 threads (9 604 waves) living ~6 us each; its per-block
// timeline (tools/timeline.py) shows 2-5 us of ramp before the chip is full.  How much of the 18 us is the grid itself?
// Back-to-back launches of a kernel that does next to nothing (one 4-byte load + store per thread, LDS allocated like the
// sweep's staging area), for several grid shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LDS_BYTES>
__global__ void k_touch(const int* in, int* out, int n) {
    __shared__ int pad[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (LDS_BYTES > 0) pad[threadIdx.x % (LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1)] = i;
    if (i < n) out[i] = in[i] + (LDS_BYTES > 0 ? pad[0] * 0 : 0);
}

int main() {
    const int N = 1 << 20;
    int *a, *b;
    CHECK(hipMalloc(&a, N * 4)); CHECK(hipMalloc(&b, N * 4));
    CHECK(hipMemset(a, 0, N * 4));
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int K = 2000;
    struct Shape { int wgs, threads; };
    const Shape shapes[] = {{1, 256}, {240, 256}, {600, 256}, {1200, 256}, {1792, 256}, {2048, 256}, {2401, 256}, {4096, 256},
                            {1200, 512}, {600, 1024}, {2401, 128}, {4802, 128}, {9604, 64}};
    for (const Shape& sh : shapes) {
        for (int lds = 0; lds < 2; ++lds) {
            auto launch = [&]() {
                if (lds) hipLaunchKernelGGL((k_touch<16384>), dim3(sh.wgs), dim3(sh.threads), 0, s, a, b, N);
                else hipLaunchKernelGGL((k_touch<0>), dim3(sh.wgs), dim3(sh.threads), 0, s, a, b, N);
            };
            for (int i = 0; i < 200; ++i) launch();
            CHECK(hipEventRecord(e0, s));
            for (int i = 0; i < K; ++i) launch();
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"workgroups\": %d, \"threads\": %d, \"waves\": %d, \"lds_bytes\": %d, \"us_per_launch\": %.3f}\n", sh.wgs, sh.threads,
                   sh.wgs * sh.threads / 64, lds ? 16384 : 0, 1e3 * ms / K);
            fflush(stdout);
        }
    }
    return 0;
}
