// TEST INFRASTRUCTURE ONLY -- a tiny serial stand-in for <hip/hip_runtime.h>.
//
// Compiling pydcop_amd/csrc/{engine.hip,layout.cpp} with g++ and this directory
// first on the include path gives `tests/emu/_build/libmaxsum_emu.so`: the very
// same engine source (no #ifdef in the product code) whose "device" is host
// memory and whose kernel launches run block after block, each thread of a block
// as a ucontext fiber so that __syncthreads() works.  It lets the host logic
// (layout, permutations, launch sequencing, C-ABI) and the kernels' index
// arithmetic be checked against the oracle in the GPU-less build container.
// It is never loaded by pydcop_amd (which only opens csrc/libmaxsum_hip.so and
// fails without a GPU) and is not a fallback: see tests/test_emu_engine.py.
#pragma once
#include <ucontext.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <tuple>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotSupported = 801, hipErrorInvalidValue = 1 };
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void* hipGraphNode_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1, hipStreamCaptureModeThreadLocal = 1 };
struct hipDeviceProp_t {
    char gcnArchName[64];
    int multiProcessorCount = 256;
};

inline const char* hipGetErrorString(hipError_t) { return "emulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) {  // $EMU_HIP_DEVICES pretend-GPUs (tests of the multi-device path)
    const char* e = std::getenv("EMU_HIP_DEVICES");
    *n = e ? std::atoi(e) : 1;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    std::strcpy(p->gcnArchName, "gfx950:emulated-on-host");
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {  // "a device" with 64 GiB, half of it free
    *total_b = (size_t)64 << 30;
    *free_b = (size_t)32 << 30;
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    std::memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (void*)1; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
// no graph capture in the emulation: the engine falls back to plain launches
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) {
    return hipErrorNotSupported;
}
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

// ---- block execution with fibers ------------------------------------------
// A barrier (and every emulated cross-lane read) switches through all fibers of the block.
// glibc's swapcontext makes a system call per switch (signal mask); on x86-64 the switch
// below saves the callee-saved registers and the stack pointer only.
namespace hipemu {
#if defined(__x86_64__)
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .weak hipemu_switch
    .hidden hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch, .-hipemu_switch
)");
struct Context {
    void* sp = nullptr;
};
inline void switch_context(Context* from, Context* to) { hipemu_switch(&from->sp, to->sp); }
inline void make_context(Context* c, char* stack, size_t size, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;          // the entry function never returns
    *--sp = (void*)entry;     // popped by the first switch's ret: rsp = top - 8 at entry (ABI)
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    c->sp = sp;
}
#else
struct Context {
    ucontext_t uc;
};
inline void switch_context(Context* from, Context* to) { swapcontext(&from->uc, &to->uc); }
inline void make_context(Context* c, char* stack, size_t size, void (*entry)()) {
    getcontext(&c->uc);
    c->uc.uc_stack.ss_sp = stack;
    c->uc.uc_stack.ss_size = size;
    c->uc.uc_link = nullptr;
    makecontext(&c->uc, entry, 0);
}
#endif
struct Fiber {
    Context ctx;
    std::vector<char> stack;
    bool done = false;
};
struct BlockRun {
    Context sched;
    std::vector<Fiber> fibers;
    int current = 0;
    void (*body)(void*) = nullptr;
    void* arg = nullptr;
};
inline thread_local BlockRun* g_run = nullptr;

inline void trampoline() {
    BlockRun* r = g_run;
    Fiber& f = r->fibers[r->current];
    r->body(r->arg);
    f.done = true;
    for (;;) switch_context(&f.ctx, &r->sched);
}
inline void yield_barrier() {  // __syncthreads(): back to the scheduler
    BlockRun* r = g_run;
    switch_context(&r->fibers[r->current].ctx, &r->sched);
}
inline void run_block(unsigned nthreads, void (*body)(void*), void* arg) {
    static thread_local BlockRun run;
    run.body = body;
    run.arg = arg;
    if (run.fibers.size() < nthreads) run.fibers.resize(nthreads);
    g_run = &run;
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = run.fibers[t];
        if (f.stack.empty()) f.stack.resize(256 * 1024);
        f.done = false;
        make_context(&f.ctx, f.stack.data(), f.stack.size(), trampoline);
    }
    bool alive = true;
    while (alive) {  // one pass == everything up to the next barrier
        alive = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = run.fibers[t];
            if (f.done) continue;
            run.current = (int)t;
            threadIdx = dim3(t);
            switch_context(&run.sched, &f.ctx);
            if (!f.done) alive = true;
        }
    }
}
template <typename K, typename... A>
struct Call {
    K k;
    std::tuple<A...> args;
    static void invoke(void* self) {
        Call* c = (Call*)self;
        std::apply(c->k, c->args);
    }
};
}  // namespace hipemu

inline void __syncthreads() { hipemu::yield_barrier(); }

// Cross-lane read (wave = 64 lanes): publish, barrier, read, barrier.  Valid as long
// as all lanes of a (sub-)wave group reach the same __shfl together, which the
// kernels guarantee (group-uniform control flow around cross-lane reads).
namespace hipemu {
inline thread_local unsigned long long g_xchg[1024];
}
template <typename V>
inline V __shfl(V var, int src_lane, int width = 64) {
    static_assert(sizeof(V) <= 8, "emulated __shfl moves at most 8 bytes");
    const unsigned t = threadIdx.x;
    unsigned long long bits = 0;
    std::memcpy(&bits, &var, sizeof(V));
    hipemu::g_xchg[t] = bits;
    hipemu::yield_barrier();
    const unsigned lane = t % 64, wave_base = t - lane;
    const unsigned src = wave_base + (lane / width) * width + ((unsigned)src_lane % (unsigned)width);
    V out;
    bits = hipemu::g_xchg[src];
    std::memcpy(&out, &bits, sizeof(V));
    hipemu::yield_barrier();
    return out;
}

// DPP moves and the gfx950 permlane swaps the wave reductions of kernels.h use (wave_min_to_lane63,
// wave_min4): the lane-selection rules of the ISA, so that the CPU suite runs the product's reduction and
// not a stand-in.  old == what a lane keeps when its row is masked off or the control gives it no source.
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)bank_mask;
    (void)bound_ctrl;
    const unsigned t = threadIdx.x, lane = t % 64, base = t - lane, row = lane / 16, i = lane % 16;
    hipemu::g_xchg[t] = (unsigned long long)(unsigned)src;
    hipemu::yield_barrier();
    int from = -1;
    if (ctrl < 0x100) from = (int)((lane & ~3u) + (((unsigned)ctrl >> (2 * (lane & 3))) & 3u));  // quad_perm
    else if (ctrl == 0x140) from = (int)(row * 16 + 15 - i);                                        // row_mirror
    else if (ctrl == 0x141) from = (int)((lane & ~7u) + 7 - (lane & 7));                            // row_half_mirror
    else if (ctrl == 0x142) from = row > 0 ? (int)(row * 16 - 1) : -1;     // row_bcast15: lane 15 of the previous row
    else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;                     // row_bcast31: lane 31 to rows 2, 3
    else if (ctrl >= 0x150 && ctrl < 0x160) from = (int)(row * 16 + (unsigned)(ctrl - 0x150));     // row_newbcast:n
    else if (ctrl > 0x120 && ctrl < 0x130) from = (int)(row * 16 + ((i + 16 - (unsigned)(ctrl - 0x120)) & 15));  // row_ror:n
    else abort();
    int out = old;
    if (from >= 0 && ((row_mask >> row) & 1)) out = (int)(unsigned)hipemu::g_xchg[base + (unsigned)from];
    hipemu::yield_barrier();
    return out;
}
// (clang's ext_vector_type spelled for g++: every vector type of the engine sources has 4-byte elements;
// the non-temporal builtins are plain accesses here)
#define ext_vector_type(n) vector_size(4 * (n))
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
typedef unsigned int hipemu_swap2 __attribute__((vector_size(8)));
// v_permlane16_swap: the odd rows (16 lanes each) of the first operand trade places with the even rows of
// the second; v_permlane32_swap: the upper half of the first with the lower half of the second.
inline hipemu_swap2 hipemu_permlane_swap(unsigned int a, unsigned int b, bool halves) {
    const unsigned t = threadIdx.x, lane = t % 64, base = t - lane;
    hipemu::g_xchg[t] = ((unsigned long long)b << 32) | a;
    hipemu::yield_barrier();
    hipemu_swap2 r = {a, b};
    const unsigned span = halves ? 32 : 16, blk = lane / span, i = lane % span;
    if (blk & 1) {  // odd row / upper half of the first operand <- even row / lower half of the second
        r[0] = (unsigned)(hipemu::g_xchg[base + (blk - 1) * span + i] >> 32);
    } else {        // even row / lower half of the second operand <- odd row / upper half of the first
        r[1] = (unsigned)(hipemu::g_xchg[base + (blk + 1) * span + i] & 0xffffffffull);
    }
    hipemu::yield_barrier();
    return r;
}
inline hipemu_swap2 __builtin_amdgcn_permlane16_swap(unsigned int a, unsigned int b, bool, bool) { return hipemu_permlane_swap(a, b, false); }
inline hipemu_swap2 __builtin_amdgcn_permlane32_swap(unsigned int a, unsigned int b, bool, bool) { return hipemu_permlane_swap(a, b, true); }
inline int __double2loint(double x) { unsigned long long b; std::memcpy(&b, &x, 8); return (int)(unsigned)(b & 0xffffffffull); }
inline int __double2hiint(double x) { unsigned long long b; std::memcpy(&b, &x, 8); return (int)(unsigned)(b >> 32); }
inline double __hiloint2double(int hi, int lo) {
    const unsigned long long b = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    double x; std::memcpy(&x, &b, 8); return x;
}
inline int __float_as_int(float x) { int b; std::memcpy(&b, &x, 4); return b; }
inline float __int_as_float(int b) { float x; std::memcpy(&x, &b, 4); return x; }

// 64-bit mask of the lanes of the wave whose predicate is non-zero
inline unsigned long long __ballot(int pred) {
    const unsigned t = threadIdx.x;
    hipemu::g_xchg[t] = pred ? 1ull : 0ull;
    hipemu::yield_barrier();
    const unsigned wave_base = t - t % 64;
    unsigned long long mask = 0;
    for (unsigned l = 0; l < 64 && wave_base + l < blockDim.x; ++l)
        if (hipemu::g_xchg[wave_base + l]) mask |= 1ull << l;
    hipemu::yield_barrier();
    return mask;
}

template <typename U>
inline U atomicMin(U* addr, U val) {  // fibers run one at a time: plain read-modify-write
    const U old = *addr;
    if (val < old) *addr = val;
    return old;
}

namespace hipemu {
inline std::mutex& launch_mutex() {
    static std::mutex m;
    return m;
}
}  // namespace hipemu

template <typename K, typename... A>
inline void hipemu_launch(K kernel, dim3 grid, dim3 block, A... args) {
    hipemu::Call<K, A...> call{kernel, std::tuple<A...>(args...)};
    // __shared__ arrays are plain statics here: one emulated kernel at a time per process
    // (engines of several "ranks" may be driven from parallel host threads)
    std::lock_guard<std::mutex> one_kernel(hipemu::launch_mutex());
    gridDim = grid;
    blockDim = block;
    for (unsigned b = 0; b < grid.x; ++b) {
        blockIdx = dim3(b);
        hipemu::run_block(block.x, &hipemu::Call<K, A...>::invoke, &call);
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu_launch(kernel, grid, block, __VA_ARGS__)
// dynamic LDS (HIP: `extern __shared__ type var[];`): one static area, one emulated kernel at a time
namespace hipemu {
alignas(16) inline unsigned char g_dynamic_lds[160 * 1024];
}
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::g_dynamic_lds;

inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_wave_barrier() { hipemu::yield_barrier(); }  // let the other lanes catch up
// advances by 10 ms (100 MHz ticks) per reading: a kernel polling for something that never
// comes runs into its time limit at once instead of hanging the test
inline unsigned long long wall_clock64() {
    static thread_local unsigned long long t = 0;
    return t += 1000000ull;
}
#define __ATOMIC_RELAXED_HIPEMU 0
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
// "IPC" between the rank threads of one test process: the handle is the pointer itself
#define HIP_IPC_HANDLE_SIZE 64
#define hipIpcMemLazyEnablePeerAccess 1
struct hipIpcMemHandle_t { char reserved[HIP_IPC_HANDLE_SIZE]; };
inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
    std::memset(h, 0, sizeof(*h));
    const long long pid = (long long)getpid();
    std::memcpy(h->reserved, &p, sizeof(p));
    std::memcpy(h->reserved + 16, &pid, sizeof(pid));
    return hipSuccess;
}
inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) {
    long long pid = 0;
    std::memcpy(&pid, h.reserved + 16, sizeof(pid));
    if (pid != (long long)getpid()) return hipErrorNotSupported;  // host memory of another process
    std::memcpy(p, h.reserved, sizeof(*p));
    return hipSuccess;
}
inline hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }
template <typename U>
inline U __hip_atomic_load(const U* p, int, int) { return *p; }
template <typename U>
inline void __hip_atomic_store(U* p, U v, int, int) { *p = v; }
inline void __builtin_amdgcn_s_sleep(int) {}
#define __builtin_amdgcn_fence(...) ((void)0)
inline void __builtin_amdgcn_s_barrier() { hipemu::yield_barrier(); }
inline void __threadfence() {}
template <typename U>
inline U atomicOr(U* addr, U val) { const U old = *addr; *addr = old | val; return old; }
template <typename U>
inline U atomicAdd(U* addr, U val) { const U old = *addr; *addr = old + val; return old; }

// ---- gfx950 builtins the kernels use -------------------------------------------
inline int __builtin_amdgcn_readfirstlane(int x) { return x; }
inline void __builtin_amdgcn_s_setprio(int) {}  // wave priority: nothing to emulate  // callers pass wave-uniform values
