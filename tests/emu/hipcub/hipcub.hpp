// TEST ONLY: the few hipCUB device primitives amaxsum.hip uses, as serial host code, for the
// emulated engine build of the CPU tests (tests/emu/build_emu.py).  Same calling convention:
// a first call with a null temporary buffer returns the size, the second does the work.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <numeric>
#include <type_traits>
#include <vector>

#include "../hip/hip_runtime.h"

namespace hipcub {

template <typename To>
struct CastOp {
    template <typename From>
    To operator()(const From& x) const { return (To)x; }
};

template <typename V>
struct CountingInputIterator {
    V base;
    explicit CountingInputIterator(V b) : base(b) {}
    V operator[](std::ptrdiff_t k) const { return base + (V)k; }
};

template <typename V, typename Op, typename It>
struct TransformInputIterator {
    It it;
    Op op;
    TransformInputIterator(It i, Op o) : it(i), op(o) {}
    V operator[](std::ptrdiff_t k) const { return op(it[k]); }
};

struct DeviceScan {
    template <typename In, typename Out>
    static hipError_t ExclusiveSum(void* temp, size_t& bytes, In in, Out out, int n, hipStream_t = nullptr) {
        if (!temp) {
            bytes = 16;
            return hipSuccess;
        }
        auto acc = decltype(in[0] + in[0])(0);
        for (int i = 0; i < n; ++i) {
            const auto x = in[i];  // (in and out may alias)
            out[i] = acc;
            acc += x;
        }
        return hipSuccess;
    }
};

struct DeviceRadixSort {
    template <typename K, typename V>
    static hipError_t SortPairs(void* temp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, int n,
                                int begin_bit = 0, int end_bit = (int)sizeof(K) * 8, hipStream_t = nullptr) {
        if (!temp) {
            bytes = 16;
            return hipSuccess;
        }
        using U = typename std::make_unsigned<K>::type;
        const int nb = end_bit - begin_bit;
        const U mask = nb >= (int)sizeof(K) * 8 ? ~(U)0 : (U)(((U)1 << nb) - 1);
        std::vector<int> idx(n);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
            return (((U)kin[a] >> begin_bit) & mask) < (((U)kin[b] >> begin_bit) & mask);
        });
        for (int i = 0; i < n; ++i) {
            kout[i] = kin[idx[i]];
            vout[i] = vin[idx[i]];
        }
        return hipSuccess;
    }
};

}  // namespace hipcub
