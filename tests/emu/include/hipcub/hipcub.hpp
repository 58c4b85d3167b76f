// TEST INFRASTRUCTURE ONLY -- CPU stand-ins for the two hipCUB device primitives the product uses
// (see tests/emu/include/hip/hip_runtime.h for what the emulation is and is not).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace hipcub {

struct DeviceRadixSort {
    template <class K, class V, class N>
    static hipError_t SortPairs(void* tmp, size_t& tmp_bytes, const K* kin, K* kout, const V* vin, V* vout, N n, int begin_bit = 0,
                                int end_bit = sizeof(K) * 8, hipStream_t = nullptr) {
        if (tmp == nullptr) {
            tmp_bytes = 16;
            return hipSuccess;
        }
        std::vector<size_t> idx((size_t)n);
        std::iota(idx.begin(), idx.end(), 0);
        const K mask = end_bit >= (int)sizeof(K) * 8 ? ~K(0) : (K)((K(1) << end_bit) - 1);
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
            return ((kin[a] & mask) >> begin_bit) < ((kin[b] & mask) >> begin_bit);
        });
        for (size_t i = 0; i < (size_t)n; i++) {
            kout[i] = kin[idx[i]];
            vout[i] = vin[idx[i]];
        }
        return hipSuccess;
    }
};

struct DeviceScan {
    template <class In, class Out>
    static hipError_t ExclusiveSum(void* tmp, size_t& tmp_bytes, In in, Out out, int n, hipStream_t = nullptr) {
        if (tmp == nullptr) {
            tmp_bytes = 16;
            return hipSuccess;
        }
        typename std::remove_reference<decltype(out[0])>::type acc = 0;
        for (int i = 0; i < n; i++) {
            auto v = in[i];
            out[i] = acc;
            acc += v;
        }
        return hipSuccess;
    }
};

}  // namespace hipcub
