// TEST INFRASTRUCTURE ONLY -- CPU stand-in for the one rocPRIM entry point the product uses (radix_sort_pairs on
// double buffers and on separate input / output arrays); see tests/emu/include/hip/hip_runtime.h for what the emulation is and is not.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace rocprim {

struct default_config {};
enum class block_radix_rank_algorithm { basic, basic_memoize, match };
template <unsigned BlockSize, unsigned ItemsPerThread> struct kernel_config {};
template <class H, class S, unsigned RadixBits, block_radix_rank_algorithm A> struct radix_sort_onesweep_config {};
template <class A = default_config, class B = default_config, class C = default_config, size_t L = 1024 * 1024> struct radix_sort_config {};

template <class T>
class double_buffer {
    T* b_[2];
    int sel_ = 0;

public:
    double_buffer(T* current, T* alternate) : b_{current, alternate} {}
    T* current() const { return b_[sel_]; }
    T* alternate() const { return b_[sel_ ^ 1]; }
    void swap() { sel_ ^= 1; }
};

template <class Config = default_config, class K, class V>
hipError_t radix_sort_pairs(void* tmp, size_t& tmp_bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, unsigned begin_bit = 0,
                            unsigned end_bit = 8 * sizeof(K), hipStream_t = nullptr, bool = false) {
    if (tmp == nullptr) {
        tmp_bytes = 16;
        return hipSuccess;
    }
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    const K mask = end_bit >= sizeof(K) * 8 ? ~K(0) : (K)((K(1) << end_bit) - 1);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ((kin[a] & mask) >> begin_bit) < ((kin[b] & mask) >> begin_bit); });
    for (size_t i = 0; i < n; i++) {
        kout[i] = kin[idx[i]];
        vout[i] = vin[idx[i]];
    }
    return hipSuccess;
}

template <class Config = default_config, class K, class V>
hipError_t radix_sort_pairs(void* tmp, size_t& tmp_bytes, double_buffer<K>& keys, double_buffer<V>& vals, size_t n, unsigned begin_bit = 0,
                            unsigned end_bit = 8 * sizeof(K), hipStream_t = nullptr, bool = false) {
    if (tmp == nullptr) {
        tmp_bytes = 16;
        return hipSuccess;
    }
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    const K* kin = keys.current();
    const V* vin = vals.current();
    const K mask = end_bit >= sizeof(K) * 8 ? ~K(0) : (K)((K(1) << end_bit) - 1);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ((kin[a] & mask) >> begin_bit) < ((kin[b] & mask) >> begin_bit); });
    K* kout = keys.alternate();
    V* vout = vals.alternate();
    for (size_t i = 0; i < n; i++) {
        kout[i] = kin[idx[i]];
        vout[i] = vin[idx[i]];
    }
    keys.swap();
    vals.swap();
    return hipSuccess;
}

}  // namespace rocprim
