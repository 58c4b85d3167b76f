#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void fill(uint32_t* k, uint32_t* v, uint64_t m, uint32_t nb) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    k[i] = (uint32_t)(x % (nb + 1));
    v[i] = (uint32_t)i;
}

template <class Config>
int run(const char* name, uint32_t* k, uint32_t* k2, uint32_t* v, uint32_t* v2, uint64_t m, int bits, hipStream_t st) {
    size_t tb = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, tb, k, k2, v, v2, m, 0, bits, st)));
    void* tmp; CK(hipMalloc(&tmp, tb + 256));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    CK((rocprim::radix_sort_pairs<Config>(tmp, tb, k, k2, v, v2, m, 0, bits, st)));
    CK(hipStreamSynchronize(st));
    hipEventRecord(a, st);
    for (int r = 0; r < 5; r++) CK((rocprim::radix_sort_pairs<Config>(tmp, tb, k, k2, v, v2, m, 0, bits, st)));
    hipEventRecord(b, st); CK(hipStreamSynchronize(st));
    float ms; hipEventElapsedTime(&ms, a, b);
    // check sortedness
    std::vector<uint32_t> h(1 << 20);
    CK(hipMemcpy(h.data(), k2 + (m / 2), h.size() * 4, hipMemcpyDeviceToHost));
    bool ok = true; for (size_t i = 1; i < h.size(); i++) ok &= h[i - 1] <= h[i];
    printf("%-28s %.3f ms per sort (tmp %zu MiB) sorted=%d\n", name, ms / 5, tb >> 20, (int)ok);
    hipFree(tmp);
    return 0;
}

template <class Config>
int rundb(const char* name, uint32_t* k, uint32_t* k2, uint32_t* v, uint32_t* v2, uint64_t m, int bits, hipStream_t st, uint32_t nb) {
    size_t tb = 0;
    rocprim::double_buffer<uint32_t> dk(k, k2), dv(v, v2);
    CK((rocprim::radix_sort_pairs<Config>(nullptr, tb, dk, dv, m, 0, bits, st)));
    void* tmp; CK(hipMalloc(&tmp, tb + 256));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float tot = 0;
    for (int r = 0; r < 5; r++) {
        fill<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(k, v, m, nb);
        rocprim::double_buffer<uint32_t> ek(k, k2), ev(v, v2);
        hipEventRecord(a, st);
        CK((rocprim::radix_sort_pairs<Config>(tmp, tb, ek, ev, m, 0, bits, st)));
        hipEventRecord(b, st); CK(hipStreamSynchronize(st));
        float ms; hipEventElapsedTime(&ms, a, b); if (r) tot += ms;
        dk = ek;
    }
    std::vector<uint32_t> h(1 << 20);
    CK(hipMemcpy(h.data(), dk.current() + (m / 2), h.size() * 4, hipMemcpyDeviceToHost));
    bool ok = true; for (size_t i = 1; i < h.size(); i++) ok &= h[i - 1] <= h[i];
    printf("DB %-28s %.3f ms per sort (tmp %zu MiB) sorted=%d\n", name, tot / 4, tb >> 20, (int)ok);
    hipFree(tmp);
    return 0;
}
using namespace rocprim;
template <int BS, int IPT, int RB, block_radix_rank_algorithm A = block_radix_rank_algorithm::match>
using OS = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<BS, IPT>, kernel_config<BS, IPT>, RB, A>>;

int main(int argc, char** argv) {
    const uint64_t m = 12ull << 24; const uint32_t nb = 1u << 21; const int bits = 22;
    uint32_t *k, *k2, *v, *v2;
    CK(hipMalloc(&k, m * 4)); CK(hipMalloc(&k2, m * 4)); CK(hipMalloc(&v, m * 4)); CK(hipMalloc(&v2, m * 4));
    hipStream_t st; hipStreamCreate(&st);
    fill<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(k, v, m, nb);
    rundb<default_config>("default", k, k2, v, v2, m, bits, st, nb);
    rundb<OS<1024, 22, 11>>("1024x22 r11", k, k2, v, v2, m, bits, st, nb);
    rundb<OS<1024, 18, 11>>("1024x18 r11", k, k2, v, v2, m, bits, st, nb);
    rundb<OS<1024, 20, 11>>("1024x20 r11", k, k2, v, v2, m, bits, st, nb);
    rundb<OS<1024, 21, 11>>("1024x21 r11", k, k2, v, v2, m, bits, st, nb);
    rundb<OS<768, 22, 11>>("768x22 r11", k, k2, v, v2, m, bits, st, nb);
    rundb<OS<512, 22, 11>>("512x22 r11", k, k2, v, v2, m, bits, st, nb);
    return 0;
}
