// Host-side group helpers used by the proof epilogue (prove.go:185,199-200,212-214,241-269,287-292) and by the
// multi-GPU combine.  Same templates as the device code, compiled for the host: a handful of scalar
// multiplications and additions per proof, which the reference also keeps on the CPU
// (icicle.go:1096-1097,1144-1146,1249-1259,1309-1314).
#pragma once
#include "common.hip.h"

namespace ga {

template <class F>
inline XYZZ<F> host_load_jac(const void* p) {
    Jac<F> j;
    memcpy(&j, p, sizeof(j));
    return from_jac(j);
}
template <class F>
inline void host_store_jac(void* p, const XYZZ<F>& a) {
    Jac<F> j = to_jac(a);
    memcpy(p, &j, sizeof(j));
}
template <class F>
inline XYZZ<F> host_load_affine(const void* p) {
    Affine<F> a;
    memcpy(&a, p, sizeof(a));
    return to_xyzz(a);
}
template <class F>
inline void host_store_affine(void* p, const XYZZ<F>& a) {
    Affine<F> r = to_affine(a);
    memcpy(p, &r, sizeof(r));
}

// scalar: fr element in Montgomery form -> canonical 8 words
template <class FrP>
inline void host_fr_canonical(const void* fr_mont, uint32_t* out8) {
    Fe<FrP> s;
    memcpy(s.l, fr_mont, 32);
    s = from_mont(s);
    memcpy(out8, s.l, 32);
}

// result = sum_j 2^(c*j) W_j  (host; Horner from the top window)
template <class F>
XYZZ<F> host_horner(const XYZZ<F>* W, int nwin, int c) {
    XYZZ<F> acc = xyzz_inf<F>();
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) acc = dbl(acc);
        acc = add(acc, W[w]);
    }
    return acc;
}

// contiguous share `index` of `count` of nwin windows (same split as multigpu.shard_range)
inline void window_share(int nwin, uint32_t index, uint32_t count, int* lo, int* hi) {
    const int base = nwin / (int)count, rem = nwin % (int)count, k = (int)index;
    *lo = k * base + (k < rem ? k : rem);
    *hi = *lo + base + (k < rem ? 1 : 0);
}

// MSM on device + Horner on host -> XYZZ.  Long vectors are split along the point axis (the (point, window) pair space of one
// launch is 2^31): the analogue of msmChunkedG1/G2, icicle.go:362-467.  (win_index, win_count) restrict the accumulation to one
// share of the windows (multi-GPU partition A): the other windows count as zero in the Horner sum, so shares add up.
template <class C, int G>
int host_msm(Ctx* ctx, const void* d_bases, const void* d_scalars, size_t n, bool mont,
             XYZZ<typename GroupField<C, G>::F>* out, uint32_t win_index = 0, uint32_t win_count = 1) {
    typedef typename GroupField<C, G>::F F;
    int c, nwin;
    GA_CHECK(msm_plan<C>(G, n, &c, &nwin));
    int lo = 0, hi = nwin;
    window_share(nwin, win_index, win_count ? win_count : 1, &lo, &hi);
    if (hi > lo && n > 0 && n <= ((size_t)1 << 31) / (size_t)(hi - lo) - 1) {   // one launch sequence: Horner folded into the reduction's host tail
        XYZZ<F> sum;
        GA_CHECK((msm_windows_device<C, G>(ctx, d_bases, d_scalars, n, mont, c, lo, hi, &sum, true)));
        for (int k = 0; k < c * lo; k++) sum = dbl(sum);
        *out = sum;
        return GA_OK;
    }
    std::vector<XYZZ<F>> W(nwin, xyzz_inf<F>());
    if (hi > lo && n > 0) {
        const size_t max_chunk = ((size_t)1 << 31) / (size_t)(hi - lo) - 1;
        std::vector<XYZZ<F>> part(hi - lo);
        for (size_t done = 0; done < n;) {
            const size_t cn = n - done < max_chunk ? n - done : max_chunk;
            GA_CHECK((msm_windows_device<C, G>(ctx, (const char*)d_bases + done * sizeof(Affine<F>), (const char*)d_scalars + done * 32, cn, mont, c,
                                              lo, hi, part.data())));
            for (int w = lo; w < hi; w++) W[w] = add(W[w], part[w - lo]);
            done += cn;
        }
    }
    *out = host_horner(W.data(), nwin, c);
    return GA_OK;
}

}  // namespace ga
