// Groth16 prover core: device-resident proving key + one-call proof from the solver's output.
//
// Mirrors backend/groth16/bn254/prove.go:130-315 (CPU) and backend/accelerated/icicle/groth16/bn254/icicle.go:784-1360
// (GPU):  computeH -> filter wire values -> 4 G1 MSMs + 1 G2 MSM -> host epilogue with the caller-supplied randomness
// (r, s); BSB22 commitments are the two extra MSMs per commitment of ga_g16_commit (prove.go:84,114) plus the K filter.  The host side is C++ because the reference's host side is compiled Go and no Go
// toolchain exists in the build image (INTEGRATION.md shows the cgo binding that calls this file's two entry points).
#include <algorithm>
#include <string>
#include <thread>

#include "hostops.cuh"

namespace ga {

// dst[idx[i]] = src[i] for elements of `chunks` 16-byte pieces (building the wire-indexed base arrays at pin time)
static __global__ void g16_scatter_points_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, const uint32_t* __restrict__ idx,
                                                 uint64_t n, uint32_t chunks) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t i = t / chunks, k = t % chunks;
    if (i >= n) return;
    dst[(uint64_t)idx[i] * chunks + k] = src[i * chunks + k];
}

struct G16Pk {
    Ctx* ctx = nullptr;
    int curve = 0;
    uint64_t n = 0;            // domain cardinality
    uint64_t nb_wires = 0;
    Domain* dom = nullptr;
    void *d_a = nullptr, *d_b = nullptr, *d_z = nullptr, *d_k = nullptr, *d_b2 = nullptr;
    uint64_t len_a = 0, len_b = 0, len_z = 0, len_k = 0, len_b2 = 0;
    uint32_t *d_idx_a = nullptr, *d_idx_b = nullptr;   // wire indices kept for the A / B MSMs (prove.go:147-168)
    uint32_t* d_idx_k = nullptr;   // wire indices feeding the K MSM when committed wires are left out (prove.go:231-235); null = W[nbPublic:]
    uint64_t len_k_remove = 0;
    std::vector<void*> d_ck_basis, d_ck_sigma;   // pinned pedersen keys (setup.go:260-287, icicle.go:231-261)
    std::vector<uint64_t> ck_len;
    // precomputed window-multiple tables (msm.cuh): d_a.. then point to windows x len points and c_* is the window width
    bool tables = false;
    int c_a = 0, c_b = 0, c_z = 0, c_k = 0;
    // Wire-indexed tables: when a base vector covers (almost) every wire, its table is laid out by WIRE id with (0,0) at the
    // wires it lacks (infinity entries are skipped by the bucket kernel), so that the digit extraction + radix sort of the whole
    // witness is done ONCE and shared by the A, B (G1 and G2) and K MSMs instead of once per filtered copy of the witness.
    bool share_a = false, share_b = false, share_k = false;
    int c_w = 0;
    // multi-GPU partition B: this key holds slice [off, off+len) of every base vector (ga_g16_key.shard_index/count)
    uint32_t shard_index = 0, shard_count = 1;
    uint64_t off_k = 0, off_z = 0, full_len_k = 0;
    std::vector<uint8_t> alpha1, beta1, delta1, beta2, delta2;   // affine images (host)
};

static int upload(Ctx* ctx, const void* src, size_t bytes, void** dst) {
    *dst = nullptr;
    hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
    if (e != hipSuccess) {
        set_error("proving key upload: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return GA_ERR_NOMEM;
    }
    if (bytes) GA_HIP_CHECK(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GA_OK;
}

static void pk_free(G16Pk* pk) {
    if (!pk) return;
    hipFree(pk->d_a);
    hipFree(pk->d_b);
    hipFree(pk->d_z);
    hipFree(pk->d_k);
    hipFree(pk->d_b2);
    hipFree(pk->d_idx_a);
    hipFree(pk->d_idx_b);
    hipFree(pk->d_idx_k);
    for (void* p : pk->d_ck_basis) hipFree(p);
    for (void* p : pk->d_ck_sigma) hipFree(p);
    if (pk->dom) ntt_domain_delete(pk->dom);
    delete pk;
}

template <class C>
static int pk_create(Ctx* ctx, const ga_g16_key* key, G16Pk** out) {
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    const size_t s1 = sizeof(Affine<F1>), s2 = sizeof(Affine<F2>);
    if (key->len_z + 1 != key->domain_cardinality) {
        set_error("proving key: len(G1.Z)=%llu but domain cardinality is %llu (expected n-1, setup.go:248-249)",
                  (unsigned long long)key->len_z, (unsigned long long)key->domain_cardinality);
        return GA_ERR_INVALID;
    }
    if (key->len_a + key->nb_infinity_a != key->nb_wires || key->len_b + key->nb_infinity_b != key->nb_wires ||
        key->len_b2 != key->len_b) {
        set_error("proving key: len(A)+NbInfinityA, len(B)+NbInfinityB must equal nbWires and len(G2.B)==len(G1.B)");
        return GA_ERR_INVALID;
    }
    G16Pk* pk = new G16Pk();
    pk->ctx = ctx;
    pk->curve = C::ID;
    pk->n = key->domain_cardinality;
    pk->nb_wires = key->nb_wires;
    pk->shard_count = key->shard_count ? key->shard_count : 1;
    pk->shard_index = key->shard_index;
    if (pk->shard_index >= pk->shard_count) {
        set_error("proving key: shard_index %u >= shard_count %u", pk->shard_index, pk->shard_count);
        delete pk;
        return GA_ERR_INVALID;
    }
    auto slice = [&](uint64_t len, uint64_t* lo, uint64_t* cnt) {   // same split as gnark_amd/multigpu.py shard_range
        uint64_t base = len / pk->shard_count, rem = len % pk->shard_count, k = pk->shard_index;
        *lo = k * base + (k < rem ? k : rem);
        *cnt = base + (k < rem ? 1 : 0);
    };
    uint64_t lo_a, lo_b, lo_z, lo_k;
    slice(key->len_a, &lo_a, &pk->len_a);
    slice(key->len_b, &lo_b, &pk->len_b);
    slice(key->len_z, &lo_z, &pk->len_z);
    slice(key->len_k, &lo_k, &pk->len_k);
    pk->len_b2 = pk->len_b;
    pk->off_k = lo_k;
    pk->off_z = lo_z;
    pk->full_len_k = key->len_k;
    int rc = ntt_domain_new<C>(ctx, pk->n, &pk->dom);
    if (rc == GA_OK) rc = upload(ctx, (const char*)key->g1_a + lo_a * s1, pk->len_a * s1, &pk->d_a);
    if (rc == GA_OK) rc = upload(ctx, (const char*)key->g1_b + lo_b * s1, pk->len_b * s1, &pk->d_b);
    if (rc == GA_OK) rc = upload(ctx, (const char*)key->g1_z + lo_z * s1, pk->len_z * s1, &pk->d_z);
    if (rc == GA_OK) rc = upload(ctx, (const char*)key->g1_k + lo_k * s1, pk->len_k * s1, &pk->d_k);
    if (rc == GA_OK) rc = upload(ctx, (const char*)key->g2_b + lo_b * s2, pk->len_b2 * s2, &pk->d_b2);
    std::vector<uint32_t> ia, ib;
    if (rc == GA_OK) {
        ia.reserve(key->len_a);
        ib.reserve(key->len_b);
        for (uint64_t i = 0; i < key->nb_wires; i++) {
            if (!key->infinity_a[i]) ia.push_back((uint32_t)i);
            if (!key->infinity_b[i]) ib.push_back((uint32_t)i);
        }
        if (ia.size() != key->len_a || ib.size() != key->len_b) {
            set_error("proving key: InfinityA/B masks disagree with len(A)/len(B)");
            rc = GA_ERR_INVALID;
        }
    }
    if (rc == GA_OK) rc = upload(ctx, ia.data() + lo_a, pk->len_a * 4, (void**)&pk->d_idx_a);
    if (rc == GA_OK) rc = upload(ctx, ib.data() + lo_b, pk->len_b * 4, (void**)&pk->d_idx_b);
    // K filter with commitments: wireValues[nbPublic:] minus the private committed and commitment wires (prove.go:231-235)
    std::vector<uint32_t> ik;
    pk->len_k_remove = key->len_k_remove;
    if (rc == GA_OK && key->len_k_remove) {
        if (!key->k_remove || key->len_k + key->len_k_remove > key->nb_wires) {
            set_error("proving key: k_remove missing or len(K)+len(k_remove) > nbWires");
            rc = GA_ERR_INVALID;
        } else {
            const uint64_t nb_public = key->nb_wires - key->len_k - key->len_k_remove;
            ik.reserve(key->len_k);
            uint64_t j = 0;
            bool ok = true;
            for (uint64_t i = 0; i < key->len_k_remove; i++)
                ok = ok && key->k_remove[i] >= nb_public && key->k_remove[i] < key->nb_wires && (i == 0 || key->k_remove[i] > key->k_remove[i - 1]);
            for (uint64_t i = nb_public; ok && i < key->nb_wires; i++) {
                if (j < key->len_k_remove && key->k_remove[j] == i) j++;
                else ik.push_back((uint32_t)i);
            }
            if (!ok || ik.size() != key->len_k) {
                set_error("proving key: k_remove must be strictly increasing wire ids in [nbPublic, nbWires)");
                rc = GA_ERR_INVALID;
            }
        }
        if (rc == GA_OK) rc = upload(ctx, ik.data() + lo_k, pk->len_k * 4, (void**)&pk->d_idx_k);
    }
    for (uint32_t i = 0; rc == GA_OK && i < key->nb_commitments; i++) {
        if (!key->ck_basis || !key->ck_basis_exp_sigma || !key->ck_len) {
            set_error("proving key: nb_commitments > 0 but the commitment key arrays are null");
            rc = GA_ERR_INVALID;
            break;
        }
        void *db = nullptr, *ds = nullptr;
        rc = upload(ctx, key->ck_basis[i], key->ck_len[i] * s1, &db);
        if (rc == GA_OK) rc = upload(ctx, key->ck_basis_exp_sigma[i], key->ck_len[i] * s1, &ds);
        pk->d_ck_basis.push_back(db);
        pk->d_ck_sigma.push_back(ds);
        pk->ck_len.push_back(key->ck_len[i]);
    }
    if (rc == GA_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {   // no host pointer survives this call
        set_error("proving key upload: stream synchronize failed");
        rc = GA_ERR_HIP;
    }
    // ---- optional precomputation: [2^(c*w)]P for every window (one shared bucket set per MSM afterwards) ----------
    if (rc == GA_OK && key->precompute >= 0) {
        int nw;
        const size_t t1 = msm_table_point_bytes<C, GA_G1>(), t2 = msm_table_point_bytes<C, GA_G2>();
        // share the witness sort between the vectors that cover at least GA_G16_SHARE_MIN_PCT % of the wires (default 90: a
        // sparse vector would make the lanes of the bucket kernel idle on its missing wires, and waste table memory)
        int share_pct = 90;
        if (const char* e = getenv("GA_G16_SHARE_MIN_PCT")) share_pct = atoi(e);
        auto dense = [&](uint64_t len) {
            return pk->shard_count == 1 && pk->nb_wires < (1ull << 27) && len > 0 && (double)len * 100.0 >= (double)pk->nb_wires * share_pct;
        };
        pk->share_a = dense(pk->len_a);
        pk->share_b = dense(pk->len_b);
        pk->share_k = dense(pk->len_k);
        msm_plan_table<C>(pk->nb_wires, &pk->c_w, &nw);
        const uint64_t wide = (uint64_t)nw * pk->nb_wires;
        msm_plan_table<C>(pk->len_a, &pk->c_a, &nw);
        uint64_t need = pk->share_a ? wide * t1 : (uint64_t)nw * pk->len_a * t1;
        msm_plan_table<C>(pk->len_b, &pk->c_b, &nw);
        need += pk->share_b ? wide * (t1 + t2) : (uint64_t)nw * pk->len_b * (t1 + t2);
        msm_plan_table<C>(pk->len_z, &pk->c_z, &nw);
        need += (uint64_t)nw * pk->len_z * t1;
        msm_plan_table<C>(pk->len_k, &pk->c_k, &nw);
        need += pk->share_k ? wide * t1 : (uint64_t)nw * pk->len_k * t1;
        size_t free_b = 0, total_b = 0;
        hipMemGetInfo(&free_b, &total_b);
        // leave room for the per-proof scratch (~0.6 KB per constraint measured) and some slack
        const bool fits = (double)need + (double)pk->n * 1024.0 < 0.85 * (double)free_b;
        if (key->precompute > 0 || fits) {
            auto make = [&](void** slot, uint64_t len, int c, size_t psz, auto build) -> int {
                if (len == 0) return GA_OK;
                const int nwin = C::FrP::BITS / c + 1;
                void* t = nullptr;
                if (hipMalloc(&t, (uint64_t)nwin * len * psz) != hipSuccess) {
                    set_error("proving key: hipMalloc of a %llu-byte window table failed", (unsigned long long)((uint64_t)nwin * len * psz));
                    return GA_ERR_NOMEM;
                }
                int r = build(*slot, len, c, t);
                if (r == GA_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) r = GA_ERR_HIP;
                hipFree(*slot);
                *slot = t;
                return r;
            };
            auto b1 = [&](const void* src, uint64_t len, int c, void* t) { return msm_table_build<C, GA_G1>(ctx, src, len, c, t); };
            auto b2 = [&](const void* src, uint64_t len, int c, void* t) { return msm_table_build<C, GA_G2>(ctx, src, len, c, t); };
            // compact base array -> wire-indexed array with (0,0) at the missing wires
            auto widen = [&](void** slot, uint64_t len, const uint32_t* d_idx, size_t psz) -> int {
                void* wide_arr = nullptr;
                if (hipMalloc(&wide_arr, pk->nb_wires * psz) != hipSuccess) {
                    set_error("proving key: hipMalloc of a wire-indexed base array failed");
                    return GA_ERR_NOMEM;
                }
                GA_HIP_CHECK(hipMemsetAsync(wide_arr, 0, pk->nb_wires * psz, ctx->stream));
                const uint32_t chunks = (uint32_t)(psz / 16);
                const uint64_t threads = len * chunks;
                hipLaunchKernelGGL(g16_scatter_points_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (u32x4*)wide_arr, (const u32x4*)*slot, d_idx, len, chunks);
                GA_KERNEL_CHECK();
                GA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                hipFree(*slot);
                *slot = wide_arr;
                return GA_OK;
            };
            uint32_t* d_ik = pk->d_idx_k;   // wire ids of K's entries: the remove-list gather, or nbPublic + i
            if (pk->share_k && !d_ik) {
                const uint64_t nbp = pk->nb_wires - pk->len_k;
                std::vector<uint32_t> ikk(pk->len_k);
                for (uint64_t i = 0; i < pk->len_k; i++) ikk[i] = (uint32_t)(nbp + i);
                rc = upload(ctx, ikk.data(), pk->len_k * 4, (void**)&d_ik);
                if (rc == GA_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = GA_ERR_HIP;
            }
            if (rc == GA_OK && pk->share_a) rc = widen(&pk->d_a, pk->len_a, pk->d_idx_a, s1);
            if (rc == GA_OK && pk->share_b) rc = widen(&pk->d_b, pk->len_b, pk->d_idx_b, s1);
            if (rc == GA_OK && pk->share_b) rc = widen(&pk->d_b2, pk->len_b2, pk->d_idx_b, s2);
            if (rc == GA_OK && pk->share_k) rc = widen(&pk->d_k, pk->len_k, d_ik, s1);
            if (d_ik && d_ik != pk->d_idx_k) hipFree(d_ik);
            const uint64_t nwr = pk->nb_wires;
            if (rc == GA_OK) rc = pk->share_a ? make(&pk->d_a, nwr, pk->c_w, t1, b1) : make(&pk->d_a, pk->len_a, pk->c_a, t1, b1);
            if (rc == GA_OK) rc = pk->share_b ? make(&pk->d_b, nwr, pk->c_w, t1, b1) : make(&pk->d_b, pk->len_b, pk->c_b, t1, b1);
            if (rc == GA_OK) rc = make(&pk->d_z, pk->len_z, pk->c_z, t1, b1);
            if (rc == GA_OK) rc = pk->share_k ? make(&pk->d_k, nwr, pk->c_w, t1, b1) : make(&pk->d_k, pk->len_k, pk->c_k, t1, b1);
            if (rc == GA_OK) rc = pk->share_b ? make(&pk->d_b2, nwr, pk->c_w, t2, b2) : make(&pk->d_b2, pk->len_b2, pk->c_b, t2, b2);
            pk->tables = rc == GA_OK;
            if (!pk->tables) pk->share_a = pk->share_b = pk->share_k = false;
        }
    }
    if (!pk->tables) pk->share_a = pk->share_b = pk->share_k = false;
    if (rc != GA_OK) {
        pk_free(pk);
        return rc;
    }
    auto cp = [](std::vector<uint8_t>& v, const void* p, size_t n) { v.assign((const uint8_t*)p, (const uint8_t*)p + n); };
    cp(pk->alpha1, key->g1_alpha, s1);
    cp(pk->beta1, key->g1_beta, s1);
    cp(pk->delta1, key->g1_delta, s1);
    cp(pk->beta2, key->g2_beta, s2);
    cp(pk->delta2, key->g2_delta, s2);
    *out = pk;
    return GA_OK;
}

// The device part of a proof on this key's shard: computeH + the five MSMs over the pinned slices.
// Outputs (before randomisation): A-sum, B1-sum, K-sum + Z-sum (G1), B2-sum (G2) -- to be added across shards.
template <class C>
static int prove_partial(G16Pk* pk, const void* w, const void* a, const void* b, const void* c, uint64_t n_constraints,
                         uint64_t nb_public, XYZZ<Fe<typename C::FpP>>* o_ar, XYZZ<Fe<typename C::FpP>>* o_bs1,
                         XYZZ<Fe<typename C::FpP>>* o_krs, XYZZ<Fe2<typename C::FpP>>* o_bs2) {
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    Ctx* ctx = pk->ctx;
    const uint64_t n = pk->n;
    if (n_constraints > n || nb_public > pk->nb_wires || pk->nb_wires - nb_public != pk->full_len_k + pk->len_k_remove) {
        set_error("prove: inconsistent sizes (constraints %llu > n %llu, or nbWires-nbPublic != len(K)+len(k_remove))",
                  (unsigned long long)n_constraints, (unsigned long long)n);
        return GA_ERR_INVALID;
    }
    hipStream_t st = ctx->stream;
    // ---- upload the solution ----------------------------------------------------------------------
    void *d_w, *d_ha, *d_hb, *d_hc, *d_wa, *d_wb;
    GA_CHECK(ctx->scratch_get("g16_w", pk->nb_wires * 32, &d_w));
    GA_CHECK(ctx->scratch_get("h_a", n * 32, &d_ha));
    GA_CHECK(ctx->scratch_get("h_b", n * 32, &d_hb));
    GA_CHECK(ctx->scratch_get("h_c", n * 32, &d_hc));
    GA_CHECK(ctx->scratch_get("g16_wa", pk->len_a * 32 + 32, &d_wa));
    GA_CHECK(ctx->scratch_get("g16_wb", pk->len_b * 32 + 32, &d_wb));
    // W first (the four witness MSMs only need W); A, B, C are uploaded by a helper thread on a second stream while
    // those MSMs run -- pageable H2D copies block the calling thread, hence the thread.  Everything is joined before
    // this function returns, so no host pointer outlives the call.
    {
        StageTimer tm(ctx, "g16_h2d_w");
        GA_HIP_CHECK(hipMemcpyAsync(d_w, w, pk->nb_wires * 32, hipMemcpyHostToDevice, st));
    }
    hipEvent_t ev_abc;
    GA_HIP_CHECK(hipEventCreateWithFlags(&ev_abc, hipEventDisableTiming));
    int up_rc = GA_OK;
    std::string up_err;
    std::thread uploader([&]() {
        if (hipSetDevice(ctx->device) != hipSuccess) {
            up_rc = GA_ERR_HIP;
            return;
        }
        const void* src[3] = {a, b, c};
        void* dst[3] = {d_ha, d_hb, d_hc};
        hipError_t e = hipSuccess;
        for (int k = 0; k < 3 && e == hipSuccess; k++) {
            e = hipMemcpyAsync(dst[k], src[k], n_constraints * 32, hipMemcpyHostToDevice, ctx->copy_stream);
            if (e == hipSuccess && n > n_constraints)   // computeH pads to the domain size (prove.go:356-359)
                e = hipMemsetAsync((char*)dst[k] + n_constraints * 32, 0, (n - n_constraints) * 32, ctx->copy_stream);
        }
        if (e == hipSuccess) e = hipEventRecord(ev_abc, ctx->copy_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
        if (e != hipSuccess) {
            up_rc = GA_ERR_HIP;
            up_err = hipGetErrorString(e);
        }
    });
    struct Joiner {
        std::thread& t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } joiner{uploader};
    // ---- wire filtering (prove.go:147-168) ------------------------------------------------------------
    if (!pk->share_a) GA_CHECK(util_gather_fr<C>(ctx, d_wa, d_w, pk->d_idx_a, pk->len_a));
    if (!pk->share_b) GA_CHECK(util_gather_fr<C>(ctx, d_wb, d_w, pk->d_idx_b, pk->len_b));
    const void* d_wk = (const char*)d_w + (nb_public + pk->off_k) * 32;
    if (pk->d_idx_k && !pk->share_k) {
        void* g;
        GA_CHECK(ctx->scratch_get("g16_wk", pk->len_k * 32 + 32, &g));
        GA_CHECK(util_gather_fr<C>(ctx, g, d_w, pk->d_idx_k, pk->len_k));
        d_wk = g;
    }
    // ---- the four witness MSMs (prove.go:194,207,237,283) ----------------------------------------------
    XYZZ<F1> ar, bs1, krs, krs2;
    XYZZ<F2> bs2;
    MsmPrepared prep;
    auto table_msm_g1 = [&](const void* table, const void* scal, uint64_t len, int c, XYZZ<F1>* out) -> int {
        if (len == 0) {
            *out = xyzz_inf<F1>();
            return GA_OK;
        }
        GA_CHECK(msm_prepare_table_scalars<C>(ctx, scal, len, true, c, &prep));
        return msm_table_device_reuse<C, GA_G1>(ctx, table, prep, out);
    };
    if (pk->tables) {
        // digits + sort of the WHOLE witness once (scratch slot 1), reused by every wire-indexed table
        MsmPrepared prep_w;
        if (pk->share_a || pk->share_b || pk->share_k) GA_CHECK(msm_prepare_table_scalars<C>(ctx, d_w, pk->nb_wires, true, pk->c_w, &prep_w, 1));
        if (pk->share_a) GA_CHECK((msm_table_device_reuse<C, GA_G1>(ctx, pk->d_a, prep_w, &ar)));
        else GA_CHECK(table_msm_g1(pk->d_a, d_wa, pk->len_a, pk->c_a, &ar));
        if (pk->share_b) {
            GA_CHECK((msm_table_device_reuse<C, GA_G1>(ctx, pk->d_b, prep_w, &bs1)));
            GA_CHECK((msm_table_device_reuse<C, GA_G2>(ctx, pk->d_b2, prep_w, &bs2)));
        } else {
            GA_CHECK(table_msm_g1(pk->d_b, d_wb, pk->len_b, pk->c_b, &bs1));
            if (pk->len_b2) GA_CHECK((msm_table_device_reuse<C, GA_G2>(ctx, pk->d_b2, prep, &bs2)));   // same scalars wB: digits/sort shared
            else bs2 = xyzz_inf<F2>();
        }
        if (pk->share_k) GA_CHECK((msm_table_device_reuse<C, GA_G1>(ctx, pk->d_k, prep_w, &krs)));
        else GA_CHECK(table_msm_g1(pk->d_k, d_wk, pk->len_k, pk->c_k, &krs));
    } else {
        GA_CHECK((host_msm<C, GA_G1>(ctx, pk->d_a, d_wa, pk->len_a, true, &ar)));
        GA_CHECK((host_msm<C, GA_G1>(ctx, pk->d_b, d_wb, pk->len_b, true, &bs1)));
        GA_CHECK((host_msm<C, GA_G2>(ctx, pk->d_b2, d_wb, pk->len_b2, true, &bs2)));
        GA_CHECK((host_msm<C, GA_G1>(ctx, pk->d_k, d_wk, pk->len_k, true, &krs)));
    }
    // ---- H (prove.go:134,346-389), then the MSM over pk.G1.Z (prove.go:225-227) ----------------------------
    uploader.join();
    if (up_rc != GA_OK) {
        set_error("prove: uploading A,B,C failed: %s", up_err.c_str());
        hipEventDestroy(ev_abc);
        return up_rc;
    }
    GA_HIP_CHECK(hipStreamWaitEvent(st, ev_abc, 0));
    GA_CHECK(ntt_domain_compute_h<C>(pk->dom, d_ha, d_hb, d_hc));   // h in d_ha, bit-reversed like pk.G1.Z
    const void* d_hz = (const char*)d_ha + pk->off_z * 32;   // this shard's slice of h[:n-1]
    if (pk->tables) GA_CHECK(table_msm_g1(pk->d_z, d_hz, pk->len_z, pk->c_z, &krs2));
    else GA_CHECK((host_msm<C, GA_G1>(ctx, pk->d_z, d_hz, pk->len_z, true, &krs2)));
    hipEventDestroy(ev_abc);
    *o_ar = ar;
    *o_bs1 = bs1;
    *o_krs = add(krs, krs2);
    *o_bs2 = bs2;
    return GA_OK;
}

// Host epilogue with the prover's randomness (prove.go:171-185,199-200,212-214,241-269,287-292) on the SUMMED partials.
template <class C>
static int finish(G16Pk* pk, XYZZ<Fe<typename C::FpP>> ar, XYZZ<Fe<typename C::FpP>> bs1, XYZZ<Fe<typename C::FpP>> krs,
                  XYZZ<Fe2<typename C::FpP>> bs2, const void* r_mont, const void* s_mont, void* proof_out) {
    typedef typename C::FrP FrP;
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    StageTimer tm(pk->ctx, "g16_epilogue_host");
    Fe<FrP> r, s;
    memcpy(r.l, r_mont, 32);
    memcpy(s.l, s_mont, 32);
    Fe<FrP> kr = neg(mul(r, s));
    Fe<FrP> rc = from_mont(r), sc = from_mont(s), krc = from_mont(kr);
    XYZZ<F1> delta1 = host_load_affine<F1>(pk->delta1.data());
    XYZZ<F1> d_r = scalar_mul(delta1, rc.l, 8), d_s = scalar_mul(delta1, sc.l, 8), d_kr = scalar_mul(delta1, krc.l, 8);
    bs1 = add(add(bs1, host_load_affine<F1>(pk->beta1.data())), d_s);
    ar = add(add(ar, host_load_affine<F1>(pk->alpha1.data())), d_r);
    krs = add(krs, d_kr);
    krs = add(krs, scalar_mul(ar, sc.l, 8));
    krs = add(krs, scalar_mul(bs1, rc.l, 8));
    XYZZ<F2> delta2 = host_load_affine<F2>(pk->delta2.data());
    bs2 = add(add(bs2, scalar_mul(delta2, sc.l, 8)), host_load_affine<F2>(pk->beta2.data()));
    char* o = reinterpret_cast<char*>(proof_out);
    host_store_affine<F1>(o, ar);
    host_store_affine<F2>(o + sizeof(Affine<F1>), bs2);
    host_store_affine<F1>(o + sizeof(Affine<F1>) + sizeof(Affine<F2>), krs);
    return GA_OK;
}

// BSB22: commitment_i = <values, Basis_i> (pedersen Commit, prove.go:84) and its proof of knowledge <values, BasisExpSigma_i>
// (ProveKnowledge, prove.go:114) -- the ICICLE prover's two commitment MSM blocks (icicle.go:834-873,904-944) in one call.
template <class C>
static int commit(G16Pk* pk, uint32_t index, const void* values, uint64_t n_values, void* commitment_out, void* pok_out) {
    typedef Fe<typename C::FpP> F1;
    Ctx* ctx = pk->ctx;
    if (index >= pk->ck_len.size() || n_values != pk->ck_len[index]) {
        set_error("commit: commitment %u of %zu, %llu values for a basis of %llu points", index, pk->ck_len.size(),
                  (unsigned long long)n_values, (unsigned long long)(index < pk->ck_len.size() ? pk->ck_len[index] : 0));
        return GA_ERR_INVALID;
    }
    XYZZ<F1> com = xyzz_inf<F1>(), pok = xyzz_inf<F1>();
    if (n_values) {
        void* d_v;
        GA_CHECK(ctx->scratch_get("g16_commit_values", n_values * 32, &d_v));
        GA_HIP_CHECK(hipMemcpyAsync(d_v, values, n_values * 32, hipMemcpyHostToDevice, ctx->stream));
        GA_CHECK((host_msm<C, GA_G1>(ctx, pk->d_ck_basis[index], d_v, n_values, true, &com)));
        GA_CHECK((host_msm<C, GA_G1>(ctx, pk->d_ck_sigma[index], d_v, n_values, true, &pok)));
    }
    host_store_affine<F1>(commitment_out, com);
    host_store_affine<F1>(pok_out, pok);
    return GA_OK;
}

// sum_i challenge^i * poks[i]  (Horner from the last point)
template <class C>
static int fold_pok(const void* poks, uint64_t n, const void* challenge_mont, void* out) {
    typedef Fe<typename C::FpP> F1;
    uint32_t ch[8];
    host_fr_canonical<typename C::FrP>(challenge_mont, ch);
    XYZZ<F1> acc = xyzz_inf<F1>();
    for (uint64_t i = n; i-- > 0;) {
        acc = scalar_mul(acc, ch, 8);
        acc = add(acc, host_load_affine<F1>((const char*)poks + i * sizeof(Affine<F1>)));
    }
    host_store_affine<F1>(out, acc);
    return GA_OK;
}

// ---- compressed point encoding (marshal.go:33-58 via gnark-crypto's encoder [EXT]) ---------
template <class P>
static bool lex_largest(const Fe<P>& y_mont) {
    // y > (p-1)/2 on the canonical value
    Fe<P> y = from_mont(y_mont);
    uint32_t half[P::N];
    for (int i = 0; i < P::N; i++) half[i] = (P::MOD[i] >> 1) | (i + 1 < P::N ? (P::MOD[i + 1] << 31) : 0);
    for (int i = P::N - 1; i >= 0; i--) {
        if (y.l[i] > half[i]) return true;
        if (y.l[i] < half[i]) return false;
    }
    return false;
}
template <class P>
static void be_bytes(const Fe<P>& x_mont, uint8_t* out) {
    Fe<P> x = from_mont(x_mont);
    for (int i = 0; i < P::N; i++) {
        uint32_t v = x.l[P::N - 1 - i];
        out[4 * i] = v >> 24;
        out[4 * i + 1] = v >> 16;
        out[4 * i + 2] = v >> 8;
        out[4 * i + 3] = v;
    }
}
template <class C>
static void flag_bytes(uint8_t* out, bool inf, bool largest) {
    if (C::ID == GA_BN254) out[0] |= inf ? 0x40 : (largest ? 0xC0 : 0x80);
    else out[0] |= inf ? 0xC0 : (0x80 | (largest ? 0x20 : 0));
}
template <class C>
static size_t compress_g1(const void* aff, uint8_t* out) {
    typedef typename C::FpP P;
    Affine<Fe<P>> a;
    memcpy(&a, aff, sizeof(a));
    const size_t nb = P::N * 4;
    memset(out, 0, nb);
    if (is_inf(a)) {
        flag_bytes<C>(out, true, false);
        return nb;
    }
    be_bytes<P>(a.x, out);
    flag_bytes<C>(out, false, lex_largest<P>(a.y));
    return nb;
}
template <class C>
static size_t compress_g2(const void* aff, uint8_t* out) {
    typedef typename C::FpP P;
    Affine<Fe2<P>> a;
    memcpy(&a, aff, sizeof(a));
    const size_t nb = P::N * 4;
    memset(out, 0, 2 * nb);
    if (is_inf(a)) {
        flag_bytes<C>(out, true, false);
        return 2 * nb;
    }
    be_bytes<P>(a.x.c1, out);        // A1 || A0
    be_bytes<P>(a.x.c0, out + nb);
    bool largest = is_zero(a.y.c1) ? lex_largest<P>(a.y.c0) : lex_largest<P>(a.y.c1);
    flag_bytes<C>(out, false, largest);
    return 2 * nb;
}

// uncompressed encodings (curve.RawEncoding(), Proof.WriteRawTo marshal.go:25-30): x | y big-endian, G2 coordinates A1 | A0;
// infinity = flag byte 0x40 followed by zeros
template <class C>
static size_t raw_g1(const void* aff, uint8_t* out) {
    typedef typename C::FpP P;
    Affine<Fe<P>> a;
    memcpy(&a, aff, sizeof(a));
    const size_t nb = P::N * 4;
    memset(out, 0, 2 * nb);
    if (is_inf(a)) {
        out[0] = 0x40;
        return 2 * nb;
    }
    be_bytes<P>(a.x, out);
    be_bytes<P>(a.y, out + nb);
    return 2 * nb;
}
template <class C>
static size_t raw_g2(const void* aff, uint8_t* out) {
    typedef typename C::FpP P;
    Affine<Fe2<P>> a;
    memcpy(&a, aff, sizeof(a));
    const size_t nb = P::N * 4;
    memset(out, 0, 4 * nb);
    if (is_inf(a)) {
        out[0] = 0x40;
        return 4 * nb;
    }
    be_bytes<P>(a.x.c1, out);
    be_bytes<P>(a.x.c0, out + nb);
    be_bytes<P>(a.y.c1, out + 2 * nb);
    be_bytes<P>(a.y.c0, out + 3 * nb);
    return 4 * nb;
}

template <class C>
static int marshal(const void* proof, const void* commitments, uint32_t ncom, const void* pok, uint8_t* out, size_t cap, size_t* len,
                   bool raw = false) {
    typedef Fe<typename C::FpP> F1;
    typedef Fe2<typename C::FpP> F2;
    const size_t nb = C::FpP::N * 4;
    const size_t need = (raw ? 2 : 1) * (nb + 2 * nb + nb + (size_t)ncom * nb + nb) + 4;
    if (cap < need) {
        set_error("proof marshal: buffer too small (%zu < %zu)", cap, need);
        return GA_ERR_INVALID;
    }
    const char* p = reinterpret_cast<const char*>(proof);
    size_t o = 0;
    auto g1 = [&](const void* a, uint8_t* dst) { return raw ? raw_g1<C>(a, dst) : compress_g1<C>(a, dst); };
    o += g1(p, out + o);
    o += raw ? raw_g2<C>(p + sizeof(Affine<F1>), out + o) : compress_g2<C>(p + sizeof(Affine<F1>), out + o);
    o += g1(p + sizeof(Affine<F1>) + sizeof(Affine<F2>), out + o);
    out[o] = ncom >> 24;   // uint32 big-endian number of commitments (the slice encoder's length prefix)
    out[o + 1] = ncom >> 16;
    out[o + 2] = ncom >> 8;
    out[o + 3] = ncom;
    o += 4;
    for (uint32_t i = 0; i < ncom; i++) o += g1((const char*)commitments + i * sizeof(Affine<F1>), out + o);
    Affine<F1> inf;
    memset(&inf, 0, sizeof(inf));
    o += g1(pok ? pok : &inf, out + o);   // CommitmentPok (infinity without commitments)
    *len = o;
    return GA_OK;
}

}  // namespace ga

using namespace ga;

extern "C" {

int ga_g16_pk_create(ga_ctx* h, const ga_g16_key* key, ga_g16_pk** out) {
    Ctx* ctx = reinterpret_cast<Ctx*>(h);
    if (!ctx || !key || !out) {
        set_error("ga_g16_pk_create: null argument");
        return GA_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    hipSetDevice(ctx->device);
    G16Pk* pk = nullptr;
    GA_DISPATCH_CURVE(key->curve, GA_CHECK(pk_create<C>(ctx, key, &pk)));
    *out = reinterpret_cast<ga_g16_pk*>(pk);
    return GA_OK;
}

void ga_g16_pk_destroy(ga_g16_pk* p) {
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk) return;
    std::lock_guard<std::mutex> g(pk->ctx->mu);
    hipSetDevice(pk->ctx->device);
    hipStreamSynchronize(pk->ctx->stream);
    pk_free(pk);
}

int ga_g16_prove(ga_g16_pk* p, const void* w, const void* a, const void* b, const void* c, uint64_t n_constraints,
                 uint64_t nb_public, const void* r, const void* s, void* proof_out) {
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !w || !a || !b || !c || !r || !s || !proof_out) {
        set_error("ga_g16_prove: null argument");
        return GA_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(pk->ctx->mu);
    hipSetDevice(pk->ctx->device);
    if (pk->shard_count != 1) {
        set_error("ga_g16_prove: this key holds shard %u of %u; use ga_g16_prove_partial + ga_g16_finish", pk->shard_index, pk->shard_count);
        return GA_ERR_STATE;
    }
    GA_DISPATCH_CURVE(pk->curve, {
        XYZZ<Fe<typename C::FpP>> ar, bs1, krs;
        XYZZ<Fe2<typename C::FpP>> bs2;
        GA_CHECK(prove_partial<C>(pk, w, a, b, c, n_constraints, nb_public, &ar, &bs1, &krs, &bs2));
        return finish<C>(pk, ar, bs1, krs, bs2, r, s, proof_out);
    });
    return GA_OK;
}

int ga_g16_prove_partial(ga_g16_pk* p, const void* w, const void* a, const void* b, const void* c, uint64_t n_constraints,
                         uint64_t nb_public, void* partials_out) {
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !w || !a || !b || !c || !partials_out) {
        set_error("ga_g16_prove_partial: null argument");
        return GA_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(pk->ctx->mu);
    hipSetDevice(pk->ctx->device);
    GA_DISPATCH_CURVE(pk->curve, {
        typedef Fe<typename C::FpP> F1;
        typedef Fe2<typename C::FpP> F2;
        XYZZ<F1> ar, bs1, krs;
        XYZZ<F2> bs2;
        GA_CHECK(prove_partial<C>(pk, w, a, b, c, n_constraints, nb_public, &ar, &bs1, &krs, &bs2));
        char* o = reinterpret_cast<char*>(partials_out);
        host_store_jac<F1>(o, ar);
        host_store_jac<F1>(o + sizeof(Jac<F1>), bs1);
        host_store_jac<F1>(o + 2 * sizeof(Jac<F1>), krs);
        host_store_jac<F2>(o + 3 * sizeof(Jac<F1>), bs2);
    });
    return GA_OK;
}

int ga_g16_finish(ga_g16_pk* p, const void* partials_sum, const void* r, const void* s, void* proof_out) {
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || !partials_sum || !r || !s || !proof_out) {
        set_error("ga_g16_finish: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(pk->curve, {
        typedef Fe<typename C::FpP> F1;
        typedef Fe2<typename C::FpP> F2;
        const char* i = reinterpret_cast<const char*>(partials_sum);
        return finish<C>(pk, host_load_jac<F1>(i), host_load_jac<F1>(i + sizeof(Jac<F1>)), host_load_jac<F1>(i + 2 * sizeof(Jac<F1>)),
                         host_load_jac<F2>(i + 3 * sizeof(Jac<F1>)), r, s, proof_out);
    });
    return GA_OK;
}

int ga_g16_proof_marshal(int curve, const void* proof, uint8_t* out, size_t cap, size_t* len) {
    if (!proof || !out || !len) {
        set_error("ga_g16_proof_marshal: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, return marshal<C>(proof, nullptr, 0, nullptr, out, cap, len));
    return GA_OK;
}

int ga_g16_proof_marshal_bsb22(int curve, const void* proof, const void* commitments, uint32_t n, const void* pok, uint8_t* out,
                               size_t cap, size_t* len) {
    if (!proof || !out || !len || (n && !commitments)) {
        set_error("ga_g16_proof_marshal_bsb22: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, return marshal<C>(proof, commitments, n, pok, out, cap, len));
    return GA_OK;
}

int ga_g16_proof_marshal_raw(int curve, const void* proof, const void* commitments, uint32_t n, const void* pok, uint8_t* out,
                             size_t cap, size_t* len) {
    if (!proof || !out || !len || (n && !commitments)) {
        set_error("ga_g16_proof_marshal_raw: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, return marshal<C>(proof, commitments, n, pok, out, cap, len, true));
    return GA_OK;
}

int ga_g1_marshal_uncompressed(int curve, const void* affine, uint8_t* out, size_t cap, size_t* len) {
    if (!affine || !out || !len) {
        set_error("ga_g1_marshal_uncompressed: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, {
        typedef typename C::FpP P;
        const size_t nb = P::N * 4;
        if (cap < 2 * nb) {
            set_error("ga_g1_marshal_uncompressed: buffer too small");
            return GA_ERR_INVALID;
        }
        Affine<Fe<P>> a;
        memcpy(&a, affine, sizeof(a));
        memset(out, 0, 2 * nb);
        if (is_inf(a)) {
            out[0] = C::ID == GA_BN254 ? 0x40 : 0x40;   // mUncompressedInfinity: 0b01<<6 (BN254), 0b010<<5 (BLS12-381)
        } else {
            be_bytes<P>(a.x, out);
            be_bytes<P>(a.y, out + nb);
        }
        *len = 2 * nb;
    });
    return GA_OK;
}

int ga_g16_commit(ga_g16_pk* p, uint32_t index, const void* values, uint64_t n_values, void* commitment_out, void* pok_out) {
    G16Pk* pk = reinterpret_cast<G16Pk*>(p);
    if (!pk || (!values && n_values) || !commitment_out || !pok_out) {
        set_error("ga_g16_commit: null argument");
        return GA_ERR_INVALID;
    }
    std::lock_guard<std::mutex> g(pk->ctx->mu);
    hipSetDevice(pk->ctx->device);
    GA_DISPATCH_CURVE(pk->curve, return commit<C>(pk, index, values, n_values, commitment_out, pok_out));
    return GA_OK;
}

int ga_g16_fold_pok(int curve, const void* poks, uint64_t n, const void* challenge, void* out) {
    if ((!poks && n) || !challenge || !out) {
        set_error("ga_g16_fold_pok: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, return fold_pok<C>(poks, n, challenge, out));
    return GA_OK;
}

}  // extern "C"
