/* gnark_amd.h -- C ABI of libgnark_amd.so, the MI355X (gfx950) prover backend for gnark.
 *
 * This is the drop-in boundary: the surface a Go package `backend/accelerated/mi355x/groth16` binds through
 * cgo in place of the ~20 ICICLE entry points that backend/accelerated/icicle/groth16/bn254/icicle.go uses
 * (reference line numbers below are into /root/reference).  INTEGRATION.md shows the cgo stub.
 *
 * Conventions
 *  - All field elements / points are gnark-crypto memory images: little-endian 64-bit limbs in Montgomery
 *    form (fr.Element = [4]uint64; fp.Element = [4]uint64 BN254 / [6]uint64 BLS12-381); G1Affine = {X,Y};
 *    G2Affine = {X{A0,A1}, Y{A0,A1}}; point at infinity = all-zero coordinates; G1Jac/G2Jac = {X,Y,Z}.
 *    A Go slice `[]fr.Element` / `[]curve.G1Affine` can be passed as `unsafe.Pointer(&s[0])` unchanged.
 *  - No pointer passed in is retained after the call returns (cgo pointer rules): inputs are copied to
 *    device memory (or are already device pointers, see GA_*_ON_DEVICE) before the function returns.
 *  - Every entry point selects its device itself (hipSetDevice), so callers need not pin OS threads
 *    (icicle.go relies on RunOnDevice + LockOSThread instead).
 *  - Every function returns GA_OK (0) or a negative error code; ga_last_error() gives the message for the
 *    calling thread.  There is NO CPU fallback: without a usable HIP device ga_ctx_create fails.
 *  - Any host thread may call any entry point on any context at any time; results never depend on the interleaving
 *    (icicle.go:77-86,821-823 keeps a per-device prove mutex for the same guarantee).  Calls on one ga_ctx are serialised by an
 *    internal mutex, with exceptions that only add throughput: a context has four lanes (stream + scratch namespace each).
 *    A Groth16 proof runs on a pair of them -- the witness MSMs on one, its H side (computeH, the Z MSM) on the partner lane
 *    from a helper thread (GA_G16_SPLIT=0 keeps one lane) --, a second ga_g16_prove arriving while the first pair is busy runs
 *    on the second pair beside it, and a second ga_msm_table_run* (the PLONK prover commits from several goroutines) or
 *    ga_g16_h_chain* / ga_g16_h_combine (the pieces of a sharded proof) takes lane 1 (GA_G16_LANES=1 restores strict queueing).
 *    ga_g16_lane_stats says how the proofs of a context were scheduled.
 *  - Destroying an object while other threads are still inside entry points that use it is safe for proving keys:
 *    ga_g16_pk_destroy waits for them (a Go `defer pk.FreeGPUResources()` beside another goroutine's Prove); every other
 *    destroy call must come after the last use, as with any C object.
 */
#ifndef GNARK_AMD_H
#define GNARK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GA_OK 0
#define GA_ERR_INVALID (-1)   /* bad argument */
#define GA_ERR_HIP (-2)       /* HIP runtime error (message in ga_last_error) */
#define GA_ERR_NOMEM (-3)     /* device allocation failed -- or would have left less than GA_HBM_RESERVE_MB (environment, default
                                 1024) MiB of HBM free: the ROCm runtime allocates the kernels' private segments at dispatch time
                                 and aborts the PROCESS when it cannot, so the library never takes the last gigabyte.  After this
                                 error the context and its keys stay usable. */
#define GA_ERR_STATE (-4)     /* object used in the wrong state */

/* curve ids (ecc.ID analogue; only the two curves of BASELINE.json are built) */
#define GA_BN254 0
#define GA_BLS12_381 1

/* group ids */
#define GA_G1 0
#define GA_G2 1

/* flags for ga_msm */
#define GA_BASES_ON_DEVICE 0x1u        /* `bases` is a device pointer */
#define GA_SCALARS_ON_DEVICE 0x2u      /* `scalars` is a device pointer */
#define GA_TABLE_BATCHED 0x10u         /* ga_msm_table_create: the table will mostly serve ga_msm_table_run_batch (PLONK's grouped
                                         * commitments over the SRS): plan a narrower window -- k bucket sets make the sort keys
                                         * log2(k) bits wider and the reduction k times larger, which moves the optimum down
                                         * (2^22 points: c = 20 instead of 22, PLONK leg 82.9 -> 79.8 ms; a single run costs +1 %) */
#define GA_SCALARS_MONTGOMERY 0x4u     /* scalars are fr.Element images (Montgomery); else canonical LE integers
                                          (ICICLE's AreScalarsMontgomeryForm, icicle.go:861-863,1232) */
#define GA_RESULT_WINDOW_SUMS 0x8u     /* multi-GPU window sharding: see ga_msm_windows */

/* NTT direction / ordering, mirroring gnark-crypto fft.Domain.FFT / FFTInverse (prove.go:362-386) */
#define GA_FFT_FORWARD 0
#define GA_FFT_INVERSE 1
#define GA_DIF 0   /* natural in  -> bit-reversed out */
#define GA_DIT 1   /* bit-reversed in -> natural out  */

typedef struct ga_ctx ga_ctx;         /* one per (process, device) */
typedef struct ga_domain ga_domain;   /* fft.Domain analogue: twiddles for one (curve, cardinality) */
typedef struct ga_g16_pk ga_g16_pk;   /* device-resident Groth16 proving key ("PinToGPU", provingkey.go:37-42) */

/* ---- context ------------------------------------------------------------------------------------------
 * replaces: icicle runtime LoadBackend / CreateDevice / WarmUpDevice (groth16_icicle.go:38-72). */
int ga_device_count(int* count);
int ga_ctx_create(int device, ga_ctx** out);
/* ga_ctx_destroy: every object created on the context (proving keys, builders, domains, tables, PLONK keys) must have been destroyed
 * first and no entry point may be running on it -- the objects keep a plain pointer to their context.  The Go package creates one
 * context per device for the life of the process (internal/ga) and never calls this. */
void ga_ctx_destroy(ga_ctx* ctx);
const char* ga_last_error(void);
const char* ga_version(void);
/* device name / gcn arch / total and free bytes (runtime.GetAvailableMemory, icicle.go:475) */
int ga_device_info(ga_ctx* ctx, char* name, size_t name_len, uint64_t* total_bytes, uint64_t* free_bytes);

/* ---- raw device buffers (DeviceSlice analogue: icicle.go:120,321,...) ---------------------------------*/
int ga_malloc(ga_ctx* ctx, size_t bytes, void** dptr);
int ga_free(ga_ctx* ctx, void* dptr);
int ga_copy_to_device(ga_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int ga_copy_to_host(ga_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int ga_sync(ga_ctx* ctx);

/* ---- multi-scalar multiplication -----------------------------------------------------------------------
 * replaces: G1Jac.MultiExp / G2Jac.MultiExp (prove.go:194,207,227,237,283) and icicle msm.Msm / g2.G2Msm
 * (icicle.go:397,451).  Computes sum_i scalars[i] * bases[i].
 *   bases   : n affine points (Montgomery), host or device pointer per flags
 *   scalars : n fr elements, Montgomery if GA_SCALARS_MONTGOMERY
 *   out_jac : HOST buffer for one Jacobian point {X,Y,Z} (Montgomery) -- castable to curve.G1Jac / G2Jac.
 * (0,0) bases are treated as infinity and zero scalars are skipped, as gnark-crypto does.
 * Jacobian results (here and in every entry point below that returns Jacobian points or partial sums) are group elements: two
 * calls on the same inputs return the same point, but not necessarily the same representative {X,Y,Z} -- from 2^25 (bucket,
 * point) pairs the order in which a bucket's points are added is not fixed (the first sort pass ranks with atomics, msm.hip.h 1b),
 * exactly as gnark-crypto's MultiExp returns a different representative for a different NbTasks.  Affine outputs (proofs,
 * commitments, ga_jac_to_affine of any result) are bit-identical from call to call and from device to device. */
int ga_msm(ga_ctx* ctx, int curve, int group, const void* bases, const void* scalars, size_t n, unsigned flags,
           void* out_jac);

/* Window-sharded variant for multi-GPU partitioning A: only windows [win_lo, win_hi) of the
 * Pippenger decomposition are accumulated; out_windows receives (win_hi - win_lo) Jacobian window sums W_j,
 * and *window_bits / *num_windows describe the decomposition so that the caller can all-gather the sums and
 * Horner-combine them with ga_msm_combine_windows.  win_lo = 0, win_hi = -1 means "all windows". */
int ga_msm_windows(ga_ctx* ctx, int curve, int group, const void* bases, const void* scalars, size_t n,
                   unsigned flags, int win_lo, int win_hi, void* out_windows, int* window_bits, int* num_windows);
/* number of windows / window width ga_msm* will use for an n-point MSM (deterministic in (curve, group, n)) */
int ga_msm_plan(int curve, int group, size_t n, int* window_bits, int* num_windows);
/* result = sum_j 2^(window_bits*j) * windows[j]   (host arithmetic; windows are Jacobian, Montgomery) */
int ga_msm_combine_windows(int curve, int group, const void* windows, int num_windows, int window_bits,
                           void* out_jac);

/* ---- MSM over pinned bases with precomputed window multiples ---------------------------------------------
 * (ICICLE's MSMConfig.PrecomputeFactor / precompute-bases, icicle.go:507-525.)  ga_msm_table_create uploads (or takes
 * from the device) n affine bases and stores [2^(c*w)]P_i for every window w: windows x n points, sized for the
 * 288 GB of an MI355X (2^24 BN254 G1 points: 12 GiB).  All windows then share one bucket set, so ga_msm_table_run does
 * windows x n bucket additions, ONE bucket reduction and no Horner step.  Results are identical to ga_msm. */
typedef struct ga_msm_table ga_msm_table;
/* flags: GA_BASES_ON_DEVICE, GA_TABLE_BATCHED */
int ga_msm_table_create(ga_ctx* ctx, int curve, int group, const void* bases, size_t n, unsigned flags, ga_msm_table** out);
void ga_msm_table_destroy(ga_msm_table* t);
/* scalars: exactly n fr elements (the table's n); flags: GA_SCALARS_ON_DEVICE, GA_SCALARS_MONTGOMERY */
int ga_msm_table_run(ga_msm_table* t, const void* scalars, unsigned flags, void* out_jac);
int ga_msm_table_info(ga_msm_table* t, int* window_bits, int* num_windows, uint64_t* table_bytes);
/* windows [win_lo, win_hi) of the table only (multi-GPU partition A on pinned bases): the 2^(c*w) factors are part of the table,
 * so the Jacobian results of disjoint window ranges ADD UP to ga_msm_table_run's result (ga_jac_add); no Horner step.  An empty
 * range gives the point at infinity. */
int ga_msm_table_run_windows(ga_msm_table* t, const void* scalars, unsigned flags, int win_lo, int win_hi, void* out_jac);
/* k scalar vectors (1 <= k <= 16; scalars[j] -> n field elements each, all host or all device as `flags` says) over the SAME pinned
 * bases in ONE pass: one sort, one task list, one launch sequence, k bucket sets; out_jacs receives k Jacobian points in order.
 * What the three wire commitments [L],[R],[O] and the three quotient shards [H0],[H1],[H2] of a PLONK proof call
 * (backend/plonk/bn254/prove.go:404-489,558-633: three kzg.Commit on the same SRS issued together): the latency-bound tails of
 * an MSM (bucket reduction, task building) are paid once per batch instead of once per polynomial. */
int ga_msm_table_run_batch(ga_msm_table* t, const void* const* scalars, uint32_t k, unsigned flags, void* out_jacs);

/* ---- small host-side group helpers used by the Go epilogue / multi-GPU combine -------------------------
 * (curve.G1Jac.AddAssign / ScalarMultiplication / FromJacobian, prove.go:199-292).  Host arithmetic. */
int ga_jac_add(int curve, int group, const void* a_jac, const void* b_jac, void* out_jac);
int ga_jac_to_affine(int curve, int group, const void* a_jac, void* out_affine);
int ga_jac_scalar_mul(int curve, int group, const void* a_jac, const void* scalar_canonical_le32,
                      void* out_jac);

/* ---- NTT -----------------------------------------------------------------------------------------------
 * replaces: fft.NewDomain (setup.go:101), fft.Domain.FFT / FFTInverse (prove.go:362-368,386) and icicle
 * ntt.InitDomain / ntt.Ntt (icicle.go:163,1425,1428,1474). */
int ga_domain_create(ga_ctx* ctx, int curve, uint64_t cardinality, ga_domain** out);
void ga_domain_destroy(ga_domain* d);
/* In-place transform of `cardinality` fr elements (Montgomery).  direction: GA_FFT_FORWARD / GA_FFT_INVERSE
 * (inverse includes the 1/n factor); decimation: GA_DIF / GA_DIT; on_coset: fft.OnCoset() with the domain's
 * FrMultiplicativeGen (5 for BN254, 7 for BLS12-381).  data is a host pointer unless on_device != 0. */
int ga_fft(ga_domain* d, void* data, int direction, int decimation, int on_coset, int on_device);

/* computeH (prove.go:346-389 / icicle.go:1391-1488): h = coefficients of (A*B - C)/(X^n - 1) in bit-reversed
 * order.  a,b,c hold n_constraints fr elements (Montgomery), zero-padded internally to the domain size;
 * h_out receives cardinality elements.  Pointers are host unless on_device != 0 (then a is clobbered). */
int ga_compute_h(ga_domain* d, const void* a, const void* b, const void* c, uint64_t n_constraints,
                 void* h_out, int on_device);

/* ---- PLONK: quotient and grand product on the device ---------------------------------------
 * replaces: (*instance).computeNumerator + divideByZH (backend/plonk/bn254/prove.go:841-1123,1287-1350; the "we do **a lot** of
 * FFT here" loop, ~108 FFTs of size n on the CPU) and iop.BuildRatioCopyConstraint (gnark-crypto, call site prove.go:645-655).
 * domain0 / domain1 are ga_domain handles of cardinality n and rho*n (rho = 4, or 8 below 6 constraints, prove.go:247-251).
 * Every polynomial is n fr elements (Montgomery), either canonical coefficients in regular order or -- when its bit of
 * lagrange_mask is set -- evaluations on domain0 in regular order (what s.x[...] holds before computeNumerator).  Bit order:
 * L R O Z Ql Qr Qm Qo Qk S1 S2 S3, then Qcp_0, Pi2_0, Qcp_1, Pi2_1, ...  (prove.go:44-59 without ZS, which is Z shifted).
 * bl, br, bo: the blinding polynomials of L, R, O (order 1: 2 coefficients), bz: of Z (order 2: 3 coefficients), prove.go:72-75.
 * h_out: rho*n fr elements, canonical, regular order -- s.h after divideByZH; h1/h2/h3 are its slices (prove.go:691-728). */
#define GA_PLONK_ON_DEVICE 0x1u   /* polynomials and the output are device pointers */
typedef struct ga_plonk_quotient_in {
    uint32_t nb_bsb;                     /* len(proof.Bsb22Commitments), at most 16 */
    const void *l, *r, *o, *z, *ql, *qr, *qm, *qo, *qk, *s1, *s2, *s3;
    const void* const* qcp;              /* [nb_bsb] s.trace.Qcp[i] */
    const void* const* pi2;              /* [nb_bsb] s.cCommitments[i] */
    uint64_t lagrange_mask;
    const void *bl, *br, *bo, *bz;
    const void *alpha, *beta, *gamma;    /* fr, Montgomery */
    uint32_t flags;
} ga_plonk_quotient_in;
int ga_plonk_quotient(ga_domain* domain0, ga_domain* domain1, const ga_plonk_quotient_in* in, void* h_out);
/* The circuit-constant half of the quotient pinned in HBM -- the precomputation prove.go:1030-1034 rules out on a CPU ("we could
 * pre-compute these rho*2 FFTs and store them at the cost of a huge memory footprint"): the evaluations of Ql, Qr, Qm, Qo, S1,
 * S2, S3 and every Qcp_i on each of the rho cosets, plus 1/(x-1) on each coset: (8 + nb_bsb) * rho * n * 32 bytes (4.3 GB at
 * n = 2^22).  ga_plonk_pk_create reads only ql, qr, qm, qo, s1, s2, s3, qcp, nb_bsb, lagrange_mask and flags of `in`;
 * ga_plonk_quotient_pinned reads only l, r, o, z, qk, pi2, the blinding polynomials, the challenges, lagrange_mask and flags
 * (6 + nb_bsb inverse/forward transform chains per proof instead of 12 + 2*nb_bsb).  The domains must outlive the key. */
typedef struct ga_plonk_pk ga_plonk_pk;
int ga_plonk_pk_create(ga_domain* domain0, ga_domain* domain1, const ga_plonk_quotient_in* in, ga_plonk_pk** out);
void ga_plonk_pk_destroy(ga_plonk_pk* pk);
int ga_plonk_quotient_pinned(ga_plonk_pk* pk, const ga_plonk_quotient_in* in, void* h_out);
/* Z in Lagrange form, regular order (n elements): Z[0] = 1, Z[i+1] = Z[i] * prod_k (e_k[i] + beta*id_k(i) + gamma) /
 * prod_k (e_k[i] + beta*id(perm[k*n+i]) + gamma), id over {1, g, g^2} * w^i.  l, r, o: evaluations on domain0; permutation:
 * 3n int64 (s.trace.S). */
int ga_plonk_build_z(ga_domain* domain0, const void* l, const void* r, const void* o, const int64_t* permutation,
                     const void* beta, const void* gamma, int on_device, void* z_out);
/* kzg.Open(p, point, pk) -> OpeningProof{H, ClaimedValue} (gnark-crypto kzg, call sites backend/plonk/bn254/prove.go:681,788,827):
 * claimed value p(point) and H = commitment to (p(X) - p(point)) / (X - point) over the pinned monomial SRS.  The reference's
 * division is a sequential Horner recurrence on the CPU; here it is a scaling, a suffix sum and a scaling on the device, followed
 * by the table MSM.  srs: ga_msm_table over pk.Kzg.G1 with at least len(p) - 1 points; poly: n fr coefficients (Montgomery; device
 * pointer with GA_SCALARS_ON_DEVICE); point, claimed_value_out: fr Montgomery; h_out: G1Jac. */
int ga_kzg_open(ga_msm_table* srs, const void* poly, size_t n, unsigned flags, const void* point, void* claimed_value_out, void* h_out);
/* out[i] = sum_{j<k} scalars[j] * vecs[j][i], k <= 16 per call: the polynomial folding of kzg.BatchOpenSinglePoint and the
 * linear combinations of innerComputeLinearizedPoly (backend/plonk/bn254/prove.go:1352-1460) in one pass.  scalars: k fr
 * (Montgomery, host); vecs/out: n fr each, all host or all device (on_device). */
int ga_fr_linear_combination(ga_ctx* ctx, int curve, uint64_t n, int k, const void* const* vecs, const void* scalars, void* out,
                             int on_device);
/* p(point) of a canonical-form polynomial (iop.Polynomial.Evaluate, evaluateBlinded prove.go:1186-1215). */
int ga_fr_poly_evaluate(ga_ctx* ctx, int curve, const void* poly, uint64_t n, const void* point, void* value_out, int on_device);
/* out[i] = a[i] * b[i] over fr (fr.Vector.Mul of gnark-crypto; ICICLE's vecOps Mul of icicle.go:1453-1458).  All host or all
 * device pointers (on_device); out may alias a or b. */
int ga_fr_vec_mul(ga_ctx* ctx, int curve, const void* a, const void* b, size_t n, void* out, int on_device);
/* fr.BatchInvert in place (zeros stay zero): the batchInvert of prove.go:1134-1147 */
int ga_fr_batch_invert(ga_ctx* ctx, int curve, void* v, uint64_t n, int on_device);

/* ---- Groth16 -------------------------------------------------------------------------------------------
 * replaces: (*ProvingKey).setupDevicePointers (icicle.go:88-264) and Prove (icicle.go:784-1360).
 * ga_g16_key describes gnark's groth16 ProvingKey (setup.go:25-48) by pointer+length; everything is copied to
 * the device synchronously by ga_g16_pk_create. */
typedef struct ga_g16_key {
    int curve;
    uint64_t domain_cardinality;     /* pk.Domain.Cardinality */
    const void* g1_alpha;            /* G1Affine */
    const void* g1_beta;
    const void* g1_delta;
    const void* g1_a; uint64_t len_a;   /* pk.G1.A (infinity entries removed, setup.go:195-219) */
    const void* g1_b; uint64_t len_b;   /* pk.G1.B */
    const void* g1_z; uint64_t len_z;   /* pk.G1.Z, bit-reversed, len n-1 (setup.go:247-249) */
    const void* g1_k; uint64_t len_k;   /* pk.G1.K */
    const void* g2_beta;             /* G2Affine */
    const void* g2_delta;
    const void* g2_b; uint64_t len_b2;  /* pk.G2.B */
    const uint8_t* infinity_a;       /* pk.InfinityA as bytes (Go []bool image), len nb_wires */
    const uint8_t* infinity_b;
    uint64_t nb_wires;
    uint64_t nb_infinity_a;
    uint64_t nb_infinity_b;
    int32_t precompute;              /* 0: precompute window tables for A,B,K,Z,G2.B -- those that fit comfortably in free HBM, see ga_g16_table_layout
                                        (default), 1: always, -1: never */
    uint32_t shard_index;            /* multi-GPU partition B: pin only slice shard_index of shard_count of every */
    uint32_t shard_count;            /* base vector (contiguous ranges); 0 or 1 = the whole key */
    /* BSB22 commitments (setup.go:260-287, prove.go:60-135,231-235); all zero / NULL for a circuit without api.Commit */
    uint32_t nb_commitments;             /* len(pk.CommitmentKeys) */
    const void* const* ck_basis;         /* [nb_commitments] -> pk.CommitmentKeys[i].Basis (G1Affine) */
    const void* const* ck_basis_exp_sigma; /* [nb_commitments] -> pk.CommitmentKeys[i].BasisExpSigma */
    const uint64_t* ck_len;              /* [nb_commitments] number of points in each basis */
    const uint64_t* k_remove;            /* sorted wire ids left out of the K MSM: every PrivateCommitted wire and every */
    uint64_t len_k_remove;               /* commitment wire (the toRemove list of prove.go:233-235); len_k = nbWires - nbPublic - len_k_remove */
    /* multi-GPU partition A (BASELINE config 4 "window-sharded"): the WHOLE key is pinned on every device and this one
     * accumulates share window_shard_index of window_shard_count of the Pippenger windows of every MSM -- on the pinned window
     * tables, where the 2^(c*w) factors are part of the table, so the partial results of the devices simply add.  0 / 0 or 0 / 1 =
     * all windows.  Not combinable with shard_count > 1. */
    uint32_t window_shard_index;
    uint32_t window_shard_count;
} ga_g16_key;

int ga_g16_pk_create(ga_ctx* ctx, const ga_g16_key* key, ga_g16_pk** out);
/* One proof on a key that is not kept on the device -- the default of the reference's GPU backend (PinToGPU false: the key is uploaded
 * for every proof, icicle.go:797-805) and of the Go package here.  The key (key->precompute is ignored: plain vectors, no tables; whole
 * key, no commitments needed before the proof) is uploaded by a helper thread WHILE the proof runs, each MSM waiting for its own vector
 * only, and freed before the call returns; proof bytes identical to ga_g16_pk_create + ga_g16_prove + ga_g16_pk_destroy.  Arguments
 * after `key` as ga_g16_prove.  No pointer is retained after the call. */
int ga_g16_prove_oneshot(ga_ctx* ctx, const ga_g16_key* key, const void* w, const void* a, const void* b, const void* c,
                         uint64_t n_constraints, uint64_t nb_public, const void* r, const void* s, void* proof_out);
void ga_g16_pk_destroy(ga_g16_pk* pk);   /* FreeGPUResources, icicle.go:1493-1549; waits for entry points still using the key */
/* How the ga_g16_prove calls of a context were scheduled since it was created: out6[0] on lanes 0/1 (device free), [1] on lanes
 * 2/3 beside another proof, [2] staged + queued for the device, [3] proofs whose H side ran on a partner lane; [4], [5] = bytes of
 * device scratch held by lanes 0/1 and by lanes 2/3. */
int ga_g16_lane_stats(ga_ctx* ctx, uint64_t* out6);

/* The same key, staged vector by vector (replaces loadG1 / loadG1Raw / loadG2, icicle.go:319-359, one call per host slice).
 * Every call takes ONE flat pointer to pointer-free memory and has copied what it needs when it returns, so a cgo caller never
 * stores a Go pointer inside a C struct and nothing is retained (ga_g16_key holds pointers: from Go it needs runtime.Pinner,
 * see go/backend/accelerated/mi355x).  It is also what the key-file readers below are built on.
 *   create -> reserve(which, len) -> append(which, chunk, count)* -> set_point x5 -> set_infinity x2
 *          [-> add_commitment_key* -> set_k_remove] -> finish  (finish consumes the builder; destroy abandons it) */
/* (ga_g16_builder_append with points == NULL skips `count` points that lie wholly OUTSIDE this shard's slice of the vector: a
 * caller that holds only its own slice -- a file reader seeking past the rest, a rank generating its shard -- need not produce them) */
#define GA_KEY_G1_A 0
#define GA_KEY_G1_B 1
#define GA_KEY_G1_Z 2
#define GA_KEY_G1_K 3
#define GA_KEY_G2_B 4
#define GA_KEY_NB_VECTORS 5
#define GA_KEY_G1_ALPHA 0
#define GA_KEY_G1_BETA 1
#define GA_KEY_G1_DELTA 2
#define GA_KEY_G2_BETA 3
#define GA_KEY_G2_DELTA 4
#define GA_KEY_NB_POINTS 5
typedef struct ga_g16_builder ga_g16_builder;
int ga_g16_builder_create(ga_ctx* ctx, int curve, uint64_t domain_cardinality, uint64_t nb_wires, uint32_t shard_index,
                          uint32_t shard_count, ga_g16_builder** out);
int ga_g16_builder_reserve(ga_g16_builder* b, int which_vector, uint64_t total_len);
/* the next `count` points of vector `which_vector` (affine, Montgomery); with a sharded builder only the part inside this shard's
 * range is copied, the caller still streams the whole vector */
int ga_g16_builder_append(ga_g16_builder* b, int which_vector, const void* points, uint64_t count);
int ga_g16_builder_set_point(ga_g16_builder* b, int which_point, const void* affine);
int ga_g16_builder_set_infinity(ga_g16_builder* b, int which /* 0: InfinityA, 1: InfinityB */, const uint8_t* mask, uint64_t nb_wires);
int ga_g16_builder_add_commitment_key(ga_g16_builder* b, const void* basis, const void* basis_exp_sigma, uint64_t len);
int ga_g16_builder_set_k_remove(ga_g16_builder* b, const uint64_t* wire_ids, uint64_t len);
int ga_g16_builder_set_window_shard(ga_g16_builder* b, uint32_t index, uint32_t count);   /* partition A, see ga_g16_key */
int ga_g16_builder_finish(ga_g16_builder* b, int32_t precompute, ga_g16_pk** out);
void ga_g16_builder_destroy(ga_g16_builder* b);

/* One proof, from the solver's output to the three proof points (prove.go:130-315 minus commitments):
 *   w        : solution.W, nb_wires fr elements (Montgomery)
 *   a,b,c    : solution.A/B/C, n_constraints elements each
 *   nb_public: r1cs.GetNbPublicVariables() (includes the constant-one wire; K scalars start there, prove.go:235)
 *   r,s      : the prover's randomness as fr elements in Montgomery form (prove.go:171-177 samples them; the Go
 *              shim samples and passes them so that tests can inject fixed values)
 *   proof_out: Ar (G1Affine) | Bs (G2Affine) | Krs (G1Affine), Montgomery -- castable to the Proof fields.
 * All pointers are host pointers. */
int ga_g16_prove(ga_g16_pk* pk, const void* w, const void* a, const void* b, const void* c,
                 uint64_t n_constraints, uint64_t nb_public, const void* r, const void* s, void* proof_out);

/* Multi-GPU proving with a key sharded by base-point range (one context + one shard per GPU, every rank gets the whole
 * solution): ga_g16_prove_partial runs computeH and the five MSMs over this shard and returns the sums BEFORE
 * randomisation as Jacobian points  A | B1 | K+Z (G1Jac each) | B2 (G2Jac);  the caller all-gathers them (RCCL), adds them
 * with ga_jac_add and calls ga_g16_finish on the totals with (r, s).  ga_g16_prove == partial + finish on an unsharded key. */
int ga_g16_prove_partial(ga_g16_pk* pk, const void* w, const void* a, const void* b, const void* c,
                         uint64_t n_constraints, uint64_t nb_public, void* partials_out);
int ga_g16_finish(ga_g16_pk* pk, const void* partials_sum, const void* r, const void* s, void* proof_out);

/* The same proof in pieces, for callers that orchestrate several devices themselves (gnark_amd/multigpu.py does it with one
 * process per GPU and RCCL): the witness MSMs of this shard (A | B1 | K as G1Jac, B2 as G2Jac -- W is uploaded only over the wire
 * range the shard's bases cover), one chain of computeH per call (v = the solver's A, B or C on the host; out_dev = n fr elements
 * on this key's device, holding n * FFT_coset(iFFT(v)) afterwards -- the chain is UNSCALED: the 1/n of the inverse transform is
 * not applied here, ga_g16_h_combine divides by n^2 in its point-wise step; a chain buffer is therefore only meaningful as an input
 * of ga_g16_h_combine of the SAME library build, never as the coset evaluations themselves), the combination
 * h = iFFT_coset((a*b - c)/(g^n - 1)) in a_dev from three such chains
 * (bit-reversed), and the MSM of this shard's slice of pk.G1.Z with the matching slice of h (h_slice_dev points at element off_z).
 * ga_g16_shard_layout: out8 = {off_z, len_z, w_lo, w_hi, domain cardinality, nb_wires, window_shard_index, window_shard_count}
 * (a window-sharded key covers all of h and W: off_z = 0, len_z = n - 1).
 * ga_g16_table_layout: out2[0] = which vectors carry a window table -- bit 0 G1.A, 1 G1.B, 2 G1.Z, 3 G1.K, 4 G2.B (with precompute = 0
 * the library builds as many as fit the free HBM, in the order of preference A, B, K, Z, G2.B); out2[1] = which of those tables are
 * laid out by wire id and share the single witness sort (bit 0 A, 1 B, 3 K, 4 G2.B). */
int ga_g16_shard_layout(ga_g16_pk* pk, uint64_t* out8);
int ga_g16_table_layout(ga_g16_pk* pk, uint64_t* out2);
int ga_g16_witness_partial(ga_g16_pk* pk, const void* w, uint64_t nb_public, void* partials_out);
int ga_g16_h_chain(ga_g16_pk* pk, const void* v, uint64_t n_constraints, void* out_dev);
/* the same chain on a vector that already sits on the device (buf_dev: n fr elements of room, the first n_constraints hold the
 * solver's vector): for callers that bring A, B, C to the chain's device in pieces -- every GPU of a node uploading 1/N of each
 * vector over its own PCIe link and handing it on over xGMI (gnark_amd/multigpu.py) -- instead of one 512 MiB upload per chain */
int ga_g16_h_chain_dev(ga_g16_pk* pk, void* buf_dev, uint64_t n_constraints);
int ga_g16_h_combine(ga_g16_pk* pk, void* a_dev, const void* b_dev, const void* c_dev);
int ga_g16_z_partial(ga_g16_pk* pk, const void* h_slice_dev, void* partial_out);
/* One proof over the GPUs of a node from ONE process (what a Go caller uses: mi355x.WithDevices): keys[i] = shard i of n of the
 * same proving key, each created in its own context on its own device.  One host thread per device inside the call; computeH's
 * three chains run on the first three devices, their results and the slices of h move over xGMI (hipMemcpyPeerAsync); the
 * partial sums are added on the host and finished with (r, s).  Arguments as ga_g16_prove. */
int ga_g16_prove_multi(ga_g16_pk* const* keys, uint32_t n, const void* w, const void* a, const void* b, const void* c,
                       uint64_t n_constraints, uint64_t nb_public, const void* r, const void* s, void* proof_out);

/* ---- proving-key files -----------------------------------------------------------------------
 * replaces: ProvingKey.ReadFrom / UnsafeReadFrom (marshal.go:305-373: compressed or uncompressed points, told apart by their
 * flag bits) and ReadDump (marshal.go:449-539: raw memory images), read straight into HBM: the file streams through pinned
 * staging buffers, points are decoded (big-endian -> Montgomery, on-curve check, square roots of compressed points) by a device
 * kernel, and the dump's slices are copied without any arithmetic.  The format is recognised from the stream (the 0xdeadbeef
 * marker of WriteDump).  Subgroup membership of G2 points is not checked (UnsafeReadFrom semantics).
 *   k_remove: the toRemove wire list of prove.go:231-235 -- it comes from the constraint system, not from the key file; NULL / 0
 *   for circuits without commitments.  precompute, shard_index, shard_count: as in ga_g16_key.
 * ga_g16_key_write_fd writes the host description of a key in the WriteTo (GA_KEY_FORMAT_COMPRESSED), WriteRawTo (RAW) or
 * WriteDump (DUMP) layout (marshal.go:231-300,378-445); points are encoded on the device. */
#define GA_KEY_FORMAT_COMPRESSED 0
#define GA_KEY_FORMAT_RAW 1
#define GA_KEY_FORMAT_DUMP 2
int ga_g16_pk_read_mem(ga_ctx* ctx, int curve, const uint8_t* data, size_t len, int32_t precompute, uint32_t shard_index,
                       uint32_t shard_count, const uint64_t* k_remove, uint64_t len_k_remove, ga_g16_pk** out, uint64_t* bytes_read);
int ga_g16_pk_read_fd(ga_ctx* ctx, int curve, int fd, int32_t precompute, uint32_t shard_index, uint32_t shard_count,
                      const uint64_t* k_remove, uint64_t len_k_remove, ga_g16_pk** out, uint64_t* bytes_read);
int ga_g16_key_write_fd(ga_ctx* ctx, const ga_g16_key* key, int format, int fd, uint64_t* bytes_written);
/* Proof.ReadFrom (marshal.go:62-86): Ar | Bs | Krs | u32 n | n commitments | CommitmentPok, compressed (WriteTo) or uncompressed
 * (WriteRawTo) points.  proof_out: Ar | Bs | Krs affine (Montgomery), commitments_out: room for max_commitments G1Affine. */
int ga_g16_proof_unmarshal(int curve, const uint8_t* data, size_t len, void* proof_out, void* commitments_out,
                           uint32_t max_commitments, uint32_t* n_commitments, void* pok_out, size_t* consumed);
/* one G1Affine / G2Affine from its wire encoding (curve.Decoder for a single point; host arithmetic) */
int ga_point_unmarshal(int curve, int group, const uint8_t* data, size_t len, void* affine_out, size_t* consumed);

/* Proof.WriteTo wire format (marshal.go:33-58, no commitments): compressed Ar | Bs | Krs | u32 0 | PoK(inf).
 * Returns the number of bytes written in *len (164 for BN254, 244 for BLS12-381). */
int ga_g16_proof_marshal(int curve, const void* proof, uint8_t* out, size_t cap, size_t* len);

/* ---- BSB22 commitments ------------------------------------------------------------------
 * replaces: the two commitment MSM blocks of the ICICLE prover -- pk.CommitmentKeys[i].Commit inside the solver hint
 * (prove.go:84, icicle.go:834-873) and ProveKnowledge after the solve (prove.go:112-117, icicle.go:904-944).
 * values = privateCommittedValues[i] (host, fr Montgomery, ck_len[i] elements).  Both MSMs run on the pinned bases with one
 * upload of the scalars; outputs are G1Affine (Montgomery): the commitment the hint hashes, and this commitment's proof of
 * knowledge (kept by the caller until all commitments are done, then folded). */
int ga_g16_commit(ga_g16_pk* pk, uint32_t index, const void* values, uint64_t n_values, void* commitment_out, void* pok_out);
/* proof.CommitmentPok.Fold(poks, challenge) (prove.go:127): sum_i challenge^i * poks[i]; host arithmetic (a handful of points).
 * poks: n G1Affine; challenge: fr Montgomery; out: G1Affine. */
int ga_g16_fold_pok(int curve, const void* poks, uint64_t n, const void* challenge, void* out);
/* fr.Hash(msg, dst, count) of gnark-crypto (field/hash ExpandMsgXmd with SHA-256, L = 16 + fr.Bytes = 48 bytes per element,
 * big-endian reduction mod r; same code shape as internal/smallfields/tinyfield/element.go:456-481): the commitment hint's
 * hash_to_field.New("bsb22-commitment") (prove.go:57-58,88-98) and the fold challenge fr.Hash(..., "G16-BSB22", 1) (:123).
 * out: count fr elements, Montgomery.  Host only. */
int ga_hash_to_field(int curve, const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, uint32_t count, void* out);
/* expand_message_xmd (RFC 9380 5.3.1, SHA-256) on its own, so that tests can pin it to std/hash/expand/expand_test.go:52-140 */
int ga_expand_message_xmd(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, size_t n, uint8_t* out);
/* Proof.WriteTo with commitments (marshal.go:33-58): Ar | Bs | Krs | u32be n | n compressed commitments | compressed PoK.
 * commitments: n G1Affine (may be NULL when n = 0), pok: G1Affine or NULL (= infinity). */
int ga_g16_proof_marshal_bsb22(int curve, const void* proof, const void* commitments, uint32_t n, const void* pok,
                               uint8_t* out, size_t cap, size_t* len);
/* Proof.WriteRawTo (marshal.go:25-30): the same layout with uncompressed points (G1 x | y, G2 x.A1 | x.A0 | y.A1 | y.A0). */
int ga_g16_proof_marshal_raw(int curve, const void* proof, const void* commitments, uint32_t n, const void* pok,
                             uint8_t* out, size_t cap, size_t* len);
/* G1Affine.Marshal() (uncompressed big-endian x | y; what SerializeCommitment hashes, constraint/commitment.go:76-89,
 * prove.go:88).  out: 64 bytes (BN254) / 96 bytes (BLS12-381). */
int ga_g1_marshal_uncompressed(int curve, const void* affine, uint8_t* out, size_t cap, size_t* len);

/* ---- profiling hooks (ICICLE_STEP_PROFILE analogue, icicle.go:72-75) ------------------------------------
 * When enabled, every kernel stage is bracketed by hipEvents on the stream it is launched on; ga_profile_read
 * returns "name=ms;name=ms;..." for the stages recorded since the last ga_profile_reset. */
int ga_profile_enable(ga_ctx* ctx, int on);
int ga_profile_reset(ga_ctx* ctx);
int ga_profile_read(ga_ctx* ctx, char* buf, size_t cap);

/* ---- test / bench support: on-device synthetic key material with known discrete logs --------------------
 * out[i] = [k_i]G (affine, Montgomery) where k_i = xoshiro-derived 64-bit values expanded from `seed`;
 * the same k_i are written (as canonical fr elements, 4 limbs) to dlogs_out (device) so that a test can check
 * MSM(s, P) == [sum s_i k_i] G with a field dot product (SURVEY 8c "known discrete log").  */
int ga_gen_bases(ga_ctx* ctx, int curve, int group, uint64_t seed, size_t n, void* bases_dev, void* dlogs_dev);
/* elements [first, first + n) of the same sequence: a rank of a multi-GPU run generates only its shard of a synthetic key */
int ga_gen_bases_at(ga_ctx* ctx, int curve, int group, uint64_t seed, uint64_t first, size_t n, void* bases_dev, void* dlogs_dev);
/* out[i] = uniform fr element (Montgomery) from a counter-based generator keyed by seed (device) */
int ga_gen_scalars(ga_ctx* ctx, int curve, uint64_t seed, size_t n, void* scalars_dev);
/* dot = sum a_i * b_i over fr; a Montgomery, b canonical (the dlogs above); result canonical LE (32 bytes, host) */
int ga_fr_dot(ga_ctx* ctx, int curve, const void* a_dev, const void* b_dev, size_t n, void* out_host);
/* [k]G for the group generator, k canonical LE 32 bytes (host arithmetic) -> Jacobian */
int ga_generator_mul(int curve, int group, const void* k_canonical_le32, void* out_jac);

/* integer-multiplier / FMA issue-rate microbenchmarks (SURVEY 8d asks for the v_mad_u64_u32 rate);
 * writes "name=Gops;..." */
int ga_microbench(ga_ctx* ctx, char* buf, size_t cap);
/* the shader clock the device holds right now, in MHz: one wave on a stream of its own compares the shader cycle counter with the
 * constant-rate one for `micros` microseconds.  Takes no context lock -- call it from a second thread while the load of interest runs
 * (tools/clock_probe.py: the bucket kernels run below the clock short microbenchmarks see). */
int ga_clock_probe(ga_ctx* ctx, uint32_t micros, double* mhz_out);

#ifdef __cplusplus
}
#endif
#endif /* GNARK_AMD_H */
