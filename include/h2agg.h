/* h2agg.h — C ABI of libh2agg.so: the MI355X (gfx950) backend for the "pure calculation context" hot
 * path of scroll-tech/halo2-snark-aggregator.
 *
 * The reference has no FFI of its own (it is 100 % Rust).  The boundary this header
 * defines is the one a GPU-backed implementation of the reference's plugin traits would bind: every
 * entry point names the reference interface (file:line under /root/reference) whose arithmetic it
 * replaces.  The Rust-side binding a maintainer would add is shown in INTEGRATION.md and
 * halo2-snark-aggregator_amd/rust-shim/.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary, caller owns every buffer;
 *   - return 0 (H2AGG_OK) on success, otherwise an H2AGG_ERR_* code with a message available from
 *     h2agg_last_error(); the Mock chips cannot construct their generic error type `E`, they panic —
 *     the shim turns a non-zero status into a panic (mock/arith/field.rs:113, mock/arith/ecc.rs:128);
 *   - a context is bound to ONE device and is single-thread-affine (the reference's chips are used
 *     from one thread: SURVEY.md §8b); one context per (host thread, GPU);
 *   - there is NO CPU mode: creating a context without a usable HIP device fails (H2AGG_ERR_HIP).
 *
 * Encodings (all little-endian, canonical = fully reduced integer, NOT Montgomery form)
 *   FR / FQ      32 bytes, integer < modulus (what halo2curves `to_repr()` yields)
 *   G1 affine    x || y, 64 bytes; identity = 64 zero bytes      (halo2curves G1Affine, `C`)
 *   G1 jacobian  x || y || z, 96 bytes; identity has z = 0       (halo2curves G1, `C::CurveExt`)
 *   Jacobian outputs are a valid representative of the result point, not a canonical one: compare
 *   after h2agg_g1_batch_to_affine (the reference compares `to_affine()` values too,
 *   halo2-snark-aggregator-circuit/src/verify_circuit.rs:180,200).
 */
#ifndef H2AGG_H
#define H2AGG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct h2agg_ctx h2agg_ctx;

enum {
    H2AGG_OK = 0,
    H2AGG_ERR_INVALID = 1,      /* bad argument (null pointer, unknown op, bad handle, bad window size) */
    H2AGG_ERR_DIV_ZERO = 2,     /* inversion of zero: reference panics, mock/arith/field.rs:113 */
    H2AGG_ERR_EMPTY = 3,        /* multi_exp of zero pairs: reference panics, mock/arith/ecc.rs:128 */
    H2AGG_ERR_HIP = 4,          /* HIP runtime / device failure */
    H2AGG_ERR_NONCANONICAL = 5, /* an input integer was >= its modulus */
    H2AGG_ERR_NOMEM = 6,
    H2AGG_ERR_BAD_POINT = 7,    /* a compressed point does not decode: "invalid point encoding in proof", transcript.rs:65-70;
                                   a partial accumulator received from a peer rank is not on the curve */
    H2AGG_ERR_PEER = 8          /* sharded aggregation: another rank failed (its code is in the message); this rank's inputs were fine */
};

/* field ops of h2agg_fr_batch_op */
enum { H2AGG_OP_ADD = 0, H2AGG_OP_SUB = 1, H2AGG_OP_MUL = 2, H2AGG_OP_SQR = 3, H2AGG_OP_INV = 4, H2AGG_OP_DIV = 5 };

/* ---- lifetime -------------------------------------------------------------------------------------
 * replaces: `MockEccChip::default()` / `MockFieldChip::default()` / `MockChipCtx::default()`
 * (halo2-snark-aggregator-circuit/src/verify_circuit.rs:115-118). */
int h2agg_create(int device_ordinal, h2agg_ctx** out);
void h2agg_destroy(h2agg_ctx* ctx);
const char* h2agg_last_error(const h2agg_ctx* ctx);
/* Use an existing HIP stream (hipStream_t) for every launch of this context; NULL = the context's own (a non-blocking stream:
 * it does NOT synchronise with the legacy default stream).  Device buffers handed to the asynchronous entry points must be
 * complete on THAT stream (or the device synchronised) before the call and must not change until the result has been joined:
 * the sort reads the scalars more than once.  Note that a framework's "default stream" usually has the handle 0 (= NULL here).
 * The call takes a few milliseconds: the context measures which of its other streams share a hardware queue with the new main
 * stream and hands their roles (tails, copies, ...) out again, so that none of them queues in front of the main stream's work. */
int h2agg_set_stream(h2agg_ctx* ctx, void* hip_stream);
/* Block until everything queued on the context's stream (and its tail streams) has finished.  Also where device-side
 * status raised by ASYNCHRONOUS calls surfaces: a non-canonical scalar handed to h2agg_g1_msm_device_async / _batch_async is
 * reported here (or by the next synchronous entry point, whichever comes first) as H2AGG_ERR_NONCANONICAL, once. */
int h2agg_synchronize(h2agg_ctx* ctx);
/* Library / device description, e.g. "h2agg 0.1 gfx950 cu=256"; valid until the context is destroyed. */
const char* h2agg_describe(h2agg_ctx* ctx);

/* ---- Fr batch ops (host buffers) ------------------------------------------------------------------
 * replaces: MockFieldChip::{add,sub,mul,square,div} element-wise over n operands
 * (halo2-snark-aggregator-api/src/mock/arith/field.rs:39-55, 98-122).  `b` is ignored for SQR / INV.
 * INV of 0 and DIV by 0 -> H2AGG_ERR_DIV_ZERO (reference: `invert().unwrap()` panics).  n == 0 is a no-op. */
int h2agg_fr_batch_op(h2agg_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out);
/* replaces: ArithFieldChip::pow_constant (arith/field.rs:83-104) over n bases with one exponent >= 1
 * (exponent 0 -> H2AGG_ERR_INVALID: the reference asserts). */
int h2agg_fr_batch_pow_constant(h2agg_ctx* ctx, const uint8_t* a, size_t n, uint64_t exponent, uint8_t* out);

/* replaces: ArithFieldChip::mul_add_accumulate default (Horner: acc = acc*b + v_i, acc_0 = 0)
 * (halo2-snark-aggregator-api/src/arith/field.rs:68-81). */
int h2agg_fr_mul_add_accumulate(h2agg_ctx* ctx, const uint8_t* v, size_t n, const uint8_t b[32], uint8_t out[32]);

/* replaces: MockFieldChip::sum_with_coeff_and_constant (acc = b + sum x_i*coeff_i)
 * (halo2-snark-aggregator-api/src/mock/arith/field.rs:124-135). */
int h2agg_fr_sum_with_coeff_and_constant(h2agg_ctx* ctx, const uint8_t* x, const uint8_t* coeff, size_t n,
                                         const uint8_t b[32], uint8_t out[32]);

/* ---- G1 batch ops (host buffers) ------------------------------------------------------------------
 * replaces: MockEccChip::add / sub (`*a + *b`, `*a - *b`) over n Jacobian pairs
 * (halo2-snark-aggregator-api/src/mock/arith/ecc.rs:30-46). */
int h2agg_g1_batch_add(h2agg_ctx* ctx, const uint8_t* a_jac, const uint8_t* b_jac, size_t n, int subtract,
                       uint8_t* out_jac);

/* replaces: MockEccChip::scalar_mul / scalar_mul_constant (`rhs * lhs`) over n (affine base, scalar) pairs
 * (halo2-snark-aggregator-api/src/mock/arith/ecc.rs:88-104); also the loop body of
 * assign_instance_commitment (halo2-snark-aggregator-api/src/systems/halo2/verify.rs:623-635). */
int h2agg_g1_batch_scalar_mul(h2agg_ctx* ctx, const uint8_t* bases_aff, const uint8_t* scalars, size_t n,
                              uint8_t* out_jac);

/* replaces: MockEccChip::to_value = `to_affine` (halo2-snark-aggregator-api/src/mock/arith/ecc.rs:64-66). */
int h2agg_g1_batch_to_affine(h2agg_ctx* ctx, const uint8_t* in_jac, size_t n, uint8_t* out_aff);

/* h2agg_g1_batch_to_affine for Jacobian points already in device memory (e.g. the outputs of
 * h2agg_g1_msm_device_async: N instance-column commitments become affine with one launch and one download). */
int h2agg_g1_batch_to_affine_device(h2agg_ctx* ctx, const uint8_t* d_in_jac, size_t n, uint8_t* out_aff);
/* ---- proof wire format (the data format on the input side of the path) ----------
 * replaces: the `C::from_bytes(&compressed)` of TranscriptRead::read_point / read_constant_point
 * (halo2-snark-aggregator-api/src/systems/halo2/transcript.rs:56-99), for all the points of a proof (or of N proofs) at once.
 * Encoding: 32 bytes little-endian x, parity of y in bit 7 of byte 31, identity = 32 zero bytes (halo2curves 0.2.1, recalled
 * from upstream: the crate is not vendored in the reference).  out_aff: canonical affine, 64 B each; ok (optional): one byte
 * per point, 1 = decoded.  Any point that does not decode (x >= p, or x^3 + 3 not a square) -> H2AGG_ERR_BAD_POINT with that
 * point's output zeroed and ok = 0; the others are still written. */
int h2agg_g1_batch_decompress(h2agg_ctx* ctx, const uint8_t* in, size_t n, uint8_t* out_aff, uint8_t* ok);
/* the inverse (G1Affine::to_bytes): canonical affine points -> 32-byte encodings */
int h2agg_g1_batch_compress(h2agg_ctx* ctx, const uint8_t* aff, size_t n, uint8_t* out);
/* Sum of n Jacobian points (the local fold after the multi-GPU all-gather of partial (W_x, W_g),
 * SURVEY.md §8e; arithmetic = MockEccChip::add, mock/arith/ecc.rs:30-37).  n == 0 -> identity. */
int h2agg_g1_sum(h2agg_ctx* ctx, const uint8_t* in_jac, size_t n, uint8_t out_jac[96]);

/* Page-locked host memory for the buffers handed to the host-buffer entry points (h2agg_g1_msm, h2agg_eval_flat, ...):
 * from pageable memory the 96 B/point cross PCIe through the runtime's bounce buffers (~15 GB/s); from these buffers the
 * copies run at link rate and asynchronously.  A binding marshals its points / scalars straight into them. */
int h2agg_host_alloc(h2agg_ctx* ctx, size_t bytes, void** out);
int h2agg_host_free(h2agg_ctx* ctx, void* p);
/* ---- multi-scalar multiplication ------------------------------------------------------------------
 * replaces: MockEccChip::multi_exp / ArithEccChip::multi_exp default — sum_i scalars[i] * bases[i]
 * (halo2-snark-aggregator-api/src/mock/arith/ecc.rs:106-129, .../arith/ecc.rs:42-60), called from
 * EvaluationQuerySchema::eval (halo2-snark-aggregator-api/src/systems/halo2/evaluation.rs:197).
 * The reference computes n independent double-and-add products; this computes the same group element
 * with a Pippenger bucket method on the GPU.  n == 0 -> H2AGG_ERR_EMPTY (the reference panics); the
 * output buffer is then set to the identity so a caller that prefers "empty sum = identity" can
 * ignore that one status. */
int h2agg_g1_msm(h2agg_ctx* ctx, const uint8_t* bases_aff, const uint8_t* scalars, size_t n, uint8_t out_jac[96]);
/* The same with the points as the trait hands them over: `Vec<C::CurveExt>`, projective (x || y || z, 96 B each, canonical;
 * z = 0: identity).  They are normalised on the device (one inversion per 8 points per lane), so the binding does not
 * have to run `batch_normalize` over a million points on one host core first (mock/arith/ecc.rs:106-129 takes
 * `points: Vec<Self::AssignedPoint>` = `C::CurveExt`); 128 instead of 96 bytes per point cross PCIe. */
int h2agg_g1_msm_jac(h2agg_ctx* ctx, const uint8_t* points_jac, const uint8_t* scalars, size_t n, uint8_t out_jac[96]);

/* replaces: eval()'s flat tail — multi_exp over the entries that carry a scalar, then `pchip.add` of
 * every scalar-less point (halo2-snark-aggregator-api/src/systems/halo2/evaluation.rs:189-200).
 * has_scalar[i] != 0 marks entries with a scalar.  No scalar-carrying entry -> H2AGG_ERR_EMPTY. */
int h2agg_eval_flat(h2agg_ctx* ctx, const uint8_t* pts_aff, const uint8_t* scalars, const uint8_t* has_scalar,
                    size_t n, uint8_t out_jac[96]);

/* ---- device-resident bases / scalars (SRS-style fixed bases; inputs stay in HBM) ------------------
 * A base table lives on the device in Montgomery form (64 B / point).  Used for the instance-commitment
 * MSM against `params.g_lagrange` (halo2-snark-aggregator-api/src/systems/halo2/verify.rs:623-635) and
 * for benchmarking with inputs already resident. */
int h2agg_bases_upload(h2agg_ctx* ctx, const uint8_t* bases_aff, size_t n, uint64_t* handle_out);
/* Fixed-base acceleration of a resident table (an SRS such as ParamsKZG.g_lagrange, the bases of
 * assign_instance_commitment, verify.rs:574-649): stores 2^(c*w) * P_i for every digit position w beside the table
 * (ceil(255 / c) x 64 B per point).  Every later MSM over this handle (h2agg_g1_msm_preloaded / _device / _device_async /
 * _device_batch_async, h2agg_instance_commitment) then drops all digits of a scalar into ONE bucket set: one bucket
 * reduction instead of one per window, no doubling chain, and a wider window.  window_bits: 0 = chosen from the table
 * size, else 4..20.  Tables up to ~2^18 points (ceil(255 / c) * n <= 2^22) take the width that minimises the work; larger
 * ones, up to 2^22 points, take c = 20: 13 levels = 832 B per point beside the table — 3.25 GiB for a 2^22-point g_lagrange,
 * built in ~80 ms — and a (level, point) sort of their own (csrc/fb_sort_kernels.hpp).  An explicit width other than 20 whose
 * levels exceed 2^22 entries, and tables beyond 2^22 points, are refused (H2AGG_ERR_INVALID) and keep the ordinary path.
 * window_bits = 0 is the opportunistic call: it returns H2AGG_ERR_NOMEM (table untouched, ordinary path kept) when the
 * levels would take more than a quarter of the device memory that is free; an explicit width allocates regardless.
 * Results are the same points (h2agg_msm_configure's explicit window_bits disables the fast path).  The c = 20 levels are
 * used by MSMs their own sort takes — one MSM at a time (batches of >= 2^18 scalars run one after another), no sort /
 * segment knobs configured; any other call over such a table runs the ordinary path and ignores the levels. */
int h2agg_bases_precompute(h2agg_ctx* ctx, uint64_t bases_handle, int window_bits);
/* bases[i] = k_i * G for n canonical Fr scalars held in DEVICE memory (workload generation: the
 * expected MSM is then (sum k_i * s_i) * G).  Arithmetic = scalar_mul_constant + to_affine. */
int h2agg_bases_generate(h2agg_ctx* ctx, const void* d_k_scalars, size_t n, uint64_t* handle_out);
int h2agg_bases_download(h2agg_ctx* ctx, uint64_t handle, size_t first, size_t n, uint8_t* out_aff);
int h2agg_bases_free(h2agg_ctx* ctx, uint64_t handle);

/* scalars in host memory, bases preloaded; uses the first n bases of the table */
int h2agg_g1_msm_preloaded(h2agg_ctx* ctx, uint64_t bases_handle, const uint8_t* scalars, size_t n,
                           uint8_t out_jac[96]);
/* replaces: the per-column body of assign_instance_commitment
 * (halo2-snark-aggregator-api/src/systems/halo2/verify.rs:601-603 bound, :623-635 sum inst_i * g_lagrange[i],
 * :637-640 identity for an empty column) against a preloaded `params.g_lagrange` table.  max_len =
 * params.n() - (blinding_factors + 1); len > max_len -> H2AGG_ERR_INVALID (the reference's assert!). */
int h2agg_instance_commitment(h2agg_ctx* ctx, uint64_t g_lagrange_handle, const uint8_t* instance, size_t len,
                              size_t max_len, uint8_t out_jac[96]);
/* scalars in DEVICE memory (n x 32 B canonical); synchronous, result to host */
int h2agg_g1_msm_device(h2agg_ctx* ctx, uint64_t bases_handle, const void* d_scalars, size_t n,
                        uint8_t out_jac[96]);
/* as above but asynchronous on the context's stream: the 96-byte canonical Jacobian result is written
 * to DEVICE memory at d_out_jac; nothing is copied to the host and the call does not synchronise. */
int h2agg_g1_msm_device_async(h2agg_ctx* ctx, uint64_t bases_handle, const void* d_scalars, size_t n,
                              void* d_out_jac);
/* `batch` MSMs over the SAME first n bases of the table, scalars laid out [batch][n] in device memory, results
 * d_out_jac[96 * q].  One set of launches computes all of them (the MSMs become extra windows of one bucket sort), so
 * many medium-sized MSMs fill the chip like one large one: the N instance-column commitments of N proofs against
 * ParamsKZG.g_lagrange (assign_instance_commitment, verify.rs:574-649) are this shape. */
int h2agg_g1_msm_device_batch_async(h2agg_ctx* ctx, uint64_t bases_handle, const void* d_scalars, size_t n, size_t batch,
                                    void* d_out_jac);

/* ---- EvaluationQuerySchema (the entry point of the hot path) ---------------------------------------
 * replaces: the AST of halo2-snark-aggregator-api/src/systems/halo2/evaluation.rs:15-27 and its
 * constructors `commit!` / `eval!` / `scalar!` (:41-60), `impl Add` / `impl Mul` (:62-84), `estimate`
 * (:295-330) and `eval` (:172-203, with eval_prepare :205-293).  A schema object is an arena of nodes
 * bound to one context; node ids are indices into it.  The host walks the tree exactly as eval_prepare
 * does but only records the schip.mul / schip.add calls on a tape; the tape runs on the GPU, its
 * results feed the multi_exp directly.  Keys are NUL-terminated strings (the reference's `String` keys). */
typedef struct h2agg_schema h2agg_schema;
int h2agg_schema_create(h2agg_ctx* ctx, h2agg_schema** out);
void h2agg_schema_destroy(h2agg_schema* s);
int h2agg_schema_node_commitment(h2agg_schema* s, const char* key, const uint8_t point_aff[64], uint32_t* node_out);
int h2agg_schema_node_eval(h2agg_schema* s, const uint8_t eval[32], uint32_t* node_out);     /* eval!(cq)   */
int h2agg_schema_node_scalar(h2agg_schema* s, const uint8_t scalar[32], uint32_t* node_out); /* scalar!(s)  */
int h2agg_schema_node_add(h2agg_schema* s, uint32_t l, uint32_t r, uint32_t* node_out);      /* l + r       */
int h2agg_schema_node_mul(h2agg_schema* s, uint32_t l, uint32_t r, uint32_t* node_out);      /* l * r       */
int h2agg_schema_estimate(h2agg_schema* s, uint32_t node, size_t* out);                      /* estimate(None) */
/* n x EvaluationQuery::new (evaluation.rs:100-118): nodes_out[i] = commit!(cq_i) + eval!(cq_i). */
int h2agg_schema_evaluation_queries(h2agg_schema* s, size_t n, const char* const* keys, const uint8_t* commitments,
                                    const uint8_t* evals, uint32_t* nodes_out);
/* Replace the commitment of an EvaluationQuery node created by h2agg_schema_evaluation_queries.  Lets the host build
 * the schema while the device is still computing that commitment (assign_instance_commitment's MSM,
 * verify.rs:574-649, whose result is the first query of every proof). */
int h2agg_schema_query_set_commitment(h2agg_schema* s, uint32_t query_node, const uint8_t point_aff[64]);
/* replaces: VerifierParams::get_point_schemas + batch_multi_open_proofs
 * (halo2-snark-aggregator-api/src/systems/halo2/multiopen.rs:23-102): queries (rotation, evaluation point,
 * schema node) in VerifierParams::queries order are grouped by rotation in first-seen order, Horner-folded
 * in v per group and in u over groups; `w` holds one W commitment per group (nw must equal the number of
 * groups — the reference's assert_eq!, multiopen.rs:48).  Keys of the W commitments: "{key}_w{i}". */
int h2agg_schema_batch_multi_open(h2agg_schema* s, const char* key, size_t nq, const int32_t* rotations,
                                  const uint8_t* points, const uint32_t* query_nodes, size_t nw, const uint8_t* w,
                                  const uint8_t v[32], const uint8_t u[32], uint32_t* w_x_out, uint32_t* w_g_out);
/* eval(): out_jac = multi_exp(points with scalars) + sum(points without); *has_scalar / out_scalar = the
 * accumulated pure-scalar term (key ""), None -> *has_scalar = 0.  A Mul whose both sides hold
 * commitments, or an Add of non-singleton scalar sides, fails with H2AGG_ERR_INVALID (the reference's
 * assert!, evaluation.rs:237-238,282); no scalar-carrying commitment -> H2AGG_ERR_EMPTY (its panic). */
int h2agg_schema_eval(h2agg_schema* s, uint32_t node, uint8_t out_jac[96], int* has_scalar, uint8_t out_scalar[32]);
/* replaces: evaluate_multiopen_proof without the print-only pairing
 * (halo2-snark-aggregator-api/src/systems/halo2/verify.rs:705-731): left = eval(w_x) + e_x*G,
 * right = eval(w_g) - e_g*G, both `to_value`d; output order = verify_circuit_final_pair.data
 * (halo2-snark-aggregator-circuit/src/fs.rs:187-190). */
int h2agg_evaluate_multiopen_proof(h2agg_schema* s, uint32_t w_x, uint32_t w_g, uint8_t left_aff[64],
                                   uint8_t right_aff[64]);
/* The host half of h2agg_evaluate_multiopen_proof(s, w_x, w_g, ...) done ahead of time: both eval_prepare walks
 * (evaluation.rs:207-330) recorded on the Fr tape and the tape ordered by dependency level — no device work, no wait.
 * A following h2agg_evaluate_multiopen_proof with the same roots on an otherwise unchanged schema starts at the upload.
 * Commitment POINTS may still be replaced in between (h2agg_schema_query_set_commitment: the evaluation reads them
 * when it runs), so a host can prepare while the device is still computing the instance commitments (verify.rs:574-649).
 * Any other change to the schema (new nodes, another eval) simply makes the evaluation do its own host half again.
 * Same errors as the evaluation would report for the host half (H2AGG_ERR_INVALID, H2AGG_ERR_EMPTY). */
int h2agg_evaluate_multiopen_prepare(h2agg_schema* s, uint32_t w_x, uint32_t w_g);
/* names returned by the last eval (evaluation.rs:183) / points_wx ++ points_wg (verify.rs:711-712), and the
 * length MockChipCtx::point_list would have after the last multi_exp (mock/arith/ecc.rs:112-116). */
size_t h2agg_schema_name_count(h2agg_schema* s);
const char* h2agg_schema_name(h2agg_schema* s, size_t i);
/* all names in one call, each followed by '\n'; returns the byte count needed (nothing is written if cap is smaller) */
size_t h2agg_schema_names_joined(h2agg_schema* s, char* out, size_t cap);
size_t h2agg_schema_point_list_len(h2agg_schema* s);

/* ---- transcript side: Poseidon sponge, PoseidonEncode, PoseidonTranscriptRead ------------
 * replaces: PoseidonChip::{update, squeeze} (halo2-snark-aggregator-api/src/hash/poseidon.rs:167-191, permutation :193-230)
 * with T = 9, RATE = 8, R_F = 8, R_P = 63 (halo2-snark-aggregator-circuit/src/verify_circuit.rs:127-135), for `nproofs`
 * independent sponges at once (a sponge is a sequential chain; the batch is what runs in parallel).  elems: [nproofs][nelem]
 * canonical Fr; upto[q] = number of elements absorbed before squeeze q (non-decreasing, <= nelem; equal consecutive values
 * squeeze again without absorbing); out: [nproofs][nsq] challenges.  An element >= r -> H2AGG_ERR_NONCANONICAL. */
int h2agg_poseidon_squeeze_batch(h2agg_ctx* ctx, const uint8_t* elems, size_t nproofs, size_t nelem, const uint32_t* upto,
                                 size_t nsq, uint8_t* out);
/* replaces: PoseidonTranscriptRead (halo2-snark-aggregator-api/src/systems/halo2/transcript.rs:10-179) + PoseidonEncode
 * (mock/transcript_encode.rs:28-74) over `nproofs` proofs of ONE layout.  `script` is the sequence of calls the reader
 * receives, one character each: 'P' read_point, 'S' read_scalar, 'Q' squeeze_challenge_scalar, 'C' common_scalar(the next
 * of `consts`), 'X' common_point(the next of this proof's `ext_points_aff`, e.g. its instance commitments, verify.rs:77-97).
 * proof_len must equal 32 x (number of 'P' and 'S'): the reader's read_exact.  points_out: [nproofs][#P] decoded points
 * (canonical affine); challenges_out: [nproofs][#Q].  A point that does not decode -> H2AGG_ERR_BAD_POINT ("invalid point
 * encoding in proof"), a scalar >= r -> H2AGG_ERR_NONCANONICAL ("invalid field element encoding in proof"). */
/* The sponges are sequential chains (~136 permutations per proof): on the device a chain runs at one wave's latency whatever
 * the batch, on a host core ~an order of magnitude faster — so small batches run on host threads (one proof per thread, the
 * same generated constants, bit-identical challenges) and large ones on the device.  backend: 0 = auto (by batch size and
 * usable host threads; environment H2AGG_TRANSCRIPT=device|host overrides), 1 = device, 2 = host.  Applies to
 * h2agg_poseidon_squeeze_batch, h2agg_transcript_read_batch and h2agg_verify_aggregation(_ex) on this context.
 * h2agg_poseidon_squeeze_batch_host: the host backend on its own (no context, no device): max_threads 0 = all usable.
 * h2agg_host_threads: worker threads the library may use (affinity mask, cgroup CPU quota, H2AGG_HOST_THREADS; <= 32). */
int h2agg_transcript_configure(h2agg_ctx* ctx, int backend);
int h2agg_poseidon_squeeze_batch_host(const uint8_t* elems, size_t nproofs, size_t nelem, const uint32_t* upto, size_t nsq,
                                      uint8_t* out, int max_threads);
int h2agg_host_threads(void);
/* arithmetic of the host sponge: 1 = AVX-512 IFMA (chosen at run time when the CPU has avx512ifma), 0 = portable 4 x 64-bit.
 * In h2agg_poseidon_squeeze_batch_host, max_threads bit 16 forces the portable kernel and bit 17 the IFMA one for that call
 * (bits 0..15 stay the thread cap): the two are differential partners in tests/test_host_sponge.py. */
int h2agg_host_sponge_kind(void);
int h2agg_transcript_read_batch(h2agg_ctx* ctx, const uint8_t* proofs, size_t proof_len, size_t nproofs, const char* script,
                                size_t script_len, const uint8_t* consts, size_t nconsts, const uint8_t* ext_points_aff,
                                size_t next, uint8_t* points_out, uint8_t* challenges_out);

/* ---- verifier-params pipeline + aggregation driver (SURVEY.md 8(f) row 1 and the caller of the hot path) ----------
 * replaces, for the pure-calculation context: VerifierParamsBuilder::build_params (halo2-snark-aggregator-api/src/systems/
 * halo2/verify.rs:342-571), VerifierParams::queries (params.rs:74-224) with lagrange.rs / expression.rs / permutation.rs /
 * lookup.rs / vanish.rs, assign_instance_commitment (verify.rs:574-649), verify_aggregation_proofs_in_chip (verify.rs:
 * 835-942) and the pairing of calc_verify_circuit_final_pair (halo2-snark-aggregator-circuit/src/verify_circuit.rs:114-201).
 * The host replays the reference's control flow; every field / group operation runs on the device (instance MSMs, point
 * decompression, one Poseidon sponge per proof, ONE Fr tape for all expressions and eval_prepare scalars, the two
 * multi_exps).
 *
 * h2agg_vk: what the path reads of halo2_proofs' VerifyingKey / ConstraintSystem (unvendored), serialized little-endian:
 *   u32 magic 0x4B563248 ("H2VK"), u32 version 1,
 *   u32 k, num_advice_columns, num_instance_columns, num_challenges, degree (cs.degree()), blinding_factors,
 *   advice_column_phase: num_advice bytes; challenge_phase: num_challenges bytes           (each padded to 4 bytes)
 *   advice_queries, instance_queries, fixed_queries: u32 count, then (u32 column, i32 rotation) each
 *   permutation columns: u32 count, then (u32 kind: 0 advice / 1 fixed / 2 instance, u32 index) each
 *   fixed commitments: u32 count + 64 B each; permutation commitments: u32 count + 64 B each   (canonical affine)
 *   vk transcript scalar: 32 B  (from_bytes_wide(blake2b("Halo2-Verify-Key", pinned vk)), verify.rs:57-70)
 *   gates: u32 count; per gate u32 polys; per poly an expression;  lookups: u32 count; per lookup: u32 n + input
 *   expressions, u32 n + table expressions
 *   expression = u32 byte length + postfix bytecode (padded to 4): 0 CONST(32 B) 1 FIXED(u32 query index) 2 ADVICE(u32)
 *   3 INSTANCE(u32) 4 CHALLENGE(u32) 5 NEG 6 SUM 7 PRODUCT 8 SCALED(32 B)   (halo2_proofs::plonk::Expression; selectors
 *   are gone after keygen: expression.rs:33-35)
 * h2agg_circuit_proofs = CircuitProof (verify.rs:769-783): one verifying key with its proofs; every proof carries ONE
 * inner proof (instances[i] = that proof's instance columns back to back, instance_lens[i * num_instance_columns + col]
 * values each); all transcripts of a circuit have one length.  name: CircuitProof::name (keys are "{name}_p{i}").
 * g_lagrange: a base-table handle holding params.g_lagrange (h2agg_bases_upload; h2agg_bases_precompute speeds it up).
 * Outputs: the final pair (W_x, W_g) as fs.rs:187-190 lays it out; lambda_out (optional) = the aggregation challenge;
 * with s_g2 / g2 (both or none) *pairing_ok = e(W_x, s_g2) * e(W_g, -g2) == 1.
 * Errors: a proof point that does not decode -> H2AGG_ERR_BAD_POINT; a scalar >= r -> H2AGG_ERR_NONCANONICAL; a proof whose
 * length does not fit the key, a W count different from the number of rotation groups (multiopen.rs:48), an instance
 * column longer than n - (blinding_factors + 1) (verify.rs:601-603) -> H2AGG_ERR_INVALID; inversion of zero in the
 * Lagrange / vanishing terms -> H2AGG_ERR_DIV_ZERO (the reference's invert().unwrap()). */
typedef struct h2agg_vk h2agg_vk;
int h2agg_vk_create(h2agg_ctx* ctx, const uint8_t* blob, size_t len, h2agg_vk** out);
void h2agg_vk_destroy(h2agg_vk* vk);
typedef struct {
    const h2agg_vk* vk;
    const char* name;
    uint64_t g_lagrange;
    size_t nproofs;
    const uint8_t* const* transcripts;
    const size_t* transcript_lens;
    const uint8_t* const* instances;
    const uint32_t* instance_lens;
} h2agg_circuit_proofs;
int h2agg_verify_aggregation(h2agg_ctx* ctx, const h2agg_circuit_proofs* circuits, size_t ncircuits, const uint8_t* s_g2,
                             const uint8_t* g2, uint8_t left_aff[64], uint8_t right_aff[64], uint8_t lambda_out[32],
                             int* pairing_ok);
/* as above, plus the fourth return value of verify_aggregation_proofs_in_chip (`commits`: every proof's advice
 * commitments, halo2-snark-aggregator-api/src/systems/halo2/verify.rs:852-856,927-939, which the production caller hands to
 * `coherent`, halo2-snark-aggregator-circuit/src/verify_circuit.rs:487-492): advice_out (optional) receives, in aggregation
 * order, num_advice_columns x 64 B per proof (canonical affine, column order); advice_cap = bytes available (too small ->
 * H2AGG_ERR_INVALID before any work). */
int h2agg_verify_aggregation_ex(h2agg_ctx* ctx, const h2agg_circuit_proofs* circuits, size_t ncircuits, const uint8_t* s_g2,
                                const uint8_t* g2, uint8_t left_aff[64], uint8_t right_aff[64], uint8_t lambda_out[32],
                                int* pairing_ok, uint8_t* advice_out, size_t advice_cap);
/* The same aggregation with its proofs SHARDED over `world` ranks, one GPU each (first grain; BASELINE.json
 * configs[3], configs[4]).  This rank passes ITS proofs in `circuits` (any subset, typically round-robin; a rank may hold none)
 * and says where each sits in the aggregation order: global_index[j] for the j-th local proof, counting circuits in order and
 * proofs within a circuit in order; total_proofs = N, the length of the whole aggregation.  Two exchanges, both inside the call:
 *   1. every proof's last squeeze (32 B) goes to every rank, which absorbs them in aggregation order and squeezes the SAME
 *      lambda as the one-rank call (halo2-snark-aggregator-api/src/systems/halo2/verify.rs:909-913, :924);
 *   2. the rank's partial pair  sum_j lambda^(N-1-global_index[j]) * (W_x_j, W_g_j)  (the fold of verify.rs:926-938, which is
 *      linear) goes to every rank and is summed with the group law (RCCL has no reduction over group elements).
 * Transport: `allgather` (the host's own: MPI, a torch.distributed store, sockets — send = this rank's `bytes`, recv = [world]
 * [bytes] in rank order, returns 0) or, if NULL, the context's RCCL communicator over xGMI (h2agg_comm_init_rank with the same
 * rank / world).  Every rank returns the same pair, lambda and pairing verdict as h2agg_verify_aggregation on all N proofs in
 * one context, bit for bit.  advice_out: this rank's proofs only, local order.  Errors as h2agg_verify_aggregation; two ranks
 * claiming one position, or a position nobody holds -> H2AGG_ERR_INVALID on every rank.
 * No rank waits for a rank that failed: once its shard description is accepted a rank enters BOTH exchanges whatever happens
 * to it locally (a proof that does not decode, a non-canonical scalar, no memory, advice_cap too small) — each payload
 * carries a status word — and every rank leaves the call at the same exchange: the failing rank with its own error, the
 * others with H2AGG_ERR_PEER.  (A rank whose shard DESCRIPTION is refused — bad rank / world, repeated global_index, no
 * transport — returns before any exchange: that is a caller's bug, the same on every rank of a correct launch.)  Trust: the
 * squeezes and partial pairs of the other ranks arrive through the transport; partial pairs are checked to be canonical
 * points of the curve (H2AGG_ERR_NONCANONICAL / H2AGG_ERR_BAD_POINT), but a rank that LIES within those bounds changes the
 * result — ranks and transport are inside the verifier's trust boundary, as the threads of the reference's one process are. */
typedef int (*h2agg_allgather_fn)(void* user, const void* send, size_t bytes, void* recv);
typedef struct {
    uint32_t rank, world;
    size_t total_proofs;
    const uint32_t* global_index;
    h2agg_allgather_fn allgather;
    void* user;
} h2agg_shard;
int h2agg_verify_aggregation_sharded(h2agg_ctx* ctx, const h2agg_circuit_proofs* circuits, size_t ncircuits,
                                     const h2agg_shard* shard, const uint8_t* s_g2, const uint8_t* g2, uint8_t left_aff[64],
                                     uint8_t right_aff[64], uint8_t lambda_out[32], int* pairing_ok, uint8_t* advice_out,
                                     size_t advice_cap);
/* The host-side recording of an aggregation (every proof's queries of params.rs:74-224, the multiopen fold, both eval_prepare
 * walks of evaluation.rs:205-293) depends on the SHAPE of the call only — which keys, how many proofs of each, their length —
 * so a context keeps the last few recordings and a later call of the same shape only refills the proof scalars, challenges and
 * commitments before running the same tape and multi_exps on them (every value is still computed from that call's inputs).
 * This reports how often that happened; any pointer may be NULL.  h2agg_debug_configure(ctx, "plan_cache", 0) records every
 * call afresh. */
int h2agg_verify_plan_stats(h2agg_ctx* ctx, uint64_t* hits, uint64_t* misses, uint64_t* plans_kept);
/* After h2agg_debug_configure(ctx, "phases", 1): the wall-clock split of the context's last h2agg_verify_aggregation* call as
 * one line — " name=milliseconds" per phase in order, then the CPU the calling thread started and ended on and how often it was
 * preempted, then every host sponge chain's start offset, run time (microseconds) and CPU.  For latency reports (which phase a
 * slow call spent its time in); costs a few clock reads per call.  Owned by the context, valid until its next call. */
const char* h2agg_last_phases(h2agg_ctx* ctx);

/* ---- multi-GPU exchange -----------------------------------------------------------
 * The one collective of a sharded aggregation: every rank holds partial accumulators (the sharded form of the fold
 * `acc = acc * lambda + proof`, halo2-snark-aggregator-api/src/systems/halo2/verify.rs:926-938, evaluated per shard);
 * they are all-gathered over RCCL / xGMI and summed with the group law on every rank (RCCL has no reduction over group
 * elements), for the caller halo2-snark-aggregator-circuit/src/verify_circuit.rs:114-201.  RCCL is dlopen'ed at the
 * first call.
 *   one process per GPU : rank 0 calls h2agg_comm_unique_id and hands the 128 bytes to the other ranks by the host's own
 *                         means; every rank calls h2agg_comm_init_rank on its context; then
 *                         h2agg_allgather_add_points(&ctx, 1, my_partials, npts, out).
 *   one process, N GPUs : h2agg_comm_create(devices, n, ctxs) creates the N contexts and their communicator
 *                         (ncclCommInitAll); h2agg_allgather_add_points(ctxs, n, partials[rank][npts], npts, out).
 * partial_jac: canonical Jacobian points (96 B); out_aff[64 * k] = to_affine(sum over ranks of point k), identical on
 * every rank.  The communicator is released by h2agg_destroy. */
int h2agg_comm_unique_id(uint8_t out[128]);
int h2agg_comm_init_rank(h2agg_ctx* ctx, const uint8_t id[128], int rank, int nranks);
int h2agg_comm_create(const int* devices, int ndev, h2agg_ctx** ctxs_out);
int h2agg_comm_size(h2agg_ctx* ctx);   /* 0 = no communicator */
int h2agg_comm_rank(h2agg_ctx* ctx);   /* -1 = no communicator */
/* why the last context-less comm call failed (h2agg_comm_unique_id, or h2agg_comm_create before a context exists):
 * e.g. the dlopen error of librccl.  Thread-local storage; valid until the thread's next call of this function. */
const char* h2agg_comm_last_error(void);
int h2agg_allgather_add_points(h2agg_ctx** ctxs, int nctx, const uint8_t* partial_jac, size_t npts, uint8_t* out_aff);

/* ---- pairing check (host arithmetic, no device work) ---------------------------
 * replaces: `E::multi_miller_loop(&[(&left_v, &s_g2_prepared), (&right_v, &n_g2_prepared)]).final_exponentiation()
 * .is_identity()` (halo2-snark-aggregator-api/src/systems/halo2/verify.rs:733-739) and the production assert
 * (halo2-snark-aggregator-circuit/src/verify_circuit.rs:175-199): BN254 optimal-ate pairing, exact final exponentiation.
 * Encodings: G1 affine as everywhere (64 B); G2 affine = x.c0 || x.c1 || y.c0 || y.c1 (128 B, canonical little-endian,
 * halo2curves G2Affine { x: Fq2 { c0, c1 }, y }; identity = 128 zero bytes).  Points are validated: coordinate >= p ->
 * H2AGG_ERR_NONCANONICAL; not on the curve / twist, or a G2 point outside the order-r subgroup -> H2AGG_ERR_BAD_POINT.
 * `ctx` only receives the error message and may be NULL.
 * h2agg_pairing_check: *ok = (prod_i e(g1_i, g2_i) == 1), the shape of EIP-197's precompile (n == 0 -> 1).
 * h2agg_pairing_product: the GT element itself, 12 x 32 B canonical in tower order (c0.c0.c0, c0.c0.c1, c0.c1.c0, ...,
 *   c1.c2.c1; Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - (9 + u)), Fq2 = Fq[u]/(u^2 + 1)).
 * h2agg_final_pair_check: *ok = (e(left, s_g2) * e(right, -g2) == 1) — `params.s_g2()`, `params.g2()` as the caller
 *   holds them (ParamsKZG); the negation of g2 happens inside, as in the reference. */
int h2agg_pairing_check(h2agg_ctx* ctx, const uint8_t* g1_aff, const uint8_t* g2_aff, size_t n, int* ok);
/* 64-byte compressed G2 points (how ParamsKZG::write stores `g2` / `s_g2`; read back at halo2-snark-aggregator-circuit/
 * src/fs.rs:49-55 through halo2_proofs) -> the 128-byte affine form above.  Encoding recalled from halo2curves 0.2.1:
 * x.c0 || x.c1 little-endian, bit 7 of byte 63 = parity of y.c0, identity = zeros.  Host arithmetic; ctx may be NULL. */
int h2agg_g2_batch_decompress(h2agg_ctx* ctx, const uint8_t* in, size_t n, uint8_t* out_aff);
int h2agg_pairing_product(h2agg_ctx* ctx, const uint8_t* g1_aff, const uint8_t* g2_aff, size_t n, uint8_t out_gt[384]);
int h2agg_final_pair_check(h2agg_ctx* ctx, const uint8_t left_aff[64], const uint8_t right_aff[64],
                           const uint8_t s_g2[128], const uint8_t g2[128], int* ok);

/* ---- Fr expression tape ------------------------------------------------------
 * A straight-line program over Fr, run on the device by the interpreter EvaluationQuerySchema::eval records into:
 * registers 0 .. nconst-1 are the inputs (canonical, 32 B each), register nconst + k is the result of op k;
 * ops3 = nops x {opcode, a, b} (u32 each; opcode 0 = mul, 1 = add, 2 = sub, 3 = inverse of a (b ignored; 1/0 -> H2AGG_ERR_DIV_ZERO, the `invert().unwrap()` of MockFieldChip::div); a, b < nconst + k).  Independent ops run in
 * parallel, a chain costs one multiplication latency per link.  out_regs selects the nout registers returned in `out`
 * (canonical).  This is the shape of the verifier's scalar-side expressions (halo2-snark-aggregator-api/src/systems/halo2/
 * {params,lookup,permutation,vanish,lagrange,expression}.rs evaluate such programs through ArithFieldChip). */
int h2agg_fr_tape_eval(h2agg_ctx* ctx, const uint8_t* consts, size_t nconst, const uint32_t* ops3, size_t nops,
                       const uint32_t* out_regs, size_t nout, uint8_t* out);

/* ---- tuning / measurement -------------------------------------------------------------------------
 * window_bits: Pippenger window c in [2, 20]; 0 = the measured table (GLV: 8 / 13 / 16 for n <= 2^10 / <= 2^14 / larger;
 * plain: 8 / 15 / 16 for n <= 2^12 / < 2^19 / larger — wide windows with a uniform top window).
 * reduce_segment: buckets per running-sum segment of the bucket reduction (power of two; 0 = 2 / 4 / 8 by bucket count,
 * 32 for n >= 2^20 in overlap mode).  big_bucket_threshold: run length above which a bucket is cut into
 * workgroup-sized chunks (0 = max(256, 8 x mean)). */
int h2agg_msm_configure(h2agg_ctx* ctx, int window_bits, int reduce_segment, int big_bucket_threshold);
/* (16-bit windows reduce their buckets as 256 rows x 128 columns — plain sums, then two weighted sums per window — unless
 * reduce_segment is given, which selects the running-sum segment kernels; a -DH2AGG_MEASURE_KNOBS build also reads H2AGG_REDUCE=segments.) */
/* GLV / endomorphism split of the scalars (k = k1 + lambda*k2, |k_i| < 2^127; phi(P) = (beta*x, y)): halves the
 * number of windows — same bucket additions, half the bucket reduction and half the serial doubling chain.
 * beta*x is computed once per base (a 32 B/point column beside the table), so the price is the decomposition pass and
 * the slice-combine pass of the half-as-many buckets: mode 0 = auto turns it on unless the tail is hidden anyway
 * (overlap mode and n >= 2^20) or is a small share of the work (n >= 2^22); 1 = on, -1 = off (254-bit windows). */
int h2agg_msm_configure_glv(h2agg_ctx* ctx, int mode);
/* Lanes per bucket in the accumulation kernel (1, 2, 4, 8, 16; 0 = chosen so that ~8192 waves are launched, at most 8
 * and at most the mean bucket occupancy). */
int h2agg_msm_configure_lanes_per_bucket(h2agg_ctx* ctx, int lanes);
/* Bucket-sort knobs: low bucket bits resolved per partition in LDS (4..12) and scalars per level-1
 * workgroup; 0 = default.  tile = -1 forces the two-array direct sort kernels (otherwise used only when
 * n does not fit the packed item's index field, n > 2^(31 - sub_bits)); tile = -2 additionally stages level 1
 * through LDS (measured slower; kept as a tested variant); tile = -3 keeps the packed two-level sort where the
 * digit-major sort would apply (plain 16-bit windows over one table, 2^16 .. 2^22 points; any non-default knob
 * here selects the packed kernels as well).  (A -DH2AGG_MEASURE_KNOBS build also reads H2AGG_SORT=packed.) */
int h2agg_msm_configure_sort(h2agg_ctx* ctx, int sub_bits, int tile);
/* Environment.  The library reads these variables and no others (experiment switches exist only in builds with
 * -DH2AGG_MEASURE_KNOBS, which no shipped library is):
 *   H2AGG_HOST_THREADS=n      worker threads of the host sponge pool (default: the usable cores)
 *   H2AGG_TRANSCRIPT=device|host, H2AGG_HOST_SPONGE=ifma|portable   overrides of h2agg_transcript_configure's automatic choice
 *   H2AGG_NO_PLACE            skip the stream-placement probe of h2agg_create / h2agg_set_stream (needed when the caller's
 *                             stream is being graph-captured: the probe launches and synchronises)
 *   H2AGG_TRACE, H2AGG_TRACE_PHASES   diagnostics on stderr (schema evaluation / phases of h2agg_verify_aggregation)
 *   H2AGG_RCCL_LIB=path       the RCCL build to dlopen for the comm entry points instead of the librccl.so.1 already in the
 *                             process / on the loader path (an operator's own build; tests point it at their stand-in)
 *   H2AGG_PAIRING_PORTABLE    (any value) the host pairing in its portable build even on a CPU with BMI2 + ADX (A/B of the two
 *                             instantiations of csrc/pairing.hpp; the pairing entry points take ctx = NULL, so this is not a key)
 * Test hooks, per context and per call (tests/ exercise code paths a production call reaches only by size):
 *   key "pcie_slices" n   cut h2agg_g1_msm's host buffers into n slices (0 = automatic)
 *       "pcie_glv" -1|0|1 GLV for those slices (0 = automatic)       "pcie_chain" 0|1  slices share one bucket set (default 1)
 *       "comb_msm" 0|1    small MSMs over tables with fixed-base levels take the comb (default 1)
 *       "plan_cache" 0|1  h2agg_verify_aggregation keeps the recording of a call shape (default 1)
 *       "small_sort" 0|1  MSMs of <= 16384 scalars sort in one launch (default 1; 0 = the packed two-level sort)
 *       "eval_split" 0|1  the two multi_exps of an evaluation run as one set of launches (default 1)
 *       "lean_acc" 0|1    bucket accumulation through k_msm_accumulate_lean (default 1; 0 = the generic kernel, which only the
 *                         measure build carries: the shipped library answers H2AGG_ERR_INVALID)
 *       "tape_lds" 0|1    Fr tapes whose live values fit run with the register file in LDS (default 1; 0 = through L2)
 *       "prewake" 0|1     a from-bytes call wakes its sponge workers at entry (they spin until the chains are posted), one more
 *                         for the pairing's second Miller loop, and waits only for the element streams in front of the sponges
 *                         (default 1; 0: the three latency measures of round 6 off, for an A/B)
 *       "phases" 0|1      keep every aggregation call's wall-clock split for h2agg_last_phases
 *       "pre_big" 0|1     h2agg_bases_precompute takes any explicit width (1: levels through the two-array sort, A/B only)
 *       "shard_fail" 0..4  this rank of h2agg_verify_aggregation_sharded fails before (1) / between (2) its exchanges, or inside
 *                         exchange 1 (3) / 2 (4) before its all-gather */
int h2agg_debug_configure(h2agg_ctx* ctx, const char* key, int value);
/* Overlap the latency-shaped tail of one MSM (enable = 1: the Horner kernel, one wave; 2: bucket reduction + window
 * sums + Horner; 3: as 2, and the bucket accumulation itself leaves the context's stream, so that the NEXT MSM's sort runs
 * under it — measured: a loss at 2^20 points (1.39 -> 1.55 ms per MSM: both kernels slow each other down), +3 % at 2^22) with the bulk kernels of the next ones: the tail runs on one of three tail streams of the context.  With overlap on, a result written by
 * h2agg_g1_msm_device_async is complete after h2agg_synchronize() (or after the next synchronous call on
 * the context), not merely after the caller's stream has drained: from 2^20 points on, the tail of the LAST MSM queued is only launched by
 * the next call on the context (behind that MSM's sort, where it costs least) or by h2agg_synchronize(); smaller MSMs launch
 * theirs at once (there the tail is a large share of the MSM: waiting costs more than the next sort gains).  Default: off. */
int h2agg_msm_set_tail_overlap(h2agg_ctx* ctx, int enable);
/* When enabled, every MSM stage is bracketed by HIP events on the context's stream and per-stage times
 * are accumulated.  enable: 0 = off, 1 = every stage, 2 + s = only stage s (one event pair per MSM: the event
 * packets between kernels cost ~5 us each, which is 5-8 % of a 2^20 MSM when all nine stages are bracketed). */
int h2agg_profile_enable(h2agg_ctx* ctx, int enable);
int h2agg_profile_reset(h2agg_ctx* ctx);
/* Number of stages; name of stage i; accumulated milliseconds and launch count of stage i. */
int h2agg_profile_stage_count(h2agg_ctx* ctx);
const char* h2agg_profile_stage_name(h2agg_ctx* ctx, int i);
int h2agg_profile_stage_get(h2agg_ctx* ctx, int i, double* total_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* H2AGG_H */
