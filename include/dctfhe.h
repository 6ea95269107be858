/*
 * dctfhe.h -- C ABI of libdctfhe.so, the MI355X-native homomorphic-evaluation engine that sits
 * behind the reference's third-party boundary.
 *
 * The reference (zhiyongggggg/dct-cryptonets) is pure Python; its hot path is reached through
 * five call sites into concrete-ml / concrete-python (absent third-party wheels):
 *
 *   (R1) compile_brevitas_qat_model(feature, calib, ...)   dct-cryptonets/homomorphic_eval.py:276-285
 *        compile_torch_model(...)                          dct-cryptonets/homomorphic_eval.py:287-295
 *   (R2) q_module.fhe_circuit.graph.maximum_integer_bit_width()            homomorphic_eval.py:301
 *   (R3) q_module.fhe_circuit.keygen()                                     homomorphic_eval.py:315
 *   (R4) q_module.forward(data, fhe="simulate"|"execute")                  homomorphic_eval.py:70
 *   (R5) q_module.fhe_circuit.mlir                                         homomorphic_eval.py:311
 *
 * Nothing like a C interface exists in the reference; each entry point below names the call
 * site whose work it carries.  The Python facade that keeps the reference's names
 * (dct-cryptonets_amd/dctfhe/quantized_module.py) binds these with ctypes; INTEGRATION.md shows
 * the stub.  Conventions: opaque handles, int status (0 = OK, <0 = error, text through
 * dctfhe_last_error()), caller-allocated host buffers, no exceptions across the boundary.
 * A dctfhe_ctx is bound to one GPU and one host thread (one per rank); distinct contexts are
 * independent.  Ciphertext layout, encodings and key formats: DESIGN.md section 3.
 */
#ifndef DCTFHE_H
#define DCTFHE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DCTFHE_MAX_TIERS 12

typedef struct dctfhe_ctx dctfhe_ctx;
typedef struct dctfhe_client_key dctfhe_client_key;   /* CLIENT: secret keys + the CSPRNG keys; never needed by the server */
typedef struct dctfhe_eval_keys dctfhe_eval_keys;     /* SERVER: key-switch keys + Fourier bootstrap keys (public material) */
typedef struct dctfhe_circuit dctfhe_circuit;
typedef struct dctfhe_session dctfhe_session;

/* One bootstrapping parameter tier (what concrete-optimizer picks per partition; R1 p_error). */
typedef struct {
  int32_t n;            /* small LWE dimension (prefix of the small secret key) */
  int32_t k, logN;      /* GLWE dimension, log2 polynomial size; k*N <= D (prefix of the big key) */
  int32_t l, beta;      /* bootstrap gadget: levels, base log */
  int32_t lk, betak;    /* key-switch gadget: levels, base log */
  int32_t ksk_share;    /* >= 0: reuse the key-switch key of that (earlier) tier; -1: own key */
  int32_t unroll;       /* key bits per blind-rotate iteration: 1, or 2 (k = 1, l = 1, n even; key of 3n/2 blocks) */
  int32_t key_lds;      /* 1: the waves of a workgroup share each bootstrap-key tile through LDS (LDS-DMA ring) instead of each pulling its own
                           copy through L1 -- (k, l, N, unroll) = (2, 1, 1024, 2) only; same results, measured no faster */
  double lwe_sigma;     /* noise std of key-switch-key rows (fraction of the torus) */
  double glwe_sigma;    /* noise std of bootstrap-key rows */
} dctfhe_tier;

typedef struct {
  int32_t D;            /* big LWE dimension = length of the master binary key */
  int32_t n_max;        /* length of the small binary key */
  int32_t n_tiers;
  int32_t input_dim;    /* fresh client encryptions mask only the first input_dim words (a prefix of the big key, like
                           every bootstrap output of a smaller ring); 0 = D.  input_sigma must suit that dimension. */
  double input_sigma;   /* noise std of fresh client encryptions */
  dctfhe_tier tiers[DCTFHE_MAX_TIERS];
} dctfhe_params;

typedef struct {
  int64_t pbs_count[DCTFHE_MAX_TIERS];   /* bootstraps per image, per tier */
  int64_t ks_count[DCTFHE_MAX_TIERS];    /* key switches per image, per tier */
  int64_t conv_macs;                     /* scalar ciphertext*plaintext MACs per image */
  int64_t lut_sites, bit_steps;          /* table look-ups and one-bit rounding steps per image */
  double bytes_algorithmic;              /* SURVEY 8(d) B_img without the key passes */
  double key_bytes_per_pass;             /* sum over layer steps of |BSK|+|KSK| (divide by images per batch) */
  double flops_f64;                      /* SURVEY 8(d) F_img */
  int32_t max_bit_width;                 /* R2 */
  int32_t n_ops;
} dctfhe_stats;

/* Per-run timing (milliseconds, HIP events on the context's stream). */
typedef struct {
  double total_ms;
  double pbs_ms[DCTFHE_MAX_TIERS];
  double ks_ms;
  double linear_ms;
  int64_t pbs_launches[DCTFHE_MAX_TIERS];
  int64_t pbs_cts[DCTFHE_MAX_TIERS];
} dctfhe_timing;

const char* dctfhe_last_error(void);
int dctfhe_version(void);

int dctfhe_ctx_create(int device_id, dctfhe_ctx** out);
int dctfhe_ctx_destroy(dctfhe_ctx* ctx);
/* use an existing HIP stream (e.g. torch's current stream) instead of the context's own; NULL resets */
int dctfhe_ctx_set_stream(dctfhe_ctx* ctx, void* hip_stream);
int dctfhe_ctx_synchronize(dctfhe_ctx* ctx);

/* R3 keygen() (homomorphic_eval.py:313-317).  The reference's `fhe_circuit.keygen()` makes a client key set and the
 * evaluation keys the server needs; here they are two handles so that the secret never has to reach the server.
 *
 * Randomness: a counter-mode ChaCha20 generator on the GPU keyed by the caller's 32-byte seed (draw it from the OS:
 * os.urandom / getrandom).  The secret-key bits and every noise term come from the seed's own stream; ciphertext and key
 * MASKS come from a second ChaCha20 key that is one block of the first (public: knowing it gives nothing about the seed).
 * The KEY MATERIAL of a client key is a pure function of (params, seed): persist the 32 bytes to persist it; every rank of a
 * multi-GPU job gets the same keys from the same seed (broadcast the seed, not the keys).  Encryption randomness is NOT a
 * function of the seed alone: see dctfhe_encrypt.
 * SECURITY STATUS: parameters follow a fit through published 128-bit sets (dctfhe/params.py), not an estimator run --
 * there is none in this environment; treat the "~128-bit" figure as unverified. */
int dctfhe_client_key_create(dctfhe_ctx* ctx, const dctfhe_params* params, const uint8_t seed[32], dctfhe_client_key** out);
int dctfhe_client_key_destroy(dctfhe_client_key* client);
/* client side: evaluation keys for `client`'s secret (the expensive part of keygen; runs on the GPU) */
int dctfhe_eval_keys_generate(dctfhe_client_key* client, dctfhe_eval_keys** out);
/* both at once */
int dctfhe_keygen(dctfhe_ctx* ctx, const dctfhe_params* params, const uint8_t seed[32], dctfhe_client_key** client,
                  dctfhe_eval_keys** eval);
int dctfhe_eval_keys_destroy(dctfhe_eval_keys* eval);
/* evaluation-key persistence / shipping to the server: a flat blob (header + parameters, then per tier its key-switch key
 * and its Fourier bootstrap key).  buf == NULL: only *size is written (size query). */
int dctfhe_eval_keys_export(dctfhe_eval_keys* eval, void* buf, size_t capacity, size_t* size);
int dctfhe_eval_keys_import(dctfhe_ctx* ctx, const void* buf, size_t size, dctfhe_eval_keys** out);

/* test / client views */
int dctfhe_client_key_export_secret(dctfhe_client_key* client, uint8_t* big_key /* D */, uint8_t* small_key /* n_max */);
/* standard-domain keys: ksk [D][lk][n+1] (public, from the evaluation keys); bsk [n][(k+1)l][k+1][N], regenerated from the
 * client's streams -- exactly what dctfhe_eval_keys_generate transformed to the Fourier domain.
 * Key-switch-key words live on the torus grid 2^-(8 limbs), limbs = 2 / 4 / 8 for lk*betak + 6 <= 16 / <= 32 / more (their low
 * 64 - 8 limbs bits are zero: masks drawn on the grid, bodies rounded to it); dctfhe_eval_keys_import refuses a key off that grid. */
int dctfhe_eval_keys_export_ksk(dctfhe_eval_keys* eval, int tier, uint64_t* out);
int dctfhe_client_key_export_bsk(dctfhe_client_key* client, int tier, uint64_t* out);
/* the generator itself: `count` 64-bit outputs (key, stream, idx0 + i).  _host runs on the CPU (known-answer tests need no GPU) */
int dctfhe_rng_host(const uint8_t key[32], uint64_t stream, uint64_t idx0, size_t count, uint64_t* out);
int dctfhe_rng_device(dctfhe_ctx* ctx, const uint8_t key[32], uint64_t stream, uint64_t idx0, size_t count, uint64_t* out);

/* R4, client half: encrypt phases (already encoded) / return phases b - <a,s>.  Host buffers.  Every dctfhe_encrypt call
 * draws masks and noise from fresh generator streams (a per-handle call counter). */
int dctfhe_encrypt(dctfhe_ctx* ctx, dctfhe_client_key* client, const uint64_t* phases, size_t count,
                   uint64_t* cts /* count x (D+1) */);
int dctfhe_decrypt(dctfhe_ctx* ctx, dctfhe_client_key* client, const uint64_t* cts, size_t count, uint64_t* phases);
/* The same in the COMPACT WIRE FORM: rows of `dim` mask words + the body instead of D + 1 words.  A fresh encryption masks only
 * params.input_dim words and a circuit output only the ring of its last table tier (dctfhe_session_dims), so the host <-> device
 * and client <-> server traffic of an image shrinks by D / dim (ResNet-18 48x112^2: 39.5 GB -> 9.9 GB of input per image).
 * encrypt: input_dim <= dim <= D (words from input_dim on are zero); decrypt: dim <= D (the words a row lacks count as zero).
 * Both forms of one call hold the same ciphertexts. */
int dctfhe_encrypt_rows(dctfhe_ctx* ctx, dctfhe_client_key* client, const uint64_t* phases, size_t count, int dim,
                        uint64_t* cts /* count x (dim+1) */);
int dctfhe_decrypt_rows(dctfhe_ctx* ctx, dctfhe_client_key* client, const uint64_t* cts /* count x (dim+1) */, size_t count, int dim,
                        uint64_t* phases);
/* ENCRYPTION RANDOMNESS IS PER HANDLE.  Key material is a pure function of (params, seed) -- persist or broadcast the 32 bytes to
 * persist or share the key -- but the masks and noise of dctfhe_encrypt come from generator keys derived from the seed AND a 128-bit
 * nonce that dctfhe_client_key_create draws from the OS (getrandom), at a position given by a per-handle call counter.  Two handles
 * made from one seed (a re-created key, a second process, the ranks of a job) therefore never draw the same mask or noise.
 * set_encrypt_counter moves the position inside the handle's own streams (kept for callers that partition them); set_encrypt_nonce
 * FIXES the nonce -- reproducible experiments and tests only: handles with equal seed, nonce and counter encrypt identically. */
int dctfhe_client_key_set_encrypt_counter(dctfhe_client_key* client, uint64_t next_call);
int dctfhe_client_key_set_encrypt_nonce(dctfhe_client_key* client, const uint8_t nonce[16]);

/* R4, server half, one primitive at a time on host buffers (parity tests, integration). */
int dctfhe_keyswitch(dctfhe_ctx* ctx, dctfhe_eval_keys* keys, int tier, const uint64_t* cts, size_t count,
                     int shift, uint64_t* cts_small /* count x (n+1) */);
/* the same when the caller knows every input to be zero beyond mask word `deff` (nested keys: outputs of a ring of
 * dimension k*N <= deff): only the first deff rows of the key are used -- identical result, deff/D of the work */
int dctfhe_keyswitch_prefix(dctfhe_ctx* ctx, dctfhe_eval_keys* keys, int tier, const uint64_t* cts, size_t count,
                            int shift, int deff, uint64_t* cts_small);
/* centred mod switch, in place on small ciphertexts (count x (n+1)): half the sum of the mask words' rounding remainders comes off the
 * body, which halves the variance of the bootstrap's mod-switch error.  dctfhe_round_lut and dctfhe_session_run
 * apply it between every key switch and its bootstrap; dctfhe_keyswitch / dctfhe_pbs are the bare primitives. */
int dctfhe_modswitch_center(dctfhe_ctx* ctx, dctfhe_eval_keys* keys, int tier, uint64_t* cts_small, size_t count);
int dctfhe_pbs(dctfhe_ctx* ctx, dctfhe_eval_keys* keys, int tier, const uint64_t* cts_small, size_t count,
               const int64_t* tables /* [ntab][2^w] */, int ntab, int w, const int32_t* table_idx /* may be NULL */,
               uint64_t* cts_out /* count x (D+1) */);
int dctfhe_round_lut(dctfhe_ctx* ctx, dctfhe_eval_keys* keys, int bit_tier, int tab_tier, const uint64_t* cts,
                     size_t count, int p, int r, const int64_t* tables, int ntab, int w,
                     const int32_t* table_idx, uint64_t* cts_out);
int dctfhe_conv2d(dctfhe_ctx* ctx, int D, const uint64_t* in, int batch, int Cin, int H, int W,
                  const int8_t* weight /* [Cout][Cin][KH][KW] */, int Cout, int KH, int KW, int stride, int pad,
                  uint64_t* out);

/* K2 one kernel at a time on host buffers (reference backbone.py:102 torch.add, :276 AvgPool2d, and the shift / offset that opens a
 * rounding chain): rows of `dim` mask words + body whose mask words from `deff` on count as zero -- the storage form of a session's
 * tensors, here exposed so that the streaming kernels can be checked at mixed effective dimensions outside a circuit.
 * add: out = a + b, dim_o >= max(deff_a, deff_b).  affine: the first nwords mask words of each inout row become a << shift, the body
 * (body << shift) + body_add, the words in between are LEFT AS THEY ARE.  sum_pool: KxK window sums, floor semantics (nn.AvgPool2d(K)
 * drops the border; the 1/K^2 lives in the next table). */
int dctfhe_add_rows(dctfhe_ctx* ctx, const uint64_t* a, int dim_a, int deff_a, const uint64_t* b, int dim_b, int deff_b, size_t count,
                    int dim_o, uint64_t* out);
int dctfhe_affine_rows(dctfhe_ctx* ctx, const uint64_t* a, int dim_a, int deff_a, size_t count, int nwords, int shift, uint64_t body_add,
                       int dim_o, uint64_t* inout);
int dctfhe_sum_pool_rows(dctfhe_ctx* ctx, const uint64_t* in /* [batch][C][H][W] rows */, int dim_in, int deff_in, int batch, int C, int H,
                         int W, int K, int dim_o, uint64_t* out /* [batch][C][H/K][W/K] rows */);

/* K10, client side, plaintext: the DCT front-end of reference data/cvfunctional.py:37-74 + data/cvtransforms.py:56-64,117-208 on
 * uint8 planes (luma [batch][fs*S][fs*S]; two chroma slots [batch][fs*Sc][fs*Sc], Sc = S/2 for the 4:2:0 paths or S):
 * blockwise orthonormal DCT-II of (pixel - 128), only the kept coefficients idx_* (row-major u*fs+v), chroma grids
 * bilinearly up-sampled to S x S, channels concatenated luma | slot 1 | slot 2, (x - mean[c]) / std[c] in f32.
 * round_coeffs != 0: the JPEG-domain (filter 8) path's integer coefficient planes.  out: float32 [batch][ny+n1+n2][S][S]. */
int dctfhe_dct_frontend(dctfhe_ctx* ctx, const uint8_t* y, const uint8_t* c1, const uint8_t* c2, int batch, int S, int Sc, int fs,
                        const int32_t* idx_y, int ny, const int32_t* idx_c1, int n1, const int32_t* idx_c2, int n2,
                        const float* mean, const float* stdv, int round_coeffs, float* out);

/* host-only validators (no GPU): parameter set / circuit blob well-formed?  0 or -1 with dctfhe_last_error() */
int dctfhe_params_check(const dctfhe_params* params);
int dctfhe_circuit_validate(const void* blob, size_t size);

/* R1: load a compiled circuit description (built by dctfhe.compile, format in DESIGN.md section 4). */
int dctfhe_circuit_load(dctfhe_ctx* ctx, const void* blob, size_t size, dctfhe_circuit** out);
int dctfhe_circuit_destroy(dctfhe_circuit* circ);
int dctfhe_circuit_stats(dctfhe_circuit* circ, const dctfhe_params* params, dctfhe_stats* out);
int dctfhe_circuit_io(dctfhe_circuit* circ, int64_t* n_in_per_image, int64_t* n_out_per_image);

/* R4: evaluate the circuit on a batch of images.  A session owns the device tensors and needs the EVALUATION keys only. */
int dctfhe_session_create(dctfhe_ctx* ctx, dctfhe_circuit* circ, dctfhe_eval_keys* keys /* NULL: clear mode */,
                          int batch, dctfhe_session** out);
int dctfhe_session_destroy(dctfhe_session* s);
int dctfhe_session_upload(dctfhe_session* s, const uint64_t* cts_in /* batch x n_in x (D+1); clear: x 1 */);
/* compact wire form: host rows of dim mask words + body (clear-mode sessions ignore dim).  upload: any 1 <= dim <= D; words a row lacks
 * count as zero, words beyond the input's effective dimension must BE zero (checked).  download: dim >= the output's effective dimension.
 * dctfhe_session_dims reports the two effective dimensions (input: what upload keeps; output: the least download accepts). */
int dctfhe_session_upload_rows(dctfhe_session* s, const uint64_t* cts_in /* batch x n_in x (dim+1) */, int dim);
int dctfhe_session_download_rows(dctfhe_session* s, uint64_t* cts_out /* batch x n_out x (dim+1) */, int dim);
int dctfhe_session_dims(dctfhe_session* s, int* in_dim, int* out_dim);
/* synchronous.  The uploaded input stays resident: run may be called again without a fresh upload (same result). */
int dctfhe_session_run(dctfhe_session* s, dctfhe_timing* timing /* may be NULL */);
/* clear-mode sessions (keys == NULL) only: `simulate` with the noise model.  sigma_per_op[i] (fraction of the torus, 0 for
 * ops that are not look-ups) is added at the input of op i's table look-up, fresh draws every run; n_ops = 0 switches it off */
int dctfhe_session_set_noise(dctfhe_session* s, uint64_t seed, const double* sigma_per_op, int n_ops);
int dctfhe_session_download(dctfhe_session* s, uint64_t* cts_out /* batch x n_out x (D+1) */);

/* f64 FMA peak micro-benchmark (TFLOP/s) used to price the blind-rotate kernel in bench.py. */
int dctfhe_fp64_peak(dctfhe_ctx* ctx, double* tflops);
/* stand-alone timing of the blind-rotate kernel: count ciphertexts of tier `tier`, average ms per launch */
int dctfhe_bench_pbs(dctfhe_ctx* ctx, dctfhe_eval_keys* keys, int tier, size_t count, int reps, double* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif
