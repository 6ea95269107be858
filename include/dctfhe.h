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

#define DCTFHE_MAX_TIERS 8

typedef struct dctfhe_ctx dctfhe_ctx;
typedef struct dctfhe_keys dctfhe_keys;
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
  int32_t reserved;
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

/* R3 keygen(): client secret keys + server evaluation keys (KSK, Fourier BSK per tier), on the GPU. */
int dctfhe_keygen(dctfhe_ctx* ctx, const dctfhe_params* params, uint64_t seed, dctfhe_keys** out);
int dctfhe_keys_destroy(dctfhe_keys* keys);
/* client-side view of the keys (tests, and the client that decrypts) */
int dctfhe_keys_export_secret(dctfhe_keys* keys, uint8_t* big_key /* D */, uint8_t* small_key /* n_max */);
/* standard-domain evaluation keys (tests): ksk [D][lk][n+1], bsk [n][(k+1)l][k+1][N] */
int dctfhe_keys_export_ksk(dctfhe_keys* keys, int tier, uint64_t* out);
int dctfhe_keys_export_bsk(dctfhe_keys* keys, int tier, uint64_t* out);

/* R4, client half: encrypt phases (already encoded) / return phases b - <a,s>.  Host buffers. */
int dctfhe_encrypt(dctfhe_ctx* ctx, dctfhe_keys* keys, const uint64_t* phases, size_t count, uint64_t seed,
                   uint64_t* cts /* count x (D+1) */);
int dctfhe_decrypt(dctfhe_ctx* ctx, dctfhe_keys* keys, const uint64_t* cts, size_t count, uint64_t* phases);

/* R4, server half, one primitive at a time on host buffers (parity tests, integration). */
int dctfhe_keyswitch(dctfhe_ctx* ctx, dctfhe_keys* keys, int tier, const uint64_t* cts, size_t count,
                     int shift, uint64_t* cts_small /* count x (n+1) */);
/* the same when the caller knows every input to be zero beyond mask word `deff` (nested keys: outputs of a ring of
 * dimension k*N <= deff): only the first deff rows of the key are used -- identical result, deff/D of the work */
int dctfhe_keyswitch_prefix(dctfhe_ctx* ctx, dctfhe_keys* keys, int tier, const uint64_t* cts, size_t count,
                            int shift, int deff, uint64_t* cts_small);
int dctfhe_pbs(dctfhe_ctx* ctx, dctfhe_keys* keys, int tier, const uint64_t* cts_small, size_t count,
               const int64_t* tables /* [ntab][2^w] */, int ntab, int w, const int32_t* table_idx /* may be NULL */,
               uint64_t* cts_out /* count x (D+1) */);
int dctfhe_round_lut(dctfhe_ctx* ctx, dctfhe_keys* keys, int bit_tier, int tab_tier, const uint64_t* cts,
                     size_t count, int p, int r, const int64_t* tables, int ntab, int w,
                     const int32_t* table_idx, uint64_t* cts_out);
int dctfhe_conv2d(dctfhe_ctx* ctx, int D, const uint64_t* in, int batch, int Cin, int H, int W,
                  const int8_t* weight /* [Cout][Cin][KH][KW] */, int Cout, int KH, int KW, int stride, int pad,
                  uint64_t* out);

/* R1: load a compiled circuit description (built by dctfhe.compile, format in DESIGN.md section 4). */
int dctfhe_circuit_load(dctfhe_ctx* ctx, const void* blob, size_t size, dctfhe_circuit** out);
int dctfhe_circuit_destroy(dctfhe_circuit* circ);
int dctfhe_circuit_stats(dctfhe_circuit* circ, const dctfhe_params* params, dctfhe_stats* out);
int dctfhe_circuit_io(dctfhe_circuit* circ, int64_t* n_in_per_image, int64_t* n_out_per_image);

/* R4: evaluate the circuit on a batch of images.  A session owns the device tensors. */
int dctfhe_session_create(dctfhe_ctx* ctx, dctfhe_circuit* circ, dctfhe_keys* keys /* NULL: clear mode */,
                          int batch, dctfhe_session** out);
int dctfhe_session_destroy(dctfhe_session* s);
int dctfhe_session_upload(dctfhe_session* s, const uint64_t* cts_in /* batch x n_in x (D+1); clear: x 1 */);
int dctfhe_session_run(dctfhe_session* s, dctfhe_timing* timing /* may be NULL */);
/* clear-mode sessions (keys == NULL) only: `simulate` with the noise model.  sigma_per_op[i] (fraction of the torus, 0 for
 * ops that are not look-ups) is added at the input of op i's table look-up, fresh draws every run; n_ops = 0 switches it off */
int dctfhe_session_set_noise(dctfhe_session* s, uint64_t seed, const double* sigma_per_op, int n_ops);
int dctfhe_session_download(dctfhe_session* s, uint64_t* cts_out /* batch x n_out x (D+1) */);

/* f64 FMA peak micro-benchmark (TFLOP/s) used to price the blind-rotate kernel in bench.py. */
int dctfhe_fp64_peak(dctfhe_ctx* ctx, double* tflops);
/* stand-alone timing of the blind-rotate kernel: count ciphertexts of tier `tier`, average ms per launch */
int dctfhe_bench_pbs(dctfhe_ctx* ctx, dctfhe_keys* keys, int tier, size_t count, int reps, double* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif
