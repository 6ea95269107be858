/*
 * oracle/tfhe_ref.h -- CPU restatement ("CPU twin") of the TFHE arithmetic that the
 * reference executes inside its third-party runtime.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or called by the
 * product (dct-cryptonets_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / as the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (zhiyongggggg/dct-cryptonets) holds no source, test or
 * golden vector for this arithmetic.  Its hot path is `QuantizedModule.forward(x, fhe=...)`
 * (reference dct-cryptonets/homomorphic_eval.py:70) which runs inside concrete-ml==1.6.1 /
 * concrete-python==2.7.0 (reference env.yml:35-36); neither is vendored, installed or
 * installable here.  This file therefore restates the *published* TFHE scheme (Chillotti,
 * Gama, Georgieva, Izabachene: "TFHE: Fast Fully Homomorphic Encryption over the Torus",
 * J. Cryptology 2020; programmable bootstrapping as in Chillotti, Joye, Paillier 2021) on
 * the 64-bit torus, and is pinned only by scheme identities (tests/test_oracle_tfhe.py):
 * decrypt(PBS_f(enc(m))) == f(m) for every m, key-switch preserves m, linearity, the
 * negacyclic sign rule, and FFT external product == exact schoolbook external product.
 *
 * Conventions (shared with the product, stated in DESIGN.md section 3):
 *   torus            u64, q = 2^64, all ciphertext arithmetic wraps mod 2^64
 *   LWE ciphertext   D+1 words: a[0..D) mask, b at index D;  b = <a,s> + phase_plain + e
 *   GLWE ciphertext  k mask polys A_0..A_{k-1} then body B, each N words, X^N = -1;
 *                    B = sum_j A_j*S_j + M + E
 *   GGSW(s) / BSK    rows r = p*l + lev  (p = 0..k component, lev = 0..l-1, lev 0 most significant)
 *                    row r = GLWE(0) + s * 2^(64 - beta*(lev+1)) added to coefficient 0 of component p
 *   decomposition    signed digits in [-B/2, B/2), closest-representable rounding
 *   key switch       out = (0,..,0,b) - sum_i sum_lev dig_lev(a_i) * KSK[i][lev]
 *   KSK[i][lev]      LWE_s( S_i * 2^(64 - betak*(lev+1)) ), n+1 words
 *   mod switch       a~ = round(a * 2N / 2^64) mod 2N
 *   blind rotate     ACC = X^{-b~} * TV ; for i<n: ACC += BSK_i [x] (X^{a~_i} * ACC - ACC)
 *   sample extract   coefficient 0
 *   test vector      from a table T of 2^w entries: box = N >> w, half = box/2,
 *                    TV[j] = T[(j+half)/box] if j+half < N else -T[0]
 */
#ifndef TFHE_REF_H
#define TFHE_REF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* deterministic generator (splitmix64 stream); used only for oracle-side key material */
uint64_t ref_splitmix64(uint64_t *state);

void ref_gen_binary_key(uint64_t seed, int len, uint8_t *key);

/* phase_out[c] = b - <a, key> for count ciphertexts of D+1 words */
void ref_lwe_phase_batch(const uint8_t *key, int D, const uint64_t *cts, int count, uint64_t *phase_out);
/* fresh encryptions: ct = (a random, b = <a,key> + phases[c] + gaussian(sigma * 2^64)); mask only on a[0..dim_eff) */
void ref_lwe_encrypt_batch(const uint8_t *key, int D, int dim_eff, const uint64_t *phases, int count,
                           double sigma, uint64_t seed, uint64_t *cts);

/* key material (standard domain) */
void ref_ksk_gen(const uint8_t *S_big, int D, const uint8_t *s_small, int n, int lk, int betak,
                 double sigma, uint64_t seed, uint64_t *ksk /* [D][lk][n+1] */);
void ref_bsk_gen(const uint8_t *s_small, int n, const uint8_t *S_glwe /* k*N bits */, int k, int N,
                 int l, int beta, double sigma, uint64_t seed,
                 uint64_t *bsk /* [n][(k+1)*l][k+1][N] */);
/* standard -> Fourier (natural order, M = N/2 complex points per polynomial, re/im interleaved) */
void ref_bsk_to_fourier(const uint64_t *bsk, int n, int k, int N, int l, double *bsk_f);

/* integer kernels (bit-exact contract) */
void ref_keyswitch(const uint64_t *cts_in, int count, int D, const uint64_t *ksk, int n, int lk,
                   int betak, uint64_t *cts_out /* count x (n+1) */);
void ref_modswitch(const uint64_t *ct_small, int n, int N, uint32_t *out /* n+1, in [0,2N) */);
/* centred mod switch: body -= (1/2) sum of the mask words' rounding remainders (in place, before ref_modswitch / a bootstrap) */
void ref_ms_center(uint64_t *cts_small, int count, int n, int N);
void ref_decompose(uint64_t v, int l, int beta, int32_t *digits /* l, lev 0 first */);
void ref_build_testvector(const int64_t *table, int w, int N, uint64_t *tv);

/* programmable bootstrap over a batch.
 * cts_small: count x (n+1).  tables: [ntab][2^w] (already scaled to the output encoding),
 * table_idx[c] selects the table.  Output count x (D_out+1), mask beyond k*N zeroed.
 * use_exact != 0: schoolbook external product mod 2^64 on the standard-domain key (bsk),
 * otherwise f64 FFT on bsk_f.  Returns 0. */
int ref_pbs_batch(const uint64_t *cts_small, int count, int n,
                  const double *bsk_f, const uint64_t *bsk, int use_exact,
                  int k, int N, int l, int beta,
                  const int64_t *tables, int w, const int32_t *table_idx,
                  int D_out, uint64_t *cts_out);

/* two-bit blind rotation (the product's one-level k = 1 tiers): definition in exact arithmetic.
 * ref_pair_secret derives (s1(1-s2), (1-s1)s2, s1 s2) per pair; bsk3 = ref_bsk_gen of that 3n/2-bit secret. */
void ref_pair_secret(const uint8_t *s, int n, uint8_t *out /* 3n/2 */);
int ref_pbs_mb2_batch(const uint64_t *cts_small, int count, int n, const uint64_t *bsk3,
                      int k, int N, int l, int beta, const int64_t *tables, int w, const int32_t *table_idx,
                      int D_out, uint64_t *cts_out);

/* levelled operators on ciphertext tensors (D+1 words per element, NCHW, batch folded by caller) */
void ref_conv2d(const uint64_t *in, int Cin, int H, int W, int D,
                const int32_t *weight /* [Cout][Cin][KH][KW] */, int Cout, int KH, int KW,
                int stride, int pad, uint64_t *out /* [Cout][Ho][Wo][D+1] */);
void ref_sum_pool(const uint64_t *in, int C, int H, int W, int D, int K, uint64_t *out /* [C][H/K][W/K] */);

/* exact rounding + table: for each ciphertext (precision p, message offset-binary, delta = 2^(63-p)):
 *   add 2^(r-1)*delta; for i<r: sign-PBS (bit tier) clears bit i; then table PBS (table tier) on p-r bits.
 * Everything the server does for one "conv output -> activation" site. */
typedef struct {
  int n, k, N, l, beta, lk, betak;
  const double *bsk_f; const uint64_t *ksk;
} ref_tier_t;
int ref_round_lut_batch(const uint64_t *cts_in, int count, int D, int p, int r,
                        const ref_tier_t *bit_tier, const ref_tier_t *tab_tier,
                        const int64_t *tables, int w, const int32_t *table_idx,
                        uint64_t *cts_out);

int ref_num_threads(void);
void ref_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
