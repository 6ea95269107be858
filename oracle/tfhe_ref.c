/*
 * oracle/tfhe_ref.c -- CPU twin of the TFHE arithmetic on the reference's hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see tfhe_ref.h header for why and for the
 * conventions).  Reference call sites this stands in for:
 *   q_module.forward(data, fhe="execute")      dct-cryptonets/homomorphic_eval.py:70
 *   q_module.fhe_circuit.keygen()              dct-cryptonets/homomorphic_eval.py:315
 * whose arithmetic lives in concrete-python==2.7.0 (env.yml:36), absent from /root/reference.
 *
 * Written for clarity, not speed: radix-2 FFT, one ciphertext at a time; OpenMP over
 * ciphertexts so that the cpu_baseline leg of bench.py can use every host core.
 */
#include "tfhe_ref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ rng */
uint64_t ref_splitmix64(uint64_t *state) {
  uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

static double rng_unit(uint64_t *st) { /* (0,1] */
  return ((double)(ref_splitmix64(st) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
}

static int64_t rng_gauss_torus(uint64_t *st, double sigma) {
  if (sigma <= 0.0) return 0;
  double u1 = rng_unit(st), u2 = rng_unit(st);
  double g = sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
  return (int64_t)llround(g * sigma * 18446744073709551616.0);
}

void ref_gen_binary_key(uint64_t seed, int len, uint8_t *key) {
  uint64_t st = seed;
  for (int i = 0; i < len; i += 64) {
    uint64_t r = ref_splitmix64(&st);
    for (int b = 0; b < 64 && i + b < len; b++) key[i + b] = (uint8_t)((r >> b) & 1);
  }
}

void ref_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ LWE */
void ref_lwe_phase_batch(const uint8_t *key, int D, const uint64_t *cts, int count, uint64_t *phase_out) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < count; c++) {
    const uint64_t *ct = cts + (size_t)c * (D + 1);
    uint64_t acc = ct[D];
    for (int i = 0; i < D; i++)
      if (key[i]) acc -= ct[i];
    phase_out[c] = acc;
  }
}

void ref_lwe_encrypt_batch(const uint8_t *key, int D, int dim_eff, const uint64_t *phases, int count,
                           double sigma, uint64_t seed, uint64_t *cts) {
  for (int c = 0; c < count; c++) {
    uint64_t st = seed ^ (0xA5A5A5A5ULL + (uint64_t)c * 0x9E3779B97F4A7C15ULL);
    uint64_t *ct = cts + (size_t)c * (D + 1);
    uint64_t b = phases[c] + (uint64_t)rng_gauss_torus(&st, sigma);
    for (int i = 0; i < D; i++) {
      uint64_t a = (i < dim_eff) ? ref_splitmix64(&st) : 0;
      ct[i] = a;
      if (key[i]) b += a;
    }
    ct[D] = b;
  }
}

/* ------------------------------------------------------------------ decomposition */
void ref_decompose(uint64_t v, int l, int beta, int32_t *digits) {
  const int total = l * beta; /* 1 <= total <= 63 */
  uint64_t x = (v + (1ULL << (63 - total))) >> (64 - total);
  const uint64_t B = 1ULL << beta, half = B >> 1, mask = B - 1;
  uint64_t carry = 0;
  for (int lev = l - 1; lev >= 0; lev--) {
    uint64_t d = (x & mask) + carry;
    x >>= beta;
    if (d >= half) { digits[lev] = (int32_t)((int64_t)d - (int64_t)B); carry = 1; }
    else           { digits[lev] = (int32_t)d; carry = 0; }
  }
}

/* ------------------------------------------------------------------ key switch */
void ref_ksk_gen(const uint8_t *S_big, int D, const uint8_t *s_small, int n, int lk, int betak,
                 double sigma, uint64_t seed, uint64_t *ksk) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < D; i++) {
    for (int lev = 0; lev < lk; lev++) {
      uint64_t st = seed ^ (0x5151ULL + ((uint64_t)i * 64 + (uint64_t)lev) * 0xD1B54A32D192ED03ULL);
      uint64_t *row = ksk + ((size_t)i * lk + lev) * (n + 1);
      uint64_t b = (uint64_t)rng_gauss_torus(&st, sigma);
      if (S_big[i]) b += 1ULL << (64 - betak * (lev + 1));
      for (int j = 0; j < n; j++) {
        uint64_t a = ref_splitmix64(&st);
        row[j] = a;
        if (s_small[j]) b += a;
      }
      row[n] = b;
    }
  }
}

void ref_keyswitch(const uint64_t *cts_in, int count, int D, const uint64_t *ksk, int n, int lk,
                   int betak, uint64_t *cts_out) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < count; c++) {
    const uint64_t *ct = cts_in + (size_t)c * (D + 1);
    uint64_t *out = cts_out + (size_t)c * (n + 1);
    int32_t dig[64];
    for (int j = 0; j < n; j++) out[j] = 0;
    out[n] = ct[D];
    for (int i = 0; i < D; i++) {
      if (ct[i] == 0) continue; /* zero mask word: all digits zero */
      ref_decompose(ct[i], lk, betak, dig);
      for (int lev = 0; lev < lk; lev++) {
        const int64_t d = dig[lev];
        if (d == 0) continue;
        const uint64_t *row = ksk + ((size_t)i * lk + lev) * (n + 1);
        const uint64_t du = (uint64_t)d;
        for (int j = 0; j <= n; j++) out[j] -= du * row[j];
      }
    }
  }
}

void ref_modswitch(const uint64_t *ct_small, int n, int N, uint32_t *out) {
  int logN = 0;
  while ((1 << logN) < N) logN++;
  const int sh = 64 - logN - 2;
  const uint32_t mask = (uint32_t)(2 * N - 1);
  for (int i = 0; i <= n; i++) out[i] = (uint32_t)(((ct_small[i] >> sh) + 1) >> 1) & mask;
}

/* Centred ("mean-compensated") mod switch, applied to a small ciphertext BEFORE ref_modswitch: the rounding error of mask word i,
 * e_i = a_i 2N/q - round(a_i 2N/q), enters the switched phase as sum_i s_i e_i.  The evaluator knows every e_i and E[s_i] = 1/2, so it
 * takes (1/2) sum_i e_i off the body; what is left is sum_i (s_i - 1/2) e_i: variance n / (192 N^2) instead of n / (96 N^2)
 * (binary keys; [K] standard trick, e.g. the "mean compensation" of the TFHE parameter literature).  In place, integer arithmetic. */
void ref_ms_center(uint64_t *cts_small, int count, int n, int N) {
  int logN = 0;
  while ((1 << logN) < N) logN++;
  const int sh = 63 - logN;
  for (int c = 0; c < count; c++) {
    uint64_t *ct = cts_small + (size_t)c * (n + 1);
    int64_t R = 0;
    for (int i = 0; i < n; i++) {
      const uint64_t at = ((ct[i] >> (sh - 1)) + 1) >> 1;          /* round(a / 2^sh), as ref_modswitch before its mask */
      R += (int64_t)(ct[i] - (at << sh));                         /* signed remainder, |.| <= 2^(sh-1) */
    }
    ct[n] -= (uint64_t)(R >> 1);
  }
}

/* ------------------------------------------------------------------ negacyclic helpers */
/* out = X^r * in  in Z[X]/(X^N+1), 0 <= r < 2N */
static void nega_rotate(uint64_t *out, const uint64_t *in, int r, int N) {
  int neg = 0;
  if (r >= N) { r -= N; neg = 1; }
  for (int j = 0; j < N; j++) {
    uint64_t v = (j >= r) ? in[j - r] : (uint64_t)0 - in[j - r + N];
    out[j] = neg ? (uint64_t)0 - v : v;
  }
}

/* ------------------------------------------------------------------ FFT (size M = N/2 complex) */
typedef struct { int M; double *wre, *wim; /* e^{-2 pi i j / M}, j < M/2 */ double *tre, *tim; /* twist e^{i pi j / N}, j < M */ int *rev; } fft_plan_t;
static fft_plan_t g_plans[16];
static int g_nplans = 0;

static const fft_plan_t *get_plan(int N) {
  const int M = N / 2;
  for (int i = 0; i < g_nplans; i++)
    if (g_plans[i].M == M) return &g_plans[i];
  const fft_plan_t *res = NULL;
#pragma omp critical(ref_fft_plan)
  {
    for (int i = 0; i < g_nplans; i++)
      if (g_plans[i].M == M) res = &g_plans[i];
    if (!res) {
      fft_plan_t p;
      p.M = M;
      p.wre = (double *)malloc(sizeof(double) * (M / 2 + 1));
      p.wim = (double *)malloc(sizeof(double) * (M / 2 + 1));
      p.tre = (double *)malloc(sizeof(double) * M);
      p.tim = (double *)malloc(sizeof(double) * M);
      p.rev = (int *)malloc(sizeof(int) * M);
      for (int j = 0; j < M / 2; j++) { p.wre[j] = cos(-2.0 * M_PI * j / M); p.wim[j] = sin(-2.0 * M_PI * j / M); }
      for (int j = 0; j < M; j++) { p.tre[j] = cos(M_PI * j / N); p.tim[j] = sin(M_PI * j / N); }
      int lg = 0;
      while ((1 << lg) < M) lg++;
      for (int j = 0; j < M; j++) {
        int r = 0;
        for (int b = 0; b < lg; b++) if (j & (1 << b)) r |= 1 << (lg - 1 - b);
        p.rev[j] = r;
      }
      g_plans[g_nplans] = p;
      res = &g_plans[g_nplans];
      g_nplans++;
    }
  }
  return res;
}

/* in-place DFT, sign = -1 forward, +1 inverse (unnormalised) */
static void fft_inplace(const fft_plan_t *p, double *re, double *im, int sign) {
  const int M = p->M;
  for (int j = 0; j < M; j++) {
    int r = p->rev[j];
    if (r > j) { double t = re[j]; re[j] = re[r]; re[r] = t; t = im[j]; im[j] = im[r]; im[r] = t; }
  }
  for (int len = 2; len <= M; len <<= 1) {
    const int half = len >> 1, step = M / len;
    for (int s = 0; s < M; s += len) {
      for (int j = 0; j < half; j++) {
        const double wr = p->wre[j * step], wi = (sign < 0) ? p->wim[j * step] : -p->wim[j * step];
        const double xr = re[s + j + half], xi = im[s + j + half];
        const double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
        re[s + j + half] = re[s + j] - tr; im[s + j + half] = im[s + j] - ti;
        re[s + j] += tr; im[s + j] += ti;
      }
    }
  }
}

/* real polynomial (given as doubles, N coeffs) -> M complex evaluations */
static void nega_fft_forward(const fft_plan_t *p, const double *x, double *re, double *im) {
  const int M = p->M;
  for (int j = 0; j < M; j++) {
    const double a = x[j], b = x[j + M];
    re[j] = a * p->tre[j] - b * p->tim[j];
    im[j] = a * p->tim[j] + b * p->tre[j];
  }
  fft_inplace(p, re, im, -1);
}

/* M complex evaluations -> N real coefficients (doubles); destroys re/im */
static void nega_fft_inverse(const fft_plan_t *p, double *re, double *im, double *x) {
  const int M = p->M;
  fft_inplace(p, re, im, +1);
  const double inv = 1.0 / M;
  for (int j = 0; j < M; j++) {
    const double zr = re[j] * inv, zi = im[j] * inv;
    x[j] = zr * p->tre[j] + zi * p->tim[j];
    x[j + M] = zi * p->tre[j] - zr * p->tim[j];
  }
}

static uint64_t f64_to_torus(double d) {
  /* d mod 2^64, exact for |d| < 2^116 */
  const double two64 = 18446744073709551616.0;
  double q = nearbyint(d * (1.0 / two64));
  double r = d - q * two64; /* in [-2^63, 2^63] */
  if (r >= 9223372036854775808.0) r -= two64;
  return (uint64_t)(int64_t)r;
}

/* ------------------------------------------------------------------ bootstrapping key */
void ref_bsk_gen(const uint8_t *s_small, int n, const uint8_t *S_glwe, int k, int N, int l, int beta,
                 double sigma, uint64_t seed, uint64_t *bsk) {
  const int rows = (k + 1) * l;
  const size_t rowsz = (size_t)(k + 1) * N;
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < n; i++) {
    for (int r = 0; r < rows; r++) {
      uint64_t st = seed ^ (0xB5B5ULL + ((uint64_t)i * 256 + (uint64_t)r) * 0xC2B2AE3D27D4EB4FULL);
      uint64_t *row = bsk + ((size_t)i * rows + r) * rowsz;
      uint64_t *B = row + (size_t)k * N;
      for (int c = 0; c < N; c++) B[c] = (uint64_t)rng_gauss_torus(&st, sigma);
      for (int j = 0; j < k; j++) {
        uint64_t *A = row + (size_t)j * N;
        for (int c = 0; c < N; c++) A[c] = ref_splitmix64(&st);
        const uint8_t *S = S_glwe + (size_t)j * N;
        for (int m = 0; m < N; m++) {
          if (!S[m]) continue;
          /* B += X^m * A */
          for (int c = m; c < N; c++) B[c] += A[c - m];
          for (int c = 0; c < m; c++) B[c] -= A[c - m + N];
        }
      }
      if (s_small[i]) {
        const int p = r / l, lev = r % l;
        row[(size_t)p * N] += 1ULL << (64 - beta * (lev + 1));
      }
    }
  }
}

void ref_bsk_to_fourier(const uint64_t *bsk, int n, int k, int N, int l, double *bsk_f) {
  const fft_plan_t *plan = get_plan(N);
  const int M = N / 2;
  const long npoly = (long)n * (k + 1) * l * (k + 1);
#pragma omp parallel
  {
    double *x = (double *)malloc(sizeof(double) * N);
    double *re = (double *)malloc(sizeof(double) * M);
    double *im = (double *)malloc(sizeof(double) * M);
#pragma omp for schedule(static)
    for (long q = 0; q < npoly; q++) {
      const uint64_t *src = bsk + (size_t)q * N;
      for (int c = 0; c < N; c++) x[c] = (double)(int64_t)src[c];
      nega_fft_forward(plan, x, re, im);
      double *dst = bsk_f + (size_t)q * N; /* M complex = N doubles */
      for (int c = 0; c < M; c++) { dst[2 * c] = re[c]; dst[2 * c + 1] = im[c]; }
    }
    free(x); free(re); free(im);
  }
}

/* ------------------------------------------------------------------ test vector */
void ref_build_testvector(const int64_t *table, int w, int N, uint64_t *tv) {
  const int box = N >> w, half = box >> 1;
  for (int j = 0; j < N; j++) {
    const int jj = j + half;
    tv[j] = (jj < N) ? (uint64_t)table[jj / box] : (uint64_t)0 - (uint64_t)table[0];
  }
}

/* ------------------------------------------------------------------ PBS */
typedef struct {
  uint64_t *acc, *rot, *diff;    /* (k+1)*N each */
  double *x, *dre, *dim;         /* N, rows*M, rows*M */
  double *ore, *oim;             /* (k+1)*M */
  int32_t *dig;                  /* rows*N (exact path) */
  uint32_t *ms;                  /* n+1 */
  uint64_t *tv;                  /* N */
} pbs_ws_t;

static void ws_alloc(pbs_ws_t *w, int n, int k, int N, int l) {
  const int rows = (k + 1) * l, M = N / 2;
  w->acc = (uint64_t *)malloc(sizeof(uint64_t) * (k + 1) * N);
  w->rot = (uint64_t *)malloc(sizeof(uint64_t) * (k + 1) * N);
  w->diff = (uint64_t *)malloc(sizeof(uint64_t) * (k + 1) * N);
  w->x = (double *)malloc(sizeof(double) * N);
  w->dre = (double *)malloc(sizeof(double) * rows * M);
  w->dim = (double *)malloc(sizeof(double) * rows * M);
  w->ore = (double *)malloc(sizeof(double) * (k + 1) * M);
  w->oim = (double *)malloc(sizeof(double) * (k + 1) * M);
  w->dig = (int32_t *)malloc(sizeof(int32_t) * rows * N);
  w->ms = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
  w->tv = (uint64_t *)malloc(sizeof(uint64_t) * N);
}
static void ws_free(pbs_ws_t *w) {
  free(w->acc); free(w->rot); free(w->diff); free(w->x); free(w->dre); free(w->dim);
  free(w->ore); free(w->oim); free(w->dig); free(w->ms); free(w->tv);
}

/* ACC += BSK_i [x] diff, FFT path */
static void external_product_fft(const fft_plan_t *plan, pbs_ws_t *w, const double *bsk_i, int k, int N, int l, int beta) {
  const int rows = (k + 1) * l, M = N / 2;
  int32_t dg[64];
  for (int p = 0; p <= k; p++) {
    for (int lev = 0; lev < l; lev++) {
      for (int c = 0; c < N; c++) { /* one digit plane at a time keeps the code simple */
        ref_decompose(w->diff[(size_t)p * N + c], l, beta, dg);
        w->x[c] = (double)dg[lev];
      }
      nega_fft_forward(plan, w->x, w->dre + (size_t)(p * l + lev) * M, w->dim + (size_t)(p * l + lev) * M);
    }
  }
  for (int q = 0; q <= k; q++) {
    double *ore = w->ore + (size_t)q * M, *oim = w->oim + (size_t)q * M;
    for (int c = 0; c < M; c++) { ore[c] = 0.0; oim[c] = 0.0; }
    for (int r = 0; r < rows; r++) {
      const double *kf = bsk_i + ((size_t)r * (k + 1) + q) * N; /* M complex interleaved */
      const double *dre = w->dre + (size_t)r * M, *dim = w->dim + (size_t)r * M;
      for (int c = 0; c < M; c++) {
        const double kr = kf[2 * c], ki = kf[2 * c + 1];
        ore[c] += dre[c] * kr - dim[c] * ki;
        oim[c] += dre[c] * ki + dim[c] * kr;
      }
    }
    nega_fft_inverse(plan, ore, oim, w->x);
    for (int c = 0; c < N; c++) w->acc[(size_t)q * N + c] += f64_to_torus(w->x[c]);
  }
}

/* ACC += BSK_i [x] diff, exact schoolbook mod 2^64 on the standard-domain key */
static void external_product_exact(pbs_ws_t *w, const uint64_t *bsk_i, int k, int N, int l, int beta) {
  const int rows = (k + 1) * l;
  int32_t dg[64];
  for (int p = 0; p <= k; p++)
    for (int c = 0; c < N; c++) {
      ref_decompose(w->diff[(size_t)p * N + c], l, beta, dg);
      for (int lev = 0; lev < l; lev++) w->dig[(size_t)(p * l + lev) * N + c] = dg[lev];
    }
  for (int q = 0; q <= k; q++) {
    uint64_t *acc = w->acc + (size_t)q * N;
    for (int r = 0; r < rows; r++) {
      const uint64_t *key = bsk_i + ((size_t)r * (k + 1) + q) * N;
      const int32_t *d = w->dig + (size_t)r * N;
      for (int a = 0; a < N; a++) {
        if (d[a] == 0) continue;
        const uint64_t da = (uint64_t)(int64_t)d[a];
        for (int b = 0; b < N - a; b++) acc[a + b] += da * key[b];
        for (int b = N - a; b < N; b++) acc[a + b - N] -= da * key[b];
      }
    }
  }
}

static void pbs_one(const fft_plan_t *plan, pbs_ws_t *w, const uint64_t *ct_small, int n,
                    const double *bsk_f, const uint64_t *bsk, int use_exact,
                    int k, int N, int l, int beta, const int64_t *table, int wbits,
                    int D_out, uint64_t *out) {
  const int rows = (k + 1) * l;
  ref_modswitch(ct_small, n, N, w->ms);
  ref_build_testvector(table, wbits, N, w->tv);
  memset(w->acc, 0, sizeof(uint64_t) * (size_t)k * N);
  {
    const int bt = (int)w->ms[n];
    nega_rotate(w->acc + (size_t)k * N, w->tv, (2 * N - bt) % (2 * N), N);
  }
  for (int i = 0; i < n; i++) {
    const int a = (int)w->ms[i];
    if (a == 0) continue;
    for (int p = 0; p <= k; p++) {
      nega_rotate(w->rot + (size_t)p * N, w->acc + (size_t)p * N, a, N);
      for (int c = 0; c < N; c++) w->diff[(size_t)p * N + c] = w->rot[(size_t)p * N + c] - w->acc[(size_t)p * N + c];
    }
    if (use_exact) external_product_exact(w, bsk + (size_t)i * rows * (k + 1) * N, k, N, l, beta);
    else           external_product_fft(plan, w, bsk_f + (size_t)i * rows * (k + 1) * N, k, N, l, beta);
  }
  /* sample extract, coefficient 0 */
  for (int j = 0; j < k; j++) {
    const uint64_t *A = w->acc + (size_t)j * N;
    out[(size_t)j * N] = A[0];
    for (int m = 1; m < N; m++) out[(size_t)j * N + m] = (uint64_t)0 - A[N - m];
  }
  for (int j = k * N; j < D_out; j++) out[j] = 0;
  out[D_out] = w->acc[(size_t)k * N];
}

int ref_pbs_batch(const uint64_t *cts_small, int count, int n,
                  const double *bsk_f, const uint64_t *bsk, int use_exact,
                  int k, int N, int l, int beta,
                  const int64_t *tables, int w, const int32_t *table_idx,
                  int D_out, uint64_t *cts_out) {
  const fft_plan_t *plan = get_plan(N);
#pragma omp parallel
  {
    pbs_ws_t ws;
    ws_alloc(&ws, n, k, N, l);
#pragma omp for schedule(dynamic, 1)
    for (int c = 0; c < count; c++) {
      const int64_t *tab = tables + ((size_t)(table_idx ? table_idx[c] : 0) << w);
      pbs_one(plan, &ws, cts_small + (size_t)c * (n + 1), n, bsk_f, bsk, use_exact, k, N, l, beta,
              tab, w, D_out, cts_out + (size_t)c * (D_out + 1));
    }
    ws_free(&ws);
  }
  return 0;
}

/* ------------------------------------------------------------------ two-bit blind rotation (product: pbs_core.h, MB = 1)
 * [K: Zhou, Yang, Zhang, Wang, "Faster bootstrapping with multiple addends", 2018; Bourse et al.]
 *   X^{a1 s1 + a2 s2} = 1 + s1(1-s2)(X^{a1}-1) + (1-s1)s2 (X^{a2}-1) + s1 s2 (X^{a1+a2}-1)
 *   ACC += sum_w (X^{e_w} - 1) * (GGSW(b_w) [x] ACC)
 * The key is an ordinary bootstrapping key for the derived secret of 3n/2 bits that ref_pair_secret builds.
 * Exact arithmetic (schoolbook mod 2^64) only: this is the definition the device path is checked against. */
void ref_pair_secret(const uint8_t *s, int n, uint8_t *out /* 3n/2 */) {
  for (int i = 0; i + 1 < n; i += 2) {
    out[3 * (i / 2) + 0] = (uint8_t)(s[i] && !s[i + 1]);
    out[3 * (i / 2) + 1] = (uint8_t)(!s[i] && s[i + 1]);
    out[3 * (i / 2) + 2] = (uint8_t)(s[i] && s[i + 1]);
  }
}

int ref_pbs_mb2_batch(const uint64_t *cts_small, int count, int n, const uint64_t *bsk3 /* [3n/2][rows][k+1][N] */,
                      int k, int N, int l, int beta, const int64_t *tables, int w, const int32_t *table_idx,
                      int D_out, uint64_t *cts_out) {
  if (n % 2) return -1;
  const int rows = (k + 1) * l;
  const size_t blk = (size_t)rows * (k + 1) * N;
#pragma omp parallel
  {
    pbs_ws_t ws;
    ws_alloc(&ws, n, k, N, l);
    uint64_t *sum = (uint64_t *)malloc(sizeof(uint64_t) * (k + 1) * N);
    uint64_t *keep = (uint64_t *)malloc(sizeof(uint64_t) * (k + 1) * N);
#pragma omp for schedule(dynamic, 1)
    for (int c = 0; c < count; c++) {
      const uint64_t *ct = cts_small + (size_t)c * (n + 1);
      uint64_t *out = cts_out + (size_t)c * (D_out + 1);
      ref_modswitch(ct, n, N, ws.ms);
      ref_build_testvector(tables + ((size_t)(table_idx ? table_idx[c] : 0) << w), w, N, ws.tv);
      memset(ws.acc, 0, sizeof(uint64_t) * (size_t)k * N);
      nega_rotate(ws.acc + (size_t)k * N, ws.tv, (2 * N - (int)ws.ms[n]) % (2 * N), N);
      for (int i = 0; i < n; i += 2) {
        const int e[3] = {(int)ws.ms[i], (int)ws.ms[i + 1], (int)((ws.ms[i] + ws.ms[i + 1]) % (2u * N))};
        memcpy(ws.diff, ws.acc, sizeof(uint64_t) * (k + 1) * N);   /* the gadget decomposition is of ACC itself */
        memcpy(keep, ws.acc, sizeof(uint64_t) * (k + 1) * N);
        memset(sum, 0, sizeof(uint64_t) * (k + 1) * N);
        for (int v = 0; v < 3; v++) {
          memset(ws.acc, 0, sizeof(uint64_t) * (k + 1) * N);
          external_product_exact(&ws, bsk3 + (size_t)(3 * (i / 2) + v) * blk, k, N, l, beta);   /* acc = GGSW(b_v) [x] ACC */
          for (int p = 0; p <= k; p++) {
            nega_rotate(ws.rot + (size_t)p * N, ws.acc + (size_t)p * N, e[v], N);
            for (int x = 0; x < N; x++) sum[(size_t)p * N + x] += ws.rot[(size_t)p * N + x] - ws.acc[(size_t)p * N + x];
          }
        }
        for (int x = 0; x < (k + 1) * N; x++) ws.acc[x] = keep[x] + sum[x];
      }
      for (int j = 0; j < k; j++) {
        const uint64_t *A = ws.acc + (size_t)j * N;
        out[(size_t)j * N] = A[0];
        for (int m = 1; m < N; m++) out[(size_t)j * N + m] = (uint64_t)0 - A[N - m];
      }
      for (int j = k * N; j < D_out; j++) out[j] = 0;
      out[D_out] = ws.acc[(size_t)k * N];
    }
    free(sum); free(keep);
    ws_free(&ws);
  }
  return 0;
}

/* ------------------------------------------------------------------ levelled ops */
void ref_conv2d(const uint64_t *in, int Cin, int H, int W, int D, const int32_t *weight, int Cout,
                int KH, int KW, int stride, int pad, uint64_t *out) {
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const size_t L = (size_t)D + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int co = 0; co < Cout; co++)
    for (int y = 0; y < Ho; y++)
      for (int x = 0; x < Wo; x++) {
        uint64_t *o = out + (((size_t)co * Ho + y) * Wo + x) * L;
        memset(o, 0, sizeof(uint64_t) * L);
        for (int ci = 0; ci < Cin; ci++)
          for (int ky = 0; ky < KH; ky++)
            for (int kx = 0; kx < KW; kx++) {
              const int iy = y * stride + ky - pad, ix = x * stride + kx - pad;
              if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue; /* zero padding = trivial zero ct */
              const int32_t wv = weight[(((size_t)co * Cin + ci) * KH + ky) * KW + kx];
              if (wv == 0) continue;
              const uint64_t wu = (uint64_t)(int64_t)wv;
              const uint64_t *s = in + (((size_t)ci * H + iy) * W + ix) * L;
              for (size_t t = 0; t < L; t++) o[t] += wu * s[t];
            }
      }
}

void ref_sum_pool(const uint64_t *in, int C, int H, int W, int D, int K, uint64_t *out) {
  const int Ho = H / K, Wo = W / K;
  const size_t L = (size_t)D + 1;
  for (int c = 0; c < C; c++)
    for (int y = 0; y < Ho; y++)
      for (int x = 0; x < Wo; x++) {
        uint64_t *o = out + (((size_t)c * Ho + y) * Wo + x) * L;
        memset(o, 0, sizeof(uint64_t) * L);
        for (int ky = 0; ky < K; ky++)
          for (int kx = 0; kx < K; kx++) {
            const uint64_t *s = in + (((size_t)c * H + y * K + ky) * W + x * K + kx) * L;
            for (size_t t = 0; t < L; t++) o[t] += s[t];
          }
      }
}

/* ------------------------------------------------------------------ exact rounding + table */
int ref_round_lut_batch(const uint64_t *cts_in, int count, int D, int p, int r,
                        const ref_tier_t *bt, const ref_tier_t *tt,
                        const int64_t *tables, int w, const int32_t *table_idx, uint64_t *cts_out) {
  const size_t L = (size_t)D + 1;
  const fft_plan_t *plan_b = (r > 0) ? get_plan(bt->N) : NULL;
  const fft_plan_t *plan_t = get_plan(tt->N);
#pragma omp parallel
  {
    pbs_ws_t wsb, wst;
    if (r > 0) ws_alloc(&wsb, bt->n, bt->k, bt->N, bt->l);
    ws_alloc(&wst, tt->n, tt->k, tt->N, tt->l);
    uint64_t *c = (uint64_t *)malloc(sizeof(uint64_t) * L);
    uint64_t *d = (uint64_t *)malloc(sizeof(uint64_t) * L);
    uint64_t *R = (uint64_t *)malloc(sizeof(uint64_t) * L);
    uint64_t *sm = (uint64_t *)malloc(sizeof(uint64_t) * ((r > 0 && bt->n > tt->n ? bt->n : tt->n) + 1));
#pragma omp for schedule(dynamic, 1)
    for (int e = 0; e < count; e++) {
      memcpy(c, cts_in + (size_t)e * L, sizeof(uint64_t) * L);
      if (r > 0) c[D] += 1ULL << (63 - p + r - 1);
      for (int i = 0; i < r; i++) {
        const int sh = p - i;
        for (size_t t = 0; t < L; t++) d[t] = c[t] << sh;
        ref_keyswitch(d, 1, D, bt->ksk, bt->n, bt->lk, bt->betak, sm);
        const int64_t v = (int64_t)(1ULL << (62 - p + i));
        pbs_one(plan_b, &wsb, sm, bt->n, bt->bsk_f, NULL, 0, bt->k, bt->N, bt->l, bt->beta, &v, 0, D, R);
        for (size_t t = 0; t < L; t++) c[t] += R[t];
        c[D] -= (uint64_t)v;
      }
      ref_keyswitch(c, 1, D, tt->ksk, tt->n, tt->lk, tt->betak, sm);
      const int64_t *tab = tables + ((size_t)(table_idx ? table_idx[e] : 0) << w);
      pbs_one(plan_t, &wst, sm, tt->n, tt->bsk_f, NULL, 0, tt->k, tt->N, tt->l, tt->beta, tab, w, D,
              cts_out + (size_t)e * L);
    }
    free(c); free(d); free(R); free(sm);
    if (r > 0) ws_free(&wsb);
    ws_free(&wst);
  }
  return 0;
}
