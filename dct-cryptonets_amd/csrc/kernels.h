// kernels.h -- gfx950 kernels of the homomorphic-evaluation engine.
//   K1 conv2d / K2 add, sum-pool, affine      ciphertext streaming, u64 wrap arithmetic (HBM bound)
//   K3 key switch = decompose + integer GEMM  (ks_decompose, ks_gemm)
//   K4+K5+K6 programmable bootstrap           (pbs_kernel: mod-switch, blind rotate, sample extract)
//   K9 keygen / encrypt / decrypt             client side, off the timed path
// Semantics: oracle/tfhe_ref.h (the CPU restatement these are checked against).
#pragma once
#include <hip/hip_runtime.h>
#include "pbs_core.h"

namespace dctfhe {

// ------------------------------------------------------------------------------------------ rng
// Counter-based CSPRNG: ChaCha20 (D. J. Bernstein's layout: 256-bit key, 64-bit block counter, 64-bit stream id), one 64-byte
// block = eight u64 outputs; (key, stream, index) -> 64 bits, any index in any order, so every kernel draws exactly the
// values keygen drew (the test views regenerate keys from the client handle).  Two keys per client: the SECRET key feeds the
// secret-key bits and every noise term; the PUBLIC key (one ChaCha block of the secret one) feeds the ciphertext / key masks.
// The reference's runtime (Concrete) seeds an AES-CTR generator the same way.
struct rng_key { uint32_t k[8]; };

#define DCTFHE_QR(a, b, c, d)                                   \
  a += b; d ^= a; d = (d << 16) | (d >> 16);                    \
  c += d; b ^= c; b = (b << 12) | (b >> 20);                    \
  a += b; d ^= a; d = (d << 8) | (d >> 24);                     \
  c += d; b ^= c; b = (b << 7) | (b >> 25);

// one ChaCha20 block; out[16] little-endian words
__host__ __device__ inline void chacha20_block(const rng_key& key, uint64_t stream, uint64_t block, uint32_t* out) {
  const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                           key.k[4], key.k[5], key.k[6], key.k[7], (uint32_t)block, (uint32_t)(block >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
  uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7], x8 = in[8], x9 = in[9],
           x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
#pragma unroll
  for (int r = 0; r < 10; r++) {
    DCTFHE_QR(x0, x4, x8, x12) DCTFHE_QR(x1, x5, x9, x13) DCTFHE_QR(x2, x6, x10, x14) DCTFHE_QR(x3, x7, x11, x15)
    DCTFHE_QR(x0, x5, x10, x15) DCTFHE_QR(x1, x6, x11, x12) DCTFHE_QR(x2, x7, x8, x13) DCTFHE_QR(x3, x4, x9, x14)
  }
  out[0] = x0 + in[0]; out[1] = x1 + in[1]; out[2] = x2 + in[2]; out[3] = x3 + in[3]; out[4] = x4 + in[4]; out[5] = x5 + in[5];
  out[6] = x6 + in[6]; out[7] = x7 + in[7]; out[8] = x8 + in[8]; out[9] = x9 + in[9]; out[10] = x10 + in[10]; out[11] = x11 + in[11];
  out[12] = x12 + in[12]; out[13] = x13 + in[13]; out[14] = x14 + in[14]; out[15] = x15 + in[15];
}
// (key, stream, index) -> 64 uniform bits: word pair (index & 7) of block (index >> 3)
__host__ __device__ inline uint64_t rnd64(const rng_key& key, uint64_t stream, uint64_t idx) {
  uint32_t o[16];
  chacha20_block(key, stream, idx >> 3, o);
  uint64_t v = 0;
#pragma unroll
  for (int w = 0; w < 8; w++)
    if ((int)(idx & 7) == w) v = (uint64_t)o[2 * w] | ((uint64_t)o[2 * w + 1] << 32);
  return v;
}
__device__ __forceinline__ int64_t gauss_torus(const rng_key& key, uint64_t stream, uint64_t idx, double sigma) {
  if (sigma <= 0.0) return 0;
  const double u1 = ((double)(rnd64(key, stream, 2 * idx) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(rnd64(key, stream, 2 * idx + 1) >> 11)) * (1.0 / 9007199254740992.0);
  const double g = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
  return (int64_t)rint(g * sigma * 18446744073709551616.0);
}

// stream ids.  Masks are drawn from the PUBLIC key at stream s, the matching noise from the SECRET key at stream s + 1.
enum : uint64_t { STREAM_BIGKEY = 1, STREAM_SMALLKEY = 2, STREAM_PUBKEY = 3, STREAM_ENCKDF = 4, STREAM_KSK = 16, STREAM_BSK_MASK = 64, STREAM_ENC = 256 };

__global__ void k_gen_bits(rng_key key, uint64_t stream, uint8_t* out, int len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) out[i] = (uint8_t)(rnd64(key, stream, (uint64_t)i) >> 63);
}
__global__ void k_rng_fill(rng_key key, uint64_t stream, uint64_t idx0, uint64_t* out, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = rnd64(key, stream, idx0 + i);
}

__device__ __forceinline__ uint64_t block_reduce_add(uint64_t v, uint64_t* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[wv] = v;
  __syncthreads();
  uint64_t s = 0;
  if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
  return s;  // valid in thread 0
}

// ------------------------------------------------------------------------------------------ client
// one block per ciphertext: a random on [0,dim_eff), b = <a,S> + phase + e
// Rows of `dim` mask words + the body (dim = D: the full-width form; dim < D: the compact wire form -- every mask word from dim_eff on
// is zero anyway).  The draws are indexed as in the full-width form, so both forms of one call hold the same ciphertexts.
__global__ void k_lwe_encrypt(const uint8_t* __restrict__ S, int D, int dim, int dim_eff, const uint64_t* __restrict__ phases,
                              double sigma, rng_key pub, rng_key sec, uint64_t stream, uint64_t* __restrict__ cts) {
  __shared__ uint64_t red[16];
  const size_t c = blockIdx.x;
  uint64_t* ct = cts + c * (size_t)(dim + 1);
  uint64_t part = 0;
  for (int j = threadIdx.x; j < dim; j += blockDim.x) {
    const uint64_t a = (j < dim_eff) ? rnd64(pub, stream, c * (uint64_t)(D + 1) + j) : 0;
    ct[j] = a;
    if (S[j]) part += a;
  }
  const uint64_t s = block_reduce_add(part, red);
  if (threadIdx.x == 0) ct[dim] = s + phases[c] + (uint64_t)gauss_torus(sec, stream + 1, c, sigma);
}

__global__ void k_lwe_phase(const uint8_t* __restrict__ S, int dim, const uint64_t* __restrict__ cts, uint64_t* __restrict__ phases) {
  __shared__ uint64_t red[16];
  const size_t c = blockIdx.x;
  const uint64_t* ct = cts + c * (size_t)(dim + 1);
  uint64_t part = 0;
  for (int j = threadIdx.x; j < dim; j += blockDim.x)
    if (S[j]) part += ct[j];
  const uint64_t s = block_reduce_add(part, red);
  if (threadIdx.x == 0) phases[c] = ct[dim] - s;
}

// ------------------------------------------------------------------------------------------ keygen
// key-switch key: one block per row (i, lev): LWE_s(S_i * 2^(64 - betak (lev+1)))
// The key lives on a COARSE TORUS GRID of 2^-(8 limbs): its mask words are drawn on that grid and the body is rounded to it (a rounding
// of 2^-(8 limbs) / sqrt(12), far below the row's noise sigma: dctfhe.hip ks_limbs() picks limbs >= (lk betak + 6) / 8).  A key-switch
// output only ever meets a mod switch to 2N <= 2^14 levels under a noise of 2^-5 .. 2^-11: the low 32-48 bits of a 64-bit key word
// carried nothing, and the matrix-core GEMM spent half to three quarters of its work multiplying them (round 2: 8 byte limbs per word).
__global__ void k_ksk_gen(const uint8_t* __restrict__ S, const uint8_t* __restrict__ s, int n, int lk, int betak, int limbs,
                          double sigma, rng_key pub, rng_key sec, uint64_t stream, uint64_t* __restrict__ ksk) {
  __shared__ uint64_t red[16];
  const size_t row = blockIdx.x;
  const int i = (int)(row / lk), lev = (int)(row % lk);
  uint64_t* dst = ksk + row * (size_t)(n + 1);
  const int drop = 64 - 8 * limbs;                                   // low bits that are not stored (0 for limbs = 8)
  const uint64_t mask = drop ? ~0ULL << drop : ~0ULL, half = drop ? 1ULL << (drop - 1) : 0;
  uint64_t part = 0;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const uint64_t a = rnd64(pub, stream, row * (uint64_t)(n + 1) + j) & mask;
    dst[j] = a;
    if (s[j]) part += a;
  }
  const uint64_t sum = block_reduce_add(part, red);
  if (threadIdx.x == 0) {
    uint64_t b = sum + (uint64_t)gauss_torus(sec, stream + 1, row, sigma);
    if (S[i]) b += 1ULL << (64 - betak * (lev + 1));
    dst[n] = (b + half) & mask;
  }
}
// any key word off the 2^-(8 limbs) grid?  (an imported key must be what k_ksk_gen makes: the limb form drops the low bytes unseen)
__global__ void k_ksk_off_grid(const uint64_t* __restrict__ ksk, size_t words, int limbs, int* __restrict__ flag) {
  const uint64_t low = limbs >= 8 ? 0 : ~(~0ULL << (64 - 8 * limbs));
  int bad = 0;
  for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < words; x += (size_t)gridDim.x * blockDim.x) bad |= (ksk[x] & low) != 0;
  if (bad) atomicOr(flag, 1);
}

// secret of the two-bit blind rotation: per pair (s1, s2) the three products s1(1-s2), (1-s1)s2, s1 s2
__global__ void k_pair_secret(const uint8_t* __restrict__ s, int n, uint8_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const uint8_t a = s[2 * i], b = s[2 * i + 1];
  out[3 * i] = a && !b; out[3 * i + 1] = !a && b; out[3 * i + 2] = a && b;
}

// bootstrap key, standard domain, rows [i0, i0+ni) x rows_per_bit: GLWE(0) + s_i * gadget.
// One block per row; the mask polynomial is parked in LDS while the body accumulates A * S.
__global__ void k_bsk_gen_std(const uint8_t* __restrict__ s_small, const uint8_t* __restrict__ S_glwe, int i0, int k, int N, int l,
                              int beta, double sigma, rng_key pub, rng_key sec, uint64_t stream, uint64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* A = reinterpret_cast<uint64_t*>(smem_raw);
  const int rows = (k + 1) * l;
  const int i = i0 + (int)(blockIdx.x / rows), r = (int)(blockIdx.x % rows);
  const uint64_t grow = (uint64_t)i * rows + r;  // global row id: randomness does not depend on chunking
  uint64_t* row = out + (size_t)blockIdx.x * (k + 1) * N;
  uint64_t* B = row + (size_t)k * N;
  for (int c = threadIdx.x; c < N; c += blockDim.x)
    B[c] = (uint64_t)gauss_torus(sec, stream + 1, grow * (uint64_t)N + c, sigma);
  for (int j = 0; j < k; j++) {
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
      const uint64_t a = rnd64(pub, stream, (grow * (uint64_t)k + j) * (uint64_t)N + c);
      A[c] = a;
      row[(size_t)j * N + c] = a;
    }
    __syncthreads();
    const uint8_t* Sj = S_glwe + (size_t)j * N;
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
      uint64_t acc = 0;
      for (int m = 0; m < N; m++) {
        if (!Sj[m]) continue;  // uniform branch
        const int src = c - m;
        acc += (src >= 0) ? A[src] : (uint64_t)0 - A[src + N];
      }
      B[c] += acc;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_small[i]) {
    const int p = r / l, lev = r % l;
    row[(size_t)p * N] += 1ULL << (64 - beta * (lev + 1));
  }
}

// standard-domain key polynomials -> Fourier device layout; GROUPS polynomials per block
template <int LOGN, int P, int GROUPS>
__global__ void __launch_bounds__((fft_geom<LOGN - 1, P>::T * GROUPS))
k_bsk_fourier(const uint64_t* __restrict__ polys, size_t npoly, const cplx* __restrict__ tw_g, cplx* __restrict__ out) {
  using F = fft_geom<LOGN - 1, P>;
  constexpr int T = F::T, N = 1 << LOGN, M = N / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  cplx* tw = reinterpret_cast<cplx*>(smem_raw);
  cplx* exch_all = tw + F::TW_ELEMS;
  for (int x = threadIdx.x; x < F::TW_ELEMS; x += blockDim.x) tw[x] = tw_g[x];
  __syncthreads();
  const int g = threadIdx.x / T, t = threadIdx.x % T;
  size_t q = (size_t)blockIdx.x * GROUPS + g;
  if (q >= npoly) q = npoly - 1;  // redundant work keeps every thread on the barriers
  cplx* exch = exch_all + (size_t)g * F::EXCH_ELEMS;
  key_poly_to_fourier<LOGN, P>(polys + q * N, out + q * M, t, tw, exch, [] { __syncthreads(); }, [] { __builtin_amdgcn_wave_barrier(); });
}

// ------------------------------------------------------------------------------------------ K3 key switch
// Step 1: digits of (ct << shift) for every mask word, offset to unsigned: dig' = dig + B/2 in [0,B).
// Layout digits[c][i*lk + lev] (u8).  Also copies the (shifted) body.
// Only the first Deff mask words are decomposed (digits [count][Deff*lk]): the caller knows the rest to be zero (nested
// keys: a ciphertext that came out of a ring of dimension kN <= Deff has a zero tail), and a zero word contributes nothing.
__global__ void k_ks_decompose(const uint64_t* __restrict__ cts, size_t count, size_t L /* row stride in words; body at L-1 */, int Deff, int shift,
                               uint64_t body_add, int lk, int betak, uint8_t* __restrict__ digits, uint64_t* __restrict__ bodies) {
  const size_t total = count * (size_t)Deff;
  const int half = 1 << (betak - 1);
  for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (size_t)gridDim.x * blockDim.x) {
    const size_t c = x / Deff;
    const int i = (int)(x % Deff);
    const uint64_t v = cts[c * L + i] << shift;
    const int tot = lk * betak;
    uint64_t xx = (v + (1ULL << (63 - tot))) >> (64 - tot);
    const uint64_t B = 1ULL << betak, mask = B - 1;
    uint64_t carry = 0;
    uint8_t* dst = digits + (c * (size_t)Deff + i) * lk;
    for (int lev = lk - 1; lev >= 0; lev--) {
      uint64_t d = (xx & mask) + carry;
      xx >>= betak;
      int dv;
      if (d >= (uint64_t)half) { dv = (int)d - (int)B; carry = 1; } else { dv = (int)d; carry = 0; }
      dst[lev] = (uint8_t)(dv + half);
    }
    if (i == 0) bodies[c] = (cts[c * L + L - 1] << shift) + body_add;      // the affine step of a look-up without rounding steps rides along
  }
}

// Step 2: out[c][j] = body_c*[j==n] - sum_r (dig'[c][r] - B/2) * ksk[r][j]
//       = body_c*[j==n] + (B/2) * colsum[j] - sum_r dig'[c][r] * ksk[r][j],   colsum[j] = sum_r ksk[r][j].
// Block: CT ciphertexts x 256 columns; digits are wave-uniform (scalar loads), the key is read
// once per block -- coalesced 8 B per lane -- and reused for CT ciphertexts.
template <int CT>
__global__ void __launch_bounds__(256)
k_ks_gemm(const uint8_t* __restrict__ digits, const uint64_t* __restrict__ bodies, size_t count, int R /* D*lk */,
          const uint64_t* __restrict__ ksk, const uint64_t* __restrict__ colsum, int n, int betak,
          uint64_t* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const size_t c0 = (size_t)blockIdx.y * CT;
  const bool live = j <= n;
  const int jj = live ? j : n;
  uint64_t acc[CT];
#pragma unroll
  for (int c = 0; c < CT; c++) acc[c] = 0;
  const uint8_t* dbase = digits + c0 * (size_t)R;
  for (int r = 0; r < R; r += 4) {
    uint64_t kv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) kv[u] = ksk[(size_t)(r + u) * (n + 1) + jj];
#pragma unroll
    for (int c = 0; c < CT; c++) {
      const size_t cc = (c0 + c < count) ? (size_t)c : 0;
      const uint32_t d4 = *reinterpret_cast<const uint32_t*>(dbase + cc * (size_t)R + r);  // wave-uniform
#pragma unroll
      for (int u = 0; u < 4; u++) acc[c] += (uint64_t)((d4 >> (8 * u)) & 0xFF) * kv[u];
    }
  }
  if (!live) return;
  const uint64_t fix = ((uint64_t)1 << (betak - 1)) * colsum[j];
#pragma unroll
  for (int c = 0; c < CT; c++) {
    if (c0 + c >= count) break;
    out[(c0 + c) * (size_t)(n + 1) + j] = (j == n ? bodies[c0 + c] : 0) + fix - acc[c];
  }
}

// ---- key switch on the matrix cores -----------------------------------------------------------------
// out[c][j] = fix[j] - sum_r dig'[c][r] * ksk[r][j] is a GEMM with a tiny-integer left operand (digits in
// [0, 2^betak)) and a u64 right operand.  The key is re-expressed once, at keygen, as 8 signed byte limbs
// (ksk = sum_k s_k 2^(8k), s_k in [-128,127], carries propagated), stored K-contiguous per output column:
// kskT[(j*8 + k) * R + r].  Then C[c][j*8+k] = sum_r dig'[c][r] * s_k[r][j] is an i8 x i8 -> i32 GEMM for
// v_mfma_i32_32x32x32_i8 (|C| <= R * 127 * 128 < 2^31 for R <= 2^17), and out = fix - sum_k C_k << 8k (mod 2^64).
// Only the TOP `limbs` byte limbs of a key word are stored (the key sits on the 2^-(8 limbs) grid: the others are zero and carry nothing
// up): column j*limbs + k' holds limb 8 - limbs + k'.
__global__ void k_ksk_to_limbs(const uint64_t* __restrict__ ksk, int R, int n, int limbs, int ncol_pad /* limbs*(n+1) rounded up */, int8_t* __restrict__ kskT) {
  const size_t total = (size_t)R * (ncol_pad / limbs);
  for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (size_t)gridDim.x * blockDim.x) {
    const size_t r = x % R;
    const size_t j = x / R;
    uint64_t v = (j <= (size_t)n) ? ksk[r * (size_t)(n + 1) + j] : 0;
    v >>= 64 - 8 * limbs;                                              // (limbs = 8: a shift by 0)
    for (int k = 0; k < limbs; k++) {
      int b = (int)(v & 0xFF);
      v >>= 8;
      if (b >= 128) { b -= 256; v += 1; }
      kskT[(j * limbs + k) * (size_t)R + r] = (int8_t)b;
    }
  }
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Block: 128 ciphertexts x 128 limb columns (16 output words), K step 64, 4 waves each owning a 64 x 64 quadrant
// (2 x 2 MFMA tiles).  Both operands are K-contiguous, so a lane's 16-byte fragment is one ds_read_b128 and
// the A and B fragments of a lane cover the same 16 k's whatever order the hardware walks them in.
template <int LIMBS>        // byte limbs stored per key word: 2, 4 or 8 (the top ones)
__global__ void __launch_bounds__(256)
k_ks_mfma(const uint8_t* __restrict__ digits, const uint64_t* __restrict__ bodies, size_t count, int R /* rows used: Deff*lk */,
          const int8_t* __restrict__ kskT, int ldk /* row stride of kskT: D*lk */, int ncol_pad, const uint64_t* __restrict__ colsum /* over the R rows used */,
          int n, int betak, uint64_t* __restrict__ out) {
  constexpr int BM = 128, BN = 128, BK = 128, LD = BK + 16;    // +16 B per row: rows land on different bank groups
  constexpr int SEG = BK / 16, RPP = 256 / SEG, NU = BM / RPP;  // 16-byte segments per row, rows staged per pass, passes
  __shared__ __attribute__((aligned(16))) int8_t As[BM * LD];
  __shared__ __attribute__((aligned(16))) int8_t Bs[BN * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (1-D grid of 8 * cpx * nrb workgroups; workgroups b and b + 8 share an XCD and its 4 MB L2): every XCD
  // owns a strip of cpx column blocks of the key and walks the ciphertext row blocks with the columns fastest, so the ~32 tiles
  // an XCD runs at a time are ~4 row blocks x its 7 column blocks -- each K-slice of the digits serves 7 tiles out of L2 and each
  // K-slice of the key 4-5.  (Round 1: x = column, y = row over all XCDs: the 83 MB limb key was re-read from HBM once per row
  // block, 10.6 GB per launch.)
  const int ncb = ncol_pad / BN, nrb = (int)((count + BM - 1) / BM), cpx = (ncb + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cb = xcd * cpx + slot % cpx, rb = slot / cpx;
  if (cb >= ncb || rb >= nrb) return;
  const size_t c0 = (size_t)rb * BM;
  const size_t col0 = (size_t)cb * BN;
  v16i acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[a][b][e] = 0;
  // staging assignment: NU x 16 B of A and NU x 16 B of B per thread per K step (K step 128: half the barriers of 64; every
  // shipped K = Deff * lk is a multiple of 128 or falls back to the remainder handling below)
  const int srow = tid / SEG, sseg = (tid % SEG) * 16;
  const uint8_t* a_src[NU];
  const int8_t* b_src[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) {
    size_t c = c0 + srow + RPP * u;
    if (c >= count) c = count - 1;
    a_src[u] = digits + c * (size_t)R + sseg;
    b_src[u] = kskT + (col0 + srow + RPP * u) * (size_t)ldk + sseg;
  }
  const int fr = lane & 31, fh = (lane >> 5) * 16;
  // software pipeline: the global loads of K-step k0 + BK are issued before the matrix instructions of step k0 and land under
  // them (round 1 loaded, waited, stored, computed: every K-step exposed a global round trip)
  // R is a multiple of 64 (checked by the host); a trailing half step reads zeros for its upper 64 bytes
  const v4i zero4 = {0, 0, 0, 0};
  auto fetch = [&](int k0, v4i* ga, v4i* gb) {
    const bool live = k0 + sseg < R;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      ga[u] = live ? *reinterpret_cast<const v4i*>(a_src[u] + k0) : zero4;
      gb[u] = live ? *reinterpret_cast<const v4i*>(b_src[u] + k0) : zero4;
    }
  };
  v4i ga[NU], gb[NU];
  fetch(0, ga, gb);
  for (int k0 = 0; k0 < R; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NU; u++) {
      *reinterpret_cast<v4i*>(&As[(srow + RPP * u) * LD + sseg]) = ga[u];
      *reinterpret_cast<v4i*>(&Bs[(srow + RPP * u) * LD + sseg]) = gb[u];
    }
    __syncthreads();
    if (k0 + BK < R) fetch(k0 + BK, ga, gb);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 32) {
      v4i fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; a++) fa[a] = *reinterpret_cast<const v4i*>(&As[(wm * 64 + a * 32 + fr) * LD + kk + fh]);
#pragma unroll
      for (int b = 0; b < 2; b++) fb[b] = *reinterpret_cast<const v4i*>(&Bs[(wn * 64 + b * 32 + fr) * LD + kk + fh]);
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  }
  // epilogue: C tile layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); limb k' = col % LIMBS stands for 2^(8 (8 - LIMBS + k'))
  static_assert(LIMBS == 2 || LIMBS == 4 || LIMBS == 8, "limb groups are folded with lane shuffles");
  const int limb = lane & (LIMBS - 1);
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const size_t word = (col0 + wn * 64 + b * 32 + (lane & 31)) / LIMBS;     // output word j of this lane's column
#pragma unroll
      for (int e = 0; e < 16; e++) {
        uint64_t v = (uint64_t)(int64_t)acc[a][b][e] << (8 * (8 - LIMBS + limb));
        v += __shfl_xor(v, 1);
        if constexpr (LIMBS >= 4) v += __shfl_xor(v, 2);
        if constexpr (LIMBS >= 8) v += __shfl_xor(v, 4);
        const size_t c = c0 + wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (limb == 0 && c < count && word <= (size_t)n) {
          const uint64_t fix = ((uint64_t)1 << (betak - 1)) * colsum[word] + (word == (size_t)n ? bodies[c] : 0);
          out[c * (size_t)(n + 1) + word] = fix - v;
        }
      }
    }
}

__global__ void k_ksk_colsum(const uint64_t* __restrict__ ksk, int R, int n, uint64_t* __restrict__ colsum) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > n) return;
  uint64_t s = 0;
  for (int r = 0; r < R; r++) s += ksk[(size_t)r * (n + 1) + j];
  colsum[j] = s;
}

// Centred mod switch (semantics: oracle/tfhe_ref.c ref_ms_center).  The bootstrap rounds every word of the small ciphertext to 2N
// levels; the evaluator knows each mask word's rounding remainder and that a key bit is 1 half of the time, so half the sum of the
// remainders comes off the body first: the mod-switch error keeps sum_i (s_i - 1/2) e_i, half the variance.  One wave per ciphertext, in
// place on the key switch's output, ~6.5 KB read per ciphertext.
__global__ void __launch_bounds__(256) k_ms_center(uint64_t* __restrict__ small, size_t count, int n, int logN) {
  const size_t c = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= count) return;
  uint64_t* ct = small + c * (size_t)(n + 1);
  const int sh = 63 - logN;
  int64_t part = 0;
  for (int i = lane; i < n; i += 64) {
    const uint64_t a = ct[i];
    const uint64_t at = ((a >> (sh - 1)) + 1) >> 1;
    part += (int64_t)(a - (at << sh));
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  if (lane == 0) ct[n] -= (uint64_t)(part >> 1);
}

// ------------------------------------------------------------------------------------------ K4-K6 bootstrap
struct pbs_launch {
  const uint64_t* cts_small;  // count x (n+1)
  size_t count;
  int n, beta;
  const cplx* bsk;
  const cplx* tw;             // twiddle table in global memory
  const cplx* wtab;           // two-bit kernels: e^{i pi m / N}, m < 2N
  const int64_t* tables;      // [ntab][2^w]
  int w;
  const int32_t* table_idx;   // optional explicit index per ciphertext
  int hw, nchan;              // else table = ((e_offset + e) / hw) % nchan  (nchan == 1: single table)
  size_t e_offset;
  uint64_t* out;              // count x (D_out+1)
  int D_out;
  int accumulate;
  uint64_t body_add;
  uint64_t* dummy;            // D_out+1 words: sink for the padded groups of the last workgroup
  int bsk_wrap;               // cache experiments only (0 in the library)
  int pf_parts;               // 0: no L2 warm-up; else each workgroup touches 1/pf_parts of the next key rows
};

// KLDS > 0: the GROUPS waves of a workgroup share each key tile through a ring of KLDS tiles in LDS (pbs_core.h, pbs_thread)
template <int LOGN, int K, int L, int P, int MB, int KLDS>
constexpr size_t pbs_lds_bytes(int groups) {
  using G = pbs_geom<LOGN, K, L, P, MB>;
  return (size_t)G::TW_BYTES + (size_t)groups * G::GROUP_BYTES + (size_t)KLDS * 3 * (K + 1) * G::T * 16;
}

template <int LOGN, int K, int L, int P, int GROUPS, int MB = 0, int KLDS = 0>
__global__ void __launch_bounds__((pbs_geom<LOGN, K, L, P, MB>::T * GROUPS), ((P >= 16 || (pbs_geom<LOGN, K, L, P, MB>::T * GROUPS) >= 512) ? 1 : 2))
pbs_kernel(pbs_launch a) {
  using G = pbs_geom<LOGN, K, L, P, MB>;
  constexpr int T = G::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  cplx* tw = reinterpret_cast<cplx*>(smem_raw);
  unsigned char* per_group = smem_raw + G::TW_BYTES;
  for (int x = threadIdx.x; x < G::TW_LDS_ELEMS; x += blockDim.x) tw[x] = a.tw[x];
  if constexpr (MB) {     // root tables of the two-bit rotation: lo[j] = zeta^j, hi[j] = zeta^(j << ZLO)
    cplx* zl = tw + G::TW_LDS_ELEMS;
    for (int x = threadIdx.x; x < G::ZLUT_ELEMS; x += blockDim.x) zl[x] = x < (1 << G::ZLO) ? a.wtab[x] : a.wtab[(size_t)(x - (1 << G::ZLO)) << G::ZLO];
  }
  __syncthreads();
#if defined(DCTFHE_WAVE_STAGGER)   // experiment: the free-running waves of a workgroup (one ciphertext each) start DCTFHE_WAVE_STAGGER x ~4 100 clocks apart
  if constexpr (T <= 64) { const int w0 = (threadIdx.x >> 6) + ((blockIdx.x >> 8) & 1) * GROUPS; for (int z = 0; z < w0 * DCTFHE_WAVE_STAGGER; z++) __builtin_amdgcn_s_sleep(64); }
#endif
#if defined(DCTFHE_STAGGER)   // experiment: desynchronise co-resident workgroups by half a transform
  if ((blockIdx.x >> 8) & 1) for (int z = 0; z < DCTFHE_STAGGER; z++) __builtin_amdgcn_s_sleep(64);
#endif
  const int g = threadIdx.x / T, t = threadIdx.x % T;
  size_t e = (size_t)blockIdx.x * GROUPS + g;
  const bool live = e < a.count;
  if (!live) e = a.count - 1;  // padded groups redo the last ciphertext and drop the result
  unsigned char* mine = per_group + (size_t)g * G::GROUP_BYTES;
  cplx* exch = reinterpret_cast<cplx*>(mine);
  uint64_t* stage = reinterpret_cast<uint64_t*>(mine + G::STAGE_OFFSET);     // aliases the exchange buffer when G::ALIAS
  uint64_t* accl = reinterpret_cast<uint64_t*>(mine + G::SHARED_BYTES);
  uint32_t* pf_dump = nullptr;     // host emulation only
  pbs_args A;
  A.ct_small = a.cts_small + e * (size_t)(a.n + 1);
  A.n = a.n; A.beta = a.beta; A.bsk = a.bsk;
  const size_t ti = a.table_idx ? (size_t)a.table_idx[e] : (a.nchan > 1 ? ((a.e_offset + e) / (size_t)a.hw) % (size_t)a.nchan : 0);
  A.table = a.tables + (ti << a.w);
  A.w = a.w;
  // padded groups of the last workgroup redo the last ciphertext (same barrier sequence) into a sink
  A.out = live ? a.out + e * (size_t)(a.D_out + 1) : a.dummy;
  A.D_out = a.D_out;
  A.accumulate = live ? a.accumulate : 0;
  A.body_add = a.body_add;
  A.bsk_wrap = a.bsk_wrap;
  A.pf_parts = a.pf_parts;
  A.wtab = a.wtab;
  A.zlut = MB ? tw + G::TW_LDS_ELEMS : nullptr;
  if constexpr (G::TWIST_LDS) A.twist = tw + G::F::TW_TOTAL; else A.twist = a.tw + G::F::TW_TOTAL;
  A.pf_rank = (int)((blockIdx.x / 8) % (unsigned)(a.pf_parts > 0 ? a.pf_parts : 1));   // blocks b and b+8 share an XCD (round-robin dispatch; speed only)
  A.kring = reinterpret_cast<const cplx*>(per_group + (size_t)GROUPS * G::GROUP_BYTES);
  A.kring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(per_group + (size_t)GROUPS * G::GROUP_BYTES);
  A.kwave = g;
#if defined(DCTFHE_ABLATE_BARRIER)   // timing experiments only (tools/exp_pbs.hip): no workgroup barriers, wrong results
  if constexpr (true) {
#else
  if constexpr (T <= 64) {
#endif
    // one ciphertext per wave (or less): every exchange and the rotation stage stay inside the wave, whose LDS
    // queue is in order -- no workgroup barrier anywhere in the loop, the waves of a workgroup run decoupled
    pbs_thread<LOGN, K, L, P, MB, KLDS, GROUPS>(A, t, tw, stage, exch, accl, pf_dump, [] { __builtin_amdgcn_wave_barrier(); }, [] { __builtin_amdgcn_wave_barrier(); });
  } else {
    static_assert(KLDS == 0 || T <= 64, "key tiles through LDS: one wave per ciphertext");
    pbs_thread<LOGN, K, L, P, MB>(A, t, tw, stage, exch, accl, pf_dump, [] { __syncthreads(); }, [] { __builtin_amdgcn_wave_barrier(); });
  }
}

// ------------------------------------------------------------------------------------------ K1 conv2d
// out[b][co][y][x][word] = sum_{ci,ky,kx} w[co][ci][ky][kx] * in[b][ci][y*s+ky-p][x*s+kx-p][word]  (mod 2^64)
// Thread = one ciphertext word of one output pixel for a tile of COT output channels; lanes run over
// words, so every load is a coalesced stream of one input ciphertext.
template <int COT>
__global__ void __launch_bounds__(256)
k_conv2d(const uint64_t* __restrict__ in, int Cin, int H, int W, size_t Lin, size_t Deff, const int8_t* __restrict__ wgt, int Cout,
         int KH, int KW, int stride, int pad, int Ho, int Wo, size_t Lout, uint64_t* __restrict__ out) {
  // tensors are stored at their effective dimension: a row is Deff mask words (everything beyond is known to be zero and is
  // not stored) then the body; thread Deff of a pixel handles the body word
  const size_t word = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int pix = blockIdx.y;           // b*Ho*Wo + y*Wo + x
  const int co0 = blockIdx.z * COT;
  const int b = pix / (Ho * Wo), y = (pix / Wo) % Ho, x = pix % Wo;
  if (word >= Lout) return;
  const bool body = word == Lout - 1;
  uint64_t acc[COT];
#pragma unroll
  for (int c = 0; c < COT; c++) acc[c] = 0;
  if (body || word < Deff) {      // (words in [Deff, Lout-1) only exist when the output row is wider than its input: zeros)
    const size_t iw = body ? Lin - 1 : word;
    const uint64_t* inb = in + (size_t)b * Cin * H * W * Lin;
    for (int ci = 0; ci < Cin; ci++)
      for (int ky = 0; ky < KH; ky++) {
        const int iy = y * stride + ky - pad;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < KW; kx++) {
          const int ix = x * stride + kx - pad;
          if (ix < 0 || ix >= W) continue;
          const uint64_t v = inb[(((size_t)ci * H + iy) * W + ix) * Lin + iw];
#pragma unroll
          for (int c = 0; c < COT; c++) {
            const int co = co0 + c;
            const int8_t wv = (co < Cout) ? wgt[(((size_t)co * Cin + ci) * KH + ky) * KW + kx] : (int8_t)0;  // uniform
            acc[c] += (uint64_t)(int64_t)wv * v;
          }
        }
      }
  }
#pragma unroll
  for (int c = 0; c < COT; c++) {
    const int co = co0 + c;
    if (co < Cout) out[((((size_t)b * Cout + co) * Ho + y) * Wo + x) * Lout + word] = acc[c];
  }
}

// ---- convolution on the matrix cores -------------------------------------------------------------------
// out[co][word] = sum_k w[co][k] * in_k[word] (mod 2^64), k = (ci, ky, kx), is a GEMM with a tiny-integer left operand (|w| <= 127)
// and a u64 right operand.  As in the key switch the u64 operand is re-expressed as 8 signed byte limbs (carries propagated),
// here on the fly while the input tile is staged: C[co][word*8 + limb] = sum_k w[co][k] * s_limb(in_k[word]) is an i8 x i8 -> i32
// GEMM for v_mfma_i32_32x32x32_i8 (|C| <= K * 127 * 128 < 2^31 for K <= 2^17) and out = sum_limb C_limb << 8 limb.
// Workgroup = one output pixel x 32 ciphertext words x 64 output channels; K in steps of 32 (ci, ky, kx) triples; wave w owns
// columns [64 w, 64 w + 64) = 8 words x 8 limbs.  Both operands K-contiguous in LDS (one ds_read_b128 per fragment).
struct conv_tap { int32_t offset; int16_t dy, dx; };     // per k = (ci, ky, kx): ci*H*W (pixels; < 0: padding row of K), ky - pad, kx - pad

__global__ void __launch_bounds__(256)
k_conv2d_mfma(const uint64_t* __restrict__ in, int H, int W, size_t Lin, size_t Deff, const int8_t* __restrict__ wpack /* [Cout_pad64][Kpad] */,
              const conv_tap* __restrict__ taps /* [Kpad], offset < 0: padding row */, int Kpad, int Cin_HW, int Cout, int stride, int Ho, int Wo,
              size_t Lout, uint64_t* __restrict__ out) {
  constexpr int BK = 32, LD = BK + 16;
  __shared__ __attribute__((aligned(16))) int8_t As[64 * LD];
  __shared__ __attribute__((aligned(16))) int8_t Bs[256 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pix = blockIdx.y;           // b*Ho*Wo + y*Wo + x
  const int b = pix / (Ho * Wo), y = (pix / Wo) % Ho, x = pix % Wo;
  const int co0 = blockIdx.z * 64;
  // staging: thread = (word w of 32, group kq of 8): 4 consecutive k for one word
  const int sw = tid & 31, kq = tid >> 5;
  const size_t wg = (size_t)blockIdx.x * 32 + sw;                    // word of the output row this thread stages
  const bool wbody = wg == Lout - 1, wlive = wbody || wg < Deff;      // (words in [Deff, Lout-1) are zero)
  const size_t iw = wbody ? Lin - 1 : wg;
  const uint64_t* inb = in + (size_t)b * Cin_HW * Lin;
  const int y0 = y * stride, x0 = x * stride;
  v16i acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[a][c][e] = 0;
  const int fr = lane & 31, fh = (lane >> 5) * 16;
  for (int k0 = 0; k0 < Kpad; k0 += BK) {
    // input words of this thread's 4 taps -> signed byte limbs, limb-major so that each limb row gets one 4-byte store
    uint32_t packed[8];
#pragma unroll
    for (int l = 0; l < 8; l++) packed[l] = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const conv_tap tp = taps[k0 + kq * 4 + u];
      const int iy = y0 + tp.dy, ix = x0 + tp.dx;
      uint64_t v = 0;
      if (wlive && tp.offset >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W)
        v = inb[((size_t)tp.offset + (size_t)(iy * W + ix)) * Lin + iw];
#pragma unroll
      for (int l = 0; l < 8; l++) {
        int byte = (int)(v & 0xFF);
        v >>= 8;
        if (byte >= 128) { byte -= 256; v += 1; }
        packed[l] |= (uint32_t)(byte & 0xFF) << (8 * u);
      }
    }
    v4i ga = {0, 0, 0, 0};
    if (tid < 128) ga = *reinterpret_cast<const v4i*>(wpack + (size_t)(co0 + (tid >> 1)) * Kpad + k0 + (tid & 1) * 16);
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 8; l++) *reinterpret_cast<uint32_t*>(&Bs[(sw * 8 + l) * LD + kq * 4]) = packed[l];
    if (tid < 128) *reinterpret_cast<v4i*>(&As[(tid >> 1) * LD + (tid & 1) * 16]) = ga;
    __syncthreads();
    v4i fa[2], fb[2];
#pragma unroll
    for (int a = 0; a < 2; a++) fa[a] = *reinterpret_cast<const v4i*>(&As[(a * 32 + fr) * LD + fh]);
#pragma unroll
    for (int c = 0; c < 2; c++) fb[c] = *reinterpret_cast<const v4i*>(&Bs[(wave * 64 + c * 32 + fr) * LD + fh]);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int c = 0; c < 2; c++) acc[a][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[c], acc[a][c], 0, 0, 0);
  }
  // epilogue: C tile layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); limb = col & 7, word = col >> 3
  const int limb = lane & 7;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const size_t word = (size_t)blockIdx.x * 32 + wave * 8 + c * 4 + ((lane & 31) >> 3);
#pragma unroll
      for (int e = 0; e < 16; e++) {
        uint64_t v = (uint64_t)(int64_t)acc[a][c][e] << (8 * limb);
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        const int co = co0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (limb == 0 && co < Cout && word < Lout) out[((((size_t)b * Cout + co) * Ho + y) * Wo + x) * Lout + word] = v;
      }
    }
}

// any non-zero mask word in [deff, D) of `count` ciphertexts?  (guards the effective-dimension shortcut at the session input)
__global__ void k_tail_nonzero(const uint64_t* __restrict__ cts, size_t count, int D /* mask words per row */, int deff, int* __restrict__ flag) {
  const size_t tail = (size_t)(D - deff), total = count * tail;
  int bad = 0;
  for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (size_t)gridDim.x * blockDim.x)
    bad |= cts[(x / tail) * (size_t)(D + 1) + deff + (x % tail)] != 0;
  if (bad) atomicOr(flag, 1);
}

// ------------------------------------------------------------------------------------------ K2 elementwise
// Rows are stored at their effective dimension (mask words the compiler knows to be zero are not stored): a row of stride L
// holds L-1 mask words and the body at L-1.  `take(row, L, d, w)` is word w of a row whose first d mask words may be non-zero.
__device__ __forceinline__ uint64_t row_word(const uint64_t* row, size_t L, size_t d, size_t w, bool body) {
  return body ? row[L - 1] : (w < d ? row[w] : 0);
}
// o = a + b; the operands may have different effective dimensions (the result has the larger)
__global__ void k_add(const uint64_t* __restrict__ a, size_t La, size_t da, const uint64_t* __restrict__ b, size_t Lb, size_t db,
                      uint64_t* __restrict__ o, size_t Lo, size_t count) {
  const size_t total = count * Lo;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t c = i / Lo, w = i % Lo;
    const bool body = w == Lo - 1;
    o[i] = row_word(a + c * La, La, da, w, body) + row_word(b + c * Lb, Lb, db, w, body);
  }
}
// o = (a << shift), body += body_add: the first nwords mask words and the body of every row (mask words only shifted); what
// lies between nwords and the end of an output row is left alone (the table bootstrap that ends the site rewrites the row)
__global__ void k_affine(const uint64_t* __restrict__ a, size_t La, size_t da, uint64_t* __restrict__ o, size_t Lo, size_t count, size_t nwords,
                         int shift, uint64_t body_add) {
  const size_t per = nwords + 1, total = count * per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t c = i / per, w = i % per;
    const bool body = w == nwords;
    uint64_t v = row_word(a + c * La, La, da, w, body) << shift;
    if (body) v += body_add;
    o[c * Lo + (body ? Lo - 1 : w)] = v;
  }
}
// rows of stride Ls -> rows of stride Ld: the first nwords mask words and the body; the other words of the output row are zero
// (session upload: host rows of D+1 words -> stored rows; download: the reverse)
__global__ void k_restride(const uint64_t* __restrict__ src, size_t Ls, uint64_t* __restrict__ dst, size_t Ld, size_t count, size_t nwords) {
  const size_t total = count * Ld;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t c = i / Ld, w = i % Ld;
    dst[i] = w == Ld - 1 ? src[c * Ls + Ls - 1] : (w < nwords ? src[c * Ls + w] : 0);
  }
}
// window-K sum pooling with floor semantics (reference nn.AvgPool2d(k): backbone.py:276; scale folded into the next table)
__global__ void k_sum_pool(const uint64_t* __restrict__ in, int C, int H, int W, size_t Lin, size_t din, int K, int Ho, int Wo,
                           uint64_t* __restrict__ out, size_t Lout, size_t total_words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_words; i += (size_t)gridDim.x * blockDim.x) {
    const size_t word = i % Lout;
    const bool body = word == Lout - 1;
    size_t e = i / Lout;
    const int x = (int)(e % Wo); e /= Wo;
    const int y = (int)(e % Ho); e /= Ho;  // e = b*C + c
    uint64_t s = 0;
    for (int ky = 0; ky < K; ky++)
      for (int kx = 0; kx < K; kx++) s += row_word(in + ((e * H + (size_t)(y * K + ky)) * W + (size_t)(x * K + kx)) * Lin, Lin, din, word, body);
    out[i] = s;
  }
}

// clear-mode table look-up on 1-word "ciphertexts" (D = 0): same arithmetic as round_lut without noise.
// sigma > 0 (`simulate` with the noise model): Gaussian noise of that standard deviation (fraction of the torus: what the
// compiler predicts at the input of this site's table bootstrap) is added where the encrypted run has it -- after the exact
// rounding steps have cleared the low bits, or, with approximate rounding, on the raw accumulator -- and the half-box
// rotation of the test vector does the rest.
__global__ void k_lut_clear(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, size_t count, int shift, uint64_t body_add,
                            int p, int r, int w, const int64_t* __restrict__ tables, int hw, int nchan, int* __restrict__ overflow,
                            double sigma, rng_key seed, uint64_t stream, int approx) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
    uint64_t v = (in[e] << shift) + body_add;
    if (r > 0 && !(approx && sigma > 0)) v += 1ULL << (63 - p + r - 1);
    if (v >> 63) atomicOr(overflow, 1);  // message left the padded range: an FHE run would wrap
    uint64_t idx = (v >> (63 - w)) & ((1ULL << w) - 1);
    if (sigma > 0) {
      const uint64_t half_box = 1ULL << (62 - w);
      // exact rounding: the value sits at the centre of its box; approximate: where its low bits put it (+ half an input unit)
      const uint64_t centre = approx && r > 0 ? v + (1ULL << (62 - p)) : (idx << (63 - w));
      const uint64_t noisy = centre + (uint64_t)gauss_torus(seed, stream, e, sigma);
      // negacyclic wrap: a value pushed across the padding bit comes back negated -- reproduce the bootstrap's behaviour
      const uint64_t pos = noisy + half_box;
      idx = (pos >> (63 - w)) & ((1ULL << w) - 1);
      const size_t ti = nchan > 1 ? (e / (size_t)hw) % (size_t)nchan : 0;
      const uint64_t t = (uint64_t)tables[(ti << w) + idx];
      out[e] = (pos >> 63) ? (uint64_t)0 - t : t;
      continue;
    }
    const size_t ti = nchan > 1 ? (e / (size_t)hw) % (size_t)nchan : 0;
    out[e] = (uint64_t)tables[(ti << w) + idx];
  }
}

// ------------------------------------------------------------------------------------------ K10 DCT front-end (client side, plaintext)
// uint8 planes -> float32 [B][channels][S][S]: (pixel - 128) blockwise orthonormal DCT-II (T B T^t per fs x fs block,
// reference data/cvfunctional.py:37-57), only the coefficients SubsetDCT keeps (cvtransforms.py:117-142), the two chroma
// coefficient grids -- DCT'd at half resolution -- bilinearly up-sampled to S x S (UpScaleDCT, cvtransforms.py:56-64;
// half-pixel centres, edge clamp), channels concatenated Y | slot 1 | slot 2 (Aggregate), then (x - mean) / std in f32
// (NormalizeDCT, cvtransforms.py:152-208; mean/std already gathered by the caller with the reference's index quirk).
// One thread per output value; f64 inside like the reference's numpy path.  A few kB per image: nothing to tune.
struct dct_args {
  const uint8_t* plane[3];      // Y [B][fs*S][fs*S]; chroma slots [B][fs*Sc][fs*Sc]
  int S, Sc, fs;                // output grid, chroma grid (S or S/2), block size (4 or 8)
  int n[3];                     // kept coefficients per plane
  const int32_t* idx[3];        // their row-major indices u*fs + v
  const float *mean, *stdv;     // per output channel
  int round_coeffs;             // JPEG-domain path (block size 8): coefficients are libjpeg's quantised integers (integer islow DCT, quality-100
                                // tables) and the up-sampled planes are rounded half to even, as the int16 arrays of the reference are
  float* out;                   // [B][n0+n1+n2][S][S]
  int batch;
};

// libjpeg's accurate integer forward DCT ("islow": Loeffler-Ligtenberg-Moschytz, 13-bit constants, 2 extra bits carried out of the row
// pass; jfdctint.c [K: restated from the published algorithm -- libjpeg-turbo is not in this environment, PARITY UNPINNED]) on one row
// or column of 8 values, in place.  TurboJPEG picks this method at quality >= 96 (reference data/cvfunctional.py:24: quality=100).
// first = 1: row pass (outputs scaled by 2^2 * sqrt(8)); first = 0: column pass (removes the 2^2; the 2-D result is 8x the DCT).
__host__ __device__ inline void jfdct_islow_1d(int32_t* d, int first) {
  constexpr int CB = 13, P1 = 2;
  constexpr int32_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                    F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  auto descale = [](int32_t x, int n) { return (x + (1 << (n - 1))) >> n; };
  int32_t t0 = d[0] + d[7], t7 = d[0] - d[7], t1 = d[1] + d[6], t6 = d[1] - d[6], t2 = d[2] + d[5], t5 = d[2] - d[5], t3 = d[3] + d[4], t4 = d[3] - d[4];
  const int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  const int sh = first ? CB - P1 : CB + P1;
  d[0] = first ? (t10 + t11) << P1 : descale(t10 + t11, P1);
  d[4] = first ? (t10 - t11) << P1 : descale(t10 - t11, P1);
  int32_t z1 = (t12 + t13) * F0_541;
  d[2] = descale(z1 + t13 * F0_765, sh);
  d[6] = descale(z1 - t12 * F1_847, sh);
  z1 = t4 + t7;
  int32_t z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int32_t z5 = (z3 + z4) * F1_175;
  t4 *= F0_298; t5 *= F2_053; t6 *= F3_072; t7 *= F1_501;
  z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
  z3 += z5; z4 += z5;
  d[7] = descale(t4 + z1 + z3, sh);
  d[5] = descale(t5 + z2 + z4, sh);
  d[3] = descale(t6 + z2 + z3, sh);
  d[1] = descale(t7 + z1 + z4, sh);
}
// quantised coefficient (u, v) of 8x8 block (by, bx): level shift, islow DCT, division by 8 * q with q = 1 (the quality-100 tables are
// all ones), rounded half away from zero as libjpeg's quantiser does (jcdctmgr.c)
__device__ inline int32_t jpeg_islow_coeff(const uint8_t* plane, int side, int by, int bx, int u, int v) {
  int32_t col[8];
  for (int i = 0; i < 8; i++) {
    int32_t row[8];
    for (int j = 0; j < 8; j++) row[j] = (int32_t)plane[(size_t)(by * 8 + i) * side + bx * 8 + j] - 128;
    jfdct_islow_1d(row, 1);
    col[i] = row[v];
  }
  jfdct_islow_1d(col, 0);
  const int32_t c = col[u];
  return c < 0 ? -((-c + 4) >> 3) : (c + 4) >> 3;
}

__device__ __forceinline__ double dct_basis(int fs, int u, int j) {
  return u == 0 ? rsqrt((double)fs) : sqrt(2.0 / fs) * cospi((double)((2 * j + 1) * u) / (double)(2 * fs));
}
__device__ inline double dct_coeff(const uint8_t* plane, int side, int fs, int by, int bx, int u, int v, int round_coeffs) {
  if (round_coeffs && fs == 8) return (double)jpeg_islow_coeff(plane, side, by, bx, u, v);     // the JPEG-domain path: integers throughout
  double acc = 0.0;
  for (int i = 0; i < fs; i++) {
    double row = 0.0;
    for (int j = 0; j < fs; j++) row += ((double)plane[(size_t)(by * fs + i) * side + bx * fs + j] - 128.0) * dct_basis(fs, v, j);
    acc += dct_basis(fs, u, i) * row;
  }
  if (round_coeffs) acc = copysign(floor(fabs(acc) + 0.5), acc);
  return acc;
}
__global__ void k_dct_frontend(dct_args a) {
  const int C = a.n[0] + a.n[1] + a.n[2];
  const size_t total = (size_t)a.batch * C * a.S * a.S;
  for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(x % a.S), oy = (int)((x / a.S) % a.S), c = (int)((x / ((size_t)a.S * a.S)) % C), b = (int)(x / ((size_t)a.S * a.S * C));
    const int pl = c < a.n[0] ? 0 : (c < a.n[0] + a.n[1] ? 1 : 2);
    const int k = a.idx[pl][c - (pl == 0 ? 0 : (pl == 1 ? a.n[0] : a.n[0] + a.n[1]))];
    const int u = k / a.fs, v = k % a.fs;
    const int grid = pl == 0 ? a.S : a.Sc, side = grid * a.fs;
    const uint8_t* plane = a.plane[pl] + (size_t)b * side * side;
    double val;
    if (grid == a.S) {
      val = dct_coeff(plane, side, a.fs, oy, ox, u, v, a.round_coeffs);
    } else {   // bilinear up-sampling of the coefficient grid
      const double sy = (oy + 0.5) * ((double)grid / a.S) - 0.5, sx = (ox + 0.5) * ((double)grid / a.S) - 0.5;
      const int y0 = (int)floor(sy), x0 = (int)floor(sx);
      const double fy = sy - y0, fx = sx - x0;
      const int y0c = min(max(y0, 0), grid - 1), y1c = min(max(y0 + 1, 0), grid - 1), x0c = min(max(x0, 0), grid - 1), x1c = min(max(x0 + 1, 0), grid - 1);
      const double top = dct_coeff(plane, side, a.fs, y0c, x0c, u, v, a.round_coeffs) * (1.0 - fx) + dct_coeff(plane, side, a.fs, y0c, x1c, u, v, a.round_coeffs) * fx;
      const double bot = dct_coeff(plane, side, a.fs, y1c, x0c, u, v, a.round_coeffs) * (1.0 - fx) + dct_coeff(plane, side, a.fs, y1c, x1c, u, v, a.round_coeffs) * fx;
      val = top * (1.0 - fy) + bot * fy;
      if (a.round_coeffs) val = rint(val);
    }
    a.out[x] = ((float)val - a.mean[c]) / a.stdv[c];
  }
}

// ------------------------------------------------------------------------------------------ f64 peak probe
__global__ void k_fp64_peak(double* out, int iters) {
  double a0 = threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-7;
  for (int i = 0; i < iters; i++) {
    a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
    a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

}  // namespace dctfhe
