// dctfhe.hip -- host side of libdctfhe.so: contexts, key generation, the layer scheduler that walks a
// compiled circuit and issues batched HIP launches, and the C ABI of include/dctfhe.h.
// gfx950 only.  No CPU fallback anywhere: every entry point needs a live HIP device.
#include <hip/hip_runtime.h>
#include <sys/random.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/dctfhe.h"
#include "kernels.h"

using namespace dctfhe;

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return -1;
}
#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) return fail("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
  } while (0)
#define CHK(x)              \
  do {                      \
    int r_ = (x);           \
    if (r_ != 0) return r_; \
  } while (0)

extern "C" const char* dctfhe_last_error(void) { return g_err.c_str(); }
extern "C" int dctfhe_version(void) { return 1; }

// ------------------------------------------------------------------------------------------ structs
struct dctfhe_ctx {
  int device = 0;
  hipStream_t stream = nullptr, own = nullptr;
  hipDeviceProp_t prop;
};

// device buffer that frees itself unless released: every early return of an entry point cleans up after itself
struct DevBuf {
  void* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct TierKeys {
  dctfhe_tier t{};
  uint64_t* d_ksk = nullptr;     // [D][lk][n+1]
  uint64_t* d_colsum = nullptr;  // [n+1], over all D*lk rows
  std::map<int, uint64_t*>* colsum_eff = nullptr;   // column sums over the first Deff*lk rows (shared with the key's owner)
  int8_t* d_kskT = nullptr;      // signed byte limbs, [limbs(n+1) padded to 128][D*lk], for the MFMA key switch
  int ncol_pad = 0;
  int limbs = 8;                 // byte limbs kept per key word (ks_limbs): the key lives on the 2^-(8 limbs) torus grid
  bool own_ksk = false;
  cplx* d_bsk = nullptr;         // [n][rows][k+1][P][T]; unroll 2: [3n/2 blocks] for the pair secret
  cplx* d_wtab = nullptr;        // unroll 2: e^{i pi m/N}, m < 2N, then e^{2 pi i k/8}, k < 8
  cplx* d_tw = nullptr;
};

// CLIENT side: everything secret.  The whole handle is a function of (params, 32-byte seed): persist the seed to persist it.
struct dctfhe_client_key {
  dctfhe_ctx* ctx = nullptr;
  dctfhe_params p{};
  uint8_t seed[32] = {};
  rng_key sec{}, pub{};          // ChaCha20 keys: secret (key bits, noise) and public (masks)
  // ENCRYPTION randomness is per HANDLE, not per seed: its two generator keys are derived from the seed AND a 128-bit nonce drawn from
  // the OS when the handle is created.  (Key material stays a pure function of the seed.  Before, a client that re-created its key from
  // the persisted seed -- or a second process holding the seed -- re-drew the very masks and noise of earlier ciphertexts: ct1 - ct2 =
  // (0, m1 - m2).  ADVICE r2.)
  uint8_t nonce[16] = {};
  rng_key enc_sec{}, enc_pub{};
  uint8_t *d_S = nullptr, *d_s = nullptr;
  uint8_t* d_spair[DCTFHE_MAX_TIERS] = {};   // unroll 2: derived secret (s1(1-s2), (1-s1)s2, s1 s2) per pair
  uint64_t enc_calls = 0;        // every dctfhe_encrypt call draws from fresh streams
  ~dctfhe_client_key() {
    if (ctx) hipSetDevice(ctx->device);
    hipFree(d_S); hipFree(d_s);
    for (auto q : d_spair) hipFree(q);
  }
};

// SERVER side: evaluation keys only (key-switch keys, Fourier bootstrap keys); nothing here depends on the secret at run time.
struct dctfhe_eval_keys {
  dctfhe_ctx* ctx = nullptr;
  dctfhe_params p{};
  TierKeys tiers[DCTFHE_MAX_TIERS];
  uint64_t* d_dummy = nullptr;   // D+1 words, sink of padded bootstrap groups
  ~dctfhe_eval_keys() {
    if (ctx) hipSetDevice(ctx->device);
    hipFree(d_dummy);
    for (int i = 0; i < DCTFHE_MAX_TIERS; i++) {
      TierKeys& tk = tiers[i];
      if (tk.own_ksk) {
        hipFree(tk.d_ksk); hipFree(tk.d_colsum); hipFree(tk.d_kskT);
        if (tk.colsum_eff) { for (auto& kv : *tk.colsum_eff) hipFree(kv.second); delete tk.colsum_eff; }
      }
      hipFree(tk.d_bsk); hipFree(tk.d_tw); hipFree(tk.d_wtab);
    }
  }
};
using dctfhe_keys = dctfhe_eval_keys;   // the server-side code below says K for the evaluation keys

enum { OP_CONV = 1, OP_ADD = 2, OP_SUMPOOL = 3, OP_LUT = 4 };
struct Op {
  int32_t type, src0, src1, dst;
  int32_t ip[12];
  int64_t lp[2];
  int64_t payload_off, payload_len;
};
struct TensorShape { int32_t C, H, W, pad; };

// a convolution's weights as the i8 MFMA kernel wants them: [Cout padded to 64][K padded to 32] K-contiguous, and per k the
// input plane offset and the tap displacement (kernels.h k_conv2d_mfma)
struct ConvPack {
  int8_t* d_w = nullptr;
  conv_tap* d_taps = nullptr;
  int kpad = 0;
};
static void conv_pack_sizes(int Cout, int Cin, int KH, int KW, size_t* wbytes, size_t* tbytes, int* kpad) {
  const int K = Cin * KH * KW;
  *kpad = (K + 31) / 32 * 32;
  *wbytes = ((size_t)((Cout + 63) / 64 * 64) * *kpad + 255) / 256 * 256;
  *tbytes = ((size_t)*kpad * sizeof(conv_tap) + 255) / 256 * 256;
}
// fills host images of the packed weights / tap table (sizes from conv_pack_sizes)
static void pack_conv_host(const int8_t* w, int Cout, int Cin, int H, int W, int KH, int KW, int pad, int kpad, int8_t* wp, conv_tap* tp) {
  const int K = Cin * KH * KW;
  for (int co = 0; co < Cout; co++) memcpy(wp + (size_t)co * kpad, w + (size_t)co * K, (size_t)K);
  for (int k = 0; k < kpad; k++) {
    if (k >= K) { tp[k] = conv_tap{-1, 0, 0}; continue; }
    const int ci = k / (KH * KW), r = k % (KH * KW);
    tp[k] = conv_tap{ci * H * W, (int16_t)(r / KW - pad), (int16_t)(r % KW - pad)};
  }
}
// every pack of a circuit (or the single one of the dctfhe_conv2d primitive) lives in ONE device allocation
struct ConvSlab {
  void* d = nullptr;
  ~ConvSlab() { hipFree(d); }
};
static int build_conv_packs(const std::vector<const int8_t*>& w, const std::vector<std::array<int, 7>>& g /* Cout,Cin,H,W,KH,KW,pad */, ConvSlab* slab,
                            std::vector<ConvPack>* out) {
  size_t total = 0;
  std::vector<size_t> off_w(w.size()), off_t(w.size());
  out->assign(w.size(), ConvPack{});
  for (size_t i = 0; i < w.size(); i++) {
    if (!w[i]) continue;
    size_t wb, tb;
    conv_pack_sizes(g[i][0], g[i][1], g[i][4], g[i][5], &wb, &tb, &(*out)[i].kpad);
    off_w[i] = total; total += wb;
    off_t[i] = total; total += tb;
  }
  if (!total) return 0;
  std::vector<char> host(total, 0);
  for (size_t i = 0; i < w.size(); i++)
    if (w[i]) pack_conv_host(w[i], g[i][0], g[i][1], g[i][2], g[i][3], g[i][4], g[i][5], g[i][6], (*out)[i].kpad, (int8_t*)&host[off_w[i]], (conv_tap*)&host[off_t[i]]);
  HIPCHK(hipMalloc(&slab->d, total));
  HIPCHK(hipMemcpy(slab->d, host.data(), total, hipMemcpyHostToDevice));
  for (size_t i = 0; i < w.size(); i++)
    if (w[i]) { (*out)[i].d_w = (int8_t*)slab->d + off_w[i]; (*out)[i].d_taps = (conv_tap*)((char*)slab->d + off_t[i]); }
  return 0;
}

struct dctfhe_circuit {
  dctfhe_ctx* ctx = nullptr;
  std::vector<TensorShape> tensors;
  std::vector<Op> ops;
  std::vector<void*> d_payload;  // per op, device copy of its payload (weights / tables)
  std::vector<ConvPack> conv;    // per op: the matrix-core form of a convolution's weights (empty for other ops)
  ConvSlab conv_slab;            // ... all of them in one allocation, made when the first encrypted session is created
  std::vector<std::vector<int8_t>> conv_w;          // host copies of the convolution weights until then
  std::vector<std::array<int, 7>> conv_geom;
  int input_tensor = 0, output_tensor = 0, max_bit_width = 0;
  ~dctfhe_circuit() {
    if (ctx) hipSetDevice(ctx->device);
    for (void* p : d_payload) if (p) hipFree(p);
  }
};

struct dctfhe_session {
  dctfhe_ctx* ctx = nullptr;
  dctfhe_circuit* circ = nullptr;
  dctfhe_keys* keys = nullptr;  // nullptr: clear mode (D = 0)
  int batch = 0;
  int D = 0;
  std::vector<uint64_t*> d_tensor;
  std::vector<std::pair<size_t, uint64_t*>> owned;
  std::vector<size_t> tensor_words;
  // encrypted tensors are stored at their effective dimension: row stride t_L[t] words = t_L[t]-1 mask words + the body;
  // t_deff[t] <= t_L[t]-1 is how many leading mask words may be non-zero.  Clear mode: one word per element.
  std::vector<size_t> t_L, t_deff;
  // scratch for LUT sites
  size_t chunk = 0;
  uint8_t* d_digits = nullptr;
  uint64_t* d_bodies = nullptr;
  uint64_t* d_small = nullptr;
  int64_t* d_bit_tables = nullptr;  // 64 single-entry tables: 2^j
  int* d_overflow = nullptr;
  // `simulate`: per-op noise (fraction of the torus) injected by the clear look-up kernel; empty = noise-free
  std::vector<double> sim_sigma;
  uint64_t sim_seed = 0, sim_run = 0;
  // timing: events are created once and reused by every run; bootstraps per tier and image are fixed by the circuit
  std::vector<hipEvent_t> ev_pool;
  int64_t pbs_per_image[DCTFHE_MAX_TIERS] = {};
  ~dctfhe_session() {
    if (ctx) hipSetDevice(ctx->device);
    for (auto& o : owned) hipFree(o.second);
    for (hipEvent_t e : ev_pool) hipEventDestroy(e);
    hipFree(d_digits); hipFree(d_bodies); hipFree(d_small); hipFree(d_bit_tables); hipFree(d_overflow);
  }
};

// ------------------------------------------------------------------------------------------ kernel dispatch
// (logN, k, l, points per thread).  8 points per thread keep the register arrays under 256 VGPRs, i.e. 2 waves
// per SIMD (measured 1.6x over 16 points per thread at 1 wave per SIMD); GROUPS packs ciphertexts up to 256
// threads per workgroup, so that two workgroups share a CU (N = 8192 needs all 512 threads for one ciphertext).
#define PBS_CASES(X)                                                                                 \
  X(8, 2, 2, 8) X(9, 1, 2, 16) X(9, 1, 3, 8) X(10, 1, 1, 8) X(10, 1, 2, 8) X(10, 2, 1, 8) X(10, 2, 2, 8)  \
  X(11, 1, 1, 8) X(11, 1, 2, 8) X(11, 1, 3, 8) X(12, 1, 1, 8) X(12, 1, 2, 8) X(12, 1, 3, 8)           \
  X(13, 1, 1, 8) X(13, 1, 2, 8) X(13, 1, 3, 8)

// two-bit blind rotation (tier.unroll == 2): k = 1, one level
#define PBS_MB_CASES(X) X(11) X(12) X(13)

template <int LOGN, int K_, int L_, int P>
constexpr int groups_for() {
  using G = pbs_geom<LOGN, K_, L_, P>;
  constexpr int by_lds = (160 * 1024 - G::TW_BYTES) / G::GROUP_BYTES;
  int g = G::T >= 256 ? 1 : 256 / G::T;   // two 256-thread workgroups per CU beat one of 512 (+10..20%, profiles/r01_exp_wg.log)
  while (g > 1 && g > by_lds) g >>= 1;
  // ... and when two such workgroups do not fit the LDS together (both mask polynomials of a k = 2, l = 2 tier live there),
  // smaller workgroups keep the CU at two waves per SIMD
  while (g > 1 && 2 * (G::TW_BYTES + g * G::GROUP_BYTES) > 160 * 1024) g >>= 1;
  return g;
}

static int tier_ppt(const dctfhe_tier& t) {
#define X(LN, K_, L_, P_) if (t.logN == LN && t.k == K_ && t.l == L_) return P_;
  PBS_CASES(X)
#undef X
  return 0;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the function on the CURRENT device: once per (call site, device), not once
// per process (a process with contexts on two GPUs would have failed on the second at every launch above 64 KB of LDS)
#define SET_LDS_ATTR(kernel, bytes)                                                                  \
  do {                                                                                               \
    static uint64_t done_mask = 0;                                                                   \
    int dev_ = 0;                                                                                    \
    HIPCHK(hipGetDevice(&dev_));                                                                     \
    if (dev_ < 0 || dev_ >= 64 || !((done_mask >> dev_) & 1)) {                                      \
      HIPCHK(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      if (dev_ >= 0 && dev_ < 64) done_mask |= 1ull << dev_;                                         \
    }                                                                                                \
  } while (0)

static int launch_pbs(const dctfhe_tier& t, const pbs_launch& a, hipStream_t st) {
  if (a.count == 0) return 0;
  // the kernel's L2 warm-up contract (pbs_core.h): callers pad the key by PBS_PF_DIST iterations (eval_alloc does)
  if (a.pf_parts != 0 && a.pf_parts < 8) return fail("bootstrap launch: pf_parts must be 0 or >= 8 (got %d)", a.pf_parts);
  if (a.bsk_wrap < 0 || (a.bsk_wrap > 0 && t.unroll == 2)) return fail("bootstrap launch: bsk_wrap is a one-bit-kernel experiment switch");
  if (!a.bsk || !a.tw || !a.cts_small || !a.out || !a.dummy || (t.unroll == 2 && !a.wtab)) return fail("bootstrap launch: null operand");
  if (a.w < 0 || a.w > t.logN - 1) return fail("bootstrap launch: table of 2^%d entries does not fit N = 2^%d", a.w, t.logN);
#define X(LN, K_, L_, P_)                                                                            \
  if (t.logN == LN && t.k == K_ && t.l == L_ && t.unroll == 1) {                                     \
    using G = pbs_geom<LN, K_, L_, P_>;                                                              \
    constexpr int GR = groups_for<LN, K_, L_, P_>();                                                 \
    const size_t lds = G::TW_BYTES + (size_t)GR * G::GROUP_BYTES;                                 \
    SET_LDS_ATTR((pbs_kernel<LN, K_, L_, P_, GR>), lds);                                             \
    const unsigned grid = (unsigned)((a.count + GR - 1) / GR);                                       \
    hipLaunchKernelGGL((pbs_kernel<LN, K_, L_, P_, GR>), dim3(grid), dim3(G::T * GR), lds, st, a);   \
    HIPCHK(hipGetLastError());                                                                       \
    return 0;                                                                                        \
  }
  PBS_CASES(X)
#undef X
#define X(LN)                                                                                        \
  if (t.logN == LN && t.k == 1 && t.l == 1 && t.unroll == 2) {                                       \
    using G = pbs_geom<LN, 1, 1, 8, 1>;                                                              \
    /* N = 4096: two ciphertexts per 512-thread workgroup share their key lines (28.5 vs 30.8 ms) */  \
    constexpr int GR = LN == 12 ? 2 : groups_for<LN, 1, 1, 8>();                                     \
    const size_t lds = G::TW_BYTES + (size_t)GR * G::GROUP_BYTES;                                    \
    SET_LDS_ATTR((pbs_kernel<LN, 1, 1, 8, GR, 1>), lds);                                             \
    const unsigned grid = (unsigned)((a.count + GR - 1) / GR);                                       \
    hipLaunchKernelGGL((pbs_kernel<LN, 1, 1, 8, GR, 1>), dim3(grid), dim3(G::T * GR), lds, st, a);   \
    HIPCHK(hipGetLastError());                                                                       \
    return 0;                                                                                        \
  }
  PBS_MB_CASES(X)
#undef X
  if (t.logN == 11 && t.k == 1 && t.l == 3 && t.unroll == 2) {
    // the general form of the two-bit rotation (one polynomial at a time, monomial factors rebuilt per gadget row), four
    // ciphertexts per 512-thread workgroup so that they share their key lines: 74.2 ms per 4096 against 82.3 for the one-bit
    // chain of the same tier (profiles/r02_exp_ablations.log); its output is 0.7 bit noisier, so the compiler only uses it
    // where the circuit's budget allows (dctfhe/params.py T4r2)
    using G = pbs_geom<11, 1, 3, 8, 1>;
    constexpr int GR = 4;
    const size_t lds = G::TW_BYTES + (size_t)GR * G::GROUP_BYTES;
    SET_LDS_ATTR((pbs_kernel<11, 1, 3, 8, GR, 1>), lds);
    const unsigned grid = (unsigned)((a.count + GR - 1) / GR);
    hipLaunchKernelGGL((pbs_kernel<11, 1, 3, 8, GR, 1>), dim3(grid), dim3(G::T * GR), lds, st, a);
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (t.logN == 10 && t.k == 2 && t.l == 1 && t.unroll == 2) {
    // general two-bit rotation for the one-level bit tier (Ba2): one wave per ciphertext, four per workgroup
    using G = pbs_geom<10, 2, 1, 8, 1>;
    if (t.key_lds) {
      // the eight ciphertexts of a 512-thread workgroup share every key tile through an LDS ring filled by LDS-DMA (pbs_core.h, KLDS):
      // one copy of the key per CU through L1 instead of eight.  Same results; measured no faster than the free-running form below
      // (47.6 against 47.5 ms per 16 384: profiles/r03_exp_t4r2_ulow_ba2_keylds.log) -- the kernel sits at its instruction-issue
      // rate at the clock the chip holds, not at the L1 path -- so the catalogue leaves it off.
      constexpr int GR8 = 8, KL = 4;
      const size_t lds8 = pbs_lds_bytes<10, 2, 1, 8, 1, KL>(GR8);
      SET_LDS_ATTR((pbs_kernel<10, 2, 1, 8, GR8, 1, KL>), lds8);
      const unsigned grid8 = (unsigned)((a.count + GR8 - 1) / GR8);
      hipLaunchKernelGGL((pbs_kernel<10, 2, 1, 8, GR8, 1, KL>), dim3(grid8), dim3(G::T * GR8), lds8, st, a);
      HIPCHK(hipGetLastError());
      return 0;
    }
    constexpr int GR = 4;
    const size_t lds = G::TW_BYTES + (size_t)GR * G::GROUP_BYTES;
    SET_LDS_ATTR((pbs_kernel<10, 2, 1, 8, GR, 1>), lds);
    const unsigned grid = (unsigned)((a.count + GR - 1) / GR);
    hipLaunchKernelGGL((pbs_kernel<10, 2, 1, 8, GR, 1>), dim3(grid), dim3(G::T * GR), lds, st, a);
    HIPCHK(hipGetLastError());
    return 0;
  }
  return fail("no bootstrap kernel instantiated for logN=%d k=%d l=%d unroll=%d", t.logN, t.k, t.l, t.unroll);
}

static int make_twiddles(const dctfhe_tier& t, std::vector<cplx>& tw) {
#define X(LN, K_, L_, P_)                                        \
  if (t.logN == LN && t.k == K_ && t.l == L_) {                  \
    tw.resize(fft_geom<LN - 1, P_>::TW_ELEMS);                   \
    fill_twiddles<LN - 1, P_>(tw.data());                        \
    return 0;                                                    \
  }
  PBS_CASES(X)
#undef X
  return fail("no bootstrap kernel instantiated for logN=%d k=%d l=%d", t.logN, t.k, t.l);
}

static int launch_bsk_fourier(const dctfhe_tier& t, const uint64_t* polys, size_t npoly, const cplx* tw, cplx* out, hipStream_t st) {
#define X(LN, K_, L_, P_)                                                                            \
  if (t.logN == LN && t.k == K_ && t.l == L_) {                                                      \
    using F = fft_geom<LN - 1, P_>;                                                                  \
    constexpr int GR = groups_for<LN, K_, L_, P_>();                                                 \
    const size_t lds = (size_t)F::TW_ELEMS * 16 + (size_t)GR * F::EXCH_ELEMS * 16;                   \
    SET_LDS_ATTR((k_bsk_fourier<LN, P_, GR>), lds);                                                  \
    const unsigned grid = (unsigned)((npoly + GR - 1) / GR);                                         \
    hipLaunchKernelGGL((k_bsk_fourier<LN, P_, GR>), dim3(grid), dim3(F::T * GR), lds, st, polys, npoly, tw, out); \
    HIPCHK(hipGetLastError());                                                                       \
    return 0;                                                                                        \
  }
  PBS_CASES(X)
#undef X
  return fail("no bootstrap kernel instantiated for logN=%d k=%d l=%d", t.logN, t.k, t.l);
}

// ------------------------------------------------------------------------------------------ context
extern "C" int dctfhe_ctx_create(int device_id, dctfhe_ctx** out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail("dctfhe: no HIP device visible -- this engine has no CPU path");
  if (device_id < 0 || device_id >= ndev) return fail("dctfhe: device %d out of range (%d visible)", device_id, ndev);
  HIPCHK(hipSetDevice(device_id));
  auto* c = new dctfhe_ctx;
  c->device = device_id;
  HIPCHK(hipGetDeviceProperties(&c->prop, device_id));
  HIPCHK(hipStreamCreate(&c->own));
  c->stream = c->own;
  *out = c;
  return 0;
}
extern "C" int dctfhe_ctx_destroy(dctfhe_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  if (c->own) hipStreamDestroy(c->own);
  delete c;
  return 0;
}
extern "C" int dctfhe_ctx_set_stream(dctfhe_ctx* c, void* s) {
  c->stream = s ? (hipStream_t)s : c->own;
  return 0;
}
extern "C" int dctfhe_ctx_synchronize(dctfhe_ctx* c) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// ------------------------------------------------------------------------------------------ keygen
static int check_params(const dctfhe_params* p) {
  // this is the only gate in front of an untrusted evaluation-key blob (dctfhe_eval_keys_import): every field gets both bounds, and
  // the caps keep every size computed from them (eval_blob_size, key allocations) far inside size_t
  if (p->n_tiers < 1 || p->n_tiers > DCTFHE_MAX_TIERS) return fail("n_tiers out of range");
  if (p->D < 4 || p->D % 4 || p->D > (1 << 16)) return fail("D must be a positive multiple of 4, at most 65536");
  if (p->n_max < 1 || p->n_max > (1 << 13)) return fail("n_max out of range (1 .. 8192)");
  if (p->input_dim < 0 || p->input_dim > p->D) return fail("input_dim out of range");
  if (!(p->input_sigma >= 0.0) || p->input_sigma > 1.0) return fail("input_sigma out of range");
  for (int i = 0; i < p->n_tiers; i++) {
    const dctfhe_tier& t = p->tiers[i];
    if (t.n < 1 || t.n > p->n_max) return fail("tier %d: n out of range", i);
    if (t.k < 1 || t.k > 2 || t.logN < 8 || t.logN > 13) return fail("tier %d: k or logN out of range", i);
    if ((t.k << t.logN) > p->D) return fail("tier %d: k*N exceeds D", i);
    if (t.l * t.beta > 63 || t.l < 1 || t.l > 3 || t.beta < 1 || (t.l >= 2 && t.beta > 16) || (t.l == 1 && t.beta > 28))
      return fail("tier %d: bad bootstrap gadget (l <= 3; beta <= 16 when l >= 2, <= 28 when l == 1: 32-bit accumulators)", i);
    if (t.unroll != 1 && t.unroll != 2) return fail("tier %d: unroll must be 1 or 2", i);
    if (t.unroll == 2) {
      const bool paired = t.k == 1 && t.l == 1 && t.logN >= 11;                                        // pair-interleaved kernels
      const bool general = (t.k == 1 && t.l == 3 && t.logN == 11) || (t.k == 2 && t.l == 1 && t.logN == 10);   // general form
      if (!(paired || general) || (t.n & 1))
        return fail("tier %d: unroll 2 needs n even and (k, l, N) = (1, 1, >= 2048), (1, 3, 2048) or (2, 1, 1024)", i);
    }
    if (t.k < 1 || t.k > 2 || t.logN < 8 || t.logN > 13) return fail("tier %d: k or logN out of range", i);
    if (t.lk < 1 || t.lk > 63 || t.betak < 1 || t.betak > 8 || t.lk * t.betak > 63) return fail("tier %d: bad key-switch gadget (1 <= betak <= 8, lk >= 1, lk * betak <= 63)", i);
    if (!(t.lwe_sigma >= 0.0) || t.lwe_sigma > 1.0 || !(t.glwe_sigma >= 0.0) || t.glwe_sigma > 1.0) return fail("tier %d: noise parameter out of range", i);
    if (t.key_lds != 0 && !(t.key_lds == 1 && t.unroll == 2 && t.k == 2 && t.l == 1 && t.logN == 10))
      return fail("tier %d: key_lds is 0, or 1 on the (k, l, N, unroll) = (2, 1, 1024, 2) tier", i);
    if (!tier_ppt(t)) return fail("tier %d: no kernel for logN=%d k=%d l=%d", i, t.logN, t.k, t.l);
    if (t.ksk_share >= i) return fail("tier %d: ksk_share must name an earlier tier", i);
    if (t.ksk_share >= 0) {
      const dctfhe_tier& o = p->tiers[t.ksk_share];
      if (o.n != t.n || o.lk != t.lk || o.betak != t.betak) return fail("tier %d: shared key-switch key has another shape", i);
    }
  }
  return 0;
}

// host-only validators: usable (and tested) without a GPU
extern "C" int dctfhe_params_check(const dctfhe_params* params) {
  if (!params) return fail("null parameters");
  return check_params(params);
}

// ---- client key -----------------------------------------------------------------------------------
static rng_key key_from_bytes(const uint8_t* b) {
  rng_key k;
  for (int i = 0; i < 8; i++) k.k[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
  return k;
}

// (seed, nonce) -> the generator keys of dctfhe_encrypt: a key-derivation key that only the seed determines (one ChaCha20 block of the
// secret generator key at a stream of its own), one block of THAT at (stream, counter) = the nonce for the noise key, and the mask key one
// block of the noise key -- the same secret -> public step as for the key material.
static void derive_encrypt_keys(dctfhe_client_key* C) {
  uint32_t o[16];
  rng_key kdf;
  chacha20_block(C->sec, STREAM_ENCKDF, 0, o);
  for (int i = 0; i < 8; i++) kdf.k[i] = o[i];
  uint64_t n0 = 0, n1 = 0;
  for (int i = 0; i < 8; i++) { n0 |= (uint64_t)C->nonce[i] << (8 * i); n1 |= (uint64_t)C->nonce[8 + i] << (8 * i); }
  chacha20_block(kdf, n0, n1, o);
  for (int i = 0; i < 8; i++) C->enc_sec.k[i] = o[i];
  chacha20_block(C->enc_sec, STREAM_PUBKEY, 0, o);
  for (int i = 0; i < 8; i++) C->enc_pub.k[i] = o[i];
}

extern "C" int dctfhe_client_key_create(dctfhe_ctx* ctx, const dctfhe_params* params, const uint8_t* seed32, dctfhe_client_key** out) {
  if (!ctx || !params || !seed32 || !out) return fail("dctfhe_client_key_create: null argument");
  CHK(check_params(params));
  HIPCHK(hipSetDevice(ctx->device));
  std::unique_ptr<dctfhe_client_key> C(new dctfhe_client_key);
  C->ctx = ctx; C->p = *params;
  memcpy(C->seed, seed32, 32);
  C->sec = key_from_bytes(seed32);
  {  // the public (mask) key is one ChaCha20 block of the secret one: knowing it says nothing about the secret key
    uint32_t o[16];
    chacha20_block(C->sec, STREAM_PUBKEY, 0, o);
    for (int i = 0; i < 8; i++) C->pub.k[i] = o[i];
  }
  {  // per-handle encryption nonce from the OS
    size_t got = 0;
    while (got < sizeof C->nonce) {
      const ssize_t r = getrandom(C->nonce + got, sizeof C->nonce - got, 0);
      if (r <= 0) return fail("dctfhe_client_key_create: the OS gave no random bytes for the encryption nonce (getrandom)");
      got += (size_t)r;
    }
    derive_encrypt_keys(C.get());
  }
  hipStream_t st = ctx->stream;
  const int D = params->D;
  HIPCHK(hipMalloc(&C->d_S, D));
  HIPCHK(hipMalloc(&C->d_s, params->n_max));
  hipLaunchKernelGGL(k_gen_bits, dim3((D + 255) / 256), dim3(256), 0, st, C->sec, (uint64_t)STREAM_BIGKEY, C->d_S, D);
  hipLaunchKernelGGL(k_gen_bits, dim3((params->n_max + 255) / 256), dim3(256), 0, st, C->sec, (uint64_t)STREAM_SMALLKEY, C->d_s, params->n_max);
  HIPCHK(hipGetLastError());
  for (int ti = 0; ti < params->n_tiers; ti++) {
    const dctfhe_tier& t = params->tiers[ti];
    if (t.unroll != 2) continue;
    HIPCHK(hipMalloc(&C->d_spair[ti], (size_t)(3 * t.n / 2)));
    hipLaunchKernelGGL(k_pair_secret, dim3((t.n / 2 + 255) / 256), dim3(256), 0, st, C->d_s, t.n, C->d_spair[ti]);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(st));
  *out = C.release();
  return 0;
}
extern "C" int dctfhe_client_key_destroy(dctfhe_client_key* C) { delete C; return 0; }

// standard-domain bootstrap key blocks [i0, i0+ni) of tier `tier` (regenerated from the client's streams whenever needed)
static int gen_bsk_std_chunk(dctfhe_client_key* C, int tier, int i0, int ni, uint64_t* d_out) {
  const dctfhe_tier& t = C->p.tiers[tier];
  const int N = 1 << t.logN, rows = (t.k + 1) * t.l;
  SET_LDS_ATTR(k_bsk_gen_std, 8192 * 8);
  // unroll 2: the "secret" is the pair secret of 3n/2 bits, i0/ni count its blocks
  const uint8_t* bits = t.unroll == 2 ? C->d_spair[tier] : C->d_s;
  hipLaunchKernelGGL(k_bsk_gen_std, dim3((unsigned)(ni * rows)), dim3(256), (size_t)N * 8, C->ctx->stream, bits, C->d_S, i0, t.k, N,
                     t.l, t.beta, t.glwe_sigma, C->pub, C->sec, (uint64_t)(STREAM_BSK_MASK + 2 * tier), d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// Precision of a key-switch key: the torus grid its words live on, 2^-(8 limbs).  The finest gadget level sits at 2^-(lk betak) and the
// row noise (sigma_lwe >= 2^-19 on every shipped tier) at least 6 bits above the grid: limbs >= (lk betak + 6) / 8, as a power of two
// (the matrix-core epilogue folds a word's limbs with lane shuffles).  Table tiers (9 levels of base 4): 4 limbs; one-bit tiers (5
// levels): 2 limbs -- a half and a quarter of round 2's GEMM.
static int ks_limbs(const dctfhe_tier& t) {
  const int bits = t.lk * t.betak + 6;
  return bits <= 16 ? 2 : bits <= 32 ? 4 : 8;
}

// ---- evaluation keys ------------------------------------------------------------------------------
static size_t tier_bsk_blocks(const dctfhe_tier& t) { return (size_t)(t.unroll == 2 ? 3 * t.n / 2 : t.n); }
static size_t tier_bsk_elems(const dctfhe_tier& t) {   // complex values of the Fourier key, without the warm-up padding
  return tier_bsk_blocks(t) * (size_t)(t.k + 1) * t.l * (t.k + 1) * ((size_t)1 << (t.logN - 1));
}
static size_t tier_ksk_words(const dctfhe_params& p, const dctfhe_tier& t) { return (size_t)p.D * t.lk * (t.n + 1); }

// allocate every array of the evaluation keys and fill what depends on the parameters only (twiddles, root tables)
static int eval_alloc(dctfhe_ctx* ctx, const dctfhe_params* params, std::unique_ptr<dctfhe_eval_keys>& E) {
  CHK(check_params(params));
  HIPCHK(hipSetDevice(ctx->device));
  E.reset(new dctfhe_eval_keys);
  E->ctx = ctx; E->p = *params;
  hipStream_t st = ctx->stream;
  const int D = params->D;
  HIPCHK(hipMalloc(&E->d_dummy, (size_t)(D + 1) * 8));
  for (int ti = 0; ti < params->n_tiers; ti++) {
    const dctfhe_tier& t = params->tiers[ti];
    TierKeys& tk = E->tiers[ti];
    tk.t = t;
    if (t.ksk_share >= 0) {
      const TierKeys& o = E->tiers[t.ksk_share];
      tk.d_ksk = o.d_ksk; tk.d_colsum = o.d_colsum; tk.d_kskT = o.d_kskT; tk.ncol_pad = o.ncol_pad; tk.colsum_eff = o.colsum_eff; tk.limbs = o.limbs;
    } else {
      tk.own_ksk = true;
      tk.limbs = ks_limbs(t);
      tk.colsum_eff = new std::map<int, uint64_t*>();
      const size_t rows = (size_t)D * t.lk;
      HIPCHK(hipMalloc(&tk.d_ksk, rows * (t.n + 1) * 8));
      HIPCHK(hipMalloc(&tk.d_colsum, (size_t)(t.n + 1) * 8));
      // the i8 MFMA key switch reads digits as SIGNED bytes: offset digits in [0, 2^betak) need betak <= 7;
      // wider gadgets and shapes the tiling does not cover go to the integer-VALU GEMM
      if (rows % 64 == 0 && rows <= (1u << 17) && t.betak <= 7) {
        tk.ncol_pad = ((tk.limbs * (t.n + 1) + 127) / 128) * 128;
        HIPCHK(hipMalloc(&tk.d_kskT, (size_t)tk.ncol_pad * rows));
      }
    }
    std::vector<cplx> tw;
    CHK(make_twiddles(t, tw));
    HIPCHK(hipMalloc(&tk.d_tw, tw.size() * sizeof(cplx)));
    HIPCHK(hipMemcpyAsync(tk.d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    const int N = 1 << t.logN, M = N / 2;
    if (t.unroll == 2) {
      std::vector<cplx> wt((size_t)2 * N + 8);
      const long double PI = 3.141592653589793238462643383279502884L;
      for (int m = 0; m < 2 * N; m++) { const long double a = PI * m / N; wt[m] = cmk((double)cosl(a), (double)sinl(a)); }
      for (int m = 0; m < 8; m++) wt[(size_t)2 * N + m] = root64(8 * m);
      // the root table is gathered from at random by every thread of the two-bit kernels, every iteration: give it an allocation that
      // maps with one large page (a 262 KB allocation lands wherever the sub-allocator has room, and the N = 8192 bootstrap measured
      // 0 / +4 / +5 / +11 % at table addresses that were 256 / 64 / 128 / 32 KB-aligned: profiles/r02_exp_ablations.log)
      const size_t wt_bytes = (((wt.size() * sizeof(cplx)) + (2u << 20) - 1) >> 21) << 21;
      HIPCHK(hipMalloc(&tk.d_wtab, wt_bytes));
      HIPCHK(hipMemcpyAsync(tk.d_wtab, wt.data(), wt.size() * sizeof(cplx), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    // PBS_PF_DIST iterations' worth of extra (zero) key blocks: the L2 warm-up of the last iterations reads past the key
    // (launch_pbs checks this contract: pbs_core.h, "L2 warm-up geometry")
    const size_t per_block = (size_t)(t.k + 1) * t.l * (t.k + 1) * M;
    const size_t pad = (size_t)PBS_PF_DIST * (t.unroll == 2 ? 3 : 1) * per_block;
    HIPCHK(hipMalloc(&tk.d_bsk, (tier_bsk_elems(t) + pad) * sizeof(cplx)));
    HIPCHK(hipMemsetAsync(tk.d_bsk + tier_bsk_elems(t), 0, pad * sizeof(cplx), st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

// what is derived from the key-switch keys once they are in place: column sums and the signed byte limbs
static int eval_finish(dctfhe_eval_keys* E) {
  hipStream_t st = E->ctx->stream;
  for (int ti = 0; ti < E->p.n_tiers; ti++) {
    const dctfhe_tier& t = E->p.tiers[ti];
    TierKeys& tk = E->tiers[ti];
    if (!tk.own_ksk) continue;
    const size_t rows = (size_t)E->p.D * t.lk;
    hipLaunchKernelGGL(k_ksk_colsum, dim3((t.n + 256) / 256), dim3(256), 0, st, tk.d_ksk, (int)rows, t.n, tk.d_colsum);
    if (tk.d_kskT) {
      // the limb form keeps the top tk.limbs bytes of every word: make sure nothing sits below them (an imported key must be on the grid)
      DevBuf flag;
      HIPCHK(flag.alloc(sizeof(int)));
      HIPCHK(hipMemsetAsync(flag.p, 0, sizeof(int), st));
      hipLaunchKernelGGL(k_ksk_off_grid, dim3(2048), dim3(256), 0, st, tk.d_ksk, rows * (size_t)(t.n + 1), tk.limbs, flag.as<int>());
      int bad = 0;
      HIPCHK(hipMemcpyAsync(&bad, flag.p, sizeof bad, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (bad) return fail("tier %d: key-switch key words off the 2^-%d torus grid (a key made by another build?)", ti, 8 * tk.limbs);
      hipLaunchKernelGGL(k_ksk_to_limbs, dim3(4096), dim3(256), 0, st, tk.d_ksk, (int)rows, t.n, tk.limbs, tk.ncol_pad, tk.d_kskT);
    }
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int dctfhe_eval_keys_generate(dctfhe_client_key* C, dctfhe_eval_keys** out) {
  if (!C || !out) return fail("dctfhe_eval_keys_generate: null argument");
  std::unique_ptr<dctfhe_eval_keys> E;
  CHK(eval_alloc(C->ctx, &C->p, E));
  hipStream_t st = C->ctx->stream;
  const int D = C->p.D;
  for (int ti = 0; ti < C->p.n_tiers; ti++) {
    const dctfhe_tier& t = C->p.tiers[ti];
    TierKeys& tk = E->tiers[ti];
    if (tk.own_ksk) {
      hipLaunchKernelGGL(k_ksk_gen, dim3((unsigned)((size_t)D * t.lk)), dim3(256), 0, st, C->d_S, C->d_s, t.n, t.lk, t.betak, ks_limbs(t), t.lwe_sigma, C->pub, C->sec,
                         (uint64_t)(STREAM_KSK + 2 * ti), tk.d_ksk);
      HIPCHK(hipGetLastError());
    }
    const int N = 1 << t.logN, M = N / 2;
    const size_t per_bit_polys = (size_t)(t.k + 1) * t.l * (t.k + 1);
    const int blocks = (int)tier_bsk_blocks(t);
    const int chunk = std::max(1, (int)std::min<size_t>(blocks, ((size_t)64 << 20) / (per_bit_polys * N * 8)));
    DevBuf d_std;
    HIPCHK(d_std.alloc((size_t)chunk * per_bit_polys * N * 8));
    for (int i0 = 0; i0 < blocks; i0 += chunk) {
      const int ni = std::min(chunk, blocks - i0);
      CHK(gen_bsk_std_chunk(C, ti, i0, ni, d_std.as<uint64_t>()));
      CHK(launch_bsk_fourier(t, d_std.as<uint64_t>(), (size_t)ni * per_bit_polys, tk.d_tw, tk.d_bsk + (size_t)i0 * per_bit_polys * M, st));
    }
    HIPCHK(hipStreamSynchronize(st));
  }
  CHK(eval_finish(E.get()));
  *out = E.release();
  return 0;
}

extern "C" int dctfhe_keygen(dctfhe_ctx* ctx, const dctfhe_params* params, const uint8_t* seed32, dctfhe_client_key** client, dctfhe_eval_keys** eval) {
  if (!client || !eval) return fail("dctfhe_keygen: null output");
  dctfhe_client_key* C = nullptr;
  CHK(dctfhe_client_key_create(ctx, params, seed32, &C));
  dctfhe_eval_keys* E = nullptr;
  if (dctfhe_eval_keys_generate(C, &E)) { delete C; return -1; }
  *client = C; *eval = E;
  return 0;
}
extern "C" int dctfhe_eval_keys_destroy(dctfhe_eval_keys* E) { delete E; return 0; }

// ---- evaluation-key persistence: header + params, then per tier its own key-switch key (u64) and its Fourier bootstrap key
// version 2: the Fourier bootstrap keys carry 2^-64 / M (accumulator updates in units of the whole torus, fft_core.h); version 1 carried 1 / M
// version 3: the key-switch keys live on the 2^-(8 ks_limbs) torus grid (version-2 keys used all 64 bits and would lose their low bytes)
static constexpr uint32_t EVAL_BLOB_VERSION = 3;
struct EvalBlobHeader { uint32_t magic, version; uint64_t total_bytes; dctfhe_params params; };
static size_t eval_blob_size(const dctfhe_params& p) {
  size_t n = sizeof(EvalBlobHeader);
  for (int ti = 0; ti < p.n_tiers; ti++) {
    const dctfhe_tier& t = p.tiers[ti];
    if (t.ksk_share < 0) n += tier_ksk_words(p, t) * 8;
    n += tier_bsk_elems(t) * sizeof(cplx);
  }
  return n;
}
extern "C" int dctfhe_eval_keys_export(dctfhe_eval_keys* E, void* buf, size_t capacity, size_t* size) {
  if (!E || !size) return fail("dctfhe_eval_keys_export: null argument");
  const size_t need = eval_blob_size(E->p);
  *size = need;
  if (!buf) return 0;                       // size query
  if (capacity < need) return fail("dctfhe_eval_keys_export: buffer of %zu bytes, %zu needed", capacity, need);
  HIPCHK(hipSetDevice(E->ctx->device));
  HIPCHK(hipStreamSynchronize(E->ctx->stream));
  EvalBlobHeader h{};
  h.magic = 0x4b564544u /* 'DEVK' */; h.version = EVAL_BLOB_VERSION; h.total_bytes = need; h.params = E->p;
  char* q = (char*)buf;
  memcpy(q, &h, sizeof h); q += sizeof h;
  for (int ti = 0; ti < E->p.n_tiers; ti++) {
    const dctfhe_tier& t = E->p.tiers[ti];
    if (t.ksk_share < 0) {
      HIPCHK(hipMemcpy(q, E->tiers[ti].d_ksk, tier_ksk_words(E->p, t) * 8, hipMemcpyDeviceToHost));
      q += tier_ksk_words(E->p, t) * 8;
    }
    HIPCHK(hipMemcpy(q, E->tiers[ti].d_bsk, tier_bsk_elems(t) * sizeof(cplx), hipMemcpyDeviceToHost));
    q += tier_bsk_elems(t) * sizeof(cplx);
  }
  return 0;
}
extern "C" int dctfhe_eval_keys_import(dctfhe_ctx* ctx, const void* buf, size_t size, dctfhe_eval_keys** out) {
  if (!ctx || !buf || !out) return fail("dctfhe_eval_keys_import: null argument");
  if (size < sizeof(EvalBlobHeader)) return fail("evaluation-key blob too short");
  EvalBlobHeader h;
  memcpy(&h, buf, sizeof h);
  if (h.magic != 0x4b564544u || h.version != EVAL_BLOB_VERSION) return fail("bad evaluation-key blob magic/version");
  CHK(check_params(&h.params));
  if (h.total_bytes != size || eval_blob_size(h.params) != size) return fail("evaluation-key blob is %zu bytes, its parameters need %zu", size, eval_blob_size(h.params));
  std::unique_ptr<dctfhe_eval_keys> E;
  CHK(eval_alloc(ctx, &h.params, E));
  const char* q = (const char*)buf + sizeof h;
  for (int ti = 0; ti < h.params.n_tiers; ti++) {
    const dctfhe_tier& t = h.params.tiers[ti];
    if (t.ksk_share < 0) {
      HIPCHK(hipMemcpy(E->tiers[ti].d_ksk, q, tier_ksk_words(h.params, t) * 8, hipMemcpyHostToDevice));
      q += tier_ksk_words(h.params, t) * 8;
    }
    HIPCHK(hipMemcpy(E->tiers[ti].d_bsk, q, tier_bsk_elems(t) * sizeof(cplx), hipMemcpyHostToDevice));
    q += tier_bsk_elems(t) * sizeof(cplx);
  }
  CHK(eval_finish(E.get()));
  *out = E.release();
  return 0;
}

// ---- test / client views ---------------------------------------------------------------------------
extern "C" int dctfhe_client_key_export_secret(dctfhe_client_key* C, uint8_t* big_key, uint8_t* small_key) {
  HIPCHK(hipSetDevice(C->ctx->device));
  HIPCHK(hipMemcpy(big_key, C->d_S, C->p.D, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(small_key, C->d_s, C->p.n_max, hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int dctfhe_eval_keys_export_ksk(dctfhe_eval_keys* K, int tier, uint64_t* out) {
  if (tier < 0 || tier >= K->p.n_tiers) return fail("tier out of range");
  const dctfhe_tier& t = K->p.tiers[tier];
  HIPCHK(hipSetDevice(K->ctx->device));
  HIPCHK(hipMemcpy(out, K->tiers[tier].d_ksk, (size_t)K->p.D * t.lk * (t.n + 1) * 8, hipMemcpyDeviceToHost));
  return 0;
}
// the standard-domain bootstrap key, regenerated from the client's streams (what dctfhe_eval_keys_generate transformed)
extern "C" int dctfhe_client_key_export_bsk(dctfhe_client_key* C, int tier, uint64_t* out) {
  if (tier < 0 || tier >= C->p.n_tiers) return fail("tier out of range");
  const dctfhe_tier& t = C->p.tiers[tier];
  HIPCHK(hipSetDevice(C->ctx->device));
  const int N = 1 << t.logN;
  const int blocks = (int)tier_bsk_blocks(t);     // unroll 2: the key of the pair secret
  const size_t words = (size_t)blocks * (t.k + 1) * t.l * (t.k + 1) * N;
  DevBuf d;
  HIPCHK(d.alloc(words * 8));
  CHK(gen_bsk_std_chunk(C, tier, 0, blocks, d.as<uint64_t>()));
  HIPCHK(hipStreamSynchronize(C->ctx->stream));
  HIPCHK(hipMemcpy(out, d.p, words * 8, hipMemcpyDeviceToHost));
  return 0;
}

// CSPRNG views: `count` outputs of (key, stream, idx0..) -- on the host (no GPU needed: known-answer tests) and on the device
extern "C" int dctfhe_rng_host(const uint8_t* key32, uint64_t stream, uint64_t idx0, size_t count, uint64_t* out) {
  const rng_key k = key_from_bytes(key32);
  for (size_t i = 0; i < count; i++) out[i] = rnd64(k, stream, idx0 + i);
  return 0;
}
extern "C" int dctfhe_rng_device(dctfhe_ctx* ctx, const uint8_t* key32, uint64_t stream, uint64_t idx0, size_t count, uint64_t* out) {
  if (count == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf d;
  HIPCHK(d.alloc(count * 8));
  hipLaunchKernelGGL(k_rng_fill, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, key_from_bytes(key32), stream, idx0, d.as<uint64_t>(), count);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, d.p, count * 8, hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------------------ client ops
// rows of `dim` mask words + body, input_dim <= dim <= D
extern "C" int dctfhe_encrypt_rows(dctfhe_ctx* ctx, dctfhe_client_key* C, const uint64_t* phases, size_t count, int dim, uint64_t* cts) {
  if (!ctx || !C || (count && (!phases || !cts))) return fail("dctfhe_encrypt: null argument");
  const int D = C->p.D, dim_eff = C->p.input_dim > 0 ? C->p.input_dim : D;
  if (dim < dim_eff || dim > D) return fail("dctfhe_encrypt: rows of %d mask words; these parameters mask %d and the key has %d", dim, dim_eff, D);
  if (count == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t L = (size_t)dim + 1;
  DevBuf d_ph, d_ct;
  HIPCHK(d_ph.alloc(count * 8));
  HIPCHK(d_ct.alloc(count * L * 8));
  HIPCHK(hipMemcpyAsync(d_ph.p, phases, count * 8, hipMemcpyHostToDevice, ctx->stream));
  // fresh streams per call: a (mask, noise) pair is never drawn twice under one handle; handles differ by their nonce
  const uint64_t stream = (uint64_t)STREAM_ENC + ((++C->enc_calls) << 16);
  hipLaunchKernelGGL(k_lwe_encrypt, dim3((unsigned)count), dim3(256), 0, ctx->stream, C->d_S, D, dim, dim_eff, d_ph.as<uint64_t>(),
                     C->p.input_sigma, C->enc_pub, C->enc_sec, stream, d_ct.as<uint64_t>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(cts, d_ct.p, count * L * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}
extern "C" int dctfhe_encrypt(dctfhe_ctx* ctx, dctfhe_client_key* C, const uint64_t* phases, size_t count, uint64_t* cts) {
  if (!C) return fail("dctfhe_encrypt: null argument");
  return dctfhe_encrypt_rows(ctx, C, phases, count, C->p.D, cts);
}
// a handle's position in its own encryption streams (the streams themselves differ from handle to handle by the nonce)
extern "C" int dctfhe_client_key_set_encrypt_counter(dctfhe_client_key* C, uint64_t next_call) {
  if (!C) return fail("null client key");
  if (next_call >= (1ULL << 47)) return fail("encrypt counter out of range");
  C->enc_calls = next_call;
  return 0;
}
// reproducible experiments and tests only: fix the nonce (and with it every mask and noise value dctfhe_encrypt will draw)
extern "C" int dctfhe_client_key_set_encrypt_nonce(dctfhe_client_key* C, const uint8_t* nonce16) {
  if (!C || !nonce16) return fail("dctfhe_client_key_set_encrypt_nonce: null argument");
  memcpy(C->nonce, nonce16, sizeof C->nonce);
  derive_encrypt_keys(C);
  return 0;
}
extern "C" int dctfhe_decrypt_rows(dctfhe_ctx* ctx, dctfhe_client_key* C, const uint64_t* cts, size_t count, int dim, uint64_t* phases) {
  if (!ctx || !C || (count && (!phases || !cts))) return fail("dctfhe_decrypt: null argument");
  if (dim < 0 || dim > C->p.D) return fail("dctfhe_decrypt: rows of %d mask words, the key has %d", dim, C->p.D);
  if (count == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t L = (size_t)dim + 1;
  DevBuf d_ph, d_ct;
  HIPCHK(d_ph.alloc(count * 8));
  HIPCHK(d_ct.alloc(count * L * 8));
  HIPCHK(hipMemcpyAsync(d_ct.p, cts, count * L * 8, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_lwe_phase, dim3((unsigned)count), dim3(256), 0, ctx->stream, C->d_S, dim, d_ct.as<uint64_t>(), d_ph.as<uint64_t>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(phases, d_ph.p, count * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}
extern "C" int dctfhe_decrypt(dctfhe_ctx* ctx, dctfhe_client_key* C, const uint64_t* cts, size_t count, uint64_t* phases) {
  if (!C) return fail("dctfhe_decrypt: null argument");
  return dctfhe_decrypt_rows(ctx, C, cts, count, C->p.D, phases);
}

// ------------------------------------------------------------------------------------------ device-level building blocks
struct Timers {
  hipStream_t st;
  bool on;
  std::vector<hipEvent_t>* pool;     // owned by the session (or by nobody: primitives pass on = false)
  struct Span { hipEvent_t a, b; int cat; };
  std::vector<Span> spans;
  size_t used = 0;
  hipEvent_t take() {
    if (used == pool->size()) { hipEvent_t e; hipEventCreate(&e); pool->push_back(e); }
    return (*pool)[used++];
  }
  int begin(int cat) {
    if (!on) return -1;
    Span s; s.cat = cat;
    s.a = take(); s.b = take();
    hipEventRecord(s.a, st);
    spans.push_back(s);
    return (int)spans.size() - 1;
  }
  void end(int h) { if (h >= 0) hipEventRecord(spans[h].b, st); }
};
enum { CAT_LINEAR = 100, CAT_KS = 101 };  // 0..7: bootstrap of tier i

// key switch of `count` ciphertexts (D+1 words each) into small ciphertexts of tier `tier`
// deff: mask words beyond it are known to be zero in every input (0 or >= D: no such knowledge).  The key switch then
// runs on the first deff rows of the key only -- the same result bit for bit, deff/D of the work.
static int dev_keyswitch(dctfhe_keys* K, int tier, const uint64_t* d_cts, size_t count, int shift, uint8_t* d_digits, uint64_t* d_bodies,
                         uint64_t* d_small, Timers* tm, int deff = 0, size_t L = 0, uint64_t body_add = 0) {
  const dctfhe_tier& t = K->p.tiers[tier];
  TierKeys& tk = K->tiers[tier];
  const int D = K->p.D;
  if (L == 0) L = (size_t)D + 1;                 // row stride of the input ciphertexts (body at L-1); host format by default
  const int Dmax = (int)std::min<size_t>(L - 1, (size_t)D);
  hipStream_t st = K->ctx->stream;
  int De = (deff > 0 && deff < Dmax && tk.d_kskT && (deff * t.lk) % 64 == 0) ? deff : Dmax;
  if (!tk.d_kskT && De != D) return fail("key switch: the integer-VALU path needs full-width rows");
  if (tk.d_kskT && (De * t.lk) % 64 != 0) return fail("key switch: %d rows x %d levels is not a multiple of 64", De, t.lk);
  const uint64_t* colsum = tk.d_colsum;
  if (De < D) {
    auto it = tk.colsum_eff->find(De);
    if (it == tk.colsum_eff->end()) {
      uint64_t* cs = nullptr;
      HIPCHK(hipMalloc(&cs, (size_t)(t.n + 1) * 8));
      hipLaunchKernelGGL(k_ksk_colsum, dim3((t.n + 256) / 256), dim3(256), 0, st, tk.d_ksk, De * t.lk, t.n, cs);
      HIPCHK(hipGetLastError());
      it = tk.colsum_eff->emplace(De, cs).first;
    }
    colsum = it->second;
  }
  const int h = tm ? tm->begin(CAT_KS) : -1;
  const size_t total = count * (size_t)De;
  const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(k_ks_decompose, dim3(grid), dim3(256), 0, st, d_cts, count, L, De, shift, body_add, t.lk, t.betak, d_digits, d_bodies);
  if (tk.d_kskT) {   // matrix-core path: i8 digits x signed byte limbs of the key
    const unsigned ncb = (unsigned)(tk.ncol_pad / 128), nrb = (unsigned)((count + 127) / 128), cpx = (ncb + 7) / 8;
#define KS_LAUNCH(LB)                                                                                                                      \
    hipLaunchKernelGGL(k_ks_mfma<LB>, dim3(8 * cpx * nrb), dim3(256), 0, st, d_digits, d_bodies, count, De * t.lk, tk.d_kskT, D * t.lk, tk.ncol_pad, \
                       colsum, t.n, t.betak, d_small)
    if (tk.limbs == 2) KS_LAUNCH(2); else if (tk.limbs == 4) KS_LAUNCH(4); else KS_LAUNCH(8);
#undef KS_LAUNCH
  } else {           // shapes the MFMA tiling does not cover (D*lk not a multiple of 64, betak = 8): integer VALU GEMM
    constexpr int CT = 16;
    dim3 g2((t.n + 1 + 255) / 256, (unsigned)((count + CT - 1) / CT));
    hipLaunchKernelGGL(k_ks_gemm<CT>, g2, dim3(256), 0, st, d_digits, d_bodies, count, D * t.lk, tk.d_ksk, tk.d_colsum, t.n, t.betak, d_small);
  }
  HIPCHK(hipGetLastError());
  if (tm) tm->end(h);
  return 0;
}

// centred mod switch on `count` small ciphertexts of tier `tier`, in place (kernels.h k_ms_center); part of the key-switch span of the timers
static int dev_ms_center(dctfhe_keys* K, int tier, uint64_t* d_small, size_t count, Timers* tm) {
  if (count == 0) return 0;
  const dctfhe_tier& t = K->p.tiers[tier];
  const int h = tm ? tm->begin(CAT_KS) : -1;
  hipLaunchKernelGGL(k_ms_center, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, K->ctx->stream, d_small, count, t.n, t.logN);
  HIPCHK(hipGetLastError());
  if (tm) tm->end(h);
  return 0;
}

static int dev_pbs(dctfhe_keys* K, int tier, const uint64_t* d_small, size_t count, const int64_t* d_tables, int w, const int32_t* d_idx,
                   int hw, int nchan, size_t e_offset, uint64_t* d_out, int accumulate, uint64_t body_add, Timers* tm, size_t L_out = 0) {
  const dctfhe_tier& t = K->p.tiers[tier];
  pbs_launch a;
  a.cts_small = d_small; a.count = count; a.n = t.n; a.beta = t.beta;
  a.bsk = K->tiers[tier].d_bsk; a.tw = K->tiers[tier].d_tw; a.wtab = K->tiers[tier].d_wtab;
  a.tables = d_tables; a.w = w; a.table_idx = d_idx; a.hw = hw; a.nchan = nchan; a.e_offset = e_offset;
  const int ring = t.k << t.logN;
  if (L_out == 0) L_out = (size_t)K->p.D + 1;
  if (L_out < (size_t)ring + 1) return fail("bootstrap output rows of %zu words cannot hold a ring of %d", L_out, ring);
  a.out = d_out; a.D_out = (int)L_out - 1; a.accumulate = accumulate; a.body_add = body_add; a.dummy = K->d_dummy; a.bsk_wrap = 0;
  // L2 warm-up: each workgroup touches 1/pf_parts of the next key blocks; the paired two-bit kernel at N = 2048 measures 3 % better
  // without (profiles/r02_exp_ablations.log: 22.5 vs 23.2 ms), every other barrier-coupled kernel better with (T4r 82.4 vs 86.3, T5a 25.7
  // vs 26.6).  The parts are dealt by blockIdx / 8, i.e. per XCD: with one 512-thread workgroup per CU an XCD holds 32 of them, and 32
  // parts -- every line touched once per XCD instead of twice -- is worth 4 % on the N = 8192 kernel (288 vs 300 ms per 12 288; 24, 40,
  // 48, 64 parts: 302, 318, 325, 330), nothing at N = 4096 (143.4 vs 144.0) and N = 2048 three levels (180.9 vs 181.1).
  a.pf_parts = (t.unroll == 2 && t.logN == 11 && t.l == 1) ? 0 : t.logN >= 13 ? 32 : 16;
  const int h = tm ? tm->begin(tier) : -1;
  CHK(launch_pbs(t, a, K->ctx->stream));
  if (tm) tm->end(h);
  return 0;
}

static int dev_conv2d(hipStream_t st, const uint64_t* in, int batch, int Cin, int H, int W, size_t Lin, size_t deff, const int8_t* d_w, const ConvPack* pk, int Cout,
                      int KH, int KW, int stride, int pad, uint64_t* out, size_t Lout) {
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (pk && pk->d_w && Lout >= 32) {
    // ciphertext rows: the contraction over (ci, ky, kx) on the matrix cores, input words split into signed byte limbs on the fly
    dim3 grid((unsigned)((Lout + 31) / 32), (unsigned)(batch * Ho * Wo), (unsigned)((Cout + 63) / 64));
    hipLaunchKernelGGL(k_conv2d_mfma, grid, dim3(256), 0, st, in, H, W, Lin, deff, pk->d_w, pk->d_taps, pk->kpad, Cin * H * W, Cout, stride, Ho, Wo, Lout, out);
    HIPCHK(hipGetLastError());
    return 0;
  }
  // clear mode (one word per element): u64 wrap MACs on the vector ALU.
  // 16 output channels per thread: 32 / 64 (fewer re-reads of the input) measured 1.45x / 3.5x SLOWER -- their per-tap weights no
  // longer fit the scalar registers (profiles/r02_exp_ablations.log)
  constexpr int COT = 16;
  dim3 grid((unsigned)((Lout + 255) / 256), (unsigned)(batch * Ho * Wo), (unsigned)((Cout + COT - 1) / COT));
  hipLaunchKernelGGL(k_conv2d<COT>, grid, dim3(256), 0, st, in, Cin, H, W, Lin, deff, d_w, Cout, KH, KW, stride, pad, Ho, Wo, Lout, out);
  HIPCHK(hipGetLastError());
  return 0;
}

static unsigned ew_grid(size_t n) { return (unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 16384)); }

// exact rounding + table look-up on `count` ciphertexts, in place on d_work (already shifted / offset)
struct LutScratch { uint8_t* digits = nullptr; uint64_t* bodies = nullptr; uint64_t* small = nullptr; int64_t* bit_tables = nullptr; size_t chunk = 0; };
// Tier of rounding step i of a look-up op: ip[5] (bit tier), from step ip[8] on its one-level twin ip[7], from step ip[11] & 255 on the
// two-bit-rotation twin ip[11] >> 8 (the compiler proves each hand-over safe; dctfhe/compile.py::step_tier is the same rule).
struct StepTiers {
  int bit = -1, coarse = -1, coarse_from = 1 << 30, coarse2 = -1, coarse2_from = 1 << 30;
  int at(int i) const { return (coarse2 >= 0 && i >= coarse2_from) ? coarse2 : (coarse >= 0 && i >= coarse_from) ? coarse : bit; }
};
static StepTiers step_tiers_of(const Op& o) {
  StepTiers s;
  s.bit = o.ip[5]; s.coarse = o.ip[7]; s.coarse_from = o.ip[8];
  if (o.ip[11] >= 0) { s.coarse2 = o.ip[11] >> 8; s.coarse2_from = o.ip[11] & 255; }
  return s;
}
// r > 0: in place on d_work (rows of Lw words, already shifted / offset; the first `deff` mask words may be non-zero).
// r == 0: nothing modifies the input, so the key switch reads d_src (rows of Ls words) directly with the site's shift and
// body offset applied on the fly, and the table bootstrap writes d_work -- no copy of the tensor at all.
static int dev_round_lut(dctfhe_keys* K, const StepTiers& stp, int tab_tier, const uint64_t* d_src, size_t Ls, int shift,
                         uint64_t body_add, uint64_t* d_work, size_t Lw, size_t count, int p, int r, const int64_t* d_tables, int w, const int32_t* d_idx,
                         int hw, int nchan, const LutScratch& sc, Timers* tm, int deff = 0) {
  for (size_t c0 = 0; c0 < count; c0 += sc.chunk) {
    const size_t cn = std::min(sc.chunk, count - c0);
    uint64_t* w0 = d_work + c0 * Lw;
    for (int i = 0; i < r; i++) {
      const int bt = stp.at(i);
      CHK(dev_keyswitch(K, bt, w0, cn, p - i, sc.digits, sc.bodies, sc.small, tm, deff, Lw));
      CHK(dev_ms_center(K, bt, sc.small, cn, tm));
      const int vlog = 62 - p + i;
      CHK(dev_pbs(K, bt, sc.small, cn, sc.bit_tables + vlog, 0, nullptr, 1, 1, 0, w0, 1, (uint64_t)0 - (1ULL << vlog), tm, Lw));
    }
    if (r > 0) CHK(dev_keyswitch(K, tab_tier, w0, cn, 0, sc.digits, sc.bodies, sc.small, tm, deff, Lw));
    else       CHK(dev_keyswitch(K, tab_tier, d_src + c0 * Ls, cn, shift, sc.digits, sc.bodies, sc.small, tm, deff, Ls, body_add));
    CHK(dev_ms_center(K, tab_tier, sc.small, cn, tm));
    CHK(dev_pbs(K, tab_tier, sc.small, cn, d_tables, w, d_idx ? d_idx + c0 : nullptr, hw, nchan, c0, w0, 0, 0, tm, Lw));
  }
  return 0;
}
// the mask words a rounding chain touches: its input's, and the rings of the bit tiers whose outputs it accumulates
static int round_chain_deff(const dctfhe_keys* K, int src_deff, const StepTiers& stp, int r) {
  int d = src_deff;
  for (int i = 0; i < r; i++) {
    const int tier = stp.at(i);
    if (tier >= 0) d = std::max(d, K->p.tiers[tier].k << K->p.tiers[tier].logN);
  }
  return d;
}

// can the key switch of `tier` run on the first deff rows of its key only?  (matrix-core path, whole 64-row steps; otherwise it walks the
// whole stored row and every word of it has to be meaningful)
static bool ks_narrows(const dctfhe_keys* K, int tier, int deff) {
  return K->tiers[tier].d_kskT != nullptr && ((size_t)deff * (size_t)K->p.tiers[tier].lk) % 64 == 0;
}

static int alloc_lut_scratch(dctfhe_keys* K, size_t chunk, LutScratch* sc) {
  int lkmax = 1, nmax = 1;
  for (int i = 0; i < K->p.n_tiers; i++) { lkmax = std::max(lkmax, K->p.tiers[i].lk); nmax = std::max(nmax, K->p.tiers[i].n); }
  sc->chunk = chunk;
  HIPCHK(hipMalloc(&sc->digits, chunk * (size_t)K->p.D * lkmax));
  HIPCHK(hipMalloc(&sc->bodies, chunk * 8));
  HIPCHK(hipMalloc(&sc->small, chunk * (size_t)(nmax + 1) * 8));
  HIPCHK(hipMalloc(&sc->bit_tables, 64 * 8));
  int64_t bt[64];
  for (int j = 0; j < 64; j++) bt[j] = (int64_t)(1ULL << j);
  HIPCHK(hipMemcpy(sc->bit_tables, bt, sizeof bt, hipMemcpyHostToDevice));
  return 0;
}
static void free_lut_scratch(LutScratch* sc) {
  hipFree(sc->digits); hipFree(sc->bodies); hipFree(sc->small); hipFree(sc->bit_tables);
  sc->digits = nullptr; sc->bodies = nullptr; sc->small = nullptr; sc->bit_tables = nullptr;
}
struct LutScratchOwner { LutScratch s{}; ~LutScratchOwner() { free_lut_scratch(&s); } };

// ------------------------------------------------------------------------------------------ primitives on host buffers
extern "C" int dctfhe_keyswitch_prefix(dctfhe_ctx* ctx, dctfhe_eval_keys* K, int tier, const uint64_t* cts, size_t count, int shift, int deff,
                                       uint64_t* cts_small) {
  if (!ctx || !K) return fail("dctfhe_keyswitch: null handle");
  if (tier < 0 || tier >= K->p.n_tiers) return fail("tier out of range");
  if (deff < 0 || deff > K->p.D) return fail("deff out of range");
  if (shift < 0 || shift > 63) return fail("shift out of range");
  if (count == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  const dctfhe_tier& t = K->p.tiers[tier];
  const size_t L = (size_t)K->p.D + 1;
  DevBuf d_in, d_small, d_bodies, d_dig;
  HIPCHK(d_in.alloc(count * L * 8));
  HIPCHK(d_small.alloc(count * (size_t)(t.n + 1) * 8));
  HIPCHK(d_bodies.alloc(count * 8));
  HIPCHK(d_dig.alloc(count * (size_t)K->p.D * t.lk));
  HIPCHK(hipMemcpy(d_in.p, cts, count * L * 8, hipMemcpyHostToDevice));
  CHK(dev_keyswitch(K, tier, d_in.as<uint64_t>(), count, shift, d_dig.as<uint8_t>(), d_bodies.as<uint64_t>(), d_small.as<uint64_t>(), nullptr, deff));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(cts_small, d_small.p, count * (size_t)(t.n + 1) * 8, hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int dctfhe_keyswitch(dctfhe_ctx* ctx, dctfhe_eval_keys* K, int tier, const uint64_t* cts, size_t count, int shift, uint64_t* cts_small) {
  return dctfhe_keyswitch_prefix(ctx, K, tier, cts, count, shift, 0, cts_small);
}

// the centred mod switch on host buffers (the scheduler applies it between every key switch and its bootstrap)
extern "C" int dctfhe_modswitch_center(dctfhe_ctx* ctx, dctfhe_eval_keys* K, int tier, uint64_t* cts_small, size_t count) {
  if (!ctx || !K || (count && !cts_small)) return fail("dctfhe_modswitch_center: null argument");
  if (tier < 0 || tier >= K->p.n_tiers) return fail("tier out of range");
  if (count == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t bytes = count * (size_t)(K->p.tiers[tier].n + 1) * 8;
  DevBuf d;
  HIPCHK(d.alloc(bytes));
  HIPCHK(hipMemcpy(d.p, cts_small, bytes, hipMemcpyHostToDevice));
  CHK(dev_ms_center(K, tier, d.as<uint64_t>(), count, nullptr));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(cts_small, d.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int dctfhe_pbs(dctfhe_ctx* ctx, dctfhe_eval_keys* K, int tier, const uint64_t* cts_small, size_t count, const int64_t* tables, int ntab,
                          int w, const int32_t* table_idx, uint64_t* cts_out) {
  if (!ctx || !K) return fail("dctfhe_pbs: null handle");
  if (tier < 0 || tier >= K->p.n_tiers) return fail("tier out of range");
  if (ntab < 1 || w < 0) return fail("dctfhe_pbs: need at least one table and w >= 0");
  if (count == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  const dctfhe_tier& t = K->p.tiers[tier];
  if (w > t.logN - 1) return fail("table of 2^%d entries does not fit N = 2^%d", w, t.logN);
  if (table_idx)
    for (size_t i = 0; i < count; i++)
      if (table_idx[i] < 0 || table_idx[i] >= ntab) return fail("table_idx[%zu] = %d out of range (%d tables)", i, table_idx[i], ntab);
  const size_t L = (size_t)K->p.D + 1;
  DevBuf d_small, d_out, d_tab, d_idx;
  HIPCHK(d_small.alloc(count * (size_t)(t.n + 1) * 8));
  HIPCHK(d_out.alloc(count * L * 8));
  HIPCHK(d_tab.alloc(((size_t)ntab << w) * 8));
  HIPCHK(hipMemcpy(d_small.p, cts_small, count * (size_t)(t.n + 1) * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_tab.p, tables, ((size_t)ntab << w) * 8, hipMemcpyHostToDevice));
  if (table_idx) {
    HIPCHK(d_idx.alloc(count * 4));
    HIPCHK(hipMemcpy(d_idx.p, table_idx, count * 4, hipMemcpyHostToDevice));
  }
  CHK(dev_pbs(K, tier, d_small.as<uint64_t>(), count, d_tab.as<int64_t>(), w, d_idx.as<int32_t>(), 1, 1, 0, d_out.as<uint64_t>(), 0, 0, nullptr));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(cts_out, d_out.p, count * L * 8, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int dctfhe_round_lut(dctfhe_ctx* ctx, dctfhe_eval_keys* K, int bit_tier, int tab_tier, const uint64_t* cts, size_t count, int p, int r,
                                const int64_t* tables, int ntab, int w, const int32_t* table_idx, uint64_t* cts_out) {
  if (!ctx || !K) return fail("dctfhe_round_lut: null handle");
  if (tab_tier < 0 || tab_tier >= K->p.n_tiers || (r > 0 && (bit_tier < 0 || bit_tier >= K->p.n_tiers))) return fail("tier out of range");
  if (p < 1 || p > 62 || r < 0 || r >= p) return fail("need 1 <= p <= 62 and 0 <= r < p");
  if (w != p - r) return fail("w must equal p - r");
  if (w > K->p.tiers[tab_tier].logN - 1) return fail("table of 2^%d entries does not fit tier %d", w, tab_tier);
  if (ntab < 1) return fail("need at least one table");
  if (count == 0) return 0;
  if (table_idx)
    for (size_t i = 0; i < count; i++)
      if (table_idx[i] < 0 || table_idx[i] >= ntab) return fail("table_idx[%zu] = %d out of range (%d tables)", i, table_idx[i], ntab);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t L = (size_t)K->p.D + 1;
  DevBuf d_work, d_tab, d_idx;
  HIPCHK(d_work.alloc(count * L * 8));
  HIPCHK(d_tab.alloc(((size_t)ntab << w) * 8));
  HIPCHK(hipMemcpy(d_work.p, cts, count * L * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_tab.p, tables, ((size_t)ntab << w) * 8, hipMemcpyHostToDevice));
  if (table_idx) {
    HIPCHK(d_idx.alloc(count * 4));
    HIPCHK(hipMemcpy(d_idx.p, table_idx, count * 4, hipMemcpyHostToDevice));
  }
  LutScratchOwner sc;
  CHK(alloc_lut_scratch(K, std::min<size_t>(count, 4096), &sc.s));
  if (r > 0) {
    hipLaunchKernelGGL(k_affine, dim3(ew_grid(count * L)), dim3(256), 0, ctx->stream, d_work.as<uint64_t>(), L, L - 1, d_work.as<uint64_t>(), L, count, L - 1, 0,
                       1ULL << (63 - p + r - 1));
    HIPCHK(hipGetLastError());
  }
  StepTiers stp;
  stp.bit = bit_tier;
  CHK(dev_round_lut(K, stp, tab_tier, d_work.as<uint64_t>(), L, 0, 0, d_work.as<uint64_t>(), L, count, p, r, d_tab.as<int64_t>(), w,
                    d_idx.as<int32_t>(), 1, 1, sc.s, nullptr));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(cts_out, d_work.p, count * L * 8, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int dctfhe_conv2d(dctfhe_ctx* ctx, int D, const uint64_t* in, int batch, int Cin, int H, int W, const int8_t* weight, int Cout,
                             int KH, int KW, int stride, int pad, uint64_t* out) {
  if (!ctx) return fail("dctfhe_conv2d: null context");
  if (D < 0 || batch < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || KH < 1 || KW < 1 || stride < 1 || pad < 0) return fail("dctfhe_conv2d: bad geometry");
  if (H + 2 * pad < KH || W + 2 * pad < KW) return fail("dctfhe_conv2d: kernel larger than the padded input");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t L = (size_t)D + 1;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const size_t nin = (size_t)batch * Cin * H * W * L, nout = (size_t)batch * Cout * Ho * Wo * L, nw = (size_t)Cout * Cin * KH * KW;
  DevBuf d_in, d_out, d_w;
  HIPCHK(d_in.alloc(nin * 8));
  HIPCHK(d_out.alloc(nout * 8));
  HIPCHK(d_w.alloc(nw));
  HIPCHK(hipMemcpy(d_in.p, in, nin * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_w.p, weight, nw, hipMemcpyHostToDevice));
  ConvSlab slab;
  std::vector<ConvPack> pk;
  CHK(build_conv_packs({weight}, {{Cout, Cin, H, W, KH, KW, pad}}, &slab, &pk));
  CHK(dev_conv2d(ctx->stream, d_in.as<uint64_t>(), batch, Cin, H, W, L, L - 1, d_w.as<int8_t>(), &pk[0], Cout, KH, KW, stride, pad, d_out.as<uint64_t>(), L));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, d_out.p, nout * 8, hipMemcpyDeviceToHost));
  return 0;
}

// ---- K2 (residual add, rounding offset / shift, window-sum pooling) one kernel at a time on host buffers.  Rows of `dim` mask words +
// body of which the first `deff` may be non-zero -- the storage form of the session's tensors, so that the kernels are checked at mixed
// effective dimensions outside a circuit (reference backbone.py:102 torch.add, :276 AvgPool2d).
static int rows_ok(const char* what, int dim, int deff) {
  if (dim < 0 || deff < 0 || deff > dim || dim > (1 << 16)) return fail("%s: rows of %d mask words with effective dimension %d", what, dim, deff);
  return 0;
}
extern "C" int dctfhe_add_rows(dctfhe_ctx* ctx, const uint64_t* a, int dim_a, int deff_a, const uint64_t* b, int dim_b, int deff_b, size_t count, int dim_o,
                               uint64_t* out) {
  if (!ctx || !a || !b || !out) return fail("dctfhe_add_rows: null argument");
  CHK(rows_ok("dctfhe_add_rows (a)", dim_a, deff_a));
  CHK(rows_ok("dctfhe_add_rows (b)", dim_b, deff_b));
  if (dim_o < std::max(deff_a, deff_b) || dim_o > (1 << 16)) return fail("dctfhe_add_rows: output rows of %d mask words cannot hold the sum (%d needed)", dim_o, std::max(deff_a, deff_b));
  if (count == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t La = (size_t)dim_a + 1, Lb = (size_t)dim_b + 1, Lo = (size_t)dim_o + 1;
  DevBuf da, db, d_o;
  HIPCHK(da.alloc(count * La * 8)); HIPCHK(db.alloc(count * Lb * 8)); HIPCHK(d_o.alloc(count * Lo * 8));
  HIPCHK(hipMemcpy(da.p, a, count * La * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db.p, b, count * Lb * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_add, dim3(ew_grid(count * Lo)), dim3(256), 0, ctx->stream, da.as<uint64_t>(), La, (size_t)deff_a, db.as<uint64_t>(), Lb, (size_t)deff_b,
                     d_o.as<uint64_t>(), Lo, count);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, d_o.p, count * Lo * 8, hipMemcpyDeviceToHost));
  return 0;
}
// inout: rows of dim_o + 1 words; the kernel writes the first nwords mask words (a << shift; zero from deff_a on) and the body
// ((body << shift) + body_add) and leaves the words in between as they were (what the rounding chain relies on: dctfhe_session_run)
extern "C" int dctfhe_affine_rows(dctfhe_ctx* ctx, const uint64_t* a, int dim_a, int deff_a, size_t count, int nwords, int shift, uint64_t body_add, int dim_o,
                                  uint64_t* inout) {
  if (!ctx || !a || !inout) return fail("dctfhe_affine_rows: null argument");
  CHK(rows_ok("dctfhe_affine_rows", dim_a, deff_a));
  if (dim_o < 0 || dim_o > (1 << 16) || nwords < 0 || nwords > dim_o || shift < 0 || shift > 63) return fail("dctfhe_affine_rows: bad output geometry (nwords <= dim_o, 0 <= shift <= 63)");
  if (count == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t La = (size_t)dim_a + 1, Lo = (size_t)dim_o + 1;
  DevBuf da, d_o;
  HIPCHK(da.alloc(count * La * 8)); HIPCHK(d_o.alloc(count * Lo * 8));
  HIPCHK(hipMemcpy(da.p, a, count * La * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_o.p, inout, count * Lo * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_affine, dim3(ew_grid(count * ((size_t)nwords + 1))), dim3(256), 0, ctx->stream, da.as<uint64_t>(), La, (size_t)deff_a, d_o.as<uint64_t>(), Lo, count,
                     (size_t)nwords, shift, body_add);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(inout, d_o.p, count * Lo * 8, hipMemcpyDeviceToHost));
  return 0;
}
// in [batch][C][H][W] rows -> out [batch][C][H/K][W/K] rows: window sums with floor semantics (border rows / columns dropped)
extern "C" int dctfhe_sum_pool_rows(dctfhe_ctx* ctx, const uint64_t* in, int dim_in, int deff_in, int batch, int C, int H, int W, int K, int dim_o, uint64_t* out) {
  if (!ctx || !in || !out) return fail("dctfhe_sum_pool_rows: null argument");
  CHK(rows_ok("dctfhe_sum_pool_rows", dim_in, deff_in));
  if (batch < 1 || C < 1 || H < 1 || W < 1 || K < 1 || H / K < 1 || W / K < 1) return fail("dctfhe_sum_pool_rows: bad geometry");
  if (dim_o < deff_in || dim_o > (1 << 16)) return fail("dctfhe_sum_pool_rows: output rows of %d mask words cannot hold the sums (%d needed)", dim_o, deff_in);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t Li = (size_t)dim_in + 1, Lo = (size_t)dim_o + 1;
  const int Ho = H / K, Wo = W / K;
  const size_t nin = (size_t)batch * C * H * W * Li, nout = (size_t)batch * C * Ho * Wo * Lo;
  DevBuf di, d_o;
  HIPCHK(di.alloc(nin * 8)); HIPCHK(d_o.alloc(nout * 8));
  HIPCHK(hipMemcpy(di.p, in, nin * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sum_pool, dim3(ew_grid(nout)), dim3(256), 0, ctx->stream, di.as<uint64_t>(), C, H, W, Li, (size_t)deff_in, K, Ho, Wo, d_o.as<uint64_t>(), Lo, nout);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, d_o.p, nout * 8, hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------------------ K10 front-end
extern "C" int dctfhe_dct_frontend(dctfhe_ctx* ctx, const uint8_t* y, const uint8_t* c1, const uint8_t* c2, int batch, int S, int Sc, int fs,
                                   const int32_t* idx_y, int ny, const int32_t* idx_c1, int n1, const int32_t* idx_c2, int n2,
                                   const float* mean, const float* stdv, int round_coeffs, float* out) {
  if (!ctx || !y || !c1 || !c2 || !mean || !stdv || !out) return fail("dctfhe_dct_frontend: null argument");
  if (batch < 1 || S < 1 || (Sc != S && 2 * Sc != S) || (fs != 4 && fs != 8)) return fail("dctfhe_dct_frontend: bad geometry (block size 4 or 8; chroma grid S or S/2)");
  if (ny < 0 || n1 < 0 || n2 < 0 || ny + n1 + n2 < 1) return fail("dctfhe_dct_frontend: no coefficients selected");
  const int32_t* idx[3] = {idx_y, idx_c1, idx_c2};
  const int n[3] = {ny, n1, n2};
  for (int p = 0; p < 3; p++)
    for (int i = 0; i < n[p]; i++)
      if (!idx[p] || idx[p][i] < 0 || idx[p][i] >= fs * fs) return fail("dctfhe_dct_frontend: coefficient index out of range");
  HIPCHK(hipSetDevice(ctx->device));
  const int C = ny + n1 + n2;
  const size_t by = (size_t)batch * S * fs * S * fs, bc = (size_t)batch * Sc * fs * Sc * fs, nout = (size_t)batch * C * S * S;
  DevBuf d_y, d_c1, d_c2, d_idx, d_ms, d_out;
  HIPCHK(d_y.alloc(by)); HIPCHK(d_c1.alloc(bc)); HIPCHK(d_c2.alloc(bc));
  HIPCHK(d_idx.alloc((size_t)C * 4)); HIPCHK(d_ms.alloc((size_t)C * 8)); HIPCHK(d_out.alloc(nout * 4));
  HIPCHK(hipMemcpy(d_y.p, y, by, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_c1.p, c1, bc, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_c2.p, c2, bc, hipMemcpyHostToDevice));
  dct_args a;
  a.plane[0] = d_y.as<uint8_t>(); a.plane[1] = d_c1.as<uint8_t>(); a.plane[2] = d_c2.as<uint8_t>();
  a.S = S; a.Sc = Sc; a.fs = fs; a.round_coeffs = round_coeffs; a.batch = batch;
  int off = 0;
  for (int p = 0; p < 3; p++) {
    a.n[p] = n[p];
    a.idx[p] = d_idx.as<int32_t>() + off;
    if (n[p]) HIPCHK(hipMemcpy(d_idx.as<int32_t>() + off, idx[p], (size_t)n[p] * 4, hipMemcpyHostToDevice));
    off += n[p];
  }
  HIPCHK(hipMemcpy(d_ms.p, mean, (size_t)C * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_ms.as<float>() + C, stdv, (size_t)C * 4, hipMemcpyHostToDevice));
  a.mean = d_ms.as<float>(); a.stdv = d_ms.as<float>() + C; a.out = d_out.as<float>();
  hipLaunchKernelGGL(k_dct_frontend, dim3(ew_grid(nout)), dim3(256), 0, ctx->stream, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, d_out.p, nout * 4, hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------------------ circuit
struct BlobHeader { uint32_t magic, version; int32_t n_tensors, n_ops, input_tensor, output_tensor, max_bit_width, reserved; };

// parse + validate a circuit blob on the host (no GPU): header, tensor table, op records, payload ranges, per-op shapes
static int parse_circuit(const void* blob, size_t size, dctfhe_circuit* c) {
  if (!blob) return fail("null circuit blob");
  if (size < sizeof(BlobHeader)) return fail("circuit blob too short");
  BlobHeader h;
  memcpy(&h, blob, sizeof h);
  if (h.magic != 0x46544344u /* 'DCTF' */ || h.version != 1) return fail("bad circuit blob magic/version");
  if (h.n_tensors < 1 || h.n_ops < 0 || h.n_tensors > (1 << 20) || h.n_ops > (1 << 20)) return fail("circuit blob: bad tensor/op count");
  const size_t need = sizeof h + (size_t)h.n_tensors * sizeof(TensorShape) + (size_t)h.n_ops * sizeof(Op);
  if (size < need) return fail("circuit blob truncated");
  auto bad_t = [&](int t) { return t < 0 || t >= h.n_tensors; };
  if (bad_t(h.input_tensor) || bad_t(h.output_tensor)) return fail("circuit blob: input/output tensor id out of range");
  c->tensors.resize(h.n_tensors);
  c->ops.resize(h.n_ops);
  const char* p = (const char*)blob + sizeof h;
  memcpy(c->tensors.data(), p, (size_t)h.n_tensors * sizeof(TensorShape));
  p += (size_t)h.n_tensors * sizeof(TensorShape);
  memcpy(c->ops.data(), p, (size_t)h.n_ops * sizeof(Op));
  c->input_tensor = h.input_tensor; c->output_tensor = h.output_tensor; c->max_bit_width = h.max_bit_width;
  for (const TensorShape& t : c->tensors)
    if (t.C < 1 || t.H < 1 || t.W < 1) return fail("circuit blob: empty tensor shape");
  for (int i = 0; i < h.n_ops; i++) {
    const Op& o = c->ops[i];
    if (o.type < OP_CONV || o.type > OP_LUT) return fail("op %d: unknown type %d", i, o.type);
    if (bad_t(o.src0) || bad_t(o.dst) || (o.type == OP_ADD && bad_t(o.src1))) return fail("op %d: tensor id out of range", i);
    if (o.payload_len < 0 || (o.payload_len > 0 && (o.payload_off < (int64_t)need || (size_t)o.payload_off + (size_t)o.payload_len > size)))
      return fail("op %d: payload out of range", i);
    const TensorShape& a = c->tensors[o.src0];
    const TensorShape& d = c->tensors[o.dst];
    switch (o.type) {
      case OP_CONV: {
        const int Cout = o.ip[0], KH = o.ip[1], KW = o.ip[2], st = o.ip[3], pad = o.ip[4];
        if (Cout < 1 || KH < 1 || KW < 1 || st < 1 || pad < 0 || a.H + 2 * pad < KH || a.W + 2 * pad < KW) return fail("op %d: bad convolution geometry", i);
        if (d.C != Cout || d.H != (a.H + 2 * pad - KH) / st + 1 || d.W != (a.W + 2 * pad - KW) / st + 1) return fail("op %d: convolution output shape mismatch", i);
        if (o.payload_len != (int64_t)Cout * a.C * KH * KW) return fail("op %d: weight payload of %lld bytes, expected %lld", i, (long long)o.payload_len, (long long)Cout * a.C * KH * KW);
        break;
      }
      case OP_ADD: {
        const TensorShape& b2 = c->tensors[o.src1];
        if (a.C != b2.C || a.H != b2.H || a.W != b2.W || a.C != d.C || a.H != d.H || a.W != d.W) return fail("op %d: add operands differ in shape", i);
        break;
      }
      case OP_SUMPOOL: {
        const int K = o.ip[0];
        if (K < 1 || d.C != a.C || d.H != a.H / K || d.W != a.W / K || d.H < 1 || d.W < 1) return fail("op %d: bad pooling geometry", i);
        break;
      }
      case OP_LUT: {
        const int pp = o.ip[0], r = o.ip[1], w = o.ip[2], shift = o.ip[3], ntab = o.ip[6];
        if (pp < 1 || pp > 62 || r < 0 || r >= pp || w != pp - r || shift < 0 || shift > 63) return fail("op %d: bad look-up precision (p=%d r=%d w=%d shift=%d)", i, pp, r, w, shift);
        if (ntab != 1 && ntab != a.C) return fail("op %d: %d tables for %d channels", i, ntab, a.C);
        if (o.payload_len != ((int64_t)ntab << w) * 8) return fail("op %d: table payload of %lld bytes, expected %lld", i, (long long)o.payload_len, (long long)(((int64_t)ntab << w) * 8));
        if (a.C != d.C || a.H != d.H || a.W != d.W) return fail("op %d: look-up changes the shape", i);
        break;
      }
    }
  }
  return 0;
}

extern "C" int dctfhe_circuit_validate(const void* blob, size_t size) {
  dctfhe_circuit c;
  return parse_circuit(blob, size, &c);
}

extern "C" int dctfhe_circuit_load(dctfhe_ctx* ctx, const void* blob, size_t size, dctfhe_circuit** out) {
  if (!ctx || !out) return fail("dctfhe_circuit_load: null argument");
  std::unique_ptr<dctfhe_circuit> c(new dctfhe_circuit);
  CHK(parse_circuit(blob, size, c.get()));
  HIPCHK(hipSetDevice(ctx->device));
  c->ctx = ctx;
  c->d_payload.assign(c->ops.size(), nullptr);
  std::vector<const int8_t*> cw(c->ops.size(), nullptr);
  std::vector<std::array<int, 7>> cg(c->ops.size());
  for (size_t i = 0; i < c->ops.size(); i++) {
    const Op& o = c->ops[i];
    if (o.payload_len > 0) {
      HIPCHK(hipMalloc(&c->d_payload[i], (size_t)o.payload_len));
      HIPCHK(hipMemcpy(c->d_payload[i], (const char*)blob + o.payload_off, (size_t)o.payload_len, hipMemcpyHostToDevice));
    }
    if (o.type == OP_CONV) {
      const TensorShape& a = c->tensors[o.src0];
      cw[i] = (const int8_t*)blob + o.payload_off;
      cg[i] = {o.ip[0], a.C, a.H, a.W, o.ip[1], o.ip[2], o.ip[4]};
    }
  }
  c->conv.assign(c->ops.size(), ConvPack{});
  c->conv_w.resize(c->ops.size());
  for (size_t i = 0; i < c->ops.size(); i++)
    if (cw[i]) c->conv_w[i].assign(cw[i], cw[i] + c->ops[i].payload_len);
  c->conv_geom = cg;
  *out = c.release();
  return 0;
}
extern "C" int dctfhe_circuit_destroy(dctfhe_circuit* c) { delete c; return 0; }
extern "C" int dctfhe_circuit_io(dctfhe_circuit* c, int64_t* n_in, int64_t* n_out) {
  const TensorShape& a = c->tensors[c->input_tensor];
  const TensorShape& b = c->tensors[c->output_tensor];
  *n_in = (int64_t)a.C * a.H * a.W;
  *n_out = (int64_t)b.C * b.H * b.W;
  return 0;
}

extern "C" int dctfhe_circuit_stats(dctfhe_circuit* c, const dctfhe_params* P, dctfhe_stats* s) {
  memset(s, 0, sizeof *s);
  s->max_bit_width = c->max_bit_width;
  s->n_ops = (int)c->ops.size();
  const double Lb = (P->D + 1) * 8.0;
  auto elems = [&](int t) { const TensorShape& x = c->tensors[t]; return (double)x.C * x.H * x.W; };
  auto tier_flops = [&](const dctfhe_tier& t) {
    const double N = (double)(1 << t.logN), M = N / 2;
    const double fft = 5.0 * M * std::log2(M);
    if (t.unroll == 2)   // per PAIR of key bits: the same transforms, three key blocks folded with their monomials (22 + 8 flops per
                         // point and key polynomial), the monomials themselves (18 per point)
      return (t.n / 2) * ((t.k + 1) * t.l * fft + (t.k + 1) * fft + (double)(t.k + 1) * (t.k + 1) * t.l * M * 30.0 + M * 18.0);
    return t.n * ((t.k + 1) * t.l * fft + (t.k + 1) * fft + (double)(t.k + 1) * (t.k + 1) * t.l * M * 8.0);
  };
  auto key_bytes = [&](const dctfhe_tier& t) {
    const double N = (double)(1 << t.logN);
    return (double)(t.unroll == 2 ? 3 * t.n / 2 : t.n) * t.l * (t.k + 1) * (t.k + 1) * N * 8.0 + (double)P->D * t.lk * (t.n + 1) * 8.0;
  };
  for (const Op& o : c->ops) {
    const double ein = elems(o.src0), eout = elems(o.dst);
    switch (o.type) {
      case OP_CONV: {
        const TensorShape& a = c->tensors[o.src0];
        s->conv_macs += (int64_t)(eout * a.C * o.ip[1] * o.ip[2]);
        s->bytes_algorithmic += (ein + eout) * Lb;
        break;
      }
      case OP_ADD: s->bytes_algorithmic += 3 * eout * Lb; break;
      case OP_SUMPOOL: s->bytes_algorithmic += (ein + eout) * Lb; break;
      case OP_LUT: {
        const int r = o.ip[9] ? 0 : o.ip[1], tt = o.ip[4], bt = o.ip[5];     // approximate rounding: no one-bit steps
        s->lut_sites += (int64_t)ein;
        s->bit_steps += (int64_t)(ein * r);
        s->bytes_algorithmic += 2 * ein * Lb * (1 + r);
        if (tt >= 0 && tt < P->n_tiers) {
          s->pbs_count[tt] += (int64_t)ein; s->ks_count[tt] += (int64_t)ein;
          s->flops_f64 += ein * tier_flops(P->tiers[tt]);
          s->key_bytes_per_pass += key_bytes(P->tiers[tt]);
        }
        const StepTiers stp = step_tiers_of(o);
        for (int st = 0; st < r; st++) {
          const int b2 = stp.at(st);
          if (b2 < 0 || b2 >= P->n_tiers) continue;
          s->pbs_count[b2] += (int64_t)ein; s->ks_count[b2] += (int64_t)ein;
          s->flops_f64 += ein * tier_flops(P->tiers[b2]);
          s->key_bytes_per_pass += key_bytes(P->tiers[b2]);
        }
        break;
      }
      default: break;
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------ session
extern "C" int dctfhe_session_create(dctfhe_ctx* ctx, dctfhe_circuit* circ, dctfhe_keys* keys, int batch, dctfhe_session** out) {
  if (batch < 1) return fail("batch must be >= 1");
  HIPCHK(hipSetDevice(ctx->device));
  std::unique_ptr<dctfhe_session> s(new dctfhe_session);
  s->ctx = ctx; s->circ = circ; s->keys = keys; s->batch = batch;
  s->D = keys ? keys->p.D : 0;
  // validate tiers named by the circuit
  if (keys)
    for (size_t i = 0; i < circ->ops.size(); i++) {
      const Op& o = circ->ops[i];
      if (o.type != OP_LUT) continue;
      const int tt = o.ip[4], bt = o.ip[5], r = o.ip[9] ? 0 : o.ip[1], w = o.ip[2];
      if (tt < 0 || tt >= keys->p.n_tiers || (r > 0 && (bt < 0 || bt >= keys->p.n_tiers))) return fail("op %zu names a tier the keys lack", i);
      if (w > keys->p.tiers[tt].logN - 1) return fail("op %zu: table of 2^%d entries does not fit tier %d", i, w, tt);
      const StepTiers stp = step_tiers_of(o);
      for (int st = 0; st < r; st++)
        if (stp.at(st) < 0 || stp.at(st) >= keys->p.n_tiers) return fail("op %zu names a coarse bit tier the keys lack", i);
    }
  if (keys && !circ->conv_slab.d) {
    // the matrix-core form of the convolution weights, once per circuit and only for encrypted evaluation (clear mode runs the VALU
    // kernel), in one allocation
    std::vector<const int8_t*> cw(circ->ops.size(), nullptr);
    for (size_t i = 0; i < circ->ops.size(); i++)
      if (i < circ->conv_w.size() && !circ->conv_w[i].empty()) cw[i] = circ->conv_w[i].data();
    CHK(build_conv_packs(cw, circ->conv_geom, &circ->conv_slab, &circ->conv));
  }
  // tensor liveness: free a buffer after its last reader; reuse freed buffers of sufficient size
  const int nt = (int)circ->tensors.size();
  std::vector<int> last_use(nt, -1);
  for (int i = 0; i < (int)circ->ops.size(); i++) {
    const Op& o = circ->ops[i];
    last_use[o.src0] = i;
    if (o.type == OP_ADD) last_use[o.src1] = i;
  }
  last_use[circ->output_tensor] = 1 << 30;
  // the input stays resident too: dctfhe_session_run may be called again without a fresh upload (bench.py does)
  last_use[circ->input_tensor] = 1 << 30;
  s->d_tensor.assign(nt, nullptr);
  s->tensor_words.assign(nt, 0);
  // row layout per tensor.  Encrypted: a tensor is stored at its effective dimension -- the compiler's `deff` (ip[10] of the
  // consuming / producing op: mask words beyond it are zero in every ciphertext of the tensor); the output of a look-up is as
  // wide as its table tier's ring, or as the rounding chain that works in place on it.  75 % of a ResNet tensor used to be
  // stored, written and streamed zeros.
  s->t_L.assign(nt, 1);
  s->t_deff.assign(nt, 0);
  if (keys) {
    const size_t D = (size_t)s->D;
    // (the matrix-core key switch walks the key in steps of 64 rows: an effective dimension it cannot take is kept at full width)
    // ... and the integer-VALU key switch (betak = 8, or a key the matrix-core tiling does not cover) walks whole rows of D words: a
    // circuit that uses such a tier keeps every tensor at full width
    bool all_mfma = true;
    for (const Op& o : circ->ops) {
      if (o.type != OP_LUT) continue;
      all_mfma = all_mfma && keys->tiers[o.ip[4]].d_kskT != nullptr;
      const int r = o.ip[9] ? 0 : o.ip[1];
      const StepTiers stp = step_tiers_of(o);
      for (int st_i = 0; st_i < r; st_i++) all_mfma = all_mfma && keys->tiers[stp.at(st_i)].d_kskT != nullptr;
    }
    auto clampd = [&](int d) { return (size_t)((all_mfma && d > 0 && (size_t)d < D && d % 64 == 0) ? d : D); };
    std::vector<char> known(nt, 0);
    auto set = [&](int t, size_t deff, size_t width) { s->t_deff[t] = deff; s->t_L[t] = (all_mfma ? std::max(deff, width) : D) + 1; known[t] = 1; };
    for (const Op& o : circ->ops)
      if (o.src0 == circ->input_tensor) { set(circ->input_tensor, clampd(o.ip[10]), 0); break; }
    if (!known[circ->input_tensor]) set(circ->input_tensor, D, 0);
    for (size_t i = 0; i < circ->ops.size(); i++) {
      const Op& o = circ->ops[i];
      if (!known[o.src0] || (o.type == OP_ADD && !known[o.src1])) return fail("op %zu reads a tensor no earlier op wrote", i);
      if (o.type == OP_LUT) {
        const int r = o.ip[9] ? 0 : o.ip[1], tt = o.ip[4];
        const size_t ring = (size_t)keys->p.tiers[tt].k << keys->p.tiers[tt].logN;
        if (clampd(o.ip[10]) < s->t_deff[o.src0]) return fail("op %zu: look-up compiled for effective dimension %d, its input has %zu", i, o.ip[10], s->t_deff[o.src0]);
        const size_t chain = r > 0 ? (size_t)round_chain_deff(keys, (int)clampd(o.ip[10]), step_tiers_of(o), r) : 0;
        set(o.dst, ring, chain);
      } else if (o.type == OP_ADD) {
        set(o.dst, std::max(s->t_deff[o.src0], s->t_deff[o.src1]), 0);
      } else {
        set(o.dst, s->t_deff[o.src0], 0);
      }
    }
  }
  for (int t = 0; t < nt; t++) {
    const TensorShape& x = circ->tensors[t];
    s->tensor_words[t] = (size_t)batch * x.C * x.H * x.W * s->t_L[t];
  }
  std::vector<std::pair<size_t, uint64_t*>> freelist;
  auto get = [&](size_t words, uint64_t** p) -> int {
    int best = -1;
    for (int i = 0; i < (int)freelist.size(); i++)
      if (freelist[i].first >= words && (best < 0 || freelist[i].first < freelist[best].first)) best = i;
    if (best >= 0) { *p = freelist[best].second; freelist.erase(freelist.begin() + best); return 0; }
    HIPCHK(hipMalloc(p, words * 8));
    s->owned.push_back({words, *p});
    return 0;
  };
  std::map<uint64_t*, size_t> cap;
  auto alloc_t = [&](int t) -> int {
    if (s->d_tensor[t]) return 0;
    uint64_t* p = nullptr;
    CHK(get(s->tensor_words[t], &p));
    s->d_tensor[t] = p;
    if (!cap.count(p)) cap[p] = s->tensor_words[t];
    return 0;
  };
  CHK(alloc_t(circ->input_tensor));
  for (int i = 0; i < (int)circ->ops.size(); i++) {
    const Op& o = circ->ops[i];
    CHK(alloc_t(o.dst));
    auto release = [&](int t) {
      if (last_use[t] == i && t != o.dst && s->d_tensor[t]) freelist.push_back({cap[s->d_tensor[t]], s->d_tensor[t]});
    };
    release(o.src0);
    if (o.type == OP_ADD && o.src1 != o.src0) release(o.src1);
  }
  if (keys) {
    size_t maxe = 1;
    for (const Op& o : circ->ops)
      if (o.type == OP_LUT) { const TensorShape& x = circ->tensors[o.src0]; maxe = std::max(maxe, (size_t)batch * x.C * x.H * x.W); }
    LutScratchOwner sc;
    CHK(alloc_lut_scratch(keys, std::min<size_t>(maxe, 16384), &sc.s));
    s->chunk = sc.s.chunk; s->d_digits = sc.s.digits; s->d_bodies = sc.s.bodies; s->d_small = sc.s.small; s->d_bit_tables = sc.s.bit_tables;
    sc.s = LutScratch{};      // the session owns them now
  }
  HIPCHK(hipMalloc(&s->d_overflow, sizeof(int)));
  HIPCHK(hipMemset(s->d_overflow, 0, sizeof(int)));
  if (keys) {
    dctfhe_stats stt;
    dctfhe_circuit_stats(circ, &keys->p, &stt);
    for (int i = 0; i < DCTFHE_MAX_TIERS; i++) s->pbs_per_image[i] = stt.pbs_count[i];
  }
  *out = s.release();
  return 0;
}

extern "C" int dctfhe_session_destroy(dctfhe_session* s) { delete s; return 0; }

// host rows of `dim` mask words + body (dim = D: the full-width form; smaller: the compact wire form of dctfhe_encrypt_rows)
extern "C" int dctfhe_session_upload_rows(dctfhe_session* s, const uint64_t* cts_in, int dim) {
  if (!s || !cts_in) return fail("dctfhe_session_upload: null argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  const int t = s->circ->input_tensor;
  hipStream_t st = s->ctx->stream;
  if (!s->keys) {      // clear mode: one word per element, stored as given
    HIPCHK(hipMemcpyAsync(s->d_tensor[t], cts_in, s->tensor_words[t] * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  if (dim < 1 || dim > s->D) return fail("dctfhe_session_upload: rows of %d mask words, the key has %d", dim, s->D);
  // the tensor is stored at its effective dimension.  The circuit was compiled for inputs that are zero beyond it (client
  // encryption on a key prefix, dctfhe_params.input_dim): the key switch and the convolutions never look at that tail, so make
  // sure it really is empty before dropping it
  const size_t Lh = (size_t)dim + 1, Ls = s->t_L[t], deff = s->t_deff[t];
  const size_t count = s->tensor_words[t] / Ls;
  DevBuf tmp;
  HIPCHK(tmp.alloc(count * Lh * 8));
  HIPCHK(hipMemcpyAsync(tmp.p, cts_in, count * Lh * 8, hipMemcpyHostToDevice, st));
  if (deff < (size_t)dim) {
    HIPCHK(hipMemsetAsync(s->d_overflow, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_tail_nonzero, dim3(ew_grid(count * ((size_t)dim - deff))), dim3(256), 0, st, tmp.as<uint64_t>(), count, dim, (int)deff, s->d_overflow);
    HIPCHK(hipGetLastError());
    int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, s->d_overflow, sizeof bad, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (bad) return fail("input ciphertexts have non-zero mask words beyond %zu: encrypt them with these parameters (input_dim)", deff);
  }
  hipLaunchKernelGGL(k_restride, dim3(ew_grid(count * Ls)), dim3(256), 0, st, tmp.as<uint64_t>(), Lh, s->d_tensor[t], Ls, count, std::min(deff, (size_t)dim));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
extern "C" int dctfhe_session_upload(dctfhe_session* s, const uint64_t* cts_in) {
  if (!s) return fail("dctfhe_session_upload: null argument");
  return dctfhe_session_upload_rows(s, cts_in, s->D);
}
// mask words a compact row of the session's input / output needs (what upload_rows / download_rows accept as `dim` at least / at most
// usefully); clear-mode sessions report 0
extern "C" int dctfhe_session_dims(dctfhe_session* s, int* in_dim, int* out_dim) {
  if (!s || !in_dim || !out_dim) return fail("dctfhe_session_dims: null argument");
  *in_dim = s->keys ? (int)s->t_deff[s->circ->input_tensor] : 0;
  *out_dim = s->keys ? (int)s->t_deff[s->circ->output_tensor] : 0;
  return 0;
}
extern "C" int dctfhe_session_set_noise(dctfhe_session* s, uint64_t seed, const double* sigma_per_op, int n_ops) {
  if (s->keys) return fail("noise simulation is a clear-mode feature: create the session without keys");
  if (n_ops != 0 && n_ops != (int)s->circ->ops.size()) return fail("expected one sigma per op (%zu), got %d", s->circ->ops.size(), n_ops);
  s->sim_sigma.assign(sigma_per_op, sigma_per_op + n_ops);
  s->sim_seed = seed;
  return 0;
}

extern "C" int dctfhe_session_download_rows(dctfhe_session* s, uint64_t* cts_out, int dim) {
  if (!s || !cts_out) return fail("dctfhe_session_download: null argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  const int t = s->circ->output_tensor;
  hipStream_t st = s->ctx->stream;
  if (!s->keys) {
    HIPCHK(hipMemcpyAsync(cts_out, s->d_tensor[t], s->tensor_words[t] * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
  if (dim < (int)s->t_deff[t] || dim > s->D) return fail("dctfhe_session_download: rows of %d mask words; the output needs %zu and the key has %d", dim, s->t_deff[t], s->D);
  const size_t Lh = (size_t)dim + 1, Ls = s->t_L[t];
  const size_t count = s->tensor_words[t] / Ls;
  DevBuf tmp;      // to host rows of dim + 1 words, zero beyond the effective dimension
  HIPCHK(tmp.alloc(count * Lh * 8));
  hipLaunchKernelGGL(k_restride, dim3(ew_grid(count * Lh)), dim3(256), 0, st, s->d_tensor[t], Ls, tmp.as<uint64_t>(), Lh, count, s->t_deff[t]);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(cts_out, tmp.p, count * Lh * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}
extern "C" int dctfhe_session_download(dctfhe_session* s, uint64_t* cts_out) {
  if (!s) return fail("dctfhe_session_download: null argument");
  return dctfhe_session_download_rows(s, cts_out, s->D);
}

extern "C" int dctfhe_session_run(dctfhe_session* s, dctfhe_timing* timing) {
  HIPCHK(hipSetDevice(s->ctx->device));
  hipStream_t st = s->ctx->stream;
  dctfhe_circuit* c = s->circ;
  dctfhe_keys* K = s->keys;
  const int B = s->batch;
  Timers tm{st, timing != nullptr, &s->ev_pool, {}};
  const hipEvent_t e0 = tm.take(), e1 = tm.take();
  HIPCHK(hipEventRecord(e0, st));
  if (timing) memset(timing, 0, sizeof *timing);
  LutScratch sc{s->d_digits, s->d_bodies, s->d_small, s->d_bit_tables, s->chunk};
  for (size_t i = 0; i < c->ops.size(); i++) {
    const Op& o = c->ops[i];
    const TensorShape& a = c->tensors[o.src0];
    const TensorShape& d = c->tensors[o.dst];
    uint64_t* src = s->d_tensor[o.src0];
    uint64_t* dst = s->d_tensor[o.dst];
    const size_t Ls = s->t_L[o.src0], Ld = s->t_L[o.dst], ds = s->t_deff[o.src0];
    switch (o.type) {
      case OP_CONV: {
        const int h = tm.begin(CAT_LINEAR);
        CHK(dev_conv2d(st, src, B, a.C, a.H, a.W, Ls, ds, (const int8_t*)c->d_payload[i], &c->conv[i], o.ip[0], o.ip[1], o.ip[2], o.ip[3], o.ip[4], dst, Ld));
        tm.end(h);
        break;
      }
      case OP_ADD: {
        const int h = tm.begin(CAT_LINEAR);
        const size_t count = s->tensor_words[o.dst] / Ld;
        hipLaunchKernelGGL(k_add, dim3(ew_grid(s->tensor_words[o.dst])), dim3(256), 0, st, src, Ls, ds, s->d_tensor[o.src1], s->t_L[o.src1], s->t_deff[o.src1], dst, Ld,
                           count);
        HIPCHK(hipGetLastError());
        tm.end(h);
        break;
      }
      case OP_SUMPOOL: {
        const int h = tm.begin(CAT_LINEAR);
        const size_t tw = s->tensor_words[o.dst];
        hipLaunchKernelGGL(k_sum_pool, dim3(ew_grid(tw)), dim3(256), 0, st, src, a.C, a.H, a.W, Ls, ds, o.ip[0], d.H, d.W, dst, Ld, tw);
        HIPCHK(hipGetLastError());
        tm.end(h);
        break;
      }
      case OP_LUT: {
        const int p = o.ip[0], r = o.ip[1], w = o.ip[2], shift = o.ip[3], tt = o.ip[4], bt = o.ip[5], nchan = o.ip[6];
        const size_t E = (size_t)B * a.C * a.H * a.W;
        const uint64_t body_add = (uint64_t)o.lp[0];
        const int hw = a.H * a.W;
        if (!K) {
          const double sg = i < s->sim_sigma.size() ? s->sim_sigma[i] : 0.0;
          hipLaunchKernelGGL(k_lut_clear, dim3(ew_grid(E)), dim3(256), 0, st, src, dst, E, shift, body_add, p, r, w, (const int64_t*)c->d_payload[i],
                             hw, nchan, s->d_overflow, sg, rng_key{{(uint32_t)s->sim_seed, (uint32_t)(s->sim_seed >> 32), 0x73696d75u, 0, 0, 0, 0, 0}},
                             (uint64_t)(0x51D0000 + (s->sim_run << 12) + i), (int)(o.ip[9] != 0));
          HIPCHK(hipGetLastError());
        } else {
          // exact rounding: + half of what is removed, then r one-bit steps clear the low bits, in place on a shifted copy of the
          // input.  Approximate rounding (ip[9], the reference README's {"method": "approximate"}): no steps -- the low bits stay
          // and the half-box rotation of the test vector does the rounding; + half an input unit puts the two inputs next to a
          // rounding boundary at equal distance from it.  Without steps nothing is copied: the key switch reads the input tensor.
          const bool approx = o.ip[9] != 0 && r > 0;
          const int steps = approx ? 0 : r;
          const uint64_t add = body_add + (approx ? (1ULL << (62 - p)) : (r > 0 ? (1ULL << (63 - p + r - 1)) : 0));
          int deff = (int)ds;
          if (steps > 0) {
            const StepTiers stp = step_tiers_of(o);
            deff = round_chain_deff(K, (int)ds, stp, steps);
            // the work row: the first deff mask words and the body.  What lies between deff and the end of the row is left alone when
            // every key switch of the chain narrows to deff (the table bootstrap that ends the site rewrites the row); a key switch
            // that cannot narrow reads the whole row, so those words are zeroed (they used to be whatever the recycled
            // buffer held)
            bool narrow = ks_narrows(K, tt, deff);
            for (int st_i = 0; st_i < steps; st_i++) narrow = narrow && ks_narrows(K, stp.at(st_i), deff);
            const size_t fill = narrow ? (size_t)deff : Ld - 1;
            const int h = tm.begin(CAT_LINEAR);
            hipLaunchKernelGGL(k_affine, dim3(ew_grid(E * (fill + 1))), dim3(256), 0, st, src, Ls, ds, dst, Ld, E, fill, shift, add);
            HIPCHK(hipGetLastError());
            tm.end(h);
          }
          CHK(dev_round_lut(K, step_tiers_of(o), tt, src, Ls, shift, add, dst, Ld, E, p, steps, (const int64_t*)c->d_payload[i], w, nullptr, hw, nchan, sc, &tm, deff));
        }
        break;
      }
      default: return fail("op %zu: unknown type %d", i, o.type);
    }
  }
  s->sim_run++;
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipEventSynchronize(e1));
  if (timing) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    timing->total_ms = ms;
    for (auto& sp : tm.spans) {
      float t = 0;
      hipEventElapsedTime(&t, sp.a, sp.b);
      if (sp.cat == CAT_LINEAR) timing->linear_ms += t;
      else if (sp.cat == CAT_KS) timing->ks_ms += t;
      else if (sp.cat >= 0 && sp.cat < DCTFHE_MAX_TIERS) { timing->pbs_ms[sp.cat] += t; timing->pbs_launches[sp.cat]++; }
    }
    for (int i = 0; i < DCTFHE_MAX_TIERS; i++) timing->pbs_cts[i] = s->pbs_per_image[i] * B;
  }
  if (!K) {
    int ov = 0;
    HIPCHK(hipMemcpy(&ov, s->d_overflow, sizeof ov, hipMemcpyDeviceToHost));
    if (ov) { HIPCHK(hipMemset(s->d_overflow, 0, sizeof(int))); return fail("clear run: a message left its padded range (calibration too tight)"); }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------ probes
extern "C" int dctfhe_fp64_peak(dctfhe_ctx* ctx, double* tflops) {
  HIPCHK(hipSetDevice(ctx->device));
  const int blocks = ctx->prop.multiProcessorCount * 8, threads = 256, iters = 1 << 14;
  double* d;
  HIPCHK(hipMalloc(&d, (size_t)blocks * threads * 8));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_fp64_peak, dim3(blocks), dim3(threads), 0, ctx->stream, d, iters);
  hipEventRecord(a, ctx->stream);
  for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_fp64_peak, dim3(blocks), dim3(threads), 0, ctx->stream, d, iters);
  hipEventRecord(b, ctx->stream);
  HIPCHK(hipEventSynchronize(b));
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  *tflops = 4.0 * blocks * threads * (double)iters * 8 * 2 / (ms * 1e-3) / 1e12;
  hipEventDestroy(a); hipEventDestroy(b);
  hipFree(d);
  return 0;
}

extern "C" int dctfhe_bench_pbs(dctfhe_ctx* ctx, dctfhe_keys* K, int tier, size_t count, int reps, double* ms_per_launch) {
  if (tier < 0 || tier >= K->p.n_tiers) return fail("tier out of range");
  HIPCHK(hipSetDevice(ctx->device));
  const dctfhe_tier& t = K->p.tiers[tier];
  const size_t L = (size_t)K->p.D + 1;
  uint64_t *d_small, *d_out; int64_t* d_tab;
  HIPCHK(hipMalloc(&d_small, count * (size_t)(t.n + 1) * 8));
  HIPCHK(hipMalloc(&d_out, count * L * 8));
  HIPCHK(hipMalloc(&d_tab, 8 * 16));
  std::vector<uint64_t> h(count * (size_t)(t.n + 1));
  uint64_t stt = 12345;
  for (auto& v : h) { stt = stt * 6364136223846793005ULL + 1442695040888963407ULL; v = stt; }
  HIPCHK(hipMemcpy(d_small, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  int64_t tab[16];
  for (int i = 0; i < 16; i++) tab[i] = (int64_t)i << 58;
  HIPCHK(hipMemcpy(d_tab, tab, sizeof tab, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  CHK(dev_pbs(K, tier, d_small, count, d_tab, 4, nullptr, 1, 1, 0, d_out, 0, 0, nullptr));
  hipEventRecord(a, ctx->stream);
  for (int r = 0; r < reps; r++) CHK(dev_pbs(K, tier, d_small, count, d_tab, 4, nullptr, 1, 1, 0, d_out, 0, 0, nullptr));
  hipEventRecord(b, ctx->stream);
  HIPCHK(hipEventSynchronize(b));
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  *ms_per_launch = ms / reps;
  hipEventDestroy(a); hipEventDestroy(b);
  hipFree(d_small); hipFree(d_out); hipFree(d_tab);
  return 0;
}
