// Host emulation of the blind-rotate thread program (dct-cryptonets_amd/csrc/pbs_core.h):
// T std::threads play the T lanes of one ciphertext group, a std::barrier plays s_barrier,
// a heap array plays LDS.  Checks the FFT index math and the whole PBS against the CPU
// oracle (oracle/tfhe_ref.c) without a GPU.  Test infrastructure only.
#include <barrier>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>

#include "../../dct-cryptonets_amd/csrc/pbs_core.h"
#include "../../oracle/tfhe_ref.h"

using namespace dctfhe;

// MB = 1: the two-bit blind rotation, checked against its exact-arithmetic definition (ref_pbs_mb2_batch)
template <int LOGN, int K, int L, int P, int MB = 0>
static int run_case(int n_in, int beta, int w, double sigma_bsk) {
  using G = pbs_geom<LOGN, K, L, P, MB>;
  constexpr int N = G::N, M = G::M, T = G::T;
  const int D = K * N, count = 6;
  std::vector<uint8_t> S(D), s(n_in);
  ref_gen_binary_key(11, D, S.data());
  ref_gen_binary_key(12, n_in, s.data());
  const int rows = (K + 1) * L;
  const int n = MB ? 3 * n_in / 2 : n_in;          // key blocks
  std::vector<uint8_t> skey(n);
  if (MB) ref_pair_secret(s.data(), n_in, skey.data()); else skey = s;
  std::vector<uint64_t> bsk((size_t)n * rows * (K + 1) * N);
  ref_bsk_gen(skey.data(), n, S.data(), K, N, L, beta, sigma_bsk, 13, bsk.data());
  std::vector<double> bskf(bsk.size());
  if (!MB) ref_bsk_to_fourier(bsk.data(), n, K, N, L, bskf.data());

  // inputs: small-LWE encryptions of messages on w bits (+padding)
  std::vector<uint64_t> phases(count), cts((size_t)count * (n_in + 1));
  for (int c = 0; c < count; c++) phases[c] = (uint64_t)((c * 5 + 1) % (1 << w)) << (63 - w);
  ref_lwe_encrypt_batch(s.data(), n_in, n_in, phases.data(), count, 1e-9, 14, cts.data());
  std::vector<int64_t> table(1 << w);
  for (int x = 0; x < (1 << w); x++) table[x] = (int64_t)(((x * 3 + 2) % (1 << w))) << (63 - w - 1);

  std::vector<uint64_t> ref_out((size_t)count * (D + 1));
  if (MB) ref_pbs_mb2_batch(cts.data(), count, n_in, bsk.data(), K, N, L, beta, table.data(), w, nullptr, D, ref_out.data());
  else ref_pbs_batch(cts.data(), count, n, bskf.data(), bsk.data(), 0, K, N, L, beta, table.data(), w, nullptr, D, ref_out.data());

  // ---- emulated device path
  std::vector<cplx> tw(G::F::TW_ELEMS);
  fill_twiddles<G::LOGM, P>(tw.data());
  std::vector<cplx> bsk_dev((size_t)(n + PBS_PF_DIST * G::KEY_BLOCKS) * G::BSK_ELEMS_PER_KEYBIT);
  std::vector<cplx> wtab(2 * N + 8);
  for (int m = 0; m < 2 * N; m++) { const long double a = 3.141592653589793238462643383279502884L * m / N; wtab[m] = cmk((double)cosl(a), (double)sinl(a)); }
  for (int m = 0; m < 8; m++) wtab[2 * N + m] = root64(8 * m);
  std::vector<cplx> zlut;      // the device's two-table form of the same roots (pbs_geom::ZLO / ZHI)
  if (MB) {
    for (int j = 0; j < (1 << G::ZLO); j++) zlut.push_back(wtab[j]);
    for (int j = 0; j < (1 << G::ZHI); j++) zlut.push_back(wtab[(size_t)j << G::ZLO]);
  }
  std::vector<unsigned char> shared(G::SHARED_BYTES);   // rotation stage aliases the exchange buffer, as on the device
  std::vector<uint64_t> accl((size_t)G::NL * N + 1);
  std::vector<uint32_t> pfd(T);
  std::vector<uint64_t> emu_out((size_t)count * (D + 1), 0x1234);
  {
    std::barrier bar(T);
    auto worker = [&](int t) {
      auto sync = [&] { bar.arrive_and_wait(); };
      const size_t npoly = (size_t)n * rows * (K + 1);
      for (size_t q = 0; q < npoly; q++)
        key_poly_to_fourier<LOGN, P>(bsk.data() + q * N, bsk_dev.data() + q * M, t, tw.data(), reinterpret_cast<cplx*>(shared.data()), sync, sync);
      sync();   // on the device the key conversion is a separate kernel: nobody is still gathering when the bootstrap starts
      for (int c = 0; c < count; c++) {
        pbs_args A;
        A.ct_small = cts.data() + (size_t)c * (n_in + 1); A.n = n_in; A.wtab = wtab.data(); A.beta = beta; A.bsk = bsk_dev.data();
        A.table = table.data(); A.w = w; A.out = emu_out.data() + (size_t)c * (D + 1); A.D_out = D;
        A.accumulate = 0; A.body_add = 0; A.bsk_wrap = 0; A.pf_parts = 0; A.twist = tw.data() + G::F::TW_TOTAL; A.pf_rank = 0; A.zlut = zlut.empty() ? nullptr : zlut.data();
        pbs_thread<LOGN, K, L, P, MB>(A, t, tw.data(), reinterpret_cast<uint64_t*>(shared.data() + G::STAGE_OFFSET), reinterpret_cast<cplx*>(shared.data()), accl.data(), pfd.data(), sync, sync);
        sync();
      }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
  }
  std::vector<uint64_t> ph_ref(count), ph_emu(count);
  ref_lwe_phase_batch(S.data(), D, ref_out.data(), count, ph_ref.data());
  ref_lwe_phase_batch(S.data(), D, emu_out.data(), count, ph_emu.data());
  int bad = 0;
  double maxd = 0, maxe = 0;
  for (int c = 0; c < count; c++) {
    const int m = (int)(phases[c] >> (63 - w));
    const uint64_t want = (uint64_t)table[m];
    const double d = std::fabs((double)(int64_t)(ph_emu[c] - ph_ref[c])) / 18446744073709551616.0;
    const double e = std::fabs((double)(int64_t)(ph_emu[c] - want)) / 18446744073709551616.0;
    if (d > maxd) maxd = d;
    if (e > maxe) maxe = e;
    const int got = (int)(((ph_emu[c] + (1ULL << (63 - w - 2))) >> (63 - w - 1)) & ((1 << (w + 1)) - 1));
    if (got != (int)(want >> (63 - w - 1))) bad++;
  }
  std::printf("%sN=%d k=%d l=%d P=%d T=%d n=%d: wrong=%d max|emu-ref|=%.3g max|emu-ideal|=%.3g\n", MB ? "two-bit " : "", N, K, L, P, T, n_in, bad, maxd, maxe);
  // a rounding flip in one decomposition gives a different but equivalent ciphertext, so the two
  // implementations agree only up to the scheme's own noise: bound both against the ideal value
  return bad || maxe > std::ldexp(1.0, -(w + 4));
}

// pure FFT round trip + comparison with an O(M^2) negacyclic evaluation
template <int LOGN, int P>
static int fft_case() {
  constexpr int N = 1 << LOGN, M = N / 2;
  using F = fft_geom<LOGN - 1, P>;
  constexpr int T = F::T;
  std::vector<cplx> tw(F::TW_ELEMS), exch(F::EXCH_ELEMS);
  fill_twiddles<LOGN - 1, P>(tw.data());
  std::vector<double> x(N);
  uint64_t st = 99;
  for (auto& v : x) v = (double)((int64_t)(ref_splitmix64(&st) >> 40) - (1 << 23));
  std::vector<cplx> spec(M), back(M);
  std::barrier bar(T);
  auto worker = [&](int t) {
    auto sync = [&] { bar.arrive_and_wait(); };
    cplx v[P];
    for (int j = 0; j < P; j++) v[j] = cmk(x[t + T * j], x[t + T * j + M]);
    fft_forward<LOGN - 1, P>(v, t, tw.data(), tw[F::TW_TOTAL + t], exch.data(), sync, sync);
    for (int j = 0; j < P; j++) spec[j * T + t] = v[j];
    fft_inverse<LOGN - 1, P>(v, t, tw.data(), tw[F::TW_TOTAL + t], exch.data(), sync, sync);
    for (int j = 0; j < P; j++) back[t + T * j] = v[j];
  };
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back(worker, t);
  for (auto& z : th) z.join();
  double err = 0;
  for (int n = 0; n < M; n++) {
    err = std::fmax(err, std::fabs(back[n].re / M - x[n]));
    err = std::fmax(err, std::fabs(back[n].im / M - x[n + M]));
  }
  // spectrum must be a permutation of the evaluations at the roots e^{i pi (1-4k)/N}
  double serr = 0;
  std::vector<char> used(M, 0);
  const double PI = 3.14159265358979323846;
  for (int k = 0; k < M; k++) {
    double re = 0, im = 0;
    for (int n = 0; n < N; n++) { const double a = PI * (double)n * (1.0 - 4.0 * k) / N; re += x[n] * std::cos(a); im += x[n] * std::sin(a); }
    double best = 1e300; int bi = -1;
    for (int q = 0; q < M; q++) { const double d = std::hypot(spec[q].re - re, spec[q].im - im); if (d < best) { best = d; bi = q; } }
    if (used[bi]) serr = 1e300;
    used[bi] = 1;
    serr = std::fmax(serr, best);
  }
  std::printf("fft N=%d P=%d T=%d passes=%d: roundtrip err=%.3g spectrum err=%.3g\n", N, P, T, F::S, err, serr);
  return err > 1e-6 || serr > 1e-3;
}

// ---- the integer/f64 helpers the kernels lean on, against independent definitions
static int torus_case() {     // y (units of the whole torus) -> floor(frac(y) 2^64), and round(frac(y) 2^32) saturating at 2^32 - 1, against 128-bit integers
  uint64_t st = 5; int bad = 0;
  for (int it = 0; it < 4000000; it++) {
    const int e = (int)(ref_splitmix64(&st) % 118) - 66;                      // |y| from 2^-66 to 2^52
    double y = std::ldexp((double)(int64_t)ref_splitmix64(&st) / 9223372036854775808.0, e);
    if (it % 7 == 0) y = std::nearbyint(y) + ((it % 3) ? 0.0 : 0.5);
    if (it % 13 == 0) y = std::ldexp((double)(int64_t)(ref_splitmix64(&st) >> 11), -53) * ((it & 1) ? 1.0 : -1.0);   // 53-bit fractions
    // exact frac(y) * 2^64 from the mantissa: y = mant * 2^(ex - 53)
    int ex; const double m = std::frexp(y, &ex);
    const __int128 mant = (__int128)std::ldexp(m, 53);                        // signed 53-bit integer
    const int sh = ex - 53 + 64;                                              // y * 2^64 = mant * 2^sh
    unsigned __int128 v;                                                      // floor(y * 2^64) mod 2^64
    if (sh >= 0) v = sh >= 64 ? 0 : (unsigned __int128)((__int128)mant << sh);
    else if (sh <= -64) v = mant < 0 ? ~(unsigned __int128)0 : 0;
    else v = (unsigned __int128)(mant >> (-sh));                              // arithmetic shift = floor
    const uint64_t want = (uint64_t)v;
    // exact for y >= 0; a negative y takes its fractional part as 1 - |frac|, which rounds to f64's grid below 1.0: 2^-53 of the torus
    const uint64_t got = f64_to_torus(y);
    const int64_t diff = (int64_t)(got - want);
    if (y >= 0 ? diff != 0 : (diff > 2048 || diff < -2048)) bad++;
    // top half, rounded to nearest: floor(frac * 2^32 + 1/2), saturating
    const uint64_t r = (want >> 32) + ((want >> 31) & 1);
    const uint32_t want32 = r >= (1ULL << 32) ? 0xFFFFFFFFu : (uint32_t)r;
    const uint32_t t32 = f64_to_torus32(y);
    if (t32 != want32) bad++;
  }
  std::printf("f64_to_torus: %d mismatches\n", bad);
  return bad != 0;
}
template <int L> static int decompose_one(int beta) {
  uint64_t st = beta * 7 + L; int bad = 0;
  for (int it = 0; it < 1000000; it++) {
    uint64_t v = ref_splitmix64(&st);
    if (it % 5 == 0) v |= ~0ULL << (it % 64);
    if (it % 11 == 0) v = ~0ULL - (ref_splitmix64(&st) & 0xffff);
    int32_t a[L], b[L];
    decompose<L>(v, beta, a); ref_decompose(v, L, beta, b);
    for (int i = 0; i < L; i++) bad += a[i] != b[i];
    if constexpr (L == 1) { int32_t c[1]; decompose<1>((uint32_t)(v >> 32), beta, c); bad += c[0] != b[0]; }
  }
  return bad;
}
static int decompose_case() {
  const int bad = decompose_one<1>(22) + decompose_one<1>(23) + decompose_one<1>(28) + decompose_one<1>(7) + decompose_one<2>(14) +
                  decompose_one<2>(16) + decompose_one<3>(11) + decompose_one<3>(12) + decompose_one<2>(5);
  std::printf("decompose vs oracle: %d mismatches\n", bad);
  return bad != 0;
}
// spectrum_freq: the transform of the polynomial X puts e^{i pi (1 - 4k)/N} at position p, k = spectrum_freq(p);
// fft_forward_n / fft_inverse_n: bit-identical to separate transforms
template <int LOGN, int P> static int layout_case() {
  constexpr int N = 1 << LOGN, M = N / 2; using F = fft_geom<LOGN - 1, P>; constexpr int T = F::T;
  std::vector<cplx> tw(F::TW_ELEMS), ex2(2 * F::EXCH_ELEMS), ex1(F::EXCH_ELEMS), spec(M), a(2 * M), b(2 * M), ia(2 * M), ib(2 * M);
  fill_twiddles<LOGN - 1, P>(tw.data());
  std::vector<double> x(2 * N);
  for (int i = 0; i < 2 * N; i++) x[i] = (double)((i * 2654435761u) % 1000) - 500;
  std::barrier bar(T);
  auto worker = [&](int t) {
    auto sync = [&] { bar.arrive_and_wait(); };
    const cplx twist = tw[F::TW_TOTAL + t];
    cplx v1[P];
    for (int j = 0; j < P; j++) v1[j] = cmk((t + T * j) == 1 ? 1.0 : 0.0, 0.0);
    fft_forward<LOGN - 1, P>(v1, t, tw.data(), twist, ex1.data(), sync, sync);
    for (int j = 0; j < P; j++) spec[P * t + j] = v1[j];
    sync();
    cplx v[2][P];
    for (int u = 0; u < 2; u++) for (int j = 0; j < P; j++) v[u][j] = cmk(x[u * N + t + T * j], x[u * N + t + T * j + M]);
    fft_forward_n<LOGN - 1, P, 2>(v, t, tw.data(), twist, ex2.data(), sync, sync);
    for (int u = 0; u < 2; u++) for (int j = 0; j < P; j++) a[u * M + j * T + t] = v[u][j];
    fft_inverse_n<LOGN - 1, P, 2>(v, t, tw.data(), twist, ex2.data(), sync, sync);
    for (int u = 0; u < 2; u++) for (int j = 0; j < P; j++) ia[u * M + j * T + t] = v[u][j];
    for (int u = 0; u < 2; u++) {
      cplx w[P];
      for (int j = 0; j < P; j++) w[j] = cmk(x[u * N + t + T * j], x[u * N + t + T * j + M]);
      fft_forward<LOGN - 1, P>(w, t, tw.data(), twist, ex1.data(), sync, sync);
      for (int j = 0; j < P; j++) b[u * M + j * T + t] = w[j];
      fft_inverse<LOGN - 1, P>(w, t, tw.data(), twist, ex1.data(), sync, sync);
      for (int j = 0; j < P; j++) ib[u * M + j * T + t] = w[j];
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back(worker, t);
  for (auto& z : th) z.join();
  double err = 0; int bad = 0;
  const double PI = 3.14159265358979323846;
  for (int p = 0; p < M; p++) {
    const double ang = PI * (1.0 - 4.0 * spectrum_freq<LOGN - 1, P>(p)) / N;
    err = std::fmax(err, std::hypot(spec[p].re - std::cos(ang), spec[p].im - std::sin(ang)));
  }
  for (int i = 0; i < 2 * M; i++) bad += (a[i].re != b[i].re) + (a[i].im != b[i].im) + (ia[i].re != ib[i].re) + (ia[i].im != ib[i].im);
  std::printf("layout N=%d P=%d: spectrum_freq err=%.3g, interleaved-vs-single mismatches=%d\n", N, P, err, bad);
  return err > 1e-9 || bad != 0;
}

// The L2 warm-up contract of pbs_core.h, asserted by the MMU: the key buffer holds exactly the key blocks plus PBS_PF_DIST
// iterations of padding and ends on an inaccessible page, so any touch past the contract faults here, on the host, instead
// of on a GPU (the recorded fault was a bsk_wrap case: the pointer walked n iterations into a buffer
// of bsk_wrap blocks).  Values are irrelevant (a zero key); n is small, the geometry is the shipped one.
template <int LOGN, int K, int L, int P, int MB = 0>
static int pf_guard_case(int n_in, int wrap, int pf_parts) {
  using G = pbs_geom<LOGN, K, L, P, MB>;
  constexpr int N = G::N, T = G::T;
  const int D = K * N;
  const size_t blocks = wrap ? (size_t)wrap : (size_t)(MB ? 3 * n_in / 2 : n_in);
  const size_t bytes = (blocks + (size_t)PBS_PF_DIST * G::KEY_BLOCKS) * G::BSK_ELEMS_PER_KEYBIT * 16;
  const size_t page = 4096, span = (bytes + page - 1) / page * page;
  unsigned char* map = (unsigned char*)mmap(nullptr, span + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (map == MAP_FAILED) { std::printf("mmap failed\n"); return 1; }
  if (mprotect(map + span, page, PROT_NONE)) { std::printf("mprotect failed\n"); return 1; }
  cplx* bsk_dev = reinterpret_cast<cplx*>(map + (span - bytes));       // the buffer ends exactly where the guard page starts
  std::vector<cplx> tw(G::F::TW_ELEMS);
  fill_twiddles<G::LOGM, P>(tw.data());
  std::vector<cplx> wtab(2 * N + 8, cmk(1.0, 0.0));
  std::vector<uint64_t> ct(n_in + 1, 0x0123456789abcdefULL), out((size_t)D + 1);
  std::vector<int64_t> table(16, 1LL << 58);
  std::vector<unsigned char> shared(G::SHARED_BYTES);
  std::vector<uint64_t> accl((size_t)G::NL * N + 1);
  std::vector<uint32_t> pfd(T);
  const int ranks[2] = {0, pf_parts > 0 ? pf_parts - 1 : 0};
  for (int pf_rank : ranks) {
    std::barrier bar(T);
    auto worker = [&](int t) {
      auto sync = [&] { bar.arrive_and_wait(); };
      pbs_args A;
      A.ct_small = ct.data(); A.n = n_in; A.wtab = wtab.data(); A.beta = L == 1 ? 20 : 10; A.bsk = bsk_dev;
      A.table = table.data(); A.w = 4; A.out = out.data(); A.D_out = D;
      A.accumulate = 0; A.body_add = 0; A.bsk_wrap = wrap; A.pf_parts = pf_parts; A.twist = tw.data() + G::F::TW_TOTAL; A.pf_rank = pf_rank; A.zlut = nullptr;
      pbs_thread<LOGN, K, L, P, MB>(A, t, tw.data(), reinterpret_cast<uint64_t*>(shared.data() + G::STAGE_OFFSET), reinterpret_cast<cplx*>(shared.data()), accl.data(), pfd.data(), sync, sync);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
  }
  munmap(map, span + page);
  std::printf("warm-up range %sN=%d k=%d l=%d n=%d wrap=%d pf_parts=%d: inside the key + %d iterations of padding\n", MB ? "two-bit " : "", N, K, L, n_in, wrap, pf_parts, PBS_PF_DIST);
  return 0;
}

int main() {
  int fail = 0;
  // every geometry the library launches (dctfhe.hip PBS_CASES used by dctfhe/params.py catalogues), pf_parts as launched (16; 32 at N = 8192),
  // the other legal settings, and the cache-experiment switch
  fail |= pf_guard_case<13, 1, 1, 8, 1>(4, 0, 16);
  fail |= pf_guard_case<13, 1, 1, 8, 1>(4, 0, 8);
  fail |= pf_guard_case<13, 1, 1, 8, 1>(4, 0, 32);
  fail |= pf_guard_case<13, 1, 1, 8, 1>(2, 0, 0);
  fail |= pf_guard_case<12, 1, 1, 8, 1>(4, 0, 16);
  fail |= pf_guard_case<11, 1, 1, 8, 1>(4, 0, 16);
  fail |= pf_guard_case<11, 1, 3, 8>(3, 0, 16);
  fail |= pf_guard_case<12, 1, 3, 8>(2, 0, 16);
  fail |= pf_guard_case<13, 1, 3, 8>(2, 0, 16);
  fail |= pf_guard_case<13, 1, 3, 8>(2, 0, 32);
  fail |= pf_guard_case<10, 2, 1, 8>(3, 0, 16);
  fail |= pf_guard_case<10, 2, 2, 8>(3, 0, 16);
  fail |= pf_guard_case<12, 1, 2, 8>(2, 0, 16);
  fail |= pf_guard_case<11, 1, 3, 8>(6, 2, 16);          // bsk_wrap with the warm-up on: the recorded fault's configuration
  fail |= pf_guard_case<13, 1, 1, 8>(5, 2, 16);
  fail |= torus_case();
  fail |= decompose_case();
  fail |= layout_case<9, 8>();
  fail |= layout_case<10, 8>();
  fail |= layout_case<11, 8>();
  fail |= layout_case<12, 8>();
  fail |= fft_case<8, 16>();
  fail |= fft_case<9, 8>();
  fail |= fft_case<9, 16>();
  fail |= fft_case<10, 16>();
  fail |= fft_case<10, 8>();
  fail |= fft_case<11, 16>();
  fail |= run_case<8, 1, 2, 16>(20, 10, 3, 1e-12);
  fail |= run_case<9, 2, 1, 8>(16, 16, 3, 1e-13);
  fail |= run_case<9, 1, 3, 8>(16, 7, 4, 1e-12);
  fail |= run_case<10, 1, 2, 16>(12, 12, 4, 1e-13);
  fail |= run_case<9, 1, 1, 8>(16, 20, 3, 1e-13);    // one level, k = 1: the pair-interleaved path
  fail |= run_case<11, 1, 1, 8>(8, 22, 4, 1e-14);
  fail |= run_case<9, 1, 1, 8, 1>(16, 20, 3, 1e-13);   // two-bit blind rotation, last pass radix 4
  fail |= run_case<10, 1, 1, 8, 1>(12, 20, 3, 1e-13);  // ... radix 8
  fail |= run_case<11, 1, 1, 8, 1>(8, 22, 4, 1e-14);   // ... radix 2
  fail |= run_case<9, 2, 1, 8, 1>(16, 16, 3, 1e-13);   // two-bit, general path: k = 2
  fail |= run_case<9, 1, 3, 8, 1>(16, 7, 4, 1e-12);    // ... three levels, first accumulator polynomial in LDS
  fail |= run_case<10, 1, 2, 16, 1>(12, 12, 4, 1e-13); // ... 16 points per thread
  std::printf(fail ? "EMUL FAIL\n" : "EMUL OK\n");
  return fail;
}
