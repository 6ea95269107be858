// fft_core.h -- per-thread building blocks of the negacyclic f64 FFT used by the blind-rotate
// kernel (K5 in DESIGN.md).  Everything here is a "thread program": it touches only the
// thread's own registers plus an LDS array, so the same code runs on a gfx950 wave and, for
// the index-math tests in tests/emul, on the host one thread at a time.
//
// Transform: a real polynomial of N coefficients in Z[X]/(X^N+1) is folded to M = N/2 complex
// points z_n = (x_n + i x_{n+M}) * e^{i pi n / N} and sent through an M-point DIF FFT that is
// kept IN PLACE in mixed radix: M = R_0 * R_1 * ... * R_{s-1}, digit i has weight
// W_i = prod_{j>i} R_j, pass i replaces time digit n_i by frequency digit k_i at the same
// position.  The spectrum therefore ends in digit-reversed order, which nobody ever undoes:
// the bootstrapping key is stored in the same order and the inverse transform (DIT, passes in
// reverse) consumes it and returns natural order.
//
// Thread layout: T = M / P threads per polynomial, P points per thread (P = 8 or 16).
//   R_i = P for every pass but possibly the last, R_last = M / P^(s-1) (divides P).
//   pass i, thread t: base = (t / W_i) * (W_i * R_i) + (t % W_i), points base + j * W_i.
//   last pass with R_last < P: thread t owns P contiguous points = P / R_last small DFTs.
// Between passes the points go through an LDS exchange buffer (one write + one read per
// pass boundary); the first forward pass reads registers, the last leaves registers.
#pragma once
#include <stdint.h>
#include <type_traits>

#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
#else
#define HD inline __attribute__((always_inline))
#endif

#ifndef DCTFHE_SHARED_TWIDDLES
#define DCTFHE_SHARED_TWIDDLES 1
#endif
#ifndef DCTFHE_PAIR_STAGGER
#define DCTFHE_PAIR_STAGGER 0      // experiment switch (fft_forward_n): s_sleep units the younger half of a 512-thread workgroup waits; OFF
#endif

// Wave-local passes of the NP-polynomial transforms, polynomial by polynomial: {butterflies; scatter; gather} of polynomial u, then of
// u + 1 -- the LDS unit moves polynomial u while the VALU runs the butterflies of u + 1 (each with a twiddle chain of its own).
#ifndef DCTFHE_PIPE_LOCAL
#define DCTFHE_PIPE_LOCAL 0
#endif
// Middle passes of the inverse transforms as fused butterflies (fused_idft below): 1 = on
#ifndef DCTFHE_FUSED_INV
#define DCTFHE_FUSED_INV 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(DCTFHE_PIPE_PIN)
#define DCTFHE_PIPE_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define DCTFHE_PIPE_SCHED_BARRIER() ((void)0)
#endif
// pinning the interleaved order with scheduling barriers measured 8% slower than leaving hipcc free (N = 8192)
#if defined(__HIP_DEVICE_COMPILE__) && defined(DCTFHE_PIN_FFT_ORDER)
#define DCTFHE_FFT_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define DCTFHE_FFT_SCHED_BARRIER() ((void)0)
#endif

namespace dctfhe {

struct cplx { double re, im; };

// Phase clock for timing experiments (tools/exp_pbs.hip, -DDCTFHE_PHASE_TIMERS): at<K>() charges the shader-clock ticks since the last
// call to phase K.  Scalar registers only; reading the clock drains the wave's LDS queue (s_memtime returns through lgkmcnt), so a
// phase includes the completion of the LDS operations issued in it.  no_tick: the shipped kernels.
struct no_tick { template <int K> HD void at() {} };
#if defined(__HIP_DEVICE_COMPILE__) && defined(DCTFHE_PHASE_TIMERS)
struct phase_clock {
  unsigned long long last, acc[12];
  __device__ __forceinline__ void start() { for (int k = 0; k < 12; k++) acc[k] = 0; last = __builtin_amdgcn_s_memtime(); }
  template <int K> __device__ __forceinline__ void at() { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc[K] += now - last; last = now; }
};
#endif

HD cplx cmk(double re, double im) { cplx c; c.re = re; c.im = im; return c; }
HD cplx cadd(cplx a, cplx b) { return cmk(a.re + b.re, a.im + b.im); }
HD cplx csub(cplx a, cplx b) { return cmk(a.re - b.re, a.im - b.im); }
HD cplx cmul(cplx a, cplx b) {
  return cmk(__builtin_fma(a.re, b.re, -(a.im * b.im)), __builtin_fma(a.re, b.im, a.im * b.re));
}
HD cplx cmulc(cplx a, cplx b) {  // a * conj(b)
  return cmk(__builtin_fma(a.re, b.re, a.im * b.im), __builtin_fma(a.im, b.re, -(a.re * b.im)));
}
HD cplx csqr(cplx a) { return cmk(__builtin_fma(a.re, a.re, -(a.im * a.im)), 2.0 * a.re * a.im); }
HD cplx cfma(cplx a, cplx b, cplx acc) {  // acc + a*b
  cplx r;
  r.re = __builtin_fma(a.re, b.re, acc.re); r.re = __builtin_fma(-a.im, b.im, r.re);
  r.im = __builtin_fma(a.re, b.im, acc.im); r.im = __builtin_fma(a.im, b.re, r.im);
  return r;
}

// compile-time loop
template <int I, int E, class F>
HD void static_for(F&& f) {
  if constexpr (I < E) { f(std::integral_constant<int, I>{}); static_for<I + 1, E>(static_cast<F&&>(f)); }
}

// cos/sin(2 pi k / 64), k = 0..16 (first quadrant); everything up to radix 16 plus the
// e^{i pi j/(2P)} twist constants comes out of this table by symmetry.
HD double q64cos(int k) {
  constexpr double t[17] = {1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494,
                            0.92387953251128675613, 0.88192126434835502971, 0.83146961230254523708,
                            0.77301045336273696081, 0.70710678118654752440, 0.63439328416364549822,
                            0.55557023301960222474, 0.47139673682599764856, 0.38268343236508977173,
                            0.29028467725446236764, 0.19509032201612826785, 0.09801714032956060199, 0.0};
  return t[k];
}
// e^{+2 pi i k / 64}
HD cplx root64(int k) {
  k &= 63;
  const int q = k >> 4, r = k & 15;
  const double c = q64cos(r), s = q64cos(16 - r);
  switch (q) {
    case 0: return cmk(c, s);
    case 1: return cmk(-s, c);
    case 2: return cmk(-c, -s);
    default: return cmk(s, -c);
  }
}

// multiply by the compile-time constant e^{SIGN * 2 pi i K / 64}
template <int K, int SIGN>
HD cplx mul_root64(cplx a) {
  constexpr int k = ((SIGN > 0 ? K : -K) % 64 + 64) % 64;
  if constexpr (k == 0) return a;
  else if constexpr (k == 16) return cmk(-a.im, a.re);
  else if constexpr (k == 32) return cmk(-a.re, -a.im);
  else if constexpr (k == 48) return cmk(a.im, -a.re);
  else if constexpr (k % 16 == 8) {  // odd multiples of pi/4
    constexpr double h = 0.70710678118654752440;
    if constexpr (k == 8) return cmk(h * (a.re - a.im), h * (a.re + a.im));
    else if constexpr (k == 24) return cmk(-h * (a.re + a.im), h * (a.re - a.im));
    else if constexpr (k == 40) return cmk(-h * (a.re - a.im), -h * (a.re + a.im));
    else return cmk(h * (a.re + a.im), -h * (a.re - a.im));
  } else {
    const cplx w = root64(k);
    return cmul(a, w);
  }
}

// Small DFT of size R (power of two <= 16) on registers, natural order in and out.
// y[k] = sum_n x[n] e^{SIGN 2 pi i n k / R};  SIGN = -1 forward, +1 inverse (unnormalised).
// Radix-2 decimation in time, fully unrolled; x is read with stride S.
template <int R, int S, int SIGN>
struct small_dft {
  static HD void run(const cplx* x, cplx* y) {
    if constexpr (R == 1) {
      y[0] = x[0];
    } else if constexpr (R == 2) {
      const cplx a = x[0], b = x[S];
      y[0] = cadd(a, b); y[1] = csub(a, b);
    } else {
      cplx e[R / 2], o[R / 2];
      small_dft<R / 2, 2 * S, SIGN>::run(x, e);
      small_dft<R / 2, 2 * S, SIGN>::run(x + S, o);
      static_for<0, R / 2>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const cplx t = mul_root64<k*(64 / R), SIGN>(o[k]);
        y[k] = cadd(e[k], t); y[k + R / 2] = csub(e[k], t);
      });
    }
  }
};

// Size-R DFT (e^{+2 pi i m k / R}) of z[m] = x[m S] c^m -- the twiddled passes of the inverse transform -- with the twiddles INSIDE the
// butterflies: decimation in time, z[2m'] = x[2m' S] (c^2)^m' and z[2m'+1] = c x[(2m'+1) S] (c^2)^m', so
//   Y[k], Y[k + R/2] = E[k] +- (c omega_R^k) O'[k]        (E, O': the two half-size transforms with base c^2)
// and a butterfly a +- t b is p = fma(t, b, a) (two fma per component) and m = 2a - p (one): 6 f64 instructions where the multiplication
// and the two additions took 8, and the 7-step power chain c^1 .. c^7 is replaced by c^2, c^4 (squarings) and c omega_8 -- the other
// factors are those times i, a matter of which operand goes where.  pw[l] = c^(2^l); cw8 = c e^{+2 pi i / 8}.
HD void fused_bf(const cplx a, const cplx b, const double tr, const double ti, cplx& p, cplx& m) {
  p.re = __builtin_fma(tr, b.re, __builtin_fma(-ti, b.im, a.re));
  p.im = __builtin_fma(tr, b.im, __builtin_fma(ti, b.re, a.im));
  m.re = __builtin_fma(2.0, a.re, -p.re);
  m.im = __builtin_fma(2.0, a.im, -p.im);
}
template <int R, int S, int LV>
struct fused_idft {
  static HD void run(const cplx* x, cplx* y, const cplx* pw, const cplx cw8) {
    static_assert(R == 2 || R == 4 || R == 8, "radix 8 passes");
    if constexpr (R == 2) {
      fused_bf(x[0], x[S], pw[LV].re, pw[LV].im, y[0], y[1]);
    } else {
      cplx e[R / 2], o[R / 2];
      fused_idft<R / 2, 2 * S, LV + 1>::run(x, e, pw, cw8);
      fused_idft<R / 2, 2 * S, LV + 1>::run(x + S, o, pw, cw8);
      const cplx c = pw[LV];
      fused_bf(e[0], o[0], c.re, c.im, y[0], y[R / 2]);
      if constexpr (R == 4) {
        fused_bf(e[1], o[1], -c.im, c.re, y[1], y[1 + R / 2]);                 // t = i c
      } else {
        fused_bf(e[1], o[1], cw8.re, cw8.im, y[1], y[1 + R / 2]);              // t = c omega_8
        fused_bf(e[2], o[2], -c.im, c.re, y[2], y[2 + R / 2]);                 // t = i c
        fused_bf(e[3], o[3], -cw8.im, cw8.re, y[3], y[3 + R / 2]);             // t = i c omega_8
      }
    }
  }
};
// the prepared factors of a pass from its twiddle base b (c = conj b)
HD void fused_idft_factors(const cplx b, cplx* pw, cplx& cw8) {
  constexpr double h = 0.70710678118654752440;
  pw[0] = cmk(b.re, -b.im);
  pw[1] = csqr(pw[0]);
  pw[2] = csqr(pw[1]);
  cw8 = cmk(h * (pw[0].re - pw[0].im), h * (pw[0].re + pw[0].im));
}

// ---------------------------------------------------------------------------------------------
// Geometry of the in-place mixed-radix transform for (log2 M, P)
constexpr int geom_logp(int P) { return P == 16 ? 4 : P == 8 ? 3 : P == 4 ? 2 : -1; }
constexpr int geom_passes(int LOGM, int P) { return LOGM / geom_logp(P) + ((LOGM % geom_logp(P)) ? 1 : 0); }
constexpr int geom_radix(int LOGM, int P, int i) {
  return (i == geom_passes(LOGM, P) - 1 && (LOGM % geom_logp(P))) ? (1 << (LOGM % geom_logp(P))) : P;
}
constexpr int geom_weight(int LOGM, int P, int i) {  // W_i = prod_{j>i} R_j
  int w = 1;
  for (int j = geom_passes(LOGM, P) - 1; j > i; j--) w *= geom_radix(LOGM, P, j);
  return w;
}
constexpr int geom_tw_offset(int LOGM, int P, int i) {  // pass i < S-1 owns W_i entries e^{-2 pi i m/(W_i R_i)}
  int o = 0;
  for (int j = 0; j < i; j++) o += geom_weight(LOGM, P, j);
  return o;
}

template <int LOGM, int P>
struct fft_geom {
  static constexpr int M = 1 << LOGM;
  static_assert(geom_logp(P) > 0, "P must be 4, 8 or 16");
  static constexpr int T = M / P;                        // threads per polynomial
  static constexpr int S = geom_passes(LOGM, P);         // number of passes
  static_assert(S >= 2, "need at least two passes");
  static constexpr int radix(int i) { return geom_radix(LOGM, P, i); }
  static constexpr int weight(int i) { return geom_weight(LOGM, P, i); }
  static constexpr int tw_offset(int i) { return geom_tw_offset(LOGM, P, i); }
  static constexpr int TW_TOTAL = geom_tw_offset(LOGM, P, S - 1);  // then T twist bases e^{i pi t/N}
  static constexpr int TW_ELEMS = TW_TOTAL + T;
  // Two images of the exchange buffer (LDS holds EXCH_ELEMS complex values either way), chosen PER EXCHANGE by the pattern that READS it.
  // gfx950 serves a wave's ds_read_b128 in four groups of 16 non-contiguous lanes over 64 banks and its ds_write_b128 in eight groups of
  // 8 contiguous lanes over 32 banks (MI355X_MICROARCH.md, LDS; model: tools/lds_model.py).  Every write pattern of the transforms is
  // conflict-free under both images; the reads differ:
  //   skew   one padding element per P: conflict-free for the strided gathers (weight < 16 lanes, or the last, smaller radix), 2-way
  //          for the lane-contiguous ones (16 contiguous elements then span 17 slots);
  //   swz    no padding, the element's position inside its row of 8 XORed with the row number: conflict-free for the lane-contiguous
  //          gathers (weight >= 16), 2-way for the strided ones.  Kept inside the same per-wave footprint as `skew` (512 elements of a
  //          wave's block -> 576 slots), so that a wave-local exchange never touches another wave's slots whichever image it uses.
  // Round 2 used `skew` throughout: 1.07 conflict cycles per LDS instruction on the N = 8192 kernel (profiles/r02_pmc_tiers*.txt), all of
  // them on the three lane-contiguous gathers of a forward + inverse pair -- the model's count exactly.
  static HD int skew(int idx) { return idx + (idx >> geom_logp(P)); }
  static HD int swz(int idx) { return (idx >> 9) * 576 + ((idx & 511) ^ ((idx >> 3) & 7)); }
  // (M = 512, one wave per ciphertext: a single gather would gain and the second set of addresses costs the k = 2 kernels registers they
  //  do not have -- 60-84 bytes of scratch per lane, +1 / +4 % time: profiles/r03_exp_lds_image.log -- so those keep `skew` throughout)
  static constexpr bool swz_reader(int i) { return P == 8 && LOGM >= 10 && geom_radix(LOGM, P, i) == P && geom_weight(LOGM, P, i) >= 16; }
  template <int IR> static HD int ex(int idx) { if constexpr (swz_reader(IR)) return swz(idx); else return skew(idx); }   // IR: the pass that gathers
  static constexpr int EXCH_ELEMS = M + (M >> geom_logp(P));
};

// thread t's first point in a radix-R pass of weight W (its points are base + j*W)
HD int pass_base(int W, int R, int t) { return (t / W) * (W * R) + (t % W); }

// address (before skew) of register j of thread t in pass i
template <int LOGM, int P, int I>
HD int pass_addr(int t, int j) {
  using G = fft_geom<LOGM, P>;
  constexpr int R = G::radix(I), W = G::weight(I);
  if constexpr (R == P) return pass_base(W, R, t) + j * W;
  else return P * t + j;  // last, smaller radix: P contiguous points = P/R small DFTs
}

// ---------------------------------------------------------------------------------------------
// Forward transform of one polynomial.
//   v[0..P)   in : folded, UN-twisted points z_n = x_n + i x_{n+M} for n = t + T*j
//             out: spectrum points at in-place addresses pass_addr<S-1>(t, j)
//   tw        : table of G::TW_ELEMS entries: pass twiddles (the first G::TW_TOTAL, all the transform reads), then
//               the T twist bases e^{i pi t/N};  twist = entry TW_TOTAL + t, handed over by value so that a kernel
//               short of LDS can keep only the pass twiddles there
//   exch      : LDS exchange buffer (G::EXCH_ELEMS), shared by the T threads of this polynomial
//   sync      : barrier for those T threads (see the barrier discipline inside).
// The exchange between pass i and i+1 only moves points among groups of W_i consecutive threads
// (pass i+1 works inside blocks of W_i points, and the W_i threads that wrote a super-block of W_i*P points
// are the ones that read it).  When W_i <= 64 that group sits inside one wave: the LDS queue of a wave is
// in order, so no workgroup barrier is needed -- `wsync` (a compiler-level wave barrier) is enough.
struct no_hook { HD void operator()() const {} };

// `before_last` runs right after the last gather, before the last pass's butterflies: the caller uses it to put
// its key loads in flight under that pass.
template <int LOGM, int P, class Sync, class WSync, class Hook = no_hook>
HD void fft_forward(cplx* v, int t, const cplx* tw, const cplx twist, cplx* exch, Sync&& sync, WSync&& wsync, Hook&& before_last = Hook{}) {
#if defined(DCTFHE_ABLATE_FFT)   // timing experiments only
  return;
#endif
  using G = fft_geom<LOGM, P>;
  constexpr int S = G::S;
  // twist part 1: compile-time factor e^{i pi j T / N} = e^{2 pi i j / (4P)} on register j
  static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[j] = mul_root64<j*(64 / (4 * P)), +1>(v[j]); });
  static_for<0, S>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int R = G::radix(i);
    constexpr int W = G::weight(i);
    if constexpr (i == S - 1) before_last();
    cplx y[P];
    if constexpr (R == P) {
      small_dft<P, 1, -1>::run(v, y);
    } else {
      static_for<0, P / R>([&](auto Gp) { constexpr int g = decltype(Gp)::value; small_dft<R, 1, -1>::run(v + g * R, y + g * R); });
    }
    if constexpr (i < S - 1) {
      // twiddle e^{-2 pi i k m / (W R)}, m = t % W; pass 0 also carries twist part 2, e^{i pi t / N}
      const cplx b = tw[G::tw_offset(i) + (t % W)];
      // twiddle powers by a running product: two live values instead of R (the tree of squarings cost 24 more
      // registers and measured no better noise: tests/emul/fftnoise.cpp)
      {
        cplx run = (i == 0) ? twist : cmk(1.0, 0.0);
        if constexpr (i == 0) y[0] = cmul(y[0], run);
        static_for<1, R>([&](auto K) { constexpr int k = decltype(K)::value; run = cmul(run, b); y[k] = cmul(y[k], run); });
      }
      // Barrier discipline.  A cross-wave exchange (W > 64) scatters over the whole buffer, so every wave must
      // be done with whatever it was reading there (the wave-local gathers of the previous transform) BEFORE
      // the first write, and the data must be complete before the gather: barrier; write; barrier; gather.
      // The gather itself, and everything a wave-local exchange touches, stays inside the wave's own block,
      // where program order (the LDS queue of a wave is in order) is enough.
      if constexpr (W <= 64) wsync(); else sync();
      static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; exch[G::template ex<i + 1>(pass_addr<LOGM, P, i>(t, j))] = y[j]; });
      if constexpr (W <= 64) wsync(); else sync();
      static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[j] = exch[G::template ex<i + 1>(pass_addr<LOGM, P, i + 1>(t, j))]; });
    } else {
      static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[j] = y[j]; });
    }
  });
}

// Inverse transform (unnormalised; the 1/M lives in the Fourier key).
//   v[0..P)   in : spectrum at addresses pass_addr<S-1>(t, j);  out: z_n for n = t + T*j, twist removed.
template <int LOGM, int P, class Sync, class WSync>
HD void fft_inverse(cplx* v, int t, const cplx* tw, const cplx twist, cplx* exch, Sync&& sync, WSync&& wsync) {
#if defined(DCTFHE_ABLATE_FFT)
  return;
#endif
  using G = fft_geom<LOGM, P>;
  constexpr int S = G::S;
  static_for<0, S>([&](auto Irev) {
    constexpr int i = S - 1 - decltype(Irev)::value;
    constexpr int R = G::radix(i);
    constexpr int W = G::weight(i);
    constexpr bool FUSED = DCTFHE_FUSED_INV && i > 0 && i < S - 1 && R == P && P == 8;
    if constexpr (i < S - 1 && !FUSED) {
      const cplx b = tw[G::tw_offset(i) + (t % W)];
      {
        cplx run = (i == 0) ? twist : cmk(1.0, 0.0);
        if constexpr (i == 0) v[0] = cmulc(v[0], run);
        static_for<1, R>([&](auto K) { constexpr int k = decltype(K)::value; run = cmul(run, b); v[k] = cmulc(v[k], run); });
      }
    }
    cplx y[P];
    if constexpr (FUSED) {
      cplx pw[3], cw8;
      fused_idft_factors(tw[G::tw_offset(i) + (t % W)], pw, cw8);
      fused_idft<P, 1, 0>::run(v, y, pw, cw8);
    } else if constexpr (R == P) {
      small_dft<P, 1, +1>::run(v, y);
    } else {
      static_for<0, P / R>([&](auto Gp) { constexpr int g = decltype(Gp)::value; small_dft<R, 1, +1>::run(v + g * R, y + g * R); });
    }
    if constexpr (i > 0) {
      constexpr int Wp = G::weight(i - 1);   // exchange between pass i-1 and i: groups of W_{i-1} threads
      static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; exch[G::template ex<i - 1>(pass_addr<LOGM, P, i>(t, j))] = y[j]; });
      if constexpr (Wp <= 64) wsync(); else sync();
      static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[j] = exch[G::template ex<i - 1>(pass_addr<LOGM, P, i - 1>(t, j))]; });
      // after a cross-wave gather other waves may still be reading this wave's block: barrier before anyone writes again
      if constexpr (Wp <= 64) wsync(); else sync();
    } else {
      static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[j] = mul_root64<j*(64 / (4 * P)), -1>(y[j]); });
    }
  });
}

// ---------------------------------------------------------------------------------------------
// NP transforms of independent polynomials, interleaved pass by pass: polynomial u exchanges through its own
// buffer exch + u * G::EXCH_ELEMS.  While the LDS unit drains the scatter of polynomial u the VALU runs the
// butterflies of polynomial u+1, and the butterflies of polynomial u start as soon as ITS gather has landed
// while the gather of u+1 is still in flight (the LDS queue of a wave returns in order, so the compiler can wait
// on a partial count).  One barrier pair per pass serves all NP polynomials.  Same arithmetic per polynomial,
// in the same order, as fft_forward / fft_inverse: results are bit-identical to NP separate calls.
template <int LOGM, int P, int NP, class Sync, class WSync, class Tick = no_tick>
HD void fft_forward_n(cplx (&v)[NP][P], int t, const cplx* tw, const cplx twist, cplx* exch, Sync&& sync, WSync&& wsync, Tick&& tick = Tick{}) {
#if defined(DCTFHE_ABLATE_FFT)
  return;
#endif
  using G = fft_geom<LOGM, P>;
  constexpr int S = G::S;
  static_for<0, NP>([&](auto U) {
    constexpr int u = decltype(U)::value;
    static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[u][j] = mul_root64<j*(64 / (4 * P)), +1>(v[u][j]); });
  });
  [[maybe_unused]] cplx bpre = cmk(1.0, 0.0);      // DCTFHE_PIPE_LOCAL: the next pass's twiddle base, read a pass ahead
  static_for<0, S>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int R = G::radix(i);
    constexpr int W = G::weight(i);
    if constexpr (i < S - 1 && W <= 64 && DCTFHE_PIPE_LOCAL) {
      // this pass's twiddle base was read one pass ahead (bpre) wherever a pass precedes this one: a read issued here would sit BEHIND the
      // previous pass's last gather in the wave's in-order LDS queue, and the first polynomial's butterflies would wait for all of it
      cplx b;
      if constexpr (i > 0) b = bpre; else b = tw[G::tw_offset(i) + (t % W)];
      const cplx run0 = (i == 0) ? twist : cmk(1.0, 0.0);
      static_for<0, NP>([&](auto U) {
        constexpr int u = decltype(U)::value;
        cplx y[P];
        small_dft<P, 1, -1>::run(v[u], y);
        cplx run = run0;
        if constexpr (i == 0) y[0] = cmul(y[0], run);
        static_for<1, R>([&](auto K) { constexpr int k = decltype(K)::value; run = cmul(run, b); y[k] = cmul(y[k], run); });
        cplx* ex = exch + u * G::EXCH_ELEMS;
        if constexpr (u == 0 && i + 1 < S - 1) bpre = tw[G::tw_offset(i + 1) + (t % G::weight(i + 1))];
        wsync();
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; ex[G::template ex<i + 1>(pass_addr<LOGM, P, i>(t, j))] = y[j]; });
        wsync();
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[u][j] = ex[G::template ex<i + 1>(pass_addr<LOGM, P, i + 1>(t, j))]; });
        DCTFHE_PIPE_SCHED_BARRIER();
      });
    } else if constexpr (i < S - 1) {
      const cplx b = tw[G::tw_offset(i) + (t % W)];
      const cplx run0 = (i == 0) ? twist : cmk(1.0, 0.0);
      if constexpr (DCTFHE_PIPE_LOCAL && i + 1 < S - 1) bpre = tw[G::tw_offset(i + 1) + (t % G::weight(i + 1))];
      if constexpr (W > 64) tick.template at<0>();        // phase 0: accumulator update, decomposition
      if constexpr (W <= 64) wsync(); else sync();      // nobody still gathers from the buffers about to be written
      if constexpr (W > 64) tick.template at<1>();        // phase 1: waiting at the leading barrier
#if DCTFHE_SHARED_TWIDDLES
      // butterflies of all NP polynomials, then ONE running twiddle product applied to all of them (the chain costs
      // as much as applying it: 4 f64 instructions per step), then the scatters
      cplx y[NP][P];
      static_for<0, NP>([&](auto U) { constexpr int u = decltype(U)::value; small_dft<P, 1, -1>::run(v[u], y[u]); });
      {
        cplx run = run0;
        if constexpr (i == 0) static_for<0, NP>([&](auto U) { constexpr int u = decltype(U)::value; y[u][0] = cmul(y[u][0], run); });
        static_for<1, R>([&](auto K) {
          constexpr int k = decltype(K)::value;
          run = cmul(run, b);
          static_for<0, NP>([&](auto U) { constexpr int u = decltype(U)::value; y[u][k] = cmul(y[u][k], run); });
        });
      }
      static_for<0, NP>([&](auto U) {
        constexpr int u = decltype(U)::value;
        cplx* ex = exch + u * G::EXCH_ELEMS;
#if defined(DCTFHE_ABLATE_EXCH)
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[u][j] = y[u][(j + 1) % P]; });
#else
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; ex[G::template ex<i + 1>(pass_addr<LOGM, P, i>(t, j))] = y[u][j]; });
#endif
      });
#else
      static_for<0, NP>([&](auto U) {
        constexpr int u = decltype(U)::value;
        cplx y[P];
        small_dft<P, 1, -1>::run(v[u], y);
        cplx run = run0;
        if constexpr (i == 0) y[0] = cmul(y[0], run);
        static_for<1, R>([&](auto K) { constexpr int k = decltype(K)::value; run = cmul(run, b); y[k] = cmul(y[k], run); });
        cplx* ex = exch + u * G::EXCH_ELEMS;
#if defined(DCTFHE_ABLATE_EXCH)   // timing experiments only: no LDS traffic, wrong results
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[u][j] = y[(j + 1) % P]; });
#else
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; ex[G::template ex<i + 1>(pass_addr<LOGM, P, i>(t, j))] = y[j]; });
#endif
        DCTFHE_FFT_SCHED_BARRIER();
      });
#endif
      if constexpr (W > 64) tick.template at<2>();        // phase 2: butterflies + twiddles + scatter of the cross-wave pass
      if constexpr (W <= 64) wsync(); else sync();
      if constexpr (W > 64) tick.template at<3>();        // phase 3: waiting at the barrier before the gather
#if defined(__HIP_DEVICE_COMPILE__)
      // Experiment, off: the two waves of a SIMD (w and w + 4 of a 512-thread workgroup) leave this barrier together and gather, compute
      // and scatter in the same phase; holding the younger half back by 64-512 cycles de-phases them for the barrier-free stretch that
      // follows: -0.5 % at N = 8192, -1.6 % at N = 4096 (profiles/r03_exp_stagger.log).  NOT shipped: the same one-line branch in the
      // single-transform path made hipcc demote that kernel's register arrays to scratch (170 -> 2 716 ms per launch, same log) -- a
      // third of a per cent is not worth standing that close to the cliff.
      if constexpr (W > 64 && DCTFHE_PAIR_STAGGER > 0) { if (threadIdx.x & 256) __builtin_amdgcn_s_sleep(DCTFHE_PAIR_STAGGER); }
#endif
      static_for<0, NP>([&](auto U) {
        constexpr int u = decltype(U)::value;
        const cplx* ex = exch + u * G::EXCH_ELEMS;
#if !defined(DCTFHE_ABLATE_EXCH)
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[u][j] = ex[G::template ex<i + 1>(pass_addr<LOGM, P, i + 1>(t, j))]; });
#endif
      });
      DCTFHE_FFT_SCHED_BARRIER();
    } else {
      static_for<0, NP>([&](auto U) {
        constexpr int u = decltype(U)::value;
        cplx y[P];
        if constexpr (R == P) {
          small_dft<P, 1, -1>::run(v[u], y);
        } else {
          static_for<0, P / R>([&](auto Gp) { constexpr int g = decltype(Gp)::value; small_dft<R, 1, -1>::run(v[u] + g * R, y + g * R); });
        }
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[u][j] = y[j]; });
        DCTFHE_FFT_SCHED_BARRIER();
      });
    }
  });
  tick.template at<4>();                                  // phase 4: gather of the cross-wave pass, the wave-local passes
}

template <int LOGM, int P, int NP, class Sync, class WSync, class Tick = no_tick>
HD void fft_inverse_n(cplx (&v)[NP][P], int t, const cplx* tw, const cplx twist, cplx* exch, Sync&& sync, WSync&& wsync, Tick&& tick = Tick{}) {
#if defined(DCTFHE_ABLATE_FFT)
  return;
#endif
  using G = fft_geom<LOGM, P>;
  constexpr int S = G::S;
  [[maybe_unused]] cplx bpre = cmk(1.0, 0.0);      // DCTFHE_PIPE_LOCAL: the next pass's twiddle base, read a pass ahead (see fft_forward_n)
  static_for<0, S>([&](auto Irev) {
    constexpr int i = S - 1 - decltype(Irev)::value;
    constexpr int R = G::radix(i);
    constexpr int W = G::weight(i);
    if constexpr (i > 0 && G::weight(i > 0 ? i - 1 : 0) > 64) tick.template at<6>();             // phase 6: the passes that end in wave-local exchanges
    constexpr bool PIPE = DCTFHE_PIPE_LOCAL && i > 0 && G::weight(i > 0 ? i - 1 : 0) <= 64;   // this pass ends in a wave-local exchange
    constexpr bool PREV_PIPE = DCTFHE_PIPE_LOCAL && i < S - 1 && W <= 64;                      // ... and so did the one before it
    cplx b = cmk(1.0, 0.0), run0 = cmk(1.0, 0.0);
    if constexpr (i < S - 1) {
      if constexpr (PREV_PIPE) b = bpre; else b = tw[G::tw_offset(i) + (t % W)];
      if constexpr (i == 0) run0 = twist;
    }
    if constexpr (PIPE) {
      static_for<0, NP>([&](auto U) {
        constexpr int u = decltype(U)::value;
        if constexpr (i < S - 1) {
          cplx run = run0;
          static_for<1, R>([&](auto K) { constexpr int k = decltype(K)::value; run = cmul(run, b); v[u][k] = cmulc(v[u][k], run); });
        }
        cplx y[P];
        if constexpr (R == P) {
          small_dft<P, 1, +1>::run(v[u], y);
        } else {
          static_for<0, P / R>([&](auto Gp) { constexpr int g = decltype(Gp)::value; small_dft<R, 1, +1>::run(v[u] + g * R, y + g * R); });
        }
        cplx* ex = exch + u * G::EXCH_ELEMS;
        if constexpr (u == 0) bpre = tw[G::tw_offset(i - 1) + (t % G::weight(i - 1))];
        wsync();
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; ex[G::template ex<i - 1>(pass_addr<LOGM, P, i>(t, j))] = y[j]; });
        wsync();
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[u][j] = ex[G::template ex<i - 1>(pass_addr<LOGM, P, i - 1>(t, j))]; });
        DCTFHE_PIPE_SCHED_BARRIER();
      });
      return;
    }
    constexpr bool FUSED = DCTFHE_FUSED_INV && i > 0 && i < S - 1 && R == P && P == 8;
    [[maybe_unused]] cplx pw[3], cw8;
    if constexpr (FUSED) fused_idft_factors(b, pw, cw8);
#if DCTFHE_SHARED_TWIDDLES
    if constexpr (i < S - 1 && !FUSED) {
      cplx run = run0;
      if constexpr (i == 0) static_for<0, NP>([&](auto U) { constexpr int u = decltype(U)::value; v[u][0] = cmulc(v[u][0], run); });
      static_for<1, R>([&](auto K) {
        constexpr int k = decltype(K)::value;
        run = cmul(run, b);
        static_for<0, NP>([&](auto U) { constexpr int u = decltype(U)::value; v[u][k] = cmulc(v[u][k], run); });
      });
    }
#endif
    static_for<0, NP>([&](auto U) {
      constexpr int u = decltype(U)::value;
#if !DCTFHE_SHARED_TWIDDLES
      if constexpr (i < S - 1 && !FUSED) {
        cplx run = run0;
        if constexpr (i == 0) v[u][0] = cmulc(v[u][0], run);
        static_for<1, R>([&](auto K) { constexpr int k = decltype(K)::value; run = cmul(run, b); v[u][k] = cmulc(v[u][k], run); });
      }
#endif
      cplx y[P];
      if constexpr (FUSED) {
        fused_idft<P, 1, 0>::run(v[u], y, pw, cw8);
      } else if constexpr (R == P) {
        small_dft<P, 1, +1>::run(v[u], y);
      } else {
        static_for<0, P / R>([&](auto Gp) { constexpr int g = decltype(Gp)::value; small_dft<R, 1, +1>::run(v[u] + g * R, y + g * R); });
      }
      if constexpr (i > 0) {
        cplx* ex = exch + u * G::EXCH_ELEMS;
#if defined(DCTFHE_ABLATE_EXCH)
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[u][j] = y[(j + 1) % P]; });
#else
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; ex[G::template ex<i - 1>(pass_addr<LOGM, P, i>(t, j))] = y[j]; });
#endif
      } else {
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[u][j] = mul_root64<j*(64 / (4 * P)), -1>(y[j]); });
      }
      DCTFHE_FFT_SCHED_BARRIER();
    });
    if constexpr (i > 0) {
      constexpr int Wp = G::weight(i - 1);
      if constexpr (Wp > 64) tick.template at<7>();       // phase 7: twiddles + butterflies + scatter of the pass before the cross-wave gather
      if constexpr (Wp <= 64) wsync(); else sync();
      if constexpr (Wp > 64) tick.template at<8>();       // phase 8: waiting at the barrier before the cross-wave gather
      static_for<0, NP>([&](auto U) {
        constexpr int u = decltype(U)::value;
        const cplx* ex = exch + u * G::EXCH_ELEMS;
#if !defined(DCTFHE_ABLATE_EXCH)
        static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; v[u][j] = ex[G::template ex<i - 1>(pass_addr<LOGM, P, i - 1>(t, j))]; });
#endif
      });
      if constexpr (Wp > 64) tick.template at<9>();       // phase 9: the cross-wave gather
      if constexpr (Wp <= 64) wsync(); else sync();
      if constexpr (Wp > 64) tick.template at<10>();      // phase 10: waiting at the trailing barrier
      DCTFHE_FFT_SCHED_BARRIER();
    }
  });
}

// Frequency index of in-place position p after fft_forward: the value there is the polynomial evaluated at
// e^{i pi (1 - 4k) / N}.  Position digit i (weight W_i) holds frequency digit k_i, whose weight is R_0 * ... * R_{i-1}.
// Thread t leaves register j at position P*t + j.
template <int LOGM, int P>
HD int spectrum_freq(int p) {
  int k = 0, fw = 1;
  for (int i = 0; i < geom_passes(LOGM, P); i++) {
    const int W = geom_weight(LOGM, P, i), R = geom_radix(LOGM, P, i);
    k += ((p / W) % R) * fw;
    fw *= R;
  }
  return k;
}

// host-side fill of the twiddle table (G::TW_ELEMS entries)
template <int LOGM, int P>
inline void fill_twiddles(cplx* tw) {
  using G = fft_geom<LOGM, P>;
  const long double PI = 3.141592653589793238462643383279502884L;
  for (int i = 0; i + 1 < G::S; i++) {
    const int W = geom_weight(LOGM, P, i), R = geom_radix(LOGM, P, i);
    for (int m = 0; m < W; m++) {
      const long double a = -2.0L * PI * m / ((long double)W * R);
      tw[geom_tw_offset(LOGM, P, i) + m] = cmk((double)__builtin_cosl(a), (double)__builtin_sinl(a));
    }
  }
  for (int t = 0; t < G::T; t++) {
    const long double a = PI * t / (long double)(2 * G::M);
    tw[G::TW_TOTAL + t] = cmk((double)__builtin_cosl(a), (double)__builtin_sinl(a));
  }
}

// ---- double -> torus.  The inverse transform hands over y in units of the WHOLE torus (1.0 = 2^64: the Fourier key carries 2^-64 / M),
// so "mod 2^64" is the fractional part -- one v_fract_f64 -- and the words fall out of two conversions.  (Rounds 1-2 kept y in units of
// 2^-64 and peeled the words off with rint / fma / magic-constant additions: 6 f64-rate instructions per 32-bit accumulator word and
// 8 + 2 per 64-bit one, 12 % of the one-level N = 8192 kernel's issue slots; now 3 and 6.)
HD double fract64(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fract(x);            // v_fract_f64: x - floor(x), clamped below 1.0
#else
  const double r = x - __builtin_floor(x);     // (a tiny negative x would round to 1.0: the instruction clamps, so does this)
  return r < 1.0 ? r : 0x1.fffffffffffffp-1;
#endif
}
// the top 32 bits of the torus value: round(frac(y) * 2^32), ROUNDED TO NEAREST -- a truncation here would bias every accumulator
// coefficient the same way, and the bias of k N mask coefficients adds up coherently in the extracted phase.  frac(y) within 2^-33 of
// 1 saturates to 2^32 - 1 instead of wrapping to 0: one unit of 2^-32, far below every tier's noise, probability 2^-33 per word.
HD uint32_t f64_to_torus32(double y) {
  const double x = __builtin_fma(fract64(y), 4294967296.0, 0.5);
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)x;                          // v_cvt_u32_f64 (truncates, saturates)
#else
  return x >= 4294967296.0 ? 0xFFFFFFFFu : (uint32_t)x;
#endif
}
// all 64 bits: floor(frac(y) * 2^64) (the truncation sits at 2^-64 of the torus, forty bits below the f64 transform's own error)
HD uint64_t f64_to_torus(double y) {
  const double x = fract64(y) * 4294967296.0;                  // < 2^32, exact
  const uint32_t hi = (uint32_t)x;
  const uint32_t lo = (uint32_t)(fract64(x) * 4294967296.0);
  return ((uint64_t)hi << 32) | lo;
}

}  // namespace dctfhe
