// pbs_core.h -- the per-thread program of the programmable bootstrap (K4 mod-switch, K5 blind
// rotate, K6 sample extract of DESIGN.md), written once for the gfx950 kernel and for the host
// emulator in tests/emul.  Semantics follow oracle/tfhe_ref.h (which cites the reference call
// site homomorphic_eval.py:70 this replaces).
//
// One "group" of T = N/(2P) threads owns one ciphertext for the whole blind rotation:
//   * the GLWE accumulator ACC (k+1 polynomials, u64) lives in registers,
//       acc[p][r], r < 2P, is coefficient  c = t + T*r  of polynomial p
//     (r < P is the low half n = t + T*r, r >= P the high half n + M: exactly the pair a folded
//      complex point needs);
//   * a rotation X^a * ACC is a permutation across threads, done through the LDS `stage` array;
//   * the FFTs exchange through the LDS `exch` array (fft_core.h);
//   * the Fourier bootstrapping key is streamed from global memory (L2 / Infinity Cache), laid out
//     [i][row][q][j][t] so that a wave reads 1 KiB contiguous per instruction.
#pragma once
#include "fft_core.h"

namespace dctfhe {

#if defined(__HIP_DEVICE_COMPILE__)
#define DCTFHE_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#define DCTFHE_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))   // the value is the same in every lane of the wave
#define DCTFHE_DEVICE 1
#else
#define DCTFHE_SCHED_BARRIER() ((void)0)
#define DCTFHE_UNIFORM(x) (x)
#endif

// key loads in flight per thread between two waits: 8 spilled more than it hid, 2 exposed the latency
// (measured on T5 / B / T4: 4 is +10..20% over 8)
// accumulator polynomials kept in LDS rather than registers: -1 = per-kernel rule (pbs_geom::NL_AUTO), 0/1 force
#ifndef PBS_LDS_POLYS
#define PBS_LDS_POLYS (-1)
#endif

#ifndef PBS_KEY_BATCH
#define PBS_KEY_BATCH 4
#endif

#ifndef PBS_PAIR
#define PBS_PAIR 1
#endif
#ifndef PBS_MB_BAR_A
#define PBS_MB_BAR_A() DCTFHE_SCHED_BARRIER()
#endif
#ifndef PBS_MB_BAR_B
#define PBS_MB_BAR_B() ((void)0)     // no barrier after the fold: hipcc may hoist the next batch's loads over it (+3 %)
#endif
#ifndef PBS_MBG_BAR_B
#define PBS_MBG_BAR_B() DCTFHE_SCHED_BARRIER()
#endif
#ifndef PBS_STD_BAR_B
#define PBS_STD_BAR_B() DCTFHE_SCHED_BARRIER()
#endif

// one-level tiers keep the accumulator as 32-bit torus values (see pbs_geom::ACC32)
#ifndef PBS_ACC32
#define PBS_ACC32 1
#endif

#ifndef PBS_PF_DIST
#define PBS_PF_DIST 2
#endif

// MB = 1: two-bit blind rotation (see pbs_thread).
template <int LOGN, int K, int L, int P, int MB = 0>
struct pbs_geom {
  static constexpr int N = 1 << LOGN;
  static constexpr int LOGM = LOGN - 1;
  static constexpr int M = N / 2;
  using F = fft_geom<LOGM, P>;
  static constexpr int T = F::T;
  static constexpr int ROWS = (K + 1) * L;
  static constexpr size_t BSK_ELEMS_PER_KEYBIT = (size_t)ROWS * (K + 1) * M;  // complex
  // LDS per group (bytes).  The rotation stage aliases the FFT exchange buffer (every stage read is over before
  // the first exchange write of the same polynomial: barrier discipline in pbs_thread), and the first NL
  // accumulator polynomials live in LDS instead of registers (32 VGPRs each at P = 8).
  // measured (profiles/r01_exp_lds_acc.log): +11..13% where the register arrays spilled (k = 2, or three levels),
  // -2..3% where they did not (one level, k = 1) -- hence the rule.  The body polynomial K always stays in registers.
  // PAIR: the K+1 = 2 forward transforms of a one-level k = 1 bootstrap run interleaved (fft_forward_n), and so do
  // the two inverse ones: LDS scatter/gather of one polynomial overlaps the butterflies of the other, and the
  // barrier count per CMUX drops from 11 to 6.  Costs a second exchange buffer; the rotation stages alias the two.
  static constexpr bool PAIR = (PBS_PAIR || MB) && K == 1 && L == 1;
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(!MB || F::T >= 64, "two-bit kernels assume one ciphertext per wave");
#endif
  static constexpr int KEY_BLOCKS = MB ? 3 : 1;        // key-bit-sized blocks read per loop iteration
  static constexpr int RL = F::radix(F::S - 1);        // radix of the last pass
  static constexpr int NG = P / RL;                    // small transforms per thread in the last pass
  // k = 2 with two levels: both mask polynomials in LDS -- with one the kernel spilled 71 VGPRs (216 B/lane of scratch, 78 GB
  // written per launch of 12 288 ciphertexts: profiles/r02_pmc_tiers.txt)
  static constexpr int NL_AUTO = (K >= 2 && L >= 2) ? 2 : (K >= 2 || L >= 3) ? 1 : 0;
  static constexpr int NL = PBS_LDS_POLYS < 0 ? NL_AUTO : (PBS_LDS_POLYS < K ? PBS_LDS_POLYS : K);
  // ACC32: a one-level gadget rounds every accumulator coefficient to 2^-(beta+1) >= 2^-29 of the torus at each step anyway
  // (beta <= 28 enforced by the library for such tiers), so the accumulator of a one-level tier is kept as the top 32 bits:
  // the rounding of each update (2^-33) is far below that, the registers and the LDS traffic of the rotation halve, and
  // the f64 -> torus conversion drops from 8+3 to 5+1 instructions.  Multi-level tiers (convolution-grade outputs) keep 64.
  static constexpr bool ACC32 = PBS_ACC32 && L == 1;
  using acc_t = std::conditional_t<ACC32, uint32_t, uint64_t>;
  static constexpr int ACC_BYTES = (int)sizeof(acc_t);
  static constexpr int STAGE_BYTES = N * ACC_BYTES;
  static constexpr int EXCH_BYTES = F::EXCH_ELEMS * 16;
  // the stage only aliases the exchange buffer when an LDS-resident polynomial needs the room: aliasing costs one
  // extra barrier per register polynomial and iteration (measured -3% on the one-level N = 8192 kernel)
  static constexpr bool ALIAS = NL > 0 || PAIR;
  static_assert(!PAIR || (NL == 0 && EXCH_BYTES >= STAGE_BYTES), "pair mode stages each polynomial in its exchange buffer");
  static constexpr int STAGE_OFFSET = ALIAS ? 0 : EXCH_BYTES;
  static constexpr int SHARED_BYTES = PAIR ? 2 * EXCH_BYTES : MB ? EXCH_BYTES /* no rotation stage */ : ALIAS ? (EXCH_BYTES > STAGE_BYTES ? EXCH_BYTES : STAGE_BYTES) : EXCH_BYTES + STAGE_BYTES;
  static constexpr int ACCL_BYTES = NL * N * ACC_BYTES;
  static constexpr int GROUP_BYTES = SHARED_BYTES + ACCL_BYTES;
  // twiddle table in LDS, shared by all groups of a workgroup; the pair kernels of the two big rings are short of
  // LDS and leave the T twist bases in global memory (read once per bootstrap)
  static constexpr bool TWIST_LDS = !(PAIR && LOGN >= 12);
  static constexpr int TW_LDS_ELEMS = TWIST_LDS ? F::TW_ELEMS : F::TW_TOTAL;
  // two-bit kernels: zeta^m = e^{i pi m / N}, m < 2N, as the product of two small LDS tables (m = hi * 2^ZLO + lo) instead of a gather
  // from a 2N-entry table in global memory: the general form needs 8 * NG monomial factors per gadget row (48 random 16-byte
  // gathers per iteration at N = 2048), and the vector memory path, not the ALU, is what those kernels wait for
  static constexpr int ZLO = MB ? (LOGN + 2) / 2 : 0, ZHI = MB ? LOGN + 1 - ZLO : 0;
  static constexpr int ZLUT_ELEMS = MB ? (1 << ZLO) + (1 << ZHI) : 0;
  static constexpr int TW_BYTES = (TW_LDS_ELEMS + ZLUT_ELEMS) * 16;
};

// signed gadget decomposition, closest-representable rounding; digs[lev], lev 0 most significant, digits in [-B/2, B/2).
// Carries come out of one addition: with B/2 added at every digit position the plain base-B digits minus B/2 are the
// balanced ones (the balanced representation is unique mod B^L).  One level is the arithmetic shift of the rounded
// high word.
template <int L>
HD void decompose(uint64_t v, int beta, int32_t* digs) {
  if constexpr (L == 1) {
    const uint32_t hi = (uint32_t)(v >> 32) + (1u << (31 - beta));      // beta <= 31: rounding touches the high word only
    digs[0] = (int32_t)hi >> (32 - beta);
  } else {
    const int total = L * beta;
    const uint64_t B = 1ULL << beta, half = B >> 1, mask = B - 1;
    uint64_t offs = 0;
    static_for<0, L>([&](auto Lv) { offs |= half << (beta * decltype(Lv)::value); });
    const uint64_t x = ((v + (1ULL << (63 - total))) >> (64 - total)) + offs;
    static_for<0, L>([&](auto Lv) {
      constexpr int lev = decltype(Lv)::value;
      digs[lev] = (int32_t)((x >> (beta * (L - 1 - lev))) & mask) - (int32_t)half;
    });
  }
}

// one level from a 32-bit accumulator word
template <int L>
HD void decompose(uint32_t v, int beta, int32_t* digs) {
  static_assert(L == 1, "32-bit accumulators are for one-level tiers");
  digs[0] = (int32_t)(v + (1u << (31 - beta))) >> (32 - beta);
}
HD void acc_add(uint64_t& a, double d) { a += f64_to_torus(d); }
HD void acc_add(uint32_t& a, double d) { a += f64_to_torus32(d); }
HD uint64_t acc_wide(uint64_t a) { return a; }
HD uint64_t acc_wide(uint32_t a) { return (uint64_t)a << 32; }

// test-vector coefficient j (0 <= j < N) of the table T (2^w entries)
HD uint64_t testvec_coeff(const int64_t* table, int w, int N, int j) {
  const int box = N >> w, half = box >> 1;
  const int jj = j + half;
  return (jj < N) ? (uint64_t)table[jj / box] : (uint64_t)0 - (uint64_t)table[0];
}

// e^{2 pi i k / 8} for a wave-uniform k: constant address space, so the device reads it with scalar loads
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ const double dctfhe_cos8[8] = {1.0, 0.70710678118654752440, 0.0, -0.70710678118654752440,
                                                       -1.0, -0.70710678118654752440, 0.0, 0.70710678118654752440};
HD cplx root8(uint32_t k) { return cmk(dctfhe_cos8[k & 7], dctfhe_cos8[(k + 6) & 7]); }
#else
HD cplx root8(uint32_t k) { return root64(8 * (int)(k & 7)); }
#endif

// Load base[lane] where `base` is the same in every lane: on the device the base goes through readfirstlane into scalar
// registers and the load takes the scalar-base + 32-bit lane-offset form -- no 64-bit VALU address arithmetic (and no
// carry-hazard nop) per load.  Global address space is stated explicitly so that it does not degrade to a flat load.
// the same pointer, known to the compiler to be wave-uniform (two readfirstlanes once, free afterwards)
HD const cplx* make_uniform(const cplx* p) {
#if defined(DCTFHE_DEVICE)
  const uint64_t v = (uint64_t)p;
  return reinterpret_cast<const cplx*>(((uint64_t)DCTFHE_UNIFORM((uint32_t)(v >> 32)) << 32) | DCTFHE_UNIFORM((uint32_t)v));
#else
  return p;
#endif
}
HD cplx load_uniform_base(const cplx* base, unsigned lane) {
#if defined(DCTFHE_DEVICE)
  typedef double v2d __attribute__((ext_vector_type(2)));
  const uint64_t v = (uint64_t)base;
  const uint32_t lo = DCTFHE_UNIFORM((uint32_t)v), hi = DCTFHE_UNIFORM((uint32_t)(v >> 32));
  const __attribute__((address_space(1))) v2d* g = (const __attribute__((address_space(1))) v2d*)(((uint64_t)hi << 32) | lo);
  const v2d x = g[lane];
  return cmk(x.x, x.y);
#else
  return base[lane];
#endif
}

#if defined(__HIP_DEVICE_COMPILE__) && defined(DCTFHE_PHASE_TIMERS) || defined(DCTFHE_PHASE_TIMERS) && defined(__HIPCC__)
__device__ unsigned long long dctfhe_phase_ticks[16 * 12];      // timing experiments: ticks per phase of the waves of workgroup DCTFHE_PHASE_TIMERS
#endif
struct pbs_args {
  const uint64_t* ct_small;   // this ciphertext: n+1 words
  int n;
  int beta;
  const cplx* bsk;            // Fourier key, [n][ROWS][K+1][P][T] complex, 1/M folded in
  const int64_t* table;       // 2^w entries, output encoding
  int w;
  uint64_t* out;              // D_out+1 words
  int D_out;
  int accumulate;             // 0: out = extract(ACC) (mask beyond K*N zeroed); 1: out += extract(ACC)
  uint64_t body_add;          // added to the body word (accumulate mode: the "- v" of a bit step)
  int bsk_wrap;               // 0 = off; >0: key bit i reads BSK[i % bsk_wrap] (cache experiments only)
  const cplx* wtab;           // MB: e^{i pi m / N}, m < 2N (monomials in the Fourier domain), then e^{2 pi i k / 8}, k < 8
  const cplx* zlut;           // MB, device: the same roots as two LDS tables, lo[2^ZLO] then hi[2^ZHI] (nullptr: gather from wtab)
  const cplx* twist;          // the T twist bases e^{i pi t/N} (entries TW_TOTAL.. of the twiddle table; LDS or global)
  int pf_rank, pf_parts;      // L2 warm-up: this workgroup touches part pf_rank of pf_parts of BSK[i + PF_DIST]
  const cplx* kring;          // KLDS: ring of key tiles in LDS shared by the KW waves of the workgroup (generic pointer for the reads)
  uint32_t kring_lds;         // ... its LDS byte address (what the LDS-DMA instruction takes in M0)
  int kwave;                  // ... this wave's index among the KW
};

// The whole bootstrap for one ciphertext, executed by thread t of its group.
//
// MB = 1, two key bits per iteration (Zhou et al. style unrolling, arranged so that the transforms are shared):
//   X^{a1 s1 + a2 s2} = 1 + s1(1-s2) (X^{a1} - 1) + (1-s1)s2 (X^{a2} - 1) + s1 s2 (X^{a1+a2} - 1)
//   ACC += sum_w (X^{e_w} - 1) * (GGSW(b_w) [x] ACC),   (b_w, e_w) as above.
// The gadget decomposition is of ACC itself -- one decomposition, K+1 forward transforms and K+1 inverse transforms per
// PAIR of key bits, no rotation through LDS -- and the monomials act in the Fourier domain, where X^e is the pointwise
// factor zeta^e, zeta = e^{i pi (1-4k)/N} the evaluation point (spectrum_freq).  Price: three key blocks per pair
// instead of two, and the key/FFT noise of three products scaled by |X^e - 1|^2 = 2 (dctfhe/params.py prices it).
// KLDS > 0 (device, general two-bit form, one wave per ciphertext): the KW waves of a workgroup -- KW ciphertexts -- walk the key in
// lock-step and share every key tile through LDS: each (gadget row, point j) step of 3 (K+1) key vectors is fetched ONCE per workgroup
// by LDS-DMA (global_load_lds_dwordx4: every wave brings 1/KW of it, no registers), KLDS - 1 steps ahead into a ring of KLDS tiles,
// and read by all KW waves with ds_read_b128.  Without it every wave pulls its own copy of the same lines through the CU's vector L1
// (64 B/clk): 8 waves x 221 KB per iteration on the k = 2 one-level tier, the path that kernel saturated (profiles/r02_exp_ablations.log).
// Protocol per step s: wait until this wave's DMAs of step s have landed (counted vmcnt: the younger ones stay in flight), workgroup
// barrier (=> everybody's part of step s is there, and everybody is done reading step s - 1), issue the DMAs of step s + KLDS - 1 into
// the tile step s - 1 used, read step s.
template <int LOGN, int K, int L, int P, int MB = 0, int KLDS = 0, int KW = 1, class Sync, class WSync>
HD void pbs_thread(const pbs_args& A, int t, const cplx* tw, uint64_t* stage_raw, cplx* exch, uint64_t* accl_raw, uint32_t* pf_dump, Sync&& sync, WSync&& wsync) {
  using G = pbs_geom<LOGN, K, L, P, MB>;
  constexpr int N = G::N, M = G::M, T = G::T, NL = G::NL;
  using acc_t = typename G::acc_t;
  acc_t* const stage = reinterpret_cast<acc_t*>(stage_raw);
  acc_t* const accl = reinterpret_cast<acc_t*>(accl_raw);
  const int n = A.n;
  const int msh = 64 - LOGN - 2;
  const cplx twist = A.twist[t];

  // L2 warm-up geometry: this workgroup owns lines [pf_line0, pf_line0 + pf_per) of every key bit.
  // CONTRACT (checked by launch_pbs on the host, asserted by tests/emul for every shipped geometry): iteration i touches
  // bytes of iteration i + PBS_PF_DIST of the key, so the key buffer carries PBS_PF_DIST iterations (KEY_BLOCKS blocks
  // each) of padding at its end; pf_parts is 0 (every thread re-touches line 0 of its iteration: no warm-up) or >= 8.
  // With bsk_wrap > 0 (cache experiments: the key buffer only holds bsk_wrap blocks) the pointer does not advance and
  // stays on block 0 -- it used to walk n iterations into a buffer of bsk_wrap blocks (the memory access fault recorded
  // in profiles/r01_exp_two_bit_rotation.log: a wrap case of the tools/exp_pbs.hip sweep with the warm-up on).
  constexpr int PF_LINES = (int)(G::KEY_BLOCKS * G::BSK_ELEMS_PER_KEYBIT * 16 / 128);
  constexpr int PF_ROUNDS = (PF_LINES / 8 + T - 1) / T;      // touches per thread per iteration; covers pf_parts >= 8
  const char* pf_ptr;
  const size_t pf_step = A.bsk_wrap > 0 ? 0 : (size_t)G::KEY_BLOCKS * G::BSK_ELEMS_PER_KEYBIT * 16;     // loop-invariant, wave-uniform
  {
    const int parts = A.pf_parts >= 8 ? A.pf_parts : PF_LINES;                 // pf_parts == 0: every thread re-touches line 0
    const int per = (PF_LINES + parts - 1) / parts;
    int line = A.pf_rank * per + (t < per ? t : per - 1);
    if (line > PF_LINES - 1 - (PF_ROUNDS - 1) * T) line = PF_LINES - 1 - (PF_ROUNDS - 1) * T;
    if (line < 0) line = 0;
    pf_ptr = reinterpret_cast<const char*>(A.bsk) + (size_t)PBS_PF_DIST * pf_step + (size_t)line * 128;
  }
  acc_t acc[K + 1][2 * P];
  {  // ACC = X^{-b~} * TV (trivial GLWE)
    const uint32_t bt = (uint32_t)(((A.ct_small[n] >> msh) + 1) >> 1) & (2 * N - 1);
    static_for<0, K>([&](auto Pp) {
      constexpr int p = decltype(Pp)::value;
      static_for<0, 2 * P>([&](auto R) {
        constexpr int r = decltype(R)::value;
        if constexpr (p < NL) accl[p * N + t + T * r] = 0; else acc[p][r] = 0;
      });
    });
    if constexpr (NL > 0) sync();
    static_for<0, 2 * P>([&](auto R) {
      constexpr int r = decltype(R)::value;
      const uint32_t idx = ((uint32_t)(t + T * r) + bt) & (2 * N - 1);
      const uint64_t v = testvec_coeff(A.table, A.w, N, (int)(idx & (N - 1)));
      acc[K][r] = (acc_t)(((idx & N) ? (uint64_t)0 - v : v) >> (64 - 8 * G::ACC_BYTES));
    });
  }

  // MB: root exponent (1 - 4k) mod 2N of this thread's first point in each small transform of the last pass; point j' of
  // small transform g sits at exponent ulow[g] - j' * (2N / RL)
  // -- ONE value per thread: small transform g starts NG*t + g in the digit of the second-last pass (no carry: NG*t is a multiple
  // of NG), whose frequency weight is M / (P * RL), so ulow[g] = ulow0 - g * USTEP.  (Holding all NG of them cost the general
  // three-level kernel four spilled registers, reloaded through the same in-order vmcnt queue as its key loads.)
  uint32_t ulow0 = 0;
  constexpr uint32_t USTEP = 4u * (uint32_t)(M / (P * G::RL));
  if constexpr (MB) ulow0 = (uint32_t)(1 - 4 * spectrum_freq<G::LOGM, P>(P * t)) & (2 * N - 1);

#if defined(DCTFHE_DEVICE)
  constexpr bool KL = KLDS > 0;
#else
  constexpr bool KL = false;       // the host emulator reads the key where it lies
#endif
  constexpr int K_SPI = G::ROWS * P;                       // steps (gadget row, point j) per iteration
  constexpr int K_STEP = 3 * (K + 1) * T;                  // complex values per step: blocks w x output polynomials q x T lanes
  constexpr int K_LPW = KL ? 3 * (K + 1) * 64 / KW : 64;   // 16-byte lanes per wave and step
  constexpr int K_CH = (K_LPW + 63) / 64, K_CHL = K_LPW / K_CH;       // DMA instructions per wave and step, active lanes in each
  static_assert(!KL || (MB && !G::PAIR && L == 1 && T == 64 && KLDS >= 2 && (3 * (K + 1) * 64) % KW == 0 && K_LPW % K_CH == 0 && K_SPI % KLDS == 0),
                "key tiles through LDS: general two-bit form, one wave per ciphertext, tile count divides the steps per iteration");
  uint32_t koff[K_CH];       // byte offset of this lane's 16 bytes of chunk i inside a pair's key, without the step's (row, j) part
#if defined(DCTFHE_DEVICE)
  [[maybe_unused]] const uint32_t kwave = KL ? DCTFHE_UNIFORM(A.kwave) : 0;
  [[maybe_unused]] auto kissue = [&](const cplx* key_pair /* wave-uniform */, auto RowJ, auto Buf) {
    constexpr int s = decltype(RowJ)::value, row = s / P, j = s % P, buf = decltype(Buf)::value;
    const uint64_t base = (uint64_t)(key_pair + ((size_t)row * (K + 1) * M + (size_t)j * T));
    static_for<0, K_CH>([&](auto I) {
      constexpr int i = decltype(I)::value;
      const uint32_t dst = A.kring_lds + (uint32_t)((buf * K_STEP + (int)(kwave * K_CH + i) * K_CHL) * 16);
      const uint64_t mask = K_CHL >= 64 ? ~0ull : ((1ull << K_CHL) - 1);
      uint32_t keep_m0; uint64_t keep_exec;
      const uint32_t vo = koff[i];
      const uint64_t sb = base;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_mov_b64 exec, %5\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0), "=&s"(keep_exec) : "v"(vo), "s"(sb), "s"(dst), "s"(mask) : "memory");
    });
  };
  if constexpr (KL) {
    static_for<0, K_CH>([&](auto I) {
      constexpr int i = decltype(I)::value;
      const uint32_t pos = (kwave * K_CH + i) * K_CHL + (uint32_t)t;          // 16-byte slot of this lane in the step's tile (t < K_CHL)
      const uint32_t vi = pos / 64, tt = pos % 64, w = vi / (K + 1), q = vi % (K + 1);
      koff[i] = ((w * G::ROWS * (K + 1) + q) * M + tt) * 16;
    });
    // steps 0 .. KLDS-2 of the first iteration
    static_for<0, KLDS - 1>([&](auto S) { kissue(make_uniform(A.bsk), S, S); });
  }
#endif

  // zeta^m: on the device the product of the two LDS tables, on the host emulator (zlut == nullptr) the full table
  auto zeta = [&](uint32_t m) -> cplx {
#if defined(DCTFHE_DEVICE) && !defined(DCTFHE_NO_ZLUT)      // (DCTFHE_NO_ZLUT: timing experiments, tools/exp_pbs.hip)
    return cmul(A.zlut[(1 << G::ZLO) + (m >> G::ZLO)], A.zlut[m & ((1u << G::ZLO) - 1)]);
#elif defined(DCTFHE_DEVICE)
    return A.wtab[m];
#else
    return A.zlut ? cmul(A.zlut[(1 << G::ZLO) + (m >> G::ZLO)], A.zlut[m & ((1u << G::ZLO) - 1)]) : A.wtab[m];
#endif
  };
#if defined(__HIP_DEVICE_COMPILE__) && defined(DCTFHE_PHASE_TIMERS)
  phase_clock tick;
  tick.start();
#else
  no_tick tick;
#endif
  for (int i = 0; i < n; i += (MB ? 2 : 1)) {
    const uint32_t a = (uint32_t)(((A.ct_small[i] >> msh) + 1) >> 1) & (2 * N - 1);
    const cplx* bsk_i = make_uniform(A.bsk + (size_t)(A.bsk_wrap > 0 ? i % A.bsk_wrap : i) * G::BSK_ELEMS_PER_KEYBIT);
    cplx out[K + 1][P];
    if constexpr (MB && !G::PAIR) {
      // general (k, l): one polynomial at a time through the single exchange buffer; the monomial factors are rebuilt per
      // gadget row (holding them for all P points would take 96 registers).  ACC polynomials that live in LDS are only
      // ever touched by their owner here -- there is no rotation -- so no barrier guards them.
      const uint32_t a2 = (uint32_t)(((A.ct_small[i + 1] >> msh) + 1) >> 1) & (2 * N - 1);
      const uint32_t au = DCTFHE_UNIFORM(a), a2u = DCTFHE_UNIFORM(a2);
      const uint32_t e1 = a * ulow0, e2 = a2 * ulow0;
      const cplx* key = A.bsk + (size_t)(3 * (i >> 1)) * G::BSK_ELEMS_PER_KEYBIT;
      static_for<0, K + 1>([&](auto Pp) {
        constexpr int p = decltype(Pp)::value;
        static_assert(L <= 3, "packing holds two deferred digits");
        uint32_t packed[2 * P];
        double first[2 * P];
        static_for<0, 2 * P>([&](auto R) {
          constexpr int r = decltype(R)::value;
          acc_t own;
          if constexpr (p < NL) own = accl[p * N + t + T * r]; else own = acc[p][r];
          int32_t dg[L];
          decompose<L>(own, A.beta, dg);
          first[r] = (double)dg[0];
          uint32_t pk = 0;
          if constexpr (L > 1) pk = (uint32_t)(uint16_t)(int16_t)dg[1];
          if constexpr (L > 2) pk |= (uint32_t)(uint16_t)(int16_t)dg[2] << 16;
          packed[r] = pk;
        });
        static_for<0, L>([&](auto Lv) {
          constexpr int lev = decltype(Lv)::value;
          constexpr int row = p * L + lev;
          cplx v[P];
          static_for<0, P>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (lev == 0) v[j] = cmk(first[j], first[P + j]);
            else v[j] = cmk((double)(int16_t)(packed[j] >> (16 * (lev - 1))), (double)(int16_t)(packed[P + j] >> (16 * (lev - 1))));
          });
          fft_forward<G::LOGM, P>(v, t, tw, twist, exch, sync, wsync);
          tick.template at<4>();                          // (general form) phase 4: decomposition + one forward transform
          cplx zb1 = cmk(1.0, 0.0), zb2 = cmk(1.0, 0.0);
          static_for<0, P>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int g = j / G::RL, jp = j % G::RL;
            if constexpr (jp == 0) {
              zb1 = zeta((e1 - au * (USTEP * (uint32_t)g)) & (2 * N - 1));      // a * ulow[g]; the second product is wave-uniform
              zb2 = zeta((e2 - a2u * (USTEP * (uint32_t)g)) & (2 * N - 1));
            }
            cplx z1 = zb1, z2 = zb2;
            if constexpr (jp > 0) {
              z1 = cmul(z1, root8((0u - au * (uint32_t)jp) * (8 / G::RL)));
              z2 = cmul(z2, root8((0u - a2u * (uint32_t)jp) * (8 / G::RL)));
            }
            cplx kk[3][K + 1];
#if defined(DCTFHE_DEVICE)
            if constexpr (KL) {
              constexpr int s = row * P + j, ahead = s + KLDS - 1;
              // this wave's DMAs of step s have landed when at most those of the KLDS - 2 younger steps are outstanding; its own
              // reads of step s - 1 are over (lgkmcnt) before anybody may overwrite that tile
              asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((KLDS - 2) * K_CH) : "memory");
              __builtin_amdgcn_s_barrier();
              asm volatile("" ::: "memory");
              kissue(make_uniform(key + (ahead >= K_SPI ? (size_t)3 * G::BSK_ELEMS_PER_KEYBIT : 0)), std::integral_constant<int, ahead % K_SPI>{},
                     std::integral_constant<int, ahead % KLDS>{});
              const cplx* tile = A.kring + (s % KLDS) * K_STEP;
              static_for<0, 3>([&](auto Ww) {
                constexpr int w = decltype(Ww)::value;
                static_for<0, K + 1>([&](auto Q) { constexpr int q = decltype(Q)::value; kk[w][q] = tile[(w * (K + 1) + q) * 64 + t]; });
              });
            } else
#endif
            {
            static_for<0, 3>([&](auto Ww) {
              constexpr int w = decltype(Ww)::value;
              static_for<0, K + 1>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
#if defined(DCTFHE_ABLATE_BSK)
                kk[w][q] = cmk(1.0 + w, 0.5 * q + row);
#else
                kk[w][q] = load_uniform_base(key + (((size_t)w * G::ROWS * (K + 1) + (size_t)row * (K + 1) + q) * M + j * T), (unsigned)t);    // uniform base + lane index
#endif
              });
            });
            }
            DCTFHE_SCHED_BARRIER();
            const cplx m1 = cmk(z1.re - 1.0, z1.im), m2 = cmk(z2.re - 1.0, z2.im);
            cplx m12 = cmul(z1, z2); m12.re -= 1.0;
            static_for<0, K + 1>([&](auto Q) {
              constexpr int q = decltype(Q)::value;
              const cplx bundle = cfma(m12, kk[2][q], cfma(m2, kk[1][q], cmul(m1, kk[0][q])));
              if constexpr (row == 0) out[q][j] = cmul(v[j], bundle); else out[q][j] = cfma(v[j], bundle, out[q][j]);
            });
            PBS_MBG_BAR_B();
          });
          tick.template at<5>();                          // phase 5: the products with the key of one gadget row
        });
      });
    } else if constexpr (MB) {
      const uint32_t a2 = (uint32_t)(((A.ct_small[i + 1] >> msh) + 1) >> 1) & (2 * N - 1);
      const cplx* key = A.bsk + (size_t)(3 * (i >> 1)) * G::BSK_ELEMS_PER_KEYBIT;     // blocks: b1 = s1(1-s2), b2 = (1-s1)s2, b12 = s1 s2
      cplx v[2][P];
      static_for<0, 2>([&](auto Pp) {
        constexpr int p = decltype(Pp)::value;
        static_for<0, 2 * P>([&](auto R) {
          constexpr int r = decltype(R)::value;
          int32_t dg[1];
          decompose<1>(acc[p][r], A.beta, dg);
          if constexpr (r < P) v[p][r].re = (double)dg[0]; else v[p][r - P].im = (double)dg[0];
        });
      });
      fft_forward_n<G::LOGM, P, 2>(v, t, tw, twist, exch, sync, wsync, tick);
      // zeta^{a} at this thread's points: one 16-byte gather per exponent and small transform from the 2N-entry root
      // table; inside a small transform the points are RL-th roots of unity apart, zeta_{j'} = zeta_0 * e^{-2 pi i a j'/RL}
      // -- a factor that only depends on a (the same for the whole ciphertext), read from the 8-entry table below.
      // (a wave holds threads of one ciphertext only: T >= 64 on the device)
      const uint32_t au = DCTFHE_UNIFORM(a), a2u = DCTFHE_UNIFORM(a2);
      const uint32_t e1 = a * ulow0, e2 = a2 * ulow0;
      cplx zb1 = cmk(1.0, 0.0), zb2 = cmk(1.0, 0.0);
      static_for<0, P>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int g = j / G::RL, jp = j % G::RL;
        if constexpr (jp == 0) {
          zb1 = zeta((e1 - au * (USTEP * (uint32_t)g)) & (2 * N - 1));
          zb2 = zeta((e2 - a2u * (USTEP * (uint32_t)g)) & (2 * N - 1));
        }
        cplx z1 = zb1, z2 = zb2;
        if constexpr (jp > 0) {
          z1 = cmul(z1, root8((0u - au * (uint32_t)jp) * (8 / G::RL)));      // wave-uniform: scalar loads
          z2 = cmul(z2, root8((0u - a2u * (uint32_t)jp) * (8 / G::RL)));
        }
        static_for<0, 2>([&](auto Q) {
          constexpr int q = decltype(Q)::value;
          cplx kk[3][2];
          static_for<0, 3>([&](auto Ww) {
            constexpr int w = decltype(Ww)::value;
            static_for<0, 2>([&](auto Rr) {
              constexpr int r = decltype(Rr)::value;
#if defined(DCTFHE_ABLATE_BSK)
              kk[w][r] = cmk(1.0 + w, 0.5 * q + r);
#else
              kk[w][r] = load_uniform_base(key + ((size_t)((w * 2 + r) * 2 + q) * M + j * T), (unsigned)t);      // uniform base + zero-extended lane index: scalar address math
#endif
            });
          });
          PBS_MB_BAR_A();
          const cplx m1 = cmk(z1.re - 1.0, z1.im), m2 = cmk(z2.re - 1.0, z2.im);
          cplx m12 = cmul(z1, z2); m12.re -= 1.0;
          static_for<0, 2>([&](auto Rr) {
            constexpr int r = decltype(Rr)::value;
            const cplx bundle = cfma(m12, kk[2][r], cfma(m2, kk[1][r], cmul(m1, kk[0][r])));
            if constexpr (r == 0) out[q][j] = cmul(v[0][j], bundle); else out[q][j] = cfma(v[1][j], bundle, out[q][j]);
          });
          PBS_MB_BAR_B();
        });
      });
    } else if constexpr (G::PAIR) {
      // both accumulator polynomials go through their stages at once (stage p = exchange buffer p; everybody is past
      // the last gather of the previous inverse transforms: their trailing barrier), one barrier, then the rotated
      // reads; the leading barrier of the forward transforms covers those reads.
      constexpr int EX = G::F::EXCH_ELEMS * (16 / G::ACC_BYTES);   // accumulator words per exchange buffer
      static_for<0, 2>([&](auto Pp) {
        constexpr int p = decltype(Pp)::value;
        static_for<0, 2 * P>([&](auto R) { constexpr int r = decltype(R)::value; stage[p * EX + t + T * r] = acc[p][r]; });
      });
      sync();
      cplx v[2][P];
      static_for<0, 2>([&](auto Pp) {
        constexpr int p = decltype(Pp)::value;
        static_for<0, 2 * P>([&](auto R) {
          constexpr int r = decltype(R)::value;
          const uint32_t src = ((uint32_t)(t + T * r) - a) & (2 * N - 1);
          const acc_t x = stage[p * EX + (src & (N - 1))];
          const acc_t m = (acc_t)0 - (acc_t)((src >> LOGN) & 1);      // all ones where the rotation wraps: -x = (x ^ m) - m
          int32_t dg[1];
          decompose<1>((acc_t)((x ^ m) - m - acc[p][r]), A.beta, dg);
          if constexpr (r < P) v[p][r].re = (double)dg[0]; else v[p][r - P].im = (double)dg[0];
        });
      });
      fft_forward_n<G::LOGM, P, 2>(v, t, tw, twist, exch, sync, wsync);
      constexpr int KB = (P > PBS_KEY_BATCH) ? PBS_KEY_BATCH : P;
      static_for<0, 2>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        static_for<0, P / KB>([&](auto Hb) {
          constexpr int j0 = decltype(Hb)::value * KB;
          cplx k0[KB], k1[KB];
          static_for<0, KB>([&](auto J) {
            constexpr int j = decltype(J)::value;
#if defined(DCTFHE_ABLATE_BSK)
            k0[j] = cmk(1.0 + j, 0.5 * q); k1[j] = cmk(0.5 * q, 1.0 + j);
#else
            k0[j] = load_uniform_base(bsk_i + ((size_t)(0 * 2 + q) * M + (j0 + j) * T), (unsigned)t);
            k1[j] = load_uniform_base(bsk_i + ((size_t)(1 * 2 + q) * M + (j0 + j) * T), (unsigned)t);
#endif
          });
          DCTFHE_SCHED_BARRIER();
          static_for<0, KB>([&](auto J) {
            constexpr int j = decltype(J)::value;
            // same order of operations as the unpaired path: ((0 + v0 k0) + v1 k1)
            out[q][j0 + j] = cfma(v[1][j0 + j], k1[j], cfma(v[0][j0 + j], k0[j], cmk(0.0, 0.0)));
          });
          DCTFHE_SCHED_BARRIER();
        });
      });
    } else {
    static_for<0, K + 1>([&](auto Q) { static_for<0, P>([&](auto J) { out[decltype(Q)::value][decltype(J)::value] = cmk(0.0, 0.0); }); });

    static_for<0, K + 1>([&](auto Pp) {
      constexpr int p = decltype(Pp)::value;
      // rotate polynomial p through LDS: every coefficient is read (rotated) and decomposed once; level 0 goes
      // straight to the transform, the other digits (<= 16 bits each: beta <= 16 whenever l >= 2) wait packed in
      // one register per coefficient.  (Re-reading + re-decomposing per level cost 15% more: integer VALU.)
      // LDS-resident polynomials are rotated straight out of their home array; the others go through the stage,
      // which shares memory with the exchange buffer: barrier before the (cross-wave) stage write so that nobody
      // is still gathering there, barrier after it; the next transform's leading barrier covers the reads.
      if constexpr (p >= NL) {
        if constexpr (G::ALIAS) sync();
        static_for<0, 2 * P>([&](auto R) { constexpr int r = decltype(R)::value; stage[t + T * r] = acc[p][r]; });
        sync();
      }
      // one rotated read + one full decomposition per coefficient; the digits of levels >= 1 wait packed in a register
      static_assert(L <= 3, "packing holds two deferred digits");
      uint32_t packed[2 * P];
      double first[2 * P];
      static_for<0, 2 * P>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const uint32_t src = ((uint32_t)(t + T * r) - a) & (2 * N - 1);
        acc_t x, own;
        if constexpr (p < NL) { x = accl[p * N + (src & (N - 1))]; own = accl[p * N + t + T * r]; }
        else                  { x = stage[src & (N - 1)]; own = acc[p][r]; }
        if (src & N) x = (acc_t)0 - x;     // (a branch-free (x ^ m) - m here makes hipcc spill 200+ bytes more per lane)
        int32_t dg[L];
        decompose<L>((acc_t)(x - own), A.beta, dg);
        first[r] = (double)dg[0];
        uint32_t pk = 0;
        if constexpr (L > 1) pk = (uint32_t)(uint16_t)(int16_t)dg[1];
        if constexpr (L > 2) pk |= (uint32_t)(uint16_t)(int16_t)dg[2] << 16;
        packed[r] = pk;
      });
      static_for<0, L>([&](auto Lv) {
        constexpr int lev = decltype(Lv)::value;
        cplx v[P];
        static_for<0, P>([&](auto J) {
          constexpr int j = decltype(J)::value;
          if constexpr (lev == 0) v[j] = cmk(first[j], first[P + j]);
          else v[j] = cmk((double)(int16_t)(packed[j] >> (16 * (lev - 1))), (double)(int16_t)(packed[P + j] >> (16 * (lev - 1))));
        });
        fft_forward<G::LOGM, P>(v, t, tw, twist, exch, sync, wsync);
        const cplx* row = bsk_i + (size_t)(p * L + lev) * (K + 1) * M;
        // Key loads are issued as one batch per output polynomial and only then consumed: left to itself hipcc
        // (at this register pressure) emits load; s_waitcnt vmcnt(0); fma -- 16 serialized L2 round trips per row
        // (measured 93 -> 72 ms per launch).  Issuing the first batch under the last FFT pass was tried and lost
        // (more scratch traffic than latency hidden).
        // key loads in flight per thread and batch: 4 everywhere but on the one-level k = 2 kernel, whose 32-bit accumulators
        // leave room for 8 (41.3 -> 37.4 ms per launch; 8 still spills on the two-level k = 2 kernel: 92 -> 101 ms)
        constexpr int KBW = (K >= 2 && L == 1 && G::ACC32) ? 8 : PBS_KEY_BATCH;
        constexpr int KB = (P > KBW) ? KBW : P;
        static_for<0, K + 1>([&](auto Q) {
          constexpr int q = decltype(Q)::value;
          static_for<0, P / KB>([&](auto Hb) {
            constexpr int j0 = decltype(Hb)::value * KB;
            cplx kb[KB];
            static_for<0, KB>([&](auto J) {
              constexpr int j = decltype(J)::value;
#if defined(DCTFHE_ABLATE_BSK)   // timing experiments only (tools/exp_pbs.hip): no key traffic
              kb[j] = cmk(1.0 + j, 0.5 * q);
#else
              kb[j] = load_uniform_base(row + ((size_t)q * M + (j0 + j) * T), (unsigned)t);
#endif
            });
            DCTFHE_SCHED_BARRIER();
            static_for<0, KB>([&](auto J) { constexpr int j = decltype(J)::value; out[q][j0 + j] = cfma(v[j0 + j], kb[j], out[q][j0 + j]); });
            PBS_STD_BAR_B();
          });
        });
      });
    });
    }  // !PAIR

    // L2 warm-up.  Every CU walks the same key in near lock-step, so without help each key line is an HBM
    // miss that all CUs of the XCD wait on together (measured: 128 ms -> 94 ms per launch with the key
    // cache-resident).  Each workgroup therefore touches 1/pf_parts of the key two iterations ahead, one
    // dword per 128-byte line; issued here, right after this iteration's last key load and before the two
    // inverse transforms, the touches are long retired when the next vmcnt wait comes (vmcnt is in-order).
    // Only for the kernels whose waves share a workgroup barrier (T > 64): they walk the key in lock-step, which is what makes
    // a line touched by one workgroup a hit for the others.  One-wave-per-ciphertext kernels (T <= 64: Ba, B) free-run; there
    // the touches were pure overhead (round 2: Ba 66.6 -> 64.3 ms per launch of 16 384, B 131.1 -> 125.9 without them).
    if constexpr (T > 64) {
       // straight-line on purpose: any branch or select chain here makes hipcc (ROCm 7.2) demote the register
       // arrays to scratch (2.4 KB/lane).  The key buffer carries PBS_PF_DIST zero key bits of padding at its end.
      uint32_t acc_pf = 0;
      static_for<0, PF_ROUNDS>([&](auto Rr) { acc_pf ^= *reinterpret_cast<const uint32_t*>(pf_ptr + (size_t)decltype(Rr)::value * T * 128); });
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" ::"v"(acc_pf));        // the touches only have to be issued
#else
      pf_dump[t] = acc_pf;
#endif
      pf_ptr += pf_step;
    }

    tick.template at<5>();                                // phase 5: the products with the key (and the L2 warm-up touches)
    if constexpr (G::PAIR) fft_inverse_n<G::LOGM, P, 2>(out, t, tw, twist, exch, sync, wsync, tick);
    static_for<0, K + 1>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      if constexpr (!G::PAIR) fft_inverse<G::LOGM, P>(out[q], t, tw, twist, exch, sync, wsync);
      static_for<0, P>([&](auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (q < NL) {
          acc_add(accl[q * N + t + T * j], out[q][j].re);
          acc_add(accl[q * N + t + T * (P + j)], out[q][j].im);
        } else {
          acc_add(acc[q][j], out[q][j].re);
          acc_add(acc[q][P + j], out[q][j].im);
        }
      });
    });
    tick.template at<11>();                               // phase 11: (last pass of) the inverse transforms, accumulator update
  }

#if defined(__HIP_DEVICE_COMPILE__) && defined(DCTFHE_PHASE_TIMERS)
  tick.template at<0>();
  if (blockIdx.x == DCTFHE_PHASE_TIMERS && (threadIdx.x & 63) == 0)          // every wave of one workgroup reports: [wave][phase]
    for (int k = 0; k < 12; k++) dctfhe_phase_ticks[(threadIdx.x >> 6) * 12 + k] = tick.acc[k];
#endif
#if defined(DCTFHE_DEVICE)
  if constexpr (KL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tiles fetched ahead of the last step (key padding) land before the wave ends
#endif
  // sample extract (coefficient 0) straight into the output LWE ciphertext
  uint64_t* o = A.out;
  static_for<0, K>([&](auto Pp) {
    constexpr int p = decltype(Pp)::value;
    static_for<0, 2 * P>([&](auto R) {
      constexpr int r = decltype(R)::value;
      const int c = t + T * r;
      const int dst = p * N + (c == 0 ? 0 : N - c);
      uint64_t av;
      if constexpr (p < NL) av = acc_wide(accl[p * N + c]); else av = acc_wide(acc[p][r]);
      const uint64_t v = (c == 0) ? av : (uint64_t)0 - av;
      if (A.accumulate) o[dst] += v; else o[dst] = v;
    });
  });
  if (!A.accumulate)
    for (int j = K * N + t; j < A.D_out; j += T) o[j] = 0;
  if (t == 0) {
    if (A.accumulate) o[A.D_out] += acc_wide(acc[K][0]) + A.body_add; else o[A.D_out] = acc_wide(acc[K][0]) + A.body_add;
  }
}

// Forward transform of one standard-domain key polynomial into the device layout [j][t], with the 1/M of the inverse transform and
// the 2^-64 that puts the accumulator updates in units of the whole torus (fft_core.h, f64_to_torus*) folded in -- both exact scalings.
// Thread t of a T-thread group.
template <int LOGN, int P, class Sync, class WSync>
HD void key_poly_to_fourier(const uint64_t* poly, cplx* dst, int t, const cplx* tw, cplx* exch, Sync&& sync, WSync&& wsync) {
  constexpr int N = 1 << LOGN, M = N / 2;
  using F = fft_geom<LOGN - 1, P>;
  constexpr int T = F::T;
  cplx v[P];
  static_for<0, P>([&](auto J) {
    constexpr int j = decltype(J)::value;
    v[j] = cmk((double)(int64_t)poly[t + T * j], (double)(int64_t)poly[t + T * j + M]);
  });
  fft_forward<LOGN - 1, P>(v, t, tw, tw[F::TW_TOTAL + t], exch, sync, wsync);
  const double inv = 0x1p-64 / (double)M;
  static_for<0, P>([&](auto J) { constexpr int j = decltype(J)::value; dst[j * T + t] = cmk(v[j].re * inv, v[j].im * inv); });
}

}  // namespace dctfhe
