// measures FFT-induced error of the external product: oracle radix-2 FFT and the device thread program, both
// against the exact schoolbook product, noise-free keys (development aid)
#include <barrier>
#include <cmath>
#include <cstdio>
#include <thread>
#include <vector>
#include "../../dct-cryptonets_amd/csrc/pbs_core.h"
#include "../../oracle/tfhe_ref.h"
using namespace dctfhe;
template <int LOGN, int K, int L, int P>
void run(int n, int beta) {
  using G = pbs_geom<LOGN, K, L, P>;
  constexpr int N = G::N, M = G::M, T = G::T;
  const int D = K * N, count = 4, w = 3;
  std::vector<uint8_t> S(D), s(n);
  ref_gen_binary_key(11, D, S.data()); ref_gen_binary_key(12, n, s.data());
  for (auto& b : s) b = 1;  // every CMUX active
  const int rows = (K + 1) * L;
  std::vector<uint64_t> bsk((size_t)n * rows * (K + 1) * N);
  ref_bsk_gen(s.data(), n, S.data(), K, N, L, beta, 0.0, 13, bsk.data());
  std::vector<double> bskf(bsk.size());
  ref_bsk_to_fourier(bsk.data(), n, K, N, L, bskf.data());
  std::vector<uint64_t> phases(count), cts((size_t)count * (n + 1));
  for (int c = 0; c < count; c++) phases[c] = (uint64_t)(c + 1) << (63 - w);
  ref_lwe_encrypt_batch(s.data(), n, n, phases.data(), count, 0.0, 14, cts.data());
  std::vector<int64_t> table(1 << w);
  for (int x = 0; x < (1 << w); x++) table[x] = (int64_t)x << 58;
  std::vector<uint64_t> o_fft((size_t)count * (D + 1)), o_ex((size_t)count * (D + 1)), o_emu((size_t)count * (D + 1));
  ref_pbs_batch(cts.data(), count, n, bskf.data(), bsk.data(), 0, K, N, L, beta, table.data(), w, nullptr, D, o_fft.data());
  ref_pbs_batch(cts.data(), count, n, bskf.data(), bsk.data(), 1, K, N, L, beta, table.data(), w, nullptr, D, o_ex.data());
  std::vector<cplx> tw(G::F::TW_ELEMS), bsk_dev((size_t)(n + PBS_PF_DIST) * G::BSK_ELEMS_PER_KEYBIT), exch(G::F::EXCH_ELEMS);
  std::vector<uint32_t> pfd(T);
  fill_twiddles<G::LOGM, P>(tw.data());
  std::vector<unsigned char> shared(G::SHARED_BYTES);   // rotation stage aliases the exchange buffer, as on the device
  std::vector<uint64_t> accl((size_t)G::NL * N + 1);
  std::barrier bar(T);
  auto worker = [&](int t) {
    auto sync = [&] { bar.arrive_and_wait(); };
    const size_t npoly = (size_t)n * rows * (K + 1);
    for (size_t q = 0; q < npoly; q++) key_poly_to_fourier<LOGN, P>(bsk.data() + q * N, bsk_dev.data() + q * M, t, tw.data(), reinterpret_cast<cplx*>(shared.data()), sync, sync);
    sync();
    for (int c = 0; c < count; c++) {
      pbs_args A; A.ct_small = cts.data() + (size_t)c * (n + 1); A.n = n; A.beta = beta; A.bsk = bsk_dev.data(); A.table = table.data(); A.w = w;
      A.out = o_emu.data() + (size_t)c * (D + 1); A.D_out = D; A.accumulate = 0; A.body_add = 0; A.bsk_wrap = 0; A.pf_parts = 0; A.twist = tw.data() + G::F::TW_TOTAL; A.pf_rank = 0;
      pbs_thread<LOGN, K, L, P>(A, t, tw.data(), reinterpret_cast<uint64_t*>(shared.data() + G::STAGE_OFFSET), reinterpret_cast<cplx*>(shared.data()), accl.data(), pfd.data(), sync, sync); sync();
    }
  };
  std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(worker, t); for (auto& x : th) x.join();
  std::vector<uint64_t> pf(count), pe(count), pm(count);
  ref_lwe_phase_batch(S.data(), D, o_fft.data(), count, pf.data());
  ref_lwe_phase_batch(S.data(), D, o_ex.data(), count, pe.data());
  ref_lwe_phase_batch(S.data(), D, o_emu.data(), count, pm.data());
  double ef = 0, em = 0, ee = 0;
  for (int c = 0; c < count; c++) {
    const uint64_t want = (uint64_t)table[(c + 1)];
    ef += std::pow((double)(int64_t)(pf[c] - want), 2); em += std::pow((double)(int64_t)(pm[c] - want), 2); ee += std::pow((double)(int64_t)(pe[c] - want), 2);
  }
  auto lg = [&](double v) { return 0.5 * std::log2(v / count) - 64; };
  std::printf("N=%d l=%d beta=%d n=%d: rms err  exact 2^%.2f  oracle-fft 2^%.2f  device-fft 2^%.2f  (per sqrt(n): fft 2^%.2f dev 2^%.2f)\n", N, L, beta, n,
              lg(ee), lg(ef), lg(em), lg(ef) - 0.5 * std::log2(n), lg(em) - 0.5 * std::log2(n));
}
int main() {
  run<12, 1, 3, 8>(8, 12);
  run<11, 1, 3, 8>(16, 12);
  return 0;
}
