// Timing harness for bootstrap-kernel variants (development aid, not part of the library).
// Random key/ciphertext contents: the kernel's work does not depend on the values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../dct-cryptonets_amd/csrc/kernels.h"
using namespace dctfhe;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LOGN, int K, int L, int P, int GR, int MB = 0, int KLDS = 0>
void run(int n, int beta, size_t count, int D, int wrap = 0, int pf = 0) {
  using G = pbs_geom<LOGN, K, L, P, MB>;
  // every case announces itself BEFORE it launches: a fault then names its case (profiles/r01_exp_two_bit_rotation.log did not)
  printf("case mb=%d klds=%d pf=%d wrap=%d N=%d k=%d l=%d P=%d groups=%d n=%d count=%zu ...\n", MB, KLDS, pf, wrap, 1 << LOGN, K, L, P, GR, n, count);
  fflush(stdout);
  if (pf != 0 && pf < 8) { printf("pf_parts must be 0 or >= 8 (kernel contract, pbs_core.h)\n"); return; }
  if (MB && wrap) { printf("wrap is not supported by the two-bit kernels\n"); return; }
  constexpr int N = G::N, M = G::M;
  std::vector<cplx> tw(G::F::TW_ELEMS);
  fill_twiddles<G::LOGM, P>(tw.data());
  const size_t bsk_elems = (size_t)((wrap ? wrap : (MB ? 3 * n / 2 : n)) + PBS_PF_DIST * G::KEY_BLOCKS) * G::BSK_ELEMS_PER_KEYBIT;
  std::vector<cplx> wtab(2 * N + 8);
  for (int m = 0; m < 2 * N; m++) wtab[m] = cmk(cos(M_PI * m / N), sin(M_PI * m / N));
  for (int m = 0; m < 8; m++) wtab[2 * N + m] = root64(8 * m);
  std::vector<cplx> bsk(bsk_elems);
  uint64_t st = 1;
  for (auto& c : bsk) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; c.re = (double)(int64_t)st * 0x1p-64 / M; st = st * 6364136223846793005ULL + 1; c.im = (double)(int64_t)st * 0x1p-64 / M; }
  std::vector<uint64_t> small(count * (n + 1));
  for (auto& v : small) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; v = st; }
  int64_t tab[16]; for (int i = 0; i < 16; i++) tab[i] = (int64_t)i << 58;
  cplx *d_tw, *d_bsk, *d_wtab;
  CK(hipMalloc(&d_wtab, wtab.size() * 16)); CK(hipMemcpy(d_wtab, wtab.data(), wtab.size() * 16, hipMemcpyHostToDevice)); uint64_t *d_small, *d_out, *d_dummy; int64_t* d_tab;
  CK(hipMalloc(&d_tw, tw.size() * 16)); CK(hipMalloc(&d_bsk, bsk_elems * 16)); CK(hipMalloc(&d_small, small.size() * 8));
  CK(hipMalloc(&d_out, count * (size_t)(D + 1) * 8)); CK(hipMalloc(&d_dummy, (size_t)(D + 1) * 8)); CK(hipMalloc(&d_tab, sizeof tab));
  CK(hipMemcpy(d_tw, tw.data(), tw.size() * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(d_bsk, bsk.data(), bsk_elems * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_small, small.data(), small.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tab, tab, sizeof tab, hipMemcpyHostToDevice));
  pbs_launch a; a.cts_small = d_small; a.count = count; a.n = n; a.beta = beta; a.bsk = d_bsk; a.tw = d_tw; a.wtab = d_wtab; a.tables = d_tab; a.w = 4; a.table_idx = nullptr;
  a.hw = 1; a.nchan = 1; a.e_offset = 0; a.out = d_out; a.D_out = D; a.accumulate = 0; a.body_add = 0; a.dummy = d_dummy; a.bsk_wrap = wrap; a.pf_parts = pf;
  const size_t lds = pbs_lds_bytes<LOGN, K, L, P, MB, KLDS>(GR);
  CK(hipFuncSetAttribute((const void*)pbs_kernel<LOGN, K, L, P, GR, MB, KLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned grid = (unsigned)((count + GR - 1) / GR);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((pbs_kernel<LOGN, K, L, P, GR, MB, KLDS>), dim3(grid), dim3(G::T * GR), lds, 0, a);
  CK(hipDeviceSynchronize());
  hipEventRecord(e0);
  hipLaunchKernelGGL((pbs_kernel<LOGN, K, L, P, GR, MB, KLDS>), dim3(grid), dim3(G::T * GR), lds, 0, a);
  hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fft = 5.0 * M * log2((double)M);
  const double fl = n * ((K + 1) * L * fft + (K + 1) * fft + (double)(K + 1) * (K + 1) * L * M * 8.0);
  printf("mb=%d klds=%d pf=%d wrap=%d N=%5d k=%d l=%d P=%2d groups=%d threads=%4d lds=%6zu: %zu cts %.1f ms -> %.0f PBS/s, %.2f TFLOP/s\n", MB, KLDS, pf, wrap, N, K, L, P, GR, G::T * GR, lds, count, ms,
         count / (ms * 1e-3), fl * count / (ms * 1e-3) / 1e12);
  fflush(stdout);
#if defined(DCTFHE_PHASE_TIMERS)
  {
    static const char* names[12] = {"acc update + decompose", "leading barrier (wait)", "fwd cross-wave pass: compute + scatter", "barrier before fwd gather (wait)",
                                    "fwd gather + wave-local passes", "key products", "inv wave-local passes", "inv pass before cross-wave gather: compute + scatter",
                                    "barrier before inv gather (wait)", "inv cross-wave gather", "trailing barrier (wait)", "inv last pass + accumulator update (general form: all inverse)"};
    unsigned long long h[16 * 12];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(dctfhe_phase_ticks), sizeof h));
    const int waves = G::T * GR / 64, iters = MB ? n / 2 : n;
    for (int k = 0; k < 12; k++) {
      printf("  phase %2d %-52s ticks/iteration per wave:", k, names[k]);
      double sum = 0;
      for (int w = 0; w < waves; w++) { printf(" %7.0f", (double)h[w * 12 + k] / iters); sum += (double)h[w * 12 + k] / iters; }
      printf("   mean %7.0f\n", sum / waves);
    }
    double tot = 0; for (int k = 0; k < 12; k++) tot += (double)h[k] / iters;
    printf("  total (wave 0) %.0f ticks/iteration\n", tot);
  }
#endif
  hipFree(d_wtab); hipFree(d_tw); hipFree(d_bsk); hipFree(d_small); hipFree(d_out); hipFree(d_dummy); hipFree(d_tab);
}

int main() {
  const int D = 8192;
  EXP_CASES
  return 0;
}
