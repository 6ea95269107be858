// How fast can every CU stream the same buffer out of L2 with 16-byte lane loads, as a function of the loads a wave
// keeps in flight?  (development aid: sizes the key stream of the two-bit N = 8192 bootstrap kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct c16 { double a, b; };
// SHARE waves of a workgroup read the same addresses (the ciphertexts of a workgroup that walk one key in lock-step: the first wave's miss
// is the others' vector-L1 hit); the rate printed counts every wave's bytes: what the L1 hands to the registers.
template <int U, int SHARE = 1>
__global__ void __launch_bounds__(512) k_stream(const c16* __restrict__ buf, size_t elems_per_iter, int iters, double* sink) {
  const int t = threadIdx.x % (512 / SHARE);
  double acc = 0;
  for (int i = 0; i < iters; i++) {
    const c16* base = buf + (size_t)i * elems_per_iter;
    for (size_t o = 0; o < elems_per_iter; o += (size_t)U * (512 / SHARE)) {
      c16 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = base[o + (size_t)u * (512 / SHARE) + t];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; u++) acc += v[u].a * v[u].b;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (acc == 1.2345) sink[0] = acc;
}
template <int U, int SHARE = 1> void run(const c16* d, size_t per_iter, int iters, double* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_stream<U, SHARE>), dim3(256), dim3(512), 0, 0, d, per_iter, iters, sink);
  CK(hipDeviceSynchronize());
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_stream<U, SHARE>), dim3(256), dim3(512), 0, 0, d, per_iter, iters, sink);
  hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)per_iter * 16 * iters * SHARE;      // delivered to registers per CU
  printf("waves sharing an address %d, loads in flight per wave %2d: %.2f ms, %.1f GB/s per CU delivered (%.1f B/clk at 2.4 GHz), %.1f GB/s per CU from L2\n", SHARE, U, ms,
         bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.4e9, bytes / SHARE / (ms * 1e-3) / 1e9);
}
int main() {
  const size_t per_iter = 12 * 4096;          // 12 key polynomials of 4096 complex points = 786 KB
  const int iters = 432;
  c16* d; double* sink;
  CK(hipMalloc(&d, per_iter * 16 * iters)); CK(hipMemset(d, 0, per_iter * 16 * iters)); CK(hipMalloc(&sink, 8));
  run<2>(d, per_iter, iters, sink); run<4>(d, per_iter, iters, sink); run<6>(d, per_iter, iters, sink); run<12>(d, per_iter, iters, sink); run<24>(d, per_iter, iters, sink);
  run<12, 2>(d, per_iter, iters, sink); run<12, 4>(d, per_iter, iters, sink); run<12, 8>(d, per_iter, iters, sink); run<24, 4>(d, per_iter, iters, sink); run<24, 8>(d, per_iter, iters, sink);
  return 0;
}
