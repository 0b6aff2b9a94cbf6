// Diagnostic: time for 250 workgroups to each pull ~21 KB from a 5 MB array, strided 128-B segments vs contiguous.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k_stage(const double* src, double* out, unsigned long long* t, int mode, int KS, int LREC, int n) {
  __shared__ double lds[8192];
  const int tid = threadIdx.x, ct = blockIdx.x % 10, ks = blockIdx.x / 10;
  const unsigned long long w0 = wall_clock64();
  double v[12];
  const int cc = tid & 15, ilg = tid >> 4, i0 = ks * KS;
#pragma unroll
  for (int u = 0; u < 12; ++u) {
    const int i = min(i0 + ilg + 16 * u, n - 1);
    const size_t off = (mode == 0) ? (size_t)i * LREC + ct * 16 + cc              // record-major (stride LREC)
                                   : ((size_t)ct * n + i) * 16 + cc;             // column-tile-major (contiguous per WG)
    v[u] = src[off];
  }
#pragma unroll
  for (int u = 0; u < 12; ++u) lds[(tid + 256 * u) & 8191] = v[u];
  __syncthreads();
  const unsigned long long w1 = wall_clock64();
  out[blockIdx.x * 256 + tid] = lds[(tid * 7) & 8191];
  if (tid == 0) t[blockIdx.x] = w1 - w0;
}
int main() {
  const int n = 4096, LREC = 152, KS = 164;
  double *src, *out; unsigned long long* t;
  hipMalloc(&src, 8 * (size_t)n * 160); hipMalloc(&out, 8 * 256 * 256); hipMalloc(&t, 8 * 256);
  hipMemset(src, 0, 8 * (size_t)n * 160);
  double* scratch; hipMalloc(&scratch, 512u << 20);     // evicts L2 / MALL between runs
  for (int mode = 0; mode < 2; ++mode) for (int flush = 0; flush < 2; ++flush) {
    for (int rep = 0; rep < 3; ++rep) {
      if (flush) hipMemset(scratch, rep, 512u << 20);
      hipLaunchKernelGGL(k_stage, dim3(250), dim3(256), 0, 0, src, out, t, mode, KS, LREC, n);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(250); hipMemcpy(h.data(), t, 8 * 250, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%s, %s: per-WG staging time min %.2f median %.2f max %.2f us\n", mode ? "tile-major (contiguous)" : "record-major (128 B segments)",
           flush ? "cold (caches flushed)" : "warm", h[0] * 0.01, h[125] * 0.01, h[249] * 0.01);
  }
  return 0;
}
