// Diagnostic: cost of the 32-term dot product of two LDS rows per lane (the quadratic-form phase of k_curve_chi / k_curve_z):
// the plain loop (the compiler forms ds_read2_b64) against lds_dot.hpp's dot32_lds (ds_read_b64, two chunks in flight).
#include "../bayesfmmm_amd/csrc/lds_dot.hpp"
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int STR = 39, ROWS = 15;
template <int MODE>
__global__ __launch_bounds__(256) void k_dot(double* out, unsigned long long* clk, int iters) {
  extern __shared__ double sm[];
  const int grp = threadIdx.x >> 5, lp = threadIdx.x & 31;
  double* tile = sm + grp * (ROWS * STR + 8);
  for (int r = 0; r < ROWS; ++r) tile[r * STR + 3 + lp] = 1.0 + 1e-3 * (r + lp);
  __syncthreads();
  const double* a = tile + (lp % 6) * STR + 3;
  const double* b = tile + (6 + lp % 7) * STR + 3;
  double acc = 0.0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double s;
    if (MODE == 0) {
      s = 0.0;
#pragma unroll
      for (int p = 0; p < 32; ++p) s += a[p] * b[p];
    } else {
      s = bfmmm::dot_lds<32>(a, b);
    }
    acc += s;
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; unsigned long long* clk;
  hipMalloc(&out, 8 * 256 * 1024); hipMalloc(&clk, 8 * 1024);
  hipFuncSetAttribute((const void*)k_dot<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipFuncSetAttribute((const void*)k_dot<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  const int iters = 200;
  for (int wg : {1, 2, 3}) for (int mode : {0, 1}) {
    const int grid = 256 * wg; const size_t lds = 46 * 1024;
    for (int rep = 0; rep < 3; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k_dot<0>, dim3(grid), dim3(256), lds, 0, out, clk, iters);
      else hipLaunchKernelGGL(k_dot<1>, dim3(grid), dim3(256), lds, 0, out, clk, iters);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid); std::vector<double> o(256);
    hipMemcpy(h.data(), clk, 8 * grid, hipMemcpyDeviceToHost); hipMemcpy(o.data(), out, 8 * 256, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%d workgroups per CU, %s: %.0f clocks per 32-term dot (median workgroup; max %.0f); check %.15g %.15g\n", wg,
           mode ? "ds_read_b64 asm " : "plain loop      ", (double)h[grid / 2] / iters, (double)h[grid - 1] / iters, o[0], o[37]);
  }
  return 0;
}
