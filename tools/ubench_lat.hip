// Diagnostic: instruction latencies seen by ONE wave per SIMD (the regime of the sweep / factor kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
#define TIMEIT(name, N, body) { __syncthreads(); unsigned long long c0 = clock64(); body; unsigned long long c1 = clock64(); if (threadIdx.x == 0) { t[slot] = (double)(c1 - c0) / (N); } ++slot; }
__global__ void k_lat(double* out, double* t, const double* g, int n1) {
  __shared__ double lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (double)((i * 7 + 1) & 8191);
  __syncthreads();
  int slot = 0;
  double f = 1.0 + threadIdx.x * 1e-9, y = 0.999, acc = 0;
  // 0: dependent FMA, unrolled 64
  TIMEIT("fma dep", 64 * 16, for (int r = 0; r < 16; ++r) { _Pragma("unroll") for (int i = 0; i < 64; ++i) f = __builtin_fma(f, y, 0.5); })
  acc += f;
  // 1: 4 independent FMA chains unrolled
  { double a = f, b = f + 1, c = f + 2, d = f + 3;
    TIMEIT("fma 4", 64 * 16 * 4, for (int r = 0; r < 16; ++r) { _Pragma("unroll") for (int i = 0; i < 64; ++i) { a = __builtin_fma(a, y, 0.5); b = __builtin_fma(b, y, 0.5); c = __builtin_fma(c, y, 0.5); d = __builtin_fma(d, y, 0.5); } })
    acc += a + b + c + d; }
  // 2: dependent LDS read chain (pointer chase)
  { int idx = threadIdx.x & 63;
    TIMEIT("lds chase", 256, for (int r = 0; r < 4; ++r) { _Pragma("unroll") for (int i = 0; i < 64; ++i) idx = (int)lds[idx]; })
    acc += idx; }
  // 3: dependent global read chain (pointer chase in L2-resident 64 KB)
  { int idx = threadIdx.x & 63;
    TIMEIT("global chase", 64, { _Pragma("unroll") for (int i = 0; i < 64; ++i) idx = (int)g[idx]; })
    acc += idx; }
  // 4: sqrt
  { double a = f + 2;
    TIMEIT("sqrt dep", 64, { _Pragma("unroll") for (int i = 0; i < 64; ++i) a = sqrt(a) + 1.5; })
    acc += a; }
  // 5: div
  { double a = f + 2;
    TIMEIT("div dep", 64, { _Pragma("unroll") for (int i = 0; i < 64; ++i) a = 3.0 / a + 1.5; })
    acc += a; }
  // 6: __syncthreads
  TIMEIT("syncthreads", 64, { _Pragma("unroll") for (int i = 0; i < 64; ++i) __syncthreads(); })
  // 7: lds write + barrier + read (typical phase hand-off)
  { double a = f;
    TIMEIT("lds handoff", 64, { _Pragma("unroll") for (int i = 0; i < 64; ++i) { lds[threadIdx.x] = a; __syncthreads(); a += lds[(threadIdx.x + 1) & (blockDim.x - 1)]; __syncthreads(); } })
    acc += a; }
  // 8: wave reduction by DPP/shuffle (6 steps)
  { double a = f;
    TIMEIT("wave reduce (6 shfl)", 16, { _Pragma("unroll") for (int i = 0; i < 16; ++i) { for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o); } })
    acc += a; }
  // 9: loop with taken branch per iteration, 1 FMA body
  { double a = f;
    TIMEIT("loop+1fma", 1024, for (int i = 0; i < n1; ++i) a = __builtin_fma(a, y, 0.5);)
    acc += a; }
  // 10: rsqrt approx
  { double a = f + 2;
    TIMEIT("rsq dep", 64, { _Pragma("unroll") for (int i = 0; i < 64; ++i) a = __builtin_amdgcn_rsq(a) + 1.5; })
    acc += a; }
  // 11: mul dep
  { double a = f;
    TIMEIT("mul dep", 64*4, for (int r = 0; r < 4; ++r) { _Pragma("unroll") for (int i = 0; i < 64; ++i) a = a * y; })
    acc += a; }
  // 12: readlane broadcast + fma (as in sequential substitution)
  { double a = f;
    TIMEIT("readlane+fma", 64, { _Pragma("unroll") for (int i = 0; i < 64; ++i) { double b = __shfl(a, i); a = __builtin_fma(b, y, a); } })
    acc += a; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  double *out, *t, *g; hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&t, 8 * 64); hipMalloc(&g, 8 * 8192);
  double hg[8192]; for (int i = 0; i < 8192; ++i) hg[i] = (double)((i * 7 + 1) & 8191);
  hipMemcpy(g, hg, sizeof hg, hipMemcpyHostToDevice);
  const char* names[] = {"fma dependent (unrolled)", "fma 4 chains (per fma)", "lds pointer chase", "global (L2) pointer chase", "sqrt(f64) dependent (+add)", "div(f64) dependent (+add)", "__syncthreads", "lds write+sync+read+sync", "wave reduce 6x shfl_xor+add", "loop iteration w/ 1 fma", "v_rsq_f64 dependent (+add)", "mul dependent", "shfl broadcast + fma"};
  for (int bs : {64, 256, 1024}) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_lat, dim3(1), dim3(bs), 0, 0, out, t, g, 1024); hipDeviceSynchronize(); }
    double h[16]; hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
    printf("block %d:\n", bs);
    for (int i = 0; i < 13; ++i) printf("  %-32s %8.1f clk  (%.1f ns)\n", names[i], h[i], h[i] / 2.4);
  }
  return 0;
}
