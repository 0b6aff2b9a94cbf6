// Micro-benchmark (diagnostic, not part of the library): shader clock vs wall clock and the issue
// rate of v_mfma_f64_16x16x4_f64 / v_fma_f64 / ds_read_b64 on one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_mfma(double* out, unsigned long long* t, int iters, int chains) {
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1e-3 * i;
  __syncthreads();
  double4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
  const double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  if (chains == 1) {
    for (int i = 0; i < iters; ++i) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
  } else if (chains == 3) {
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
    }
  } else if (chains == 4) {           // 3 chains, each MFMA fed by a fresh v_mul (as in k_pair_gram)
    double p = x, q = y;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(p * q, q, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(p * x, p, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(q * x, x, a2, 0, 0, 0);
      p += 1e-9; q -= 1e-9;
    }
  } else if (chains == 5) {           // the same with all three operands read from LDS (16-byte reads, two steps each)
    typedef double v2 __attribute__((ext_vector_type(2)));
    const v2* l2 = (const v2*)lds;
    int idx = threadIdx.x & 63;
    for (int i = 0; i < iters; i += 2) {
      const v2 w0 = l2[(idx + 0) & 2047], w1 = l2[(idx + 64) & 2047], w2 = l2[(idx + 128) & 2047];
      const v2 u0 = l2[(idx + 192) & 2047], u1 = l2[(idx + 256) & 2047], u2 = l2[(idx + 320) & 2047];
      const v2 b0 = l2[(idx + 384) & 2047], b1 = l2[(idx + 448) & 2047], b2 = l2[(idx + 512) & 2047];
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0.x * u0.x, b0.x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1.x * u1.x, b1.x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2.x * u2.x, b2.x, a2, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0.y * u0.y, b0.y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1.y * u1.y, b1.y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2.y * u2.y, b2.y, a2, 0, 0, 0);
      idx += 7;
    }
  } else if (chains == 0) {           // dependent fp64 FMA chain
    double f = x;
    for (int i = 0; i < iters; ++i) f = __builtin_fma(f, y, x);
    a0[0] = f;
  } else if (chains == -1) {          // dependent LDS read chain
    int idx = threadIdx.x;
    double f = 0;
    for (int i = 0; i < iters; ++i) { const double v = lds[idx & 4095]; idx = (int)(v * 1000.0) + threadIdx.x + i; f += v; }
    a0[0] = f;
  } else if (chains == -2) {          // 4 independent fp64 FMA chains
    double f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3;
    for (int i = 0; i < iters; ++i) { f0 = __builtin_fma(f0, y, x); f1 = __builtin_fma(f1, y, x); f2 = __builtin_fma(f2, y, x); f3 = __builtin_fma(f3, y, x); }
    a0[0] = f0 + f1 + f2 + f3;
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2];
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = w1 - w0; t[1] = c1 - c0; }
}

int main() {
  double* out; unsigned long long* t;
  hipMalloc(&out, sizeof(double) * 256 * 1024); hipMalloc(&t, 16);
  const int modes[] = {1, 3, 4, 5, -1};
  const char* names[] = {"mfma_f64_16x16x4 1 chain", "mfma_f64_16x16x4 3 chains (per 3)", "3 chains + v_mul feeding (per 3)", "3 chains, operands from LDS b128 (per 3)", "ds_read_b64 dependent"};
  for (int grid : {1, 256}) for (int bs : {256}) for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 5; ++m) {
      const int iters = 2000;
      hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(bs), 0, 0, out, t, iters, modes[m]);
      hipDeviceSynchronize();
      unsigned long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
      if (rep == 1)
        printf("grid %3d block %3d %-36s wall %7.2f us  clock64 %9llu  -> %.1f ns/iter, %.1f clk/iter, clk %.0f MHz\n", grid, bs, names[m],
               h[0] * 0.01, h[1], h[0] * 10.0 / iters, (double)h[1] / iters, h[1] / (h[0] * 0.01));
    }
  return 0;
}
