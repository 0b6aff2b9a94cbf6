// Diagnostic: single-workgroup load throughput from L2 (what one CU's texture-addresser / L1 path sustains).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int VEC>
__global__ void k_ta(const double* src, double* out, unsigned long long* t, int steps, int stride_d, int region_d) {
  const int tid = threadIdx.x;
  double acc = 0;
  __syncthreads();
  const unsigned long long c0 = clock64();
  int off = tid * VEC;
  for (int s = 0; s < steps; ++s) {
    if (VEC == 2) {
      const double2* p = (const double2*)(src + off);
      double2 a = p[0], b = p[1], c = p[2], d = p[3];
      acc += a.x + b.y + c.x + d.y;
    } else {
      const double* p = src + off;
      double a = p[0], b = p[stride_d], c = p[2 * stride_d], d = p[3 * stride_d], e = p[4 * stride_d], f = p[5 * stride_d], g = p[6 * stride_d];
      acc += a + b + c + d + e + f + g;
    }
    off += stride_d * 8; if (off > region_d) off -= region_d;
  }
  const unsigned long long c1 = clock64();
  out[tid] = acc;
  if (tid == 0) t[0] = c1 - c0;
}
int main() {
  double *src, *out; unsigned long long* t;
  const int region = 64 * 1024;  // doubles = 512 KB
  hipMalloc(&src, 8 * (region + 65536)); hipMalloc(&out, 8 * 1024); hipMalloc(&t, 8);
  hipMemset(src, 0, 8 * (region + 65536));
  for (int bs : {256, 640, 1024}) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_ta<2>, dim3(1), dim3(bs), 0, 0, src, out, t, 200, 8 * bs / 8, region);
    hipDeviceSynchronize(); unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("block %4d  4 x 16B loads/thread/step (64 B/thread contiguous): %.0f clk/step, %.1f B/clk\n", bs, h / 200.0, bs * 64.0 / (h / 200.0));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_ta<1>, dim3(1), dim3(bs), 0, 0, src, out, t, 200, bs, region);
    hipDeviceSynchronize(); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("block %4d  7 x  8B loads/thread/step (coalesced rows):          %.0f clk/step, %.1f B/clk\n", bs, h / 200.0, bs * 56.0 / (h / 200.0));
  }
  return 0;
}
