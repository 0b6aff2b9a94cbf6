// Diagnostic: how many workgroups start concurrently for a given LDS footprint / grid size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k_spin(unsigned long long* starts, unsigned long long* ends, int* xcc, int spin_ticks) {
  extern __shared__ double smem[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    starts[blockIdx.x] = t0;
    unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    xcc[blockIdx.x] = (int)((id & 0xf) << 16 | (hw & 0xffff));
  }
  smem[threadIdx.x] = (double)t0;
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) { }
  if (threadIdx.x == 0) ends[blockIdx.x] = wall_clock64();
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d, LDS/block max %zu, regs/block %d, name %s\n", p.multiProcessorCount, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.gcnArchName);
  unsigned long long *s, *e; int* x;
  hipMalloc(&s, 8 * 4096); hipMalloc(&e, 8 * 4096); hipMalloc(&x, 4 * 4096);
  hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int ldskb : {8, 48, 60, 70, 80, 100}) for (int grid : {128, 200, 250, 256, 257, 320, 512, 513}) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_spin, dim3(grid), dim3(256), ldskb * 1024, 0, s, e, x, 1000); hipDeviceSynchronize(); }
    std::vector<unsigned long long> hs(grid), he(grid); std::vector<int> hx(grid);
    hipMemcpy(hs.data(), s, 8 * grid, hipMemcpyDeviceToHost); hipMemcpy(he.data(), e, 8 * grid, hipMemcpyDeviceToHost);
    hipMemcpy(hx.data(), x, 4 * grid, hipMemcpyDeviceToHost);
    unsigned long long t0 = *std::min_element(hs.begin(), hs.end());
    int late = 0; unsigned long long mx = 0; int per_xcc[16] = {0};
    for (int i = 0; i < grid; ++i) { if (hs[i] - t0 > 500) ++late; mx = std::max(mx, hs[i] - t0); per_xcc[(hx[i] >> 16) & 15]++; }
    printf("lds %3d KB grid %3d: late WGs %3d, last start +%.2f us, total %.2f us | per-XCC:", ldskb, grid, late, mx * 0.01,
           (*std::max_element(he.begin(), he.end()) - t0) * 0.01);
    for (int q = 0; q < 8; ++q) printf(" %d", per_xcc[q]);
    printf("\n");
  }
  return 0;
}
