// Diagnostic: what makes a dependent kernel cost more than the 1.6 us of an empty one -- the size of its kernarg block (all of it
// read), the length of straight-line code a wave runs through once, or both.  Graph of 100 launches, 520 workgroups x 256.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double v[150]; };
template <int NREAD, int NCODE, int STRIDE = 1>
__global__ void k_probe(Big b, double* out) {
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NREAD; ++i) s += b.v[i * STRIDE];          // NREAD fields of the kernarg block (scalar loads), STRIDE doubles apart
  double x = (double)threadIdx.x + s;
#pragma unroll
  for (int i = 0; i < NCODE; ++i) x = x * 1.0000001 + (double)(i & 7);      // NCODE straight-line instructions (8 bytes each)
  if (x == 12345.678) out[blockIdx.x] = x;
}
template <int NREAD, int NCODE, int STRIDE = 1>
static void run(const char* what) {
  double* out; hipMalloc(&out, 8 * 4096);
  Big b; for (int i = 0; i < 150; ++i) b.v[i] = i * 1e-3;
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int r = 0; r < 100; ++r) hipLaunchKernelGGL((k_probe<NREAD, NCODE, STRIDE>), dim3(520), dim3(256), 0, st, b, out);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, st);
  hipEventRecord(e0, st);
  for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %.2f us per launch\n", what, ms * 1000.0 / 2000);
  hipGraphExecDestroy(ge); hipGraphDestroy(g); hipFree(out);
}
int main() {
  run<1, 0>("1 kernarg field, no code");
  run<16, 0>("16 kernarg fields (128 B)");
  run<64, 0>("64 kernarg fields (512 B)");
  run<150, 0>("150 kernarg fields (1200 B)");
  run<8, 0, 8>("8 fields, one per 64-byte line (first 512 B)");
  run<13, 0, 8>("13 fields, one per 64-byte line (first 832 B)");
  run<18, 0, 8>("18 fields, one per 64-byte line (1152 B)");
  run<6, 0, 24>("6 fields 192 B apart (to 1152 B)");
  run<1, 512>("1 field, 512 dependent DP instructions (8 KB)");
  run<1, 2048>("1 field, 2048 dependent DP instructions (32 KB)");
  run<150, 2048>("150 fields, 2048 dependent DP instructions");
  return 0;
}
