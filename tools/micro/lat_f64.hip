// Micro-benchmark (diagnostic, not product): dependent-chain latencies of the fp64 VALU ops, LDS round trips and
// cross-lane exchanges on gfx950, one wave per CU -- the quantities that bound k_ilqr's serial loops.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
#define N 4096
template <int OP> __global__ void k(double *out, long long *cyc, double a, double b) {
  __shared__ double lds[256];
  const int lane = threadIdx.x;
  double x = a + lane * 1e-9, y = b;
  lds[lane] = x; lds[64 + lane] = y;
  __syncthreads();
  long long t0 = clock64();
  if (OP == 0) { for (int i = 0; i < N; ++i) { x = x * y; } }                       // dependent mul
  if (OP == 1) { for (int i = 0; i < N; ++i) { x = x + y; } }                       // dependent add
  if (OP == 2) { for (int i = 0; i < N; ++i) { x = __builtin_fma(x, y, y); } }      // dependent fma
  if (OP == 3) { for (int i = 0; i < N; ++i) { x = y / x; } }                       // dependent division
  if (OP == 4) { for (int i = 0; i < N; ++i) { x = __builtin_amdgcn_rcp(x); } }     // dependent rcp
  if (OP == 5) { for (int i = 0; i < N; ++i) { lds[lane] = x; asm volatile("" ::: "memory"); x = lds[lane ^ 1] + y; asm volatile("" ::: "memory"); } }  // LDS write->read (b64) + add
  if (OP == 6) { for (int i = 0; i < N; ++i) { int lo = __builtin_amdgcn_ds_bpermute((lane ^ 1) * 4, __double2loint(x)); int hi = __builtin_amdgcn_ds_bpermute((lane ^ 1) * 4, __double2hiint(x)); x = __hiloint2double(hi, lo) + y; } }  // bpermute x2 + add
  if (OP == 7) { double z = x; for (int i = 0; i < N; ++i) { x = x * y; z = z * y; } x += z; }    // two independent mul chains
  if (OP == 8) { float f = (float)x, g = (float)y; for (int i = 0; i < N; ++i) { f = f * g; } x = f; }   // f32 dependent mul
  if (OP == 9) { for (int i = 0; i < N; ++i) { double s, c; sincos(x, &s, &c); x = s + c; } }   // sincos chain
  if (OP == 10) { for (int i = 0; i < N; ++i) { x = tan(x) + y; } }   // tan chain
  if (OP == 11) { for (int i = 0; i < N; ++i) { x = lds[(lane + (int)x) & 63]; asm volatile("" ::: "memory"); } }  // dependent LDS read (address from data)
  if (OP == 12) { double z = x, w = x, v = y; for (int i = 0; i < N; ++i) { x = x * y; z = z * y; w = w * y; v = v * y;} x += z + w + v; }  // four independent mul chains
  long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = x;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double *out; long long *cyc;
  hipMalloc(&out, 64 * 8 * 4); hipMalloc(&cyc, 8 * 4);
  const char *names[] = {"mul", "add", "fma", "div", "rcp", "lds w->r + add", "bpermute x2 + add", "2 indep mul chains", "f32 mul", "sincos + add", "tan + add", "dependent lds read", "4 indep mul chains"};
#define RUN(OP) { hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.99999999); hipDeviceSynchronize(); hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.99999999); hipDeviceSynchronize(); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-22s %8.1f cycles / iteration\n", names[OP], (double)c / N); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
  // clock rate of s_memtime vs wall clock
  { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.99999999); hipEventRecord(e1); hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("mul chain: %lld counter ticks in %.3f ms (launch incl.) => >= %.0f MHz counter\n", c, ms, c / ms / 1e3); }
  return 0;
}
