// Developer microbenchmark (gfx950): issue cost / dependent latency of the instructions the 16-lane sweeps are
// made of, measured on one wavefront with HIP events (ns per iteration) and the shader cycle counter.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_latency valu_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SRC>
__device__ inline double bc(double v) {
  const long x = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(long, v), 0x150 + SRC, 0xf, 0xf, false);
  return __builtin_bit_cast(double, x);
}
template <int SRC>
__device__ inline int bc32(int v) { return __builtin_amdgcn_mov_dpp(v, 0x150 + SRC, 0xf, 0xf, false); }
#define N_IT 100000
template <int MODE>
__global__ void k(double *out, long long *cyc, double a, double b) {
  __shared__ double lds[1024 * 8];
  double x[8];
  for (int i = 0; i < 8; i++) x[i] = out[threadIdx.x & 63] + i;
  int y[8];
  for (int i = 0; i < 8; i++) y[i] = threadIdx.x + i;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < N_IT; r++) {
    if (MODE == 0) x[0] = __builtin_fma(x[0], a, b);
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = __builtin_fma(x[i], a, b);
    }
    if (MODE == 2) x[0] = bc<5>(x[0]) + 1.0;       // dpp -> add dependent pair
    if (MODE == 3) {                               // 8 independent b64 dpp, each followed by an add on another chain
      x[0] += bc<0>(x[7]); x[1] += bc<1>(x[7]); x[2] += bc<2>(x[7]); x[3] += bc<3>(x[7]);
      x[4] += bc<4>(x[7]); x[5] += bc<5>(x[7]); x[6] += bc<6>(x[7]); x[7] = x[7] * a;
    }
    if (MODE == 4) {                               // the same adds without the dpp
      x[0] += x[7]; x[1] += x[7]; x[2] += x[7]; x[3] += x[7]; x[4] += x[7]; x[5] += x[7]; x[6] += x[7]; x[7] = x[7] * a;
    }
    if (MODE == 5) {                               // 8 independent b32 dpp + int adds
      y[0] += bc32<0>(y[7]); y[1] += bc32<1>(y[7]); y[2] += bc32<2>(y[7]); y[3] += bc32<3>(y[7]);
      y[4] += bc32<4>(y[7]); y[5] += bc32<5>(y[7]); y[6] += bc32<6>(y[7]); y[7] = y[7] * 3;
    }
    if (MODE == 8) {
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = x[i] + a;
    }
    if (MODE == 9) {
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = x[i] * a;
    }
    if (MODE == 10) {                              // 16 independent fma
#pragma unroll
      for (int i = 0; i < 8; i++) { x[i] = __builtin_fma(x[i], a, b); }
#pragma unroll
      for (int i = 0; i < 8; i++) { x[i] = __builtin_fma(x[i], b, a); }
    }
    if (MODE == 11) {                              // 8 fma + 8 independent 32-bit int ops
#pragma unroll
      for (int i = 0; i < 8; i++) { x[i] = __builtin_fma(x[i], a, b); y[i] = y[i] * 3 + 1; }
    }
    if (MODE == 6) {                               // LDS round trip on the critical path
      lds[threadIdx.x] = x[0];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      x[0] += lds[threadIdx.x ^ 1];
    }
    if (MODE == 7) {                               // 7 LDS writes + 7 transposed reads (the 7x7 transpose)
#pragma unroll
      for (int i = 0; i < 7; i++) lds[threadIdx.x * 8 + i] = x[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 7; i++) x[i] += lds[((threadIdx.x & 48) + i) * 8 + (threadIdx.x & 7)];
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0; for (int i = 0; i < 8; i++) s += x[i] + y[i];
  if (threadIdx.x < 64) out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
template <int MODE>
float run(double *d, long long *c, int threads = 64) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<1, threads>>>(d, c, 0.999, 1e-3);
  (void)hipEventRecord(e0); k<MODE><<<1, threads>>>(d, c, 0.999, 1e-3); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double *d; long long *c; (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 64 * 8); (void)hipMemset(d, 0, 512);
  float ms[8] = {run<0>(d, c), run<1>(d, c), run<2>(d, c), run<3>(d, c), run<4>(d, c), run<5>(d, c), run<6>(d, c), run<7>(d, c)};
  long long h[8]; (void)hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
  const char *nm[8] = {"1 dependent fma", "8 independent fma", "dpp b64 -> add (dependent)", "7x(dpp b64 + add) + mul", "7 add + mul",
                       "7x(dpp b32 + iadd) + imul", "lds write -> read -> add", "7 lds writes + 7 reads + 7 adds"};
  for (int i = 0; i < 8; i++) printf("%-32s %8.2f ns/iter  %8.2f counter ticks/iter\n", nm[i], ms[i] * 1e6 / N_IT, (double)h[i] / N_IT);
  printf("8 independent add                %8.2f ns/iter\n", run<8>(d, c) * 1e6 / N_IT);
  printf("8 independent mul                %8.2f ns/iter\n", run<9>(d, c) * 1e6 / N_IT);
  printf("16 independent fma               %8.2f ns/iter\n", run<10>(d, c) * 1e6 / N_IT);
  printf("8 fma + 16 int32 ops             %8.2f ns/iter\n", run<11>(d, c) * 1e6 / N_IT);
  printf("8 independent fma, 4 waves/CU    %8.2f ns/iter\n", run<1>(d, c, 256) * 1e6 / N_IT);
  printf("8 independent fma, 8 waves/CU    %8.2f ns/iter\n", run<1>(d, c, 512) * 1e6 / N_IT);
  printf("8 independent fma, 16 waves/CU   %8.2f ns/iter\n", run<1>(d, c, 1024) * 1e6 / N_IT);
  printf("16 independent fma, 8 waves/CU   %8.2f ns/iter\n", run<10>(d, c, 512) * 1e6 / N_IT);
  return 0;
}
