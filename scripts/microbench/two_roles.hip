// Developer microbenchmark: a sweep-like wavefront (serial steps: LDS gathers, DPP broadcasts, three LDS round trips, ~120 dependent FP64
// operations -- the shape of p_solve's factor step) and an evaluation-like wavefront (loads, ~1000 FP64 operations with instruction-level
// parallelism, LDS and global stores -- the shape of a node-parallel phase) sharing a SIMD: how much do they slow each other down?
// Workgroups of two wavefronts (wave 0 sweeps, wave 1 evaluates) with 39 KB of LDS: four per CU, one of each role per SIMD
// (scripts/microbench/wave_placement.hip).  mode 1: only the sweepers run; 2: only the evaluators; 3: both.
// hipcc --offload-arch=gfx950 -O3 two_roles.hip -o two_roles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SRC>
__device__ __forceinline__ double bc(double v) {
  const long x = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(long, v), 0x150 + SRC, 0xf, 0xf, false);
  return __builtin_bit_cast(double, x);
}
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int LDW = 65, ROWS = 55;

__global__ __launch_bounds__(128) void k_roles(int mode, int steps, int chunks, const double *gin, double *gout, long long *cyc) {
  __shared__ double stage[ROWS * LDW];
  __shared__ double tr[4][8][8];
  __shared__ double pad[1200];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, role = lane & 15;
  for (int i = threadIdx.x; i < ROWS * LDW; i += 128) stage[i] = 1.0 + 1e-3 * (i % 97);
  for (int i = threadIdx.x; i < 1200; i += 128) pad[i] = 0.5;
  __syncthreads();
  const long long t0 = clock64();
  if (wave == 0 && (mode & 1)) {
    // ---- sweeper: `steps` serial steps ----
    double a[7];
    for (int i = 0; i < 7; i++) a[i] = 1e-3 * (role + i);
    double U = 0.0;
    for (int s = 0; s < steps; s++) {
      const int cj = grp * 16 + (s & 15);
      double gq[7];
#pragma unroll
      for (int i = 0; i < 7; i++) gq[i] = stage[((role + i) % ROWS) * LDW + cj];
      const double gA = stage[(20 + role) * LDW + cj], gB = stage[(36 + role) * LDW + cj];
      const double G[8] = {bc<0>(gA), bc<1>(gA), bc<2>(gA), bc<3>(gA), bc<4>(gA), bc<5>(gA), bc<6>(gA), bc<7>(gA)};
      const double E[4] = {bc<8>(gA), bc<9>(gA), bc<10>(gA), bc<11>(gA)};
      const double cc[7] = {bc<0>(gB), bc<1>(gB), bc<2>(gB), bc<3>(gB), bc<4>(gB), bc<5>(gB), bc<6>(gB)};
      const double rc[7] = {bc<7>(gB), bc<8>(gB), bc<9>(gB), bc<10>(gB), bc<11>(gB), bc<12>(gB), bc<13>(gB)};
#pragma unroll
      for (int i = 0; i < 7; i++) a[i] += 1e-3 * gq[i];
      double b[7];
      for (int pass = 0; pass < 2; pass++) {      // two structured solves around an LDS transpose
        const double s1 = E[0] * a[2] + E[1] * a[3], s2 = E[2] * a[2] + E[3] * a[3];
        b[2] = s1 + 1e-3 * (G[0] * a[0] + G[1] * a[1]); b[3] = s2 + 1e-3 * (G[4] * a[0] + G[5] * a[1]);
        b[0] = a[0] + 1e-3 * b[2]; b[1] = a[1] + 1e-3 * b[3];
        b[4] = a[4] + 1e-3 * (G[2] * b[2] + G[6] * b[3]); b[5] = a[5] + 1e-3 * b[4]; b[6] = a[6] + 1e-3 * (G[3] * b[2] + G[7] * b[3]);
        if (pass == 0) {
          if (role < 7) {
#pragma unroll
            for (int i = 0; i < 7; i++) tr[grp][role][i] = b[i];
          }
          wsync();
          if (role < 7) {
#pragma unroll
            for (int i = 0; i < 7; i++) a[i] = tr[grp][i][role];
          }
          wsync();
        }
      }
      double mw[7];
#pragma unroll
      for (int i = 0; i < 7; i++) mw[i] = 1e-3 * bc<5>(b[i]);
      const double D = 1.0 + mw[5], iD = __builtin_amdgcn_rcp(D) * (2.0 - D * __builtin_amdgcn_rcp(D));
      const double coef = (1e-3 * b[5]) * iD;
#pragma unroll
      for (int i = 0; i < 7; i++) a[i] = b[i] - mw[i] * coef;
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int i = 0; i < 7; i++) { d0 -= a[i] * cc[i]; d1 += a[i] * rc[i]; }
      if (role < 7) { tr[grp][7][role] = d0 * 1e-3; tr[grp][role][7] = d1 * 1e-3; }
      wsync();
      double uu = 0.0;
#pragma unroll
      for (int i = 0; i < 7; i++) { const double pj = a[i] - tr[grp][7][i], sj = a[i] + pj; uu += rc[i] * sj; a[i] = pj * 0.999; }
      U += uu;
      pad[600 + lane] = coef;
      wsync();
    }
    gout[blockIdx.x * 128 + threadIdx.x] = U + a[0];
  } else if (wave == 1 && (mode & 2)) {
    // ---- evaluator: `chunks` node-parallel evaluations ----
    double acc = 0.0;
    const double *g = gin + (size_t)blockIdx.x * 64 * 48;
    for (int c = 0; c < chunks; c++) {
      double v[40];
#pragma unroll
      for (int i = 0; i < 40; i++) v[i] = g[(size_t)((i + c) % 48) * 64 + lane];
      double x[8];
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = v[i];
#pragma unroll
      for (int r = 0; r < 30; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          x[i] = x[i] * v[(r + i) % 40] + v[(r + 2 * i + 7) % 40];
          x[i] = x[i] * 1e-3 + v[(r + 3 * i + 11) % 40];
          x[(i + 1) & 7] += x[i] * v[(r + 5 * i + 13) % 40];
          x[(i + 3) & 7] = x[(i + 3) & 7] * 0.5 + x[i] * 1e-4;
        }
      }
#pragma unroll
      for (int i = 0; i < 8; i++) { pad[i * 64 + lane] = x[i]; acc += x[i]; }
      double *o = gout + (size_t)(gridDim.x * 128) + ((size_t)blockIdx.x * 20) * 64;
#pragma unroll
      for (int i = 0; i < 20; i++) o[(size_t)i * 64 + lane] = x[i & 7] + i;
    }
    gout[blockIdx.x * 128 + threadIdx.x] = acc;
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 2 + wave] = t1 - t0;
}

int main() {
  const int nb = 1024, steps = 2000, chunks = 200;
  double *gin, *gout; long long *cyc;
  hipMalloc(&gin, (size_t)nb * 64 * 48 * 8); hipMemset(gin, 0, (size_t)nb * 64 * 48 * 8);
  hipMalloc(&gout, ((size_t)nb * 128 + (size_t)nb * 20 * 64) * 8);
  hipMalloc(&cyc, nb * 2 * sizeof(long long));
  std::vector<long long> h(nb * 2);
  for (int mode = 1; mode <= 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_roles, dim3(nb), dim3(128), 0, 0, mode, steps, chunks, (const double *)gin, gout, cyc);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), cyc, nb * 2 * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0, e = 0;
    for (int b = 0; b < nb; b++) { s += h[b * 2]; e += h[b * 2 + 1]; }
    printf("mode %d (%s): sweeper %.0f cycles per step, evaluator %.0f cycles per chunk\n", mode, mode == 1 ? "sweepers only" : mode == 2 ? "evaluators only" : "both",
           (mode & 1) ? s / nb / steps : 0.0, (mode & 2) ? e / nb / chunks : 0.0);
  }
  return 0;
}
