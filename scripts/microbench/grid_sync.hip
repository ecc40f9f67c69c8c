// Developer microbenchmark: cost of a grid-wide barrier between single-wavefront workgroups (cooperative launch), against
// the cost of ending one small kernel and starting the next.  hipcc --offload-arch=gfx950 -O3 grid_sync.hip -o grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(64) void k_sync(int n, double *buf) {
  cg::grid_group g = cg::this_grid();
  double v = buf[blockIdx.x * 64 + threadIdx.x];
  for (int i = 0; i < n; i++) {
    buf[blockIdx.x * 64 + threadIdx.x] = v + 1.0;
    g.sync();
    v = buf[((blockIdx.x + 1) % gridDim.x) * 64 + threadIdx.x];
  }
  buf[blockIdx.x * 64 + threadIdx.x] = v;
}
// hand-written barrier: one atomic counter per generation, spinning with s_sleep
__global__ __launch_bounds__(64) void k_sync2(int n, double *buf, unsigned *bar) {
  double v = buf[blockIdx.x * 64 + threadIdx.x];
  for (int i = 0; i < n; i++) {
    buf[blockIdx.x * 64 + threadIdx.x] = v + 1.0;
    __threadfence();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(i + 1) * gridDim.x;
      while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    v = __builtin_nontemporal_load(&buf[((blockIdx.x + 1) % gridDim.x) * 64 + threadIdx.x]);
  }
  buf[blockIdx.x * 64 + threadIdx.x] = v;
}
__global__ __launch_bounds__(64) void k_small(double *buf) {
  buf[blockIdx.x * 64 + threadIdx.x] += 1.0;
}

int main() {
  for (int nb : {200, 600, 2000, 4000}) {
    double *buf; unsigned *bar;
    hipMalloc(&buf, (size_t)nb * 64 * 8); hipMemset(buf, 0, (size_t)nb * 64 * 8);
    hipMalloc(&bar, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int n = 100; float ms;
    void *args[] = {&n, &buf};
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipError_t e = hipLaunchCooperativeKernel((void *)k_sync, dim3(nb), dim3(64), args, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("blocks %4d: cooperative grid.sync x%d: %s, %.2f us per sync\n", nb, n, hipGetErrorString(e), ms * 1e3 / n);
    }
    for (int rep = 0; rep < 2; rep++) {
      hipMemset(bar, 0, 4);
      void *a2[] = {&n, &buf, &bar};
      hipEventRecord(e0);
      hipError_t e = hipLaunchCooperativeKernel((void *)k_sync2, dim3(nb), dim3(64), a2, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("blocks %4d: atomic barrier x%d: %s, %.2f us per sync\n", nb, n, hipGetErrorString(e), ms * 1e3 / n);
    }
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_small, dim3(nb), dim3(64), 0, 0, buf);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("blocks %4d: %d small kernels back to back: %.2f us per kernel\n", nb, n, ms * 1e3 / n);
    }
    hipFree(buf); hipFree(bar);
  }
  return 0;
}
