// Developer microbenchmark: where do the two wavefronts of a 128-thread workgroup land?  (Input for a producer / consumer split of the
// persistent kernel: DESIGN.md section 8.)  Workgroups of two wavefronts with 39 KB of LDS each (four per CU, as p_solve's), one
// workgroup per four NLPs of a 4096-NLP batch; every wavefront reports HW_ID (SIMD, CU, SE) and XCC_ID.
// hipcc --offload-arch=gfx950 -O3 wave_placement.hip -o wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(128) void k_where(unsigned *out, int spin) {
  __shared__ double pad[39 * 128];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
  pad[threadIdx.x] = (double)hw;
  double v = pad[(threadIdx.x * 7) % 128];
  for (int i = 0; i < spin; i++) v = v * 1.0000001 + 1e-9;              // stay resident while the rest of the grid arrives
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (v == 12345.0 ? 1u : 0u);
  }
}

int main() {
  const int nb = 1024;
  unsigned *d;
  hipMalloc(&d, nb * 4 * sizeof(unsigned));
  hipLaunchKernelGGL(k_where, dim3(nb), dim3(128), 0, 0, d, 200000);
  std::vector<unsigned> h(nb * 4);
  hipMemcpy(h.data(), d, nb * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
  // per (xcc, se, cu): how many wavefronts on each SIMD, and how many workgroups have both wavefronts on the same SIMD
  std::map<unsigned, std::vector<int>> simd_count;
  int same = 0, adjacent = 0;
  for (int b = 0; b < nb; b++) {
    unsigned key[2], simd[2];
    for (int w = 0; w < 2; w++) {
      const unsigned hw = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1] & 0xf;
      simd[w] = (hw >> 4) & 3;
      key[w] = (xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xf);
      auto &c = simd_count[key[w]];
      if (c.empty()) c.assign(4, 0);
      c[simd[w]]++;
    }
    if (key[0] == key[1] && simd[0] == simd[1]) same++;
    if (key[0] == key[1] && ((simd[0] + 1) & 3) == simd[1]) adjacent++;
    if (b < 12) printf("workgroup %2d: wave 0 xcc %u se %u cu %2u simd %u | wave 1 xcc %u se %u cu %2u simd %u\n", b, key[0] >> 16, (key[0] >> 8) & 0xff,
                       key[0] & 0xff, simd[0], key[1] >> 16, (key[1] >> 8) & 0xff, key[1] & 0xff, simd[1]);
  }
  std::map<std::vector<int>, int> hist;
  for (auto &kv : simd_count) hist[kv.second]++;
  printf("%d workgroups x 2 wavefronts on %zu CUs; both wavefronts on one SIMD: %d; on consecutive SIMDs: %d\n", nb, simd_count.size(), same, adjacent);
  for (auto &kv : hist) printf("  CUs with wavefronts per SIMD [%d %d %d %d]: %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
  // wave-0 wavefronts per SIMD (what a fixed role assignment by wavefront index would give)
  std::map<unsigned, std::vector<int>> w0;
  for (int b = 0; b < nb; b++) {
    const unsigned hw = h[(b * 2) * 2], xcc = h[(b * 2) * 2 + 1] & 0xf;
    auto &c = w0[(xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xf)];
    if (c.empty()) c.assign(4, 0);
    c[(hw >> 4) & 3]++;
  }
  std::map<std::vector<int>, int> hist0;
  for (auto &kv : w0) hist0[kv.second]++;
  for (auto &kv : hist0) printf("  CUs with wave-0 wavefronts per SIMD [%d %d %d %d]: %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
  return 0;
}
