// Micro-benchmark (measurement tool, not product): on which SIMD of its CU does wave w of a 256-thread workgroup land? If the
// dispatcher always starts a workgroup on the same SIMD, the "spine" wave (wave 0) of every k_cells / k_fused tile on a CU shares
// ONE SIMD with the spine waves of the other resident tiles, and that SIMD carries ~2x the instructions of the others.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/simd_place.hip -o gpurun_out/simd_place && gpurun_out/simd_place
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(256, 4) void k_place(unsigned long long* hist, uint32_t lds_pad) {
  extern __shared__ uint8_t lds[];
  uint32_t hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const uint32_t simd = (hw >> 4) & 3u, wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) atomicAdd(&hist[wave * 4 + simd], 1ull);
  // stay resident for a while so that several workgroups share a CU, as in the real kernels
  unsigned long long t0 = clock64();
  while (clock64() - t0 < 20000) { if (lds_pad == 0xFFFFFFFFu) lds[threadIdx.x] = 1; }
}

int main() {
  unsigned long long* d; unsigned long long h[16];
  hipMalloc(&d, sizeof h);
  for (uint32_t lds : {0u, 38u * 1024u}) {
    hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(k_place, dim3(2782), dim3(256), lds, 0, d, lds);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("dynamic LDS %u: rows = wave of the workgroup, columns = SIMD id\n", lds);
    for (int w = 0; w < 4; w++) printf("  wave %d: %6llu %6llu %6llu %6llu\n", w, h[w * 4], h[w * 4 + 1], h[w * 4 + 2], h[w * 4 + 3]);
  }
  return 0;
}
