// Micro-benchmark (measurement tool, not product): what do LDS reads cost on gfx950 by width and alignment when every lane
// reads from its own frame (113-byte stride, the cfg2 layout), and does global_load_lds_dwordx4 accept per-lane unaligned sources?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_align.hip -o gpurun_out/lds_align && gpurun_out/lds_align
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef uint32_t u32a1 __attribute__((aligned(1)));
typedef uint64_t u64a1 __attribute__((aligned(1)));
typedef uint64_t u64a4 __attribute__((aligned(4)));
struct __attribute__((packed, aligned(1))) V4a1 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) V4a4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) V3a4 { uint32_t x, y, z; };
struct __attribute__((packed, aligned(4))) V2a4 { uint32_t x, y; };

constexpr int kIters = 64, kUnroll = 8;

template <int MODE>
__device__ __forceinline__ uint32_t rd(const uint8_t* p) {
  if (MODE == 0) return *(const u32a1*)p;                                   // b32 unaligned
  if (MODE == 1) return *(const uint32_t*)((uintptr_t)p & ~(uintptr_t)3);   // b32 aligned
  if (MODE == 2) { const V2a4 v = *(const V2a4*)((uintptr_t)p & ~(uintptr_t)3); return v.x ^ v.y; }      // 2 dwords, 4B aligned
  if (MODE == 3) { const uint64_t v = *(const u64a1*)p; return (uint32_t)v ^ (uint32_t)(v >> 32); }       // b64 unaligned
  if (MODE == 4) { const uint64_t v = *(const uint64_t*)((uintptr_t)p & ~(uintptr_t)7); return (uint32_t)v ^ (uint32_t)(v >> 32); }  // b64 8B aligned
  if (MODE == 5) { const V4a1 v = *(const V4a1*)p; return v.x ^ v.y ^ v.z ^ v.w; }                        // b128 unaligned
  if (MODE == 6) { const V4a4 v = *(const V4a4*)((uintptr_t)p & ~(uintptr_t)3); return v.x ^ v.y ^ v.z ^ v.w; }   // b128 4B aligned
  if (MODE == 7) { const uint4 v = *(const uint4*)((uintptr_t)p & ~(uintptr_t)15); return v.x ^ v.y ^ v.z ^ v.w; } // b128 16B aligned
  if (MODE == 8) { const V3a4 v = *(const V3a4*)((uintptr_t)p & ~(uintptr_t)3); return v.x ^ v.y ^ v.z; }          // b96 4B aligned
  return 0;
}

// stride: bytes between the lanes' frames; every iteration reads kUnroll independent positions inside the frame
template <int MODE>
__global__ void k_lds(const uint8_t* in, uint32_t* out, unsigned long long* cyc, uint32_t stride) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  for (uint32_t i = threadIdx.x; i < 64u * 160u / 16u; i += blockDim.x) ((uint4*)lds)[i] = ((const uint4*)in)[i];
  __syncthreads();
  const uint8_t* fr = lds + (threadIdx.x & 63) * stride + (threadIdx.x >> 6) * 0;
  uint32_t acc = 0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int u = 0; u < kUnroll; u++) acc ^= rd<MODE>(fr + ((it + u * 13) % 97));
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}

// LDS-DMA: lane L copies 16 bytes from src + srcoff[L] to lds + k * 1024 + 16 * L (eight pieces), then the block dumps LDS
__global__ void k_dma(const uint8_t* in, const uint32_t* srcoff, uint8_t* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t lane = threadIdx.x;
  const unsigned long long t0 = clock64();
#pragma unroll
  for (int k = 0; k < 8; k++)
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(in + srcoff[lane] + 16 * k),
                                     (void __attribute__((address_space(3)))*)(lds + k * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = clock64();
  __syncthreads();
  for (uint32_t i = lane; i < 8192 / 16; i += 64) ((uint4*)out)[(size_t)blockIdx.x * 512 + i] = ((const uint4*)lds)[i];
  if (lane == 0) atomicAdd(cyc, t1 - t0);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
static int run_lds(const char* name, const uint8_t* d_in, uint32_t* d_out, unsigned long long* d_cyc, uint32_t stride, int blocks, int threads) {
  CK(hipMemset(d_cyc, 0, 8));
  hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(threads), 64 * 160, 0, d_in, d_out, d_cyc, stride);
  CK(hipDeviceSynchronize());
  unsigned long long c = 0; CK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
  printf("%-22s stride %3u blocks %4d x %3d thr: %7.1f cycles per wave-instruction (wave 0 of each block)\n", name, stride, blocks, threads,
         (double)c / blocks / (kIters * kUnroll));
  return 0;
}

int main() {
  std::vector<uint8_t> h(1 << 20);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 2654435761u >> 13);
  uint8_t *d_in, *d_dump; uint32_t* d_out; unsigned long long* d_cyc; uint32_t* d_off;
  CK(hipMalloc(&d_in, h.size())); CK(hipMalloc(&d_out, 4 << 20)); CK(hipMalloc(&d_cyc, 8)); CK(hipMalloc(&d_off, 256)); CK(hipMalloc(&d_dump, 8192 * 1024));
  CK(hipMemcpy(d_in, h.data(), h.size(), hipMemcpyHostToDevice));
  for (uint32_t stride : {113u, 128u, 132u}) {
    for (int cfg = 0; cfg < 2; cfg++) {
      const int blocks = cfg ? 1024 : 1, threads = cfg ? 256 : 64;
      if (run_lds<0>("b32 unaligned", d_in, d_out, d_cyc, stride, blocks, threads)) return 1;
      if (run_lds<1>("b32 aligned", d_in, d_out, d_cyc, stride, blocks, threads)) return 1;
      if (run_lds<2>("2xb32 4B-aligned", d_in, d_out, d_cyc, stride, blocks, threads)) return 1;
      if (run_lds<3>("b64 unaligned", d_in, d_out, d_cyc, stride, blocks, threads)) return 1;
      if (run_lds<4>("b64 8B-aligned", d_in, d_out, d_cyc, stride, blocks, threads)) return 1;
      if (run_lds<5>("b128 unaligned", d_in, d_out, d_cyc, stride, blocks, threads)) return 1;
      if (run_lds<6>("b128 4B-aligned", d_in, d_out, d_cyc, stride, blocks, threads)) return 1;
      if (run_lds<7>("b128 16B-aligned", d_in, d_out, d_cyc, stride, blocks, threads)) return 1;
      if (run_lds<8>("b96 4B-aligned", d_in, d_out, d_cyc, stride, blocks, threads)) return 1;
    }
  }
  // LDS-DMA with per-lane sources: (a) coalesced 16-byte aligned, (b) coalesced but misaligned by 5, (c) one frame per lane (113-byte stride)
  for (int mode = 0; mode < 3; mode++) {
    uint32_t off[64];
    for (int l = 0; l < 64; l++) off[l] = mode == 0 ? 16 * l : mode == 1 ? 16 * l + 5 : 113 * l + 3;
    CK(hipMemcpy(d_off, off, 256, hipMemcpyHostToDevice));
    CK(hipMemset(d_cyc, 0, 8));
    const int blocks = 1024;
    hipLaunchKernelGGL(k_dma, dim3(blocks), dim3(64), 8192, 0, d_in, d_off, d_dump, d_cyc);
    CK(hipDeviceSynchronize());
    unsigned long long c = 0; CK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
    std::vector<uint8_t> dump(8192);
    CK(hipMemcpy(dump.data(), d_dump + 8192 * 7, 8192, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int k = 0; k < 8; k++) for (int l = 0; l < 64; l++) if (memcmp(&dump[k * 1024 + 16 * l], &h[off[l] + 16 * k], 16)) bad++;
    printf("LDS-DMA mode %d (%s): %d of 512 pieces wrong, %.0f cycles issue->landed for 8 pieces per wave\n", mode,
           mode == 0 ? "coalesced aligned" : mode == 1 ? "coalesced, source misaligned by 5" : "one 113-byte frame per lane", bad, (double)c / blocks);
  }
  return 0;
}
