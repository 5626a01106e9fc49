// TEST INFRASTRUCTURE — self-test of the SIMT emulator: the wave / workgroup primitives of codec.hip.h against scalar loops.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <vector>

#include "../../etl_amd/csrc/lookback.hip.h"

using namespace etlg;

static int g_fail = 0;
#define CHECK(c, ...) do { if (!(c)) { if (g_fail++ < 20) { printf("FAIL %s:%d ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

__global__ void k_scans(const uint32_t* in, uint32_t* out_add, uint32_t* out_max, uint32_t* out_seg, uint32_t* out_blk, uint32_t* tot) {
  __shared__ uint32_t lds[16];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t v = in[t];
  out_add[t] = wave_scan_add(v & 0xFFFF);
  out_max[t] = wave_scan_max(v & 0xFFFF);
  out_seg[t] = wave_scan_incl((v & 7) == 0 ? 0x80000001u : (v & 1), [](uint32_t a, uint32_t b) { return seg_combine(a, b); }, 0u);
  uint32_t total;
  out_blk[t] = block_scan_incl<0>(v & 0xFF, lds, &total);
  if (threadIdx.x == 0) tot[blockIdx.x] = total;
}

__global__ void k_misc(uint32_t* out) {
  const uint32_t t = threadIdx.x, lane = t & 63;
  const unsigned long long b = __ballot(lane % 3 == 0);
  out[t * 4 + 0] = (uint32_t)b ^ (uint32_t)(b >> 32);
  out[t * 4 + 1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t + 7));
  out[t * 4 + 2] = __shfl_up(t, 5, 64);
  uint32_t x = 0;
  if (lane >= 10 && lane < 40) x = (uint32_t)__ballot(1) + (uint32_t)(__ballot(1) >> 32);  // partial mask
  out[t * 4 + 3] = x + __shfl(t, 63 - lane, 64);
}

int main() {
  const uint32_t B = 256, G = 3, N = B * G;
  std::vector<uint32_t> in(N), a(N), m(N), s(N), blk(N), tot(G);
  uint64_t r = 88172645463325252ull;
  for (auto& x : in) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; x = (uint32_t)r; }
  hipLaunchKernelGGL(k_scans, dim3(G), dim3(B), 0, 0, in.data(), a.data(), m.data(), s.data(), blk.data(), tot.data());
  for (uint32_t w = 0; w < N / 64; w++) {
    uint32_t acc = 0, mx = 0, seg = 0;
    for (uint32_t l = 0; l < 64; l++) {
      const uint32_t v = in[w * 64 + l];
      acc += v & 0xFFFF; mx = std::max(mx, v & 0xFFFF);
      seg = seg_combine(seg, (v & 7) == 0 ? 0x80000001u : (v & 1));
      CHECK(a[w * 64 + l] == acc, "add w%u l%u %u != %u", w, l, a[w * 64 + l], acc);
      CHECK(m[w * 64 + l] == mx, "max w%u l%u", w, l);
      CHECK(s[w * 64 + l] == seg, "seg w%u l%u %x != %x", w, l, s[w * 64 + l], seg);
    }
  }
  for (uint32_t b = 0; b < G; b++) {
    uint32_t acc = 0;
    for (uint32_t t = 0; t < B; t++) { acc += in[b * B + t] & 0xFF; CHECK(blk[b * B + t] == acc, "blk b%u t%u", b, t); }
    CHECK(tot[b] == acc, "tot b%u", b);
  }
  std::vector<uint32_t> o(4 * 128);
  hipLaunchKernelGGL(k_misc, dim3(1), dim3(128), 0, 0, o.data());
  for (uint32_t t = 0; t < 128; t++) {
    const uint32_t lane = t & 63, w0 = t & ~63u;
    unsigned long long b = 0; for (int l = 0; l < 64; l++) if (l % 3 == 0) b |= 1ull << l;
    CHECK(o[t * 4] == ((uint32_t)b ^ (uint32_t)(b >> 32)), "ballot");
    CHECK(o[t * 4 + 1] == w0 + 7, "readfirstlane %u", o[t * 4 + 1]);
    CHECK(o[t * 4 + 2] == (lane < 5 ? t : t - 5), "shfl_up");
    unsigned long long pm = 0; for (int l = 10; l < 40; l++) pm |= 1ull << l;
    const uint32_t x = (lane >= 10 && lane < 40) ? (uint32_t)pm + (uint32_t)(pm >> 32) : 0;
    CHECK(o[t * 4 + 3] == x + (w0 + 63 - lane), "partial ballot / shfl t%u: %u", t, o[t * 4 + 3]);
  }
  printf(g_fail ? "simt selftest: %d failures\n" : "simt selftest ok\n", g_fail);
  return g_fail != 0;
}
