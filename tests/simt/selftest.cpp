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

// ---- the two-level look-back with hand-made descriptor states: workgroups run one after the other in the emulator,
// so the parity files only ever see "every predecessor already inclusive"; this drives the deeper windows.
template <class Op>
__global__ void k_lookback(unsigned long long* desc, unsigned long long* gdesc, uint32_t tile, uint64_t agg, uint64_t carry,
                           uint32_t* fail, uint64_t* out) {
  const uint64_t r = lookback<Op>(desc, gdesc, tile, agg, carry, fail);
  if (threadIdx.x == 0) *out = r;
}

template <class Op>
static void check_lookback(const char* name, uint32_t tile, int incl_group /* newest group with an INCLUSIVE descriptor, -1 = none (the carry ends the walk) */,
                           uint64_t seed) {
  const uint32_t g = tile >> 6, j = tile & 63, ntiles = (g + 1) * 64;
  std::vector<unsigned long long> desc(ntiles, 0), gdesc(g + 2, 0);
  std::vector<uint64_t> agg(ntiles);
  uint64_t r = seed;
  auto rnd = [&]() { r ^= r << 13; r ^= r >> 7; r ^= r << 17; return r; };
  for (auto& a : agg) {
    const uint64_t x = rnd();
    // payloads that are valid for every operator: small counts, bit 29 of the high word (the "Begin seen" flag of OpTxn) now and then
    a = ((uint64_t)(((x >> 40) & 0xFFF) | ((x & 15) == 0 ? (1u << 29) : 0)) << 32) | (uint32_t)((x >> 8) & 0xFFFFF);
  }
  const uint64_t carry = ((uint64_t)7 << 32) | 3;
  // scalar reference: exclusive prefix of `tile`, and the inclusive prefix of every group
  std::vector<uint64_t> gincl(g + 1);
  uint64_t run = carry;
  for (uint32_t t = 0; t < ntiles; t++) { if (t == tile) break; run = Op::f(run, agg[t]); }
  const uint64_t want = run;
  run = carry;
  for (uint32_t t = 0; t < g * 64; t++) { run = Op::f(run, agg[t]); if ((t & 63) == 63) gincl[t >> 6] = run; }
  for (uint32_t t = g * 64; t < tile; t++) desc[t] = ST_AGG | agg[t];   // earlier tiles of the own group: aggregates only
  for (uint32_t k = 0; k < g; k++) {
    uint64_t ga = agg[k * 64];
    for (uint32_t t = k * 64 + 1; t < (k + 1) * 64; t++) ga = Op::f(ga, agg[t]);
    gdesc[k] = ((int)k <= incl_group) ? (ST_INCL | gincl[k]) : (ST_AGG | ga);
  }
  uint32_t fail = 0; uint64_t out = ~0ull;
  hipLaunchKernelGGL(k_lookback<Op>, dim3(1), dim3(64), 0, 0, desc.data(), gdesc.data(), tile, agg[tile], carry, &fail, &out);
  (void)hipStreamSynchronize(0);
  CHECK(fail == 0, "%s tile %u: gave up", name, tile);
  CHECK(out == want, "%s tile %u incl_group %d: %llx != %llx", name, tile, incl_group, (unsigned long long)out, (unsigned long long)want);
  CHECK(desc[tile] == (ST_AGG | agg[tile]), "%s tile %u: own aggregate not published", name, tile);
  if (j == 63) CHECK(gdesc[g] == (ST_INCL | Op::f(want, agg[tile])), "%s tile %u: group descriptor", name, tile);
  (void)j;
}

static void lookback_tests() {
  uint64_t seed = 0x9E3779B97F4A7C15ull;
  const uint32_t tiles[] = {0, 1, 5, 63, 64, 65, 100, 127, 128, 64 * 63 + 9, 64 * 64, 64 * 64 + 63, 64 * 65 + 1, 64 * 130 + 17, 64 * 200 + 63};
  for (uint32_t tile : tiles) {
    const int g = (int)(tile >> 6);
    const int incls[] = {g - 1, g - 2, g - 40, g - 64, g - 65, g - 130, -1};
    for (int ig : incls) {
      if (ig < -1 || ig >= g) { if (ig != -1) continue; }
      seed += 0x1234567;
      check_lookback<OpAdd>("OpAdd", tile, ig, seed);
      check_lookback<OpAdd2>("OpAdd2", tile, ig, seed);
      check_lookback<OpTxn>("OpTxn", tile, ig, seed);
    }
  }
}

// ---- the lazy stream model (ETLG_SIMT_STREAMS=lazy; under immediate execution the same checks hold trivially, except the "stale" ones)
__global__ void k_fill(uint32_t* dst, uint32_t v, uint32_t n) { for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = v; }
__global__ void k_sum(const uint32_t* src, uint32_t n, uint32_t* out) {
  uint32_t acc = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) acc += src[i];
  atomicAdd(out, acc);
}
__global__ void k_wait_flag(volatile uint32_t* flag, uint32_t* out) {   // polls like a late carry: s_sleep is where another stream's work may run
  uint32_t polls = 0;
  while (*flag == 0 && polls < 100000) { polls++; __builtin_amdgcn_s_sleep(2); }
  if (threadIdx.x == 0) *out = *flag ? 1u : 2u;
}

static void stream_tests() {
  const int mode = simt::streams_mode();   // under a random schedule an unordered pair may run either way
  const bool lazy = mode == 1;
  hipStream_t a, b;
  hipEvent_t ev;
  (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  uint32_t *buf = nullptr, *out = nullptr, *pinned = nullptr;
  (void)hipMalloc((void**)&buf, 1024 * 4); (void)hipMalloc((void**)&out, 64);
  (void)hipHostMalloc((void**)&pinned, 64);
  // 1. producer on a, consumer on b behind a wait for the producer's event: the consumer sees the producer's values
  hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, a, buf, 3u, 1024u);
  (void)hipEventRecord(ev, a);
  (void)hipStreamWaitEvent(b, ev, 0);
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(64), 0, b, (const uint32_t*)buf, 1024u, out);
  (void)hipStreamSynchronize(b);
  CHECK(out[0] == 3 * 1024, "ordered by an event: %u", out[0]);
  // 2. the same WITHOUT the wait: under the lazy model the consumer runs first (nothing forces the producer) and sums the old values
  out[1] = 0;
  hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, a, buf, 5u, 1024u);
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(64), 0, b, (const uint32_t*)buf, 1024u, out + 1);
  (void)hipStreamSynchronize(b);
  CHECK(mode == 2 ? (out[1] == 3 * 1024 || out[1] == 5 * 1024) : out[1] == (lazy ? 3u : 5u) * 1024, "unordered: %u (mode %d)", out[1], mode);
  (void)hipStreamSynchronize(a);
  CHECK(buf[7] == 5, "the producer ran once its stream was synchronised");
  // 3. copies: from pageable memory the bytes are taken at the call; from pinned memory when the copy runs; into pageable memory the
  //    call completes the copy (and everything before it on the stream)
  uint32_t pageable[4] = {11, 12, 13, 14};
  pinned[0] = 21;
  (void)hipMemcpyAsync(buf, pageable, 16, hipMemcpyHostToDevice, a);
  (void)hipMemcpyAsync(buf + 4, pinned, 4, hipMemcpyHostToDevice, a);
  pageable[0] = 99; pinned[0] = 77;                 // the caller reuses both buffers before any synchronisation
  (void)hipStreamSynchronize(a);
  CHECK(buf[0] == 11, "pageable source staged at the call: %u", buf[0]);
  CHECK(mode == 2 ? (buf[4] == 77 || buf[4] == 21) : buf[4] == (lazy ? 77u : 21u), "pinned source read when the copy runs: %u", buf[4]);
  uint32_t back = 0;
  hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, a, buf, 8u, 16u);
  (void)hipMemcpyAsync(&back, buf, 4, hipMemcpyDeviceToHost, a);   // pageable destination: complete on return
  CHECK(back == 8, "copy into pageable memory completes at the call: %u", back);
  // 4. a kernel that polls for another stream's result: the other stream's queued work runs while it polls
  uint32_t* flag = buf + 512;
  *flag = 0; out[2] = 0;
  hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, a, flag, 1u, 1u);          // queued on a, not forced by anything ...
  hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, b, (volatile uint32_t*)flag, out + 2);
  (void)hipStreamSynchronize(b);                                              // ... except by the poll of the kernel on b
  CHECK(out[2] == 1, "a polling kernel lets the other stream run: %u", out[2]);
  (void)hipStreamSynchronize(a);
  // 5. hipFree drains: nothing queued may touch the block afterwards
  hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, a, buf, 1u, 1024u);
  (void)hipFree(buf);
  (void)hipFree(out); (void)hipHostFree(pinned);
  (void)hipEventDestroy(ev); (void)hipStreamDestroy(a); (void)hipStreamDestroy(b);
}

int main() {
  const uint32_t B = 256, G = 3, N = B * G;
  std::vector<uint32_t> in(N), a(N), m(N), s(N), blk(N), tot(G);
  uint64_t r = 88172645463325252ull;
  for (auto& x : in) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; x = (uint32_t)r; }
  hipLaunchKernelGGL(k_scans, dim3(G), dim3(B), 0, 0, in.data(), a.data(), m.data(), s.data(), blk.data(), tot.data());
  (void)hipStreamSynchronize(0);
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
  (void)hipStreamSynchronize(0);
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
  lookback_tests();
  stream_tests();
  printf(g_fail ? "simt selftest: %d failures\n" : "simt selftest ok\n", g_fail);
  return g_fail != 0;
}
