// Record-boundary scan for gfx950: the byte offsets of every CopyData frame of a batch,
// computed on the device when the caller has no sidecar (include/etlg.h: etlg_decode with
// frame_offsets == NULL, and etlg_scan_boundaries).
//
// Frames are a linked list through the buffer — the frame at p is 'd' | Int32-BE length L and the
// next one starts at p + 1 + L — so which boundaries lie in a tile of bytes depends on where the
// chain enters it. The kernel is optimistic and verifies itself:
//
//   1. one wave stages its tile (TB bytes + a header's worth of halo) into LDS;
//   2. it GUESSES its entry — the first position of the tile holding a well-formed XLogData /
//      keepalive frame header whose successor holds one too — and walks the chain from there to
//      the end of the tile out of LDS, recording the frame starts (tile 0 knows its entry: 0);
//   3. it publishes (frames, exit offset) and takes part in two look-backs: the usual two-level
//      decoupled prefix sum of the frame counts (its offsets' base index), and a look at the
//      nearest older tile that found frames, whose exit must be exactly this tile's guessed
//      entry (or lie beyond the tile when it found none);
//   4. offsets go straight to their final indexes.
//
// If every tile passes its check, the guesses ARE the chain (induction from tile 0). A tile that
// fails — payload bytes that look like two consecutive headers right at a tile start, or a
// malformed frame — records the entry it should have used as a hint and raises a counter; the
// host reruns the kernel with the hints (each run fixes every tile flagged by the one before,
// and every tile under a frame found to cover a flagged tile) and, after a few runs, lets one
// lane follow the chain (k_bounds_seq). Wrong guesses need
// non-text bytes that mimic a frame AND its successor within the ~100 bytes before the first
// real frame of a tile, so reruns are rare; malformed input (an error anyway) costs one.
//
// The result is exactly the sequential rule  p -> p + 1 + L  while buf[p] == 'd', L >= 4 and
// the frame fits; else the rest of the buffer is one last (malformed) frame  — the rule of the
// oracle and of the host fallback in host.cpp.
#include "lookback.hip.h"

namespace etlg {

constexpr uint32_t TB = 8192;          // bytes per tile
constexpr uint32_t HALO = 32;          // a header (and the bytes the guess inspects) may straddle the tile end
constexpr uint32_t MAXN = TB / 5 + 2;  // frames that can start in one tile (a frame is >= 5 bytes)
constexpr uint32_t NO_ENTRY = 0xFFFFFFFFu;

struct BoundsParams {
  const u8* in;
  uint64_t len;
  uint32_t* offs;              // out: nframes + 1 offsets
  uint32_t offs_cap;           // entries available in offs
  uint32_t ntiles;
  unsigned long long* vdesc;   // [ntiles]           st:2 | has:1 | n:15 | exit:32   (zeroed)
  unsigned long long* ndesc;   // [ntiles + ngroups] frame-count look-back           (zeroed)
  uint32_t* hints;             // [ntiles] entry to use instead of guessing, NO_ENTRY = guess (kept across reruns)
  uint32_t* result;            // [0] nframes  [1] failure flags (1 spin gave up, 2 offs_cap too small)  [2] tiles that failed their check
};

// Is there a well-formed frame header at absolute offset p? If so `next` = start of the following
// frame. `st` holds the input bytes from `lo` on.
DEV bool header_at(const u8* st, uint32_t lo, uint64_t len, uint32_t p, uint32_t& next) {
  if ((uint64_t)p + 5 > len) return false;
  const u8* h = st + (p - lo);
  const uint32_t L = ld_be32(h + 1);
  if (h[0] != 'd' || L < 4 || (uint64_t)p + 1 + L > len) return false;
  next = p + 1 + L;
  return true;
}

// A header that also looks like what the server sends: XLogData with a known pgoutput tag, or a
// keepalive of the right size. Only used to pick a guess (never to decide what a frame is).
DEV bool plausible_at(const u8* st, uint32_t lo, uint64_t len, uint32_t p, uint32_t& next) {
  if (!header_at(st, lo, len, p, next)) return false;
  const u8* h = st + (p - lo);
  const uint32_t L = next - p - 1;
  if (h[5] == 'k') return L == 4 + 18;
  if (h[5] != 'w' || L < 4 + 26) return false;
  const uint32_t t = h[kTagOff];
  return t == 'B' || t == 'C' || t == 'I' || t == 'U' || t == 'D' || t == 'R' || t == 'T' || t == 'M' || t == 'O' || t == 'Y';
}

// Walks the chain from `p` to the end of the tile; every lane runs the same walk (uniform control
// flow, LDS broadcast reads), lane 0 records the frame starts relative to lo.
DEV void walk(const u8* st, uint16_t* plist, uint32_t lo, uint32_t hi, uint64_t len, uint32_t p, uint32_t& n, uint32_t& exit_off) {
  n = 0;
  const bool rec = (threadIdx.x & 63) == 0;
  while (p < hi) {
    uint32_t nx;
    if (rec) plist[n] = (uint16_t)(p - lo);
    n++;
    p = header_at(st, lo, len, p, nx) ? nx : (uint32_t)len;  // a malformed header: the rest is one frame
  }
  exit_off = p;
}

__global__ __launch_bounds__(64) void k_bounds(BoundsParams q) {
  __shared__ __attribute__((aligned(16))) u8 st[TB + HALO + 16];
  __shared__ uint16_t plist[MAXN];
  const uint32_t lane = threadIdx.x;
  const uint32_t tile = blockIdx.x;
  const uint64_t lo64 = (uint64_t)tile * TB;
  const uint32_t lo = (uint32_t)lo64;
  const uint32_t hi = (uint32_t)(lo64 + TB < q.len ? lo64 + TB : q.len);
  uint32_t* fail = &q.result[1];
  // ---- stage [lo, hi + HALO), zero past the end of the input
  {
    const uint32_t want = hi - lo + HALO;
    const bool al = ((uintptr_t)q.in & 15) == 0;  // tile starts are multiples of TB
    for (uint32_t c = 16 * lane; c < want; c += 16 * 64) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (al && lo64 + c + 16 <= q.len) v = *(const uint4*)(q.in + lo64 + c);
      else {
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < 16; i++) if (lo64 + c + i < q.len) w[i >> 2] |= (uint32_t)q.in[lo64 + c + i] << (8 * (i & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
      *(uint4*)(st + c) = v;
    }
  }
  __syncthreads();
  // ---- entry: known (tile 0), hinted by an earlier run, or guessed
  uint32_t entry = NO_ENTRY;  // absolute offset
  const uint32_t hint = q.hints[tile];
  if (tile == 0) entry = 0;
  else if (hint != NO_ENTRY && hint >= lo) entry = hint;  // (a hint below the tile came from a predecessor that was wrong itself)
  else {
    // first position whose frame looks real and whose successor does too (or cannot be seen from here)
    for (uint32_t base = lo; base < hi && entry == NO_ENTRY; base += 64) {
      const uint32_t p = base + lane;
      bool good = false;
      uint32_t nx;
      if (p < hi && plausible_at(st, lo, q.len, p, nx)) {
        uint32_t nx2;
        good = nx >= hi || plausible_at(st, lo, q.len, nx, nx2);  // nx < hi: its header is inside the staged window
      }
      const unsigned long long m = __ballot(good);
      if (m) entry = base + (uint32_t)__builtin_ctzll(m);
    }
  }
  uint32_t n = 0, e = 0;
  const bool has = entry != NO_ENTRY && entry < hi;
  if (has) walk(st, plist, lo, hi, q.len, entry, n, e);
  __syncthreads();  // plist (written by lane 0) -> all lanes
  // ---- publish, prefix-sum the frame counts
  if (lane == 0) {
    const unsigned long long w = ST_AGG | ((unsigned long long)(has ? 1u : 0u) << 47) | ((unsigned long long)n << 32) | e;
    __hip_atomic_store(&q.vdesc[tile], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const uint64_t N = lookback<OpAdd>(q.ndesc, q.ndesc + q.ntiles, tile, n, 0, fail);
  // ---- check the guess against the nearest older tile that found frames
  if (tile > 0) {
    uint32_t polls = 0, e_prev = 0;
    int64_t top = (int64_t)tile - 1;
    for (;;) {
      const int64_t idx = top - lane;
      unsigned long long w = ST_AGG | (1ull << 47);  // below tile 0: a virtual tile with frames and exit 0 (never reached: tile 0 has frames)
      if (idx >= 0) w = __hip_atomic_load(&q.vdesc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long m_has = __ballot((w & ST_MASK) != 0 && ((w >> 47) & 1)), m_empty = __ballot((w & ST_MASK) == 0);
      const int first = m_has ? __builtin_ctzll(m_has) : 64;
      const unsigned long long need = first >= 63 ? ~0ull : ((2ull << first) - 1);
      if (m_empty & need) {
        if (++polls > kMaxPolls) { if (lane == 0) atomicOr(fail, 1u); return; }
        __builtin_amdgcn_s_sleep(2);
        continue;
      }
      if (first < 64) { e_prev = (uint32_t)__shfl(w, first, 64); break; }
      top -= 64;  // 64 tiles inside one frame: keep looking
    }
    const bool ok = has ? e_prev == entry : e_prev >= hi;
    if (!ok) {
      if (lane == 0) {
        q.hints[tile] = e_prev;  // what this tile should have started from (>= hi: nothing starts here)
        atomicAdd(&q.result[2], 1u);
      }
      // a frame that runs past this tile also covers every tile up to its end: tell them all now (tiles
      // inside one long value that mimics frames agree with each other and would otherwise be found
      // one per run)
      if (e_prev >= hi) {
        const uint64_t last = (uint64_t)e_prev / TB;  // tiles tile+1 .. last-1 end at or before e_prev
        for (uint64_t t = (uint64_t)tile + 1 + lane; t < last && t < q.ntiles; t += 64) q.hints[t] = e_prev;
      }
    }
  }
  // ---- offsets at their final indexes
  if (N + n + 1 > q.offs_cap) { if (lane == 0) atomicOr(fail, 2u); return; }
  for (uint32_t i = lane; i < n; i += 64) q.offs[N + i] = lo + plist[i];
  if (tile == q.ntiles - 1 && lane == 0) {
    q.offs[N + n] = (uint32_t)q.len;
    q.result[0] = (uint32_t)(N + n);
  }
}

// Cold fallback: one lane follows the whole chain out of global memory.
__global__ void k_bounds_seq(BoundsParams q) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t p = 0;
  uint32_t n = 0;
  q.result[1] = 0; q.result[2] = 0;
  while (p < q.len) {
    if (n + 2 > q.offs_cap) { q.result[1] = 2u; return; }
    q.offs[n++] = (uint32_t)p;
    uint64_t next = q.len;
    if (q.len - p >= 5) {
      const u8* h = q.in + p;
      const uint64_t L = ((uint64_t)h[1] << 24) | ((uint64_t)h[2] << 16) | ((uint64_t)h[3] << 8) | h[4];
      if (h[0] == 'd' && L >= 4 && p + 1 + L <= q.len) next = p + 1 + L;
    }
    p = next;
  }
  q.offs[n] = (uint32_t)q.len;
  q.result[0] = n;
}

}  // namespace etlg

extern "C" {

using namespace etlg;

uint32_t etlg_k_bounds_tile_bytes(void) { return TB; }

// Scratch layout (device, caller-owned): vdesc [ntiles] u64 | ndesc [ntiles + ngroups] u64 (both zeroed
// before every run) | hints [ntiles] u32 (0xFF-filled before the first run only) | result [4] u32 (zeroed).
void etlg_k_launch_bounds(const uint8_t* in, uint64_t len, uint32_t* offs, uint32_t offs_cap, void* vdesc, void* ndesc,
                          uint32_t* hints, uint32_t* result, int sequential, hipStream_t s) {
  BoundsParams q;
  q.in = in; q.len = len; q.offs = offs; q.offs_cap = offs_cap;
  q.ntiles = (uint32_t)((len + TB - 1) / TB);
  q.vdesc = (unsigned long long*)vdesc; q.ndesc = (unsigned long long*)ndesc; q.hints = hints; q.result = result;
  if (sequential) hipLaunchKernelGGL(k_bounds_seq, dim3(1), dim3(64), 0, s, q);
  else hipLaunchKernelGGL(k_bounds, dim3(q.ntiles), dim3(64), 0, s, q);
}

}  // extern "C"
