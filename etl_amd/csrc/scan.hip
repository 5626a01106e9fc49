// Record-boundary scan for gfx950: the byte offsets of every CopyData frame of a batch,
// computed on the device when the caller has no sidecar (include/etlg.h: etlg_decode with
// frame_offsets == NULL, and etlg_scan_boundaries).
//
// Frames are a linked list through the buffer — the frame at p is 'd' | Int32-BE length L and the
// next one starts at p + 1 + L — so which boundaries lie in a tile of bytes depends on where the
// chain enters it. The kernel is optimistic and verifies itself:
//
//   1. one wave stages its tile (TB bytes + a header's worth of halo) into LDS;
//   2. every lane GUESSES an entry into its own 128 bytes — the first position holding a
//      well-formed XLogData / keepalive frame header whose successor holds one too — and walks the
//      chain to the end of its 128 bytes; the lanes are then stitched (the exit of one must be the
//      guess of the lane it lands in), which yields the tile's frames and exit offset from the
//      first guess on (tile 0 knows its entry: offset 0). A tile whose lanes do not stitch walks
//      the chain with one lane instead;
//   3. ONE two-level decoupled look-back carries (frames so far: sum, furthest exit so far: max):
//      the sum is the base index of the tile's offsets, the max must be exactly the tile's
//      guessed entry (or lie beyond the tile when it found nothing);
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
constexpr uint32_t NO_ENTRY = 0xFFFFFFFFu;

struct BoundsParams {
  const u8* in;
  uint64_t len;
  uint32_t* offs;              // out: nframes + 1 offsets
  uint32_t offs_cap;           // entries available in offs
  uint32_t ntiles;
  unsigned long long* d_clear; // the OTHER descriptor buffer (result block + look-back words of the next run), zeroed by this launch
  uint32_t clear_words;        // ... its extent in 8-byte words
  uint32_t* hflag;             // pinned host memory: [0] nframes (last tile), [1] set to 1 by any tile that fails — the host reads the device
                               // result block only then
  unsigned long long* ndesc;   // [ntiles + ngroups] look-back of (frames: sum, exit: max)   (zeroed)
  uint32_t* hints;             // [ntiles] entry to use instead of guessing, NO_ENTRY = guess (kept across reruns)
  uint32_t dbg;                // 1: per-phase shader-clock sums into result[4..11] (profiling only)
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

// Walks the chain from `p` until it reaches `stop` (p < stop on entry): n = frames started, exit_off =
// where the chain stands then. `out` (optional): the absolute frame starts, written from out[0] on.
DEV void walk(const u8* st, uint32_t* out, uint32_t lo, uint32_t stop, uint64_t len, uint32_t p, uint32_t& n, uint32_t& exit_off) {
  n = 0;
  while (p < stop) {
    uint32_t nx;
    if (out) out[n] = p;
    n++;
    p = header_at(st, lo, len, p, nx) ? nx : (uint32_t)len;  // a malformed header: the rest is one frame
  }
  exit_off = p;
}

// look-back payload: frames so far (30 bits, summed) | furthest exit so far (32 bits, max)
struct OpCountExit {
  DEV static uint64_t id() { return 0; }
  DEV static uint64_t f(uint64_t a, uint64_t b) {
    const uint32_t ea = (uint32_t)a, eb = (uint32_t)b;
    return ((((a >> 32) + (b >> 32)) & 0x3FFFFFFFull) << 32) | (ea > eb ? ea : eb);
  }
};

constexpr uint32_t SUB = TB / 64;  // bytes of the tile each lane looks at (128)
#define STAMP(k) do { if (q.dbg && threadIdx.x == 0 && (blockIdx.x & 31) == 5) { const unsigned long long _t = clock64(); atomicAdd(&q.result[4 + (k)], (uint32_t)(_t - t_prev)); t_prev = _t; } } while (0)

// What a tile knows after its local phase (registers; the LDS window is free again)
struct TileLocal {
  uint32_t lo, hi, a_end;      // the tile's bytes, the end of this lane's 128
  uint32_t entry, n, e;        // guessed entry, frames and exit offset of the tile's chain
  uint32_t s_l, n_l, inc;      // this lane: its guess, the frames of its walk, inclusive prefix of the on-chain lanes' counts
  bool has, stitched, mine;
};

// ---- local phase of one tile: stage, guess, walk, stitch. Ends with the tile's aggregate published to the look-back.
DEV void bounds_local(const BoundsParams& q, u8* st, uint32_t tile, TileLocal& t, unsigned long long& t_prev) {
  const uint32_t lane = threadIdx.x;
  const uint64_t lo64 = (uint64_t)tile * TB;
  const uint32_t lo = (uint32_t)lo64;
  const uint32_t hi = (uint32_t)(lo64 + TB < q.len ? lo64 + TB : q.len);
  t.lo = lo; t.hi = hi;
  // ---- stage [lo, hi + HALO), zero past the end of the input. A full interior tile takes the fast
  //      route: all 8 of a lane's 16-byte loads are in flight before the first LDS store.
  {
    const uint32_t want = hi - lo + HALO;
    const bool al = ((uintptr_t)q.in & 15) == 0;  // tile starts are multiples of TB
    if (al && lo64 + TB + HALO + 16 <= q.len) {
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = *(const uint4*)(q.in + lo64 + 16 * lane + 1024 * k);
      uint4 vh = make_uint4(0, 0, 0, 0);
      if (lane < (HALO + 15) / 16) vh = *(const uint4*)(q.in + lo64 + TB + 16 * lane);
#pragma unroll
      for (int k = 0; k < 8; k++) *(uint4*)(st + 16 * lane + 1024 * k) = v[k];
      if (lane < (HALO + 15) / 16) *(uint4*)(st + TB + 16 * lane) = vh;
    } else {
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
  }
  __syncthreads();
  STAMP(0);
  // ---- every lane guesses an entry into ITS 128 bytes: the first position holding a plausible frame
  //      whose successor is plausible too (or lies outside the tile). 'd' bytes are found 4 at a time.
  const uint32_t a = lo + lane * SUB, a_end = a + SUB < hi ? a + SUB : hi;
  t.a_end = a_end;
  uint32_t s_l = NO_ENTRY;
  if (a < hi) {
    // all 128 bytes at once (8 independent 16-byte LDS reads): one bit per byte that is 'd', one per
    // byte that is 0. A guess is only tried where 'd' is followed by a zero byte, i.e. at frames
    // shorter than 16 MiB (text is full of 'd's; a longer frame is simply never guessed and costs
    // the tile a hinted rerun).
    unsigned long long dmask[2] = {0, 0}, zmask[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint4 v = *(const uint4*)(st + (a - lo) + 16 * k);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t x = w[j] ^ 0x64646464u;                                                  // 'd' -> 0
        const uint32_t zd = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);              // 0x80 where the byte was 'd'
        const uint32_t y = w[j];
        const uint32_t zz = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);              // 0x80 where the byte was 0
        const uint32_t bd = ((zd >> 7) & 1u) | ((zd >> 14) & 2u) | ((zd >> 21) & 4u) | ((zd >> 28) & 8u);
        const uint32_t bz = ((zz >> 7) & 1u) | ((zz >> 14) & 2u) | ((zz >> 21) & 4u) | ((zz >> 28) & 8u);
        dmask[k >> 2] |= (unsigned long long)bd << (16 * (k & 3) + 4 * j);
        zmask[k >> 2] |= (unsigned long long)bz << (16 * (k & 3) + 4 * j);
      }
    }
    const unsigned long long znext = st[(a - lo) + SUB] == 0 ? 1ull : 0ull;  // the byte after this lane's range
    dmask[0] &= (zmask[0] >> 1) | (zmask[1] << 63);
    dmask[1] &= (zmask[1] >> 1) | (znext << 63);
    for (int half = 0; half < 2 && s_l == NO_ENTRY; half++) {
      unsigned long long m = dmask[half];
      while (m && s_l == NO_ENTRY) {
        const uint32_t p = a + 64 * half + (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        uint32_t nx, nx2;
        if (p < a_end && plausible_at(st, lo, q.len, p, nx) && (nx >= hi || plausible_at(st, lo, q.len, nx, nx2))) s_l = p;
      }
    }
  }
  STAMP(1);
  // the tile's own entry: known (tile 0), hinted by an earlier run, or the first lane's guess
  const uint32_t hint = q.hints[tile];
  uint32_t entry = NO_ENTRY;
  if (tile == 0) entry = 0;
  else if (hint != NO_ENTRY && hint >= lo) entry = hint;  // (a hint below the tile came from a predecessor that was wrong itself)
  else {
    const unsigned long long m = __ballot(s_l != NO_ENTRY);
    if (m) entry = (uint32_t)__shfl(s_l, __builtin_ctzll(m), 64);
  }
  const bool has = entry != NO_ENTRY && entry < hi;
  if (has && (entry - lo) / SUB == lane) s_l = entry;  // the lane the chain enters through starts exactly there
  // local walks: from the lane's guess to the end of its 128 bytes
  uint32_t n_l = 0, e_l = 0;
  if (s_l != NO_ENTRY) walk(st, nullptr, lo, a_end, q.len, s_l, n_l, e_l);
  STAMP(2);
  // ---- stitch the lanes. The chain passes through every guessing lane from the entry lane on iff each
  //      of them starts exactly where the furthest walk before it ended (walks of on-chain lanes end
  //      further and further), and every lane without a guess lies under a frame. One prefix-max scan.
  uint32_t n = 0, e = 0, inc = 0;
  bool stitched = has;
  bool mine = false;  // this lane's walk is part of the tile's chain
  if (has) {
    const uint32_t c0 = (entry - lo) / SUB;
    mine = lane >= c0 && s_l != NO_ENTRY;
    const uint32_t mx = wave_scan_max(mine ? e_l : 0u);  // inclusive prefix max of the exits
    const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x138, 0xF, 0xF, false);  // furthest exit of the lanes before this one
    bool good = true;
    if (lane > c0 && a < hi) good = mine ? before == s_l : before >= a_end;
    stitched = __ballot(!good) == 0;
    e = wave_last(mx);
    inc = wave_scan_add(mine ? n_l : 0u);  // inclusive prefix of the on-chain lanes' frame counts
    n = wave_last(inc);
  }
  STAMP(3);
  if (has && !stitched) {
    // the lanes do not agree (bytes that mimic a frame inside a value, a malformed header ...): one lane
    // walks the tile; it walks again (out of global memory) to write the offsets once their base index is known
    if (lane == 0) walk(st, nullptr, lo, hi, q.len, entry, n, e);
    n = (uint32_t)__shfl(n, 0, 64); e = (uint32_t)__shfl(e, 0, 64);
  }
  STAMP(4);
  t.entry = entry; t.n = n; t.e = e; t.s_l = s_l; t.n_l = n_l; t.inc = inc; t.has = has; t.stitched = stitched; t.mine = mine;
  // the aggregate is published NOW; the prefix is fetched after the wave's second tile has done its local work, so the look-back's
  // round trips (half of a tile's time when waited for at once) are covered by work
  lookback_publish(q.ndesc, tile, ((uint64_t)n << 32) | e);
}

// ---- second phase: the look-back gives the frames before this tile and how far the chain has come; check the guess, write the
//      offsets. The LDS window may hold another tile by now: the short re-walks read the input itself.
DEV void bounds_finish(const BoundsParams& q, uint32_t tile, const TileLocal& t, uint32_t* lfail, unsigned long long& t_prev) {
  const uint32_t lane = threadIdx.x;
  const uint32_t hi = t.hi, n = t.n, entry = t.entry;
  const bool has = t.has;
  const uint64_t pre = lookback_resolve<OpCountExit>(q.ndesc, q.ndesc + q.ntiles, tile, ((uint64_t)n << 32) | t.e, 0, lfail);
  const uint64_t N = pre >> 32;
  if (*lfail) { if (lane == 0) { atomicOr(&q.result[1], 1u); q.hflag[1] = 1u; } }
  STAMP(5);
  if (tile > 0) {
    // If every tile passes this check the guesses are the chain: by induction the exits of the older
    // tiles are the true ones and grow with the tile index, so their maximum is where the chain enters.
    const uint32_t e_prev = (uint32_t)pre;
    const bool ok = has ? e_prev == entry : e_prev >= hi;
    if (!ok) {
      if (lane == 0) {
        q.hints[tile] = e_prev;  // what this tile should have started from (>= hi: nothing starts here)
        atomicAdd(&q.result[2], 1u);
        q.hflag[1] = 1u;
      }
      // a frame that runs past this tile also covers every tile up to its end: tell them all now (tiles
      // inside one long value that mimics frames agree with each other and would otherwise be found
      // one per run)
      if (e_prev >= hi) {
        const uint64_t last = (uint64_t)e_prev / TB;  // tiles tile+1 .. last-1 end at or before e_prev
        for (uint64_t tt = (uint64_t)tile + 1 + lane; tt < last && tt < q.ntiles; tt += 64) q.hints[tt] = e_prev;
      }
    }
  }
  // ---- offsets at their final indexes: every on-chain lane repeats its short walk (consecutive lanes
  //      write consecutive entries)
  if (N + n + 1 > q.offs_cap) { if (lane == 0) { atomicOr(&q.result[1], 2u); q.hflag[1] = 1u; } return; }
  if (has && t.stitched) {
    if (t.mine) { uint32_t n2, e2; walk(q.in, q.offs + N + (t.inc - t.n_l), 0u, t.a_end, q.len, t.s_l, n2, e2); }
  } else if (has && lane == 0) {
    uint32_t n2, e2;
    walk(q.in, q.offs + N, 0u, hi, q.len, entry, n2, e2);
  }
  if (tile == q.ntiles - 1 && lane == 0) {
    q.offs[N + n] = (uint32_t)q.len;
    q.result[0] = (uint32_t)(N + n);
    q.hflag[0] = (uint32_t)(N + n);
  }
  STAMP(6);
}

// One wave = TWO consecutive tiles through ONE LDS window: local(A), local(B), finish(A), finish(B). Half as many waves as tiles:
// a 64 MiB batch (8 192 tiles) is one round of the chip instead of 1.7, and each look-back is resolved a tile's work after it
// was published.
__global__ __launch_bounds__(64) void k_bounds(BoundsParams q) {
  __shared__ __attribute__((aligned(16))) u8 st[TB + HALO + 16];
  __shared__ uint32_t lfail;   // failure bit of THIS wave's look-backs (1 spin gave up)
  const uint32_t lane = threadIdx.x;
  if (lane == 0) lfail = 0;
  unsigned long long t_prev = q.dbg ? clock64() : 0;
  if (q.clear_words) {  // descriptors are double buffered: this launch clears the buffer the next run will use
    const uint32_t per = (q.clear_words + gridDim.x - 1) / gridDim.x;
    for (uint32_t i = lane; i < per; i += 64) { const uint32_t w = blockIdx.x * per + i; if (w < q.clear_words) q.d_clear[w] = 0; }
  }
  const uint32_t t0 = 2 * blockIdx.x, t1 = t0 + 1;
  TileLocal A, B;
  bounds_local(q, st, t0, A, t_prev);
  const bool two = t1 < q.ntiles;
  if (two) {
    __syncthreads();   // every lane is done with tile A's bytes
    bounds_local(q, st, t1, B, t_prev);
  }
  // B first: the last tile of a group (always a B) publishes the group's aggregate as soon as it has folded the group's tile words,
  // before it waits for the older groups itself — resolving A first would put A's wait for the older groups in front of that
  // publication and chain the groups one round trip after the other
  if (two) bounds_finish(q, t1, B, &lfail, t_prev);
  bounds_finish(q, t0, A, &lfail, t_prev);
}

// Cold fallback: one lane follows the whole chain out of global memory.
__global__ void k_bounds_seq(BoundsParams q) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t p = 0;
  uint32_t n = 0;
  q.result[1] = 0; q.result[2] = 0;
  while (p < q.len) {
    if (n + 2 > q.offs_cap) { q.result[1] = 2u; q.hflag[1] = 1u; return; }
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
  q.hflag[0] = n;
}

}  // namespace etlg

extern "C" {

using namespace etlg;

uint32_t etlg_k_bounds_tile_bytes(void) { return TB; }

// Scratch (device, caller-owned): two descriptor buffers, each = result block (16 u32, first) | ndesc [ntiles + ngroups] u64, zero when a
// run starts — every run zeroes the OTHER buffer (`clear`, `clear_words`) — and hints [ntiles] u32, 0xFF-filled (refilled by the
// caller after a run that set any). hflag: 2 u32 of pinned host memory, zeroed by the caller.
void etlg_k_launch_bounds(const uint8_t* in, uint64_t len, uint32_t* offs, uint32_t offs_cap, void* cur, void* clear, uint32_t clear_words,
                          uint32_t* hints, uint32_t* hflag, int sequential, hipStream_t s) {
  BoundsParams q;
  q.dbg = (sequential & 2) ? 1u : 0u; sequential &= 1;
  q.in = in; q.len = len; q.offs = offs; q.offs_cap = offs_cap;
  q.ntiles = (uint32_t)((len + TB - 1) / TB);
  q.result = (uint32_t*)cur; q.ndesc = (unsigned long long*)cur + 8; q.hints = hints; q.hflag = hflag;
  q.d_clear = (unsigned long long*)clear; q.clear_words = clear_words;
  if (sequential) hipLaunchKernelGGL(k_bounds_seq, dim3(1), dim3(64), 0, s, q);
  else hipLaunchKernelGGL(k_bounds, dim3((q.ntiles + 1) / 2), dim3(64), 0, s, q);
}

}  // extern "C"
