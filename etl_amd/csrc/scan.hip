// Record-boundary scan for gfx950: the byte offsets of every CopyData frame of a batch,
// computed on the device when the caller has no sidecar (include/etlg.h: etlg_decode with
// frame_offsets == NULL, and etlg_scan_boundaries).
//
// Frames are a linked list through the buffer — the frame at p is 'd' | Int32-BE length L and the
// next one starts at p + 1 + L — so which boundaries lie in a tile of bytes depends on where the
// chain enters it. The scan is optimistic and verifies itself, in THREE kernels on one stream and
// without a word exchanged between workgroups of one launch (round 5; rounds 2-4 carried the tile
// aggregates through a decoupled look-back inside one kernel — the scan's only source of defects
// that the emulator could not see, and two or three dependent round trips in every tile's path):
//
//   k_bounds_local    one wave per tile of TB bytes: stage the tile (+ a header's worth of halo) into
//                     LDS; every lane GUESSES an entry into its own 128 bytes — the first position
//                     holding a well-formed XLogData / keepalive frame header whose successor holds one
//                     too — and walks the chain to the end of its 128 bytes; the lanes are stitched (the
//                     exit of one must be the guess of the lane it lands in), which yields the tile's
//                     frames and exit offset from the first guess on (tile 0 knows its entry: offset 0;
//                     a tile whose lanes do not stitch walks the chain with one lane). The tile's frame
//                     starts go to a scratch row of its own (16-bit offsets inside the tile), its
//                     {entry, frames, exit} to a summary table.
//   k_bounds_resolve  the summary table is scanned (frames: sum, exit: max), one thread per tile, blocks of 256:
//                     every tile's prefix inside its block, every block's aggregate.
//   k_bounds_write    every workgroup folds the block aggregates in front of its block; the sum is the base
//                     index of a tile's offsets, the max must be exactly the tile's guessed entry (or lie
//                     beyond the tile when it found nothing); the scratch rows go to their final indexes.
//
// If every tile passes its check, the guesses ARE the chain (induction from tile 0). A tile that
// fails — payload bytes that look like two consecutive headers right at a tile start, or a
// malformed frame — gets the entry it should have used as a hint; the host reruns the kernels with
// the hints (each run fixes every tile flagged by the one before, and every tile under a frame
// found to cover a flagged tile) and, after a few runs, lets one lane follow the chain
// (k_bounds_seq). Wrong guesses need non-text bytes that mimic a frame AND its successor within the
// ~100 bytes before the first real frame of a tile, so reruns are rare; malformed input (an error
// anyway) costs one.
//
// The result is exactly the sequential rule  p -> p + 1 + L  while buf[p] == 'd', L >= 4 and
// the frame fits; else the rest of the buffer is one last (malformed) frame  — the rule of the
// oracle (there is no host-side scan: when the tiles' guesses do not settle the one-lane kernel k_bounds_seq walks the chain).
#include "lookback.hip.h"

namespace etlg {

constexpr uint32_t TB = 8192;          // bytes per tile
constexpr uint32_t HALO = 32;          // a header (and the bytes the guess inspects) may straddle the tile end
constexpr uint32_t NO_ENTRY = 0xFFFFFFFFu;

constexpr uint32_t CAP = 360;          // frame starts a tile's scratch row holds: the shortest message the server sends is a 23-byte keepalive (8192 / 23 = 356);
                                       // a tile with more (a stream of malformed 5-byte frames) sends the scan to the one-lane fallback

struct TileSum { uint32_t entry, n, e, flags; };   // flags: 1 the chain enters this tile (entry valid), 2 more than CAP frames
struct TilePre { uint32_t n, e; };                 // frames / furthest exit of a block's tiles before this one; block aggregates use the same pair

struct BoundsParams {
  const u8* in;
  uint64_t len;
  uint32_t* offs;              // out: nframes + 1 offsets
  uint32_t offs_cap;           // entries available in offs
  uint32_t ntiles;
  uint32_t* hflag;             // pinned host memory: [0] nframes, [1] set to 1 when the run did not produce the offsets — the host reads the
                               // device result block only then
  TileSum* sum;                // [ntiles]
  TilePre* pre;                // [ntiles] the tile's prefix inside its block of RB tiles (k_bounds_resolve)
  TilePre* blk;                // [ceil(ntiles / RB)] block aggregates
  uint16_t* rows;              // [ntiles][CAP] frame starts of the tile, relative to its first byte
  uint32_t* hints;             // [ntiles] entry to use instead of guessing, NO_ENTRY = guess (kept across reruns)
  uint32_t* result;            // [0] nframes  [1] failure flags (2 offs_cap too small, 4 a tile with more than CAP frames)  [2] tiles that failed their check
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

// ... the same walk, writing the frame starts relative to `lo` as 16-bit values (a tile's scratch row)
DEV void walk16(const u8* st, uint16_t* out, uint32_t lo, uint32_t stop, uint64_t len, uint32_t p) {
  uint32_t n = 0;
  while (p < stop) {
    uint32_t nx;
    out[n++] = (uint16_t)(p - lo);
    p = header_at(st, lo, len, p, nx) ? nx : (uint32_t)len;
  }
}

constexpr uint32_t SUB = TB / 64;  // bytes of the tile each lane looks at (128)

// What a tile knows after its local phase (registers; the LDS window is free again)
struct TileLocal {
  uint32_t lo, hi, a_end;      // the tile's bytes, the end of this lane's 128
  uint32_t entry, n, e;        // guessed entry, frames and exit offset of the tile's chain
  uint32_t s_l, n_l, inc;      // this lane: its guess, the frames of its walk, inclusive prefix of the on-chain lanes' counts
  bool has, stitched, mine;
};

// ---- local phase of one tile: stage, guess, walk, stitch.
DEV void bounds_local(const BoundsParams& q, u8* st, uint32_t tile, TileLocal& t) {
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
  // ---- every lane guesses an entry into ITS 128 bytes: the first position holding a plausible frame
  //      whose successor is plausible too (or lies outside the tile). 'd' bytes are found 4 at a time.
  const uint32_t a = lo + lane * SUB, a_end = a + SUB < hi ? a + SUB : hi;
  t.a_end = a_end;
  uint32_t s_l = NO_ENTRY;
  if (a < hi) {
    // all 128 bytes at once (8 independent 16-byte LDS reads + the dword behind them). A guess is only tried where 'd' is followed by a
    // zero byte, i.e. at frames shorter than 16 MiB (text is full of 'd's; a longer frame is simply never guessed and costs the tile
    // a hinted rerun). The flags stay where their bytes are (bit 7 of each byte: 'd' here AND zero in the byte behind, which for a
    // dword's last byte is the next dword's first) and only WHICH dwords hold a candidate is collected in a 32-bit mask: packing two
    // 128-bit byte masks per lane was 900 of the kernel's ~1 300 instructions, for the one or two candidates a lane has.
    uint32_t w[33];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint4 v = *(const uint4*)(st + (a - lo) + 16 * k);
      w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
    w[32] = *(const uint32_t*)(st + (a - lo) + SUB);   // (the window has a halo: readable for every lane)
    auto zero_flags = [](uint32_t y) { return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu); };   // 0x80 where the byte is 0 (exact)
    uint32_t dwords = 0;   // bit j: dword j of this lane's 128 bytes holds a candidate
    uint32_t zz_next = zero_flags(w[32]);
#pragma unroll
    for (int j = 31; j >= 0; j--) {
      const uint32_t zz = zero_flags(w[j]);
      const uint32_t zd = zero_flags(w[j] ^ 0x64646464u);                            // 0x80 where the byte is 'd'
      const uint32_t cand = zd & __builtin_amdgcn_alignbit(zz_next, zz, 8);           // ... and the byte behind it is 0
      dwords |= cand ? 1u << j : 0u;
      zz_next = zz;
    }
    while (dwords && s_l == NO_ENTRY) {
      const uint32_t j = (uint32_t)__builtin_ctz(dwords);
      dwords &= dwords - 1u;
      const u8* const at = st + (a - lo) + 4u * j;
      const uint32_t x = *(const uint32_t*)at, nx_dw = *(const uint32_t*)(at + 4);
      uint32_t cand = zero_flags(x ^ 0x64646464u) & __builtin_amdgcn_alignbit(zero_flags(nx_dw), zero_flags(x), 8);
      while (cand && s_l == NO_ENTRY) {
        const uint32_t p = a + 4u * j + ((uint32_t)__builtin_ctz(cand) >> 3);
        cand &= cand - 1u;
        uint32_t nx, nx2;
        if (p < a_end && plausible_at(st, lo, q.len, p, nx) && (nx >= hi || plausible_at(st, lo, q.len, nx, nx2))) s_l = p;
      }
    }
  }
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
  if (has && !stitched) {
    // the lanes do not agree (bytes that mimic a frame inside a value, a malformed header ...): one lane
    // walks the tile; it walks again (out of global memory) to write the offsets once their base index is known
    if (lane == 0) walk(st, nullptr, lo, hi, q.len, entry, n, e);
    n = (uint32_t)__shfl(n, 0, 64); e = (uint32_t)__shfl(e, 0, 64);
  }
  t.entry = entry; t.n = n; t.e = e; t.s_l = s_l; t.n_l = n_l; t.inc = inc; t.has = has; t.stitched = stitched; t.mine = mine;
}

// One wave = one tile: local phase, then the tile's frame starts (out of the LDS window: no second look at the input) into its scratch
// row and its {entry, frames, exit} into the summary table. Nothing here depends on another tile.
__global__ __launch_bounds__(64) void k_bounds_local(BoundsParams q) {
  __shared__ __attribute__((aligned(16))) u8 st[TB + HALO + 16];
  const uint32_t lane = threadIdx.x, tile = blockIdx.x;
  TileLocal t;
  bounds_local(q, st, tile, t);
  uint32_t flags = t.has ? 1u : 0u;
  if (t.has) {
    uint16_t* const row = q.rows + (size_t)tile * CAP;
    if (t.n > CAP) flags |= 2u;
    else if (t.stitched) { if (t.mine) walk16(st, row + (t.inc - t.n_l), t.lo, t.a_end, q.len, t.s_l); }   // consecutive lanes write consecutive entries
    else if (lane == 0) walk16(st, row, t.lo, t.hi, q.len, t.entry);
  }
  if (lane == 0) q.sum[tile] = TileSum{t.entry, t.n, t.e, flags};
}

// The exclusive scan of (frames: sum, exit: max) over the summary table, two levels. Level one: a workgroup takes RB consecutive tiles,
// one per thread (coalesced 16-byte loads: ONE workgroup walking the whole table was 15 us of a 45 us scan — every lane on a line of
// its own through one CU's memory pipeline), and leaves every tile's prefix inside its block and the block's aggregate.
constexpr int RB = 256;
__global__ __launch_bounds__(RB) void k_bounds_resolve(BoundsParams q) {
  __shared__ uint32_t s_n[RB / 64], s_e[RB / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t t = blockIdx.x * RB + tid;
  if (blockIdx.x == 0 && tid == 0) { q.result[1] = 0; q.result[2] = 0; }   // (the write kernel behind this one raises them)
  TileSum v{NO_ENTRY, 0, 0, 0};
  if (t < q.ntiles) v = q.sum[t];
  const uint32_t in_n = wave_scan_add(v.n), in_e = wave_scan_max(v.e);
  if (lane == 63) { s_n[wave] = in_n; s_e[wave] = in_e; }
  __syncthreads();
  uint32_t N = in_n - v.n;
  uint32_t E = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)in_e, 0x138, 0xF, 0xF, false);   // the lanes before this one (0 into lane 0)
  for (uint32_t w = 0; w < wave; w++) { N += s_n[w]; E = s_e[w] > E ? s_e[w] : E; }
  if (t < q.ntiles) q.pre[t] = TilePre{N, E};
  if (tid == RB - 1) q.blk[blockIdx.x] = TilePre{N + v.n, v.e > E ? v.e : E};
}

// Level two + the check + the offsets, one WAVE per tile (four tiles per workgroup: the copy wants the whole chip, 32 workgroups of 256
// tiles took 34 us): the wave folds the block aggregates in front of its tile's block (a 64 MiB batch has 32 of them, 4 GiB 2 048: a few
// trips of 64), checks the tile's guess against the chain so far and copies the scratch row to its final indexes. A tile whose guess
// does not hold leaves its hint and raises the run's flags: the host runs the kernels again and ignores what the others wrote (their
// base indexes are wrong behind the first bad tile anyway).
__global__ __launch_bounds__(256) void k_bounds_write(BoundsParams q) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= q.ntiles) return;
  const uint32_t nb = t / RB;   // blocks in front of this tile's
  uint32_t bn = 0, be = 0;
  for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
    const uint32_t b = b0 + lane;
    TilePre a{0, 0};
    if (b < nb) a = q.blk[b];
    bn += wave_last(wave_scan_add(a.n));
    const uint32_t m = wave_last(wave_scan_max(a.e));
    be = m > be ? m : be;
  }
  const TileSum v = q.sum[t];
  const TilePre pr = q.pre[t];
  const uint32_t N = bn + pr.n, E = be > pr.e ? be : pr.e;
  uint32_t fl = (v.flags & 2u) ? 4u : 0u;
  if (t > 0) {
    // If every tile passes this check the guesses are the chain: by induction the exits of the older
    // tiles are the true ones and grow with the tile index, so their maximum is where the chain enters.
    const uint64_t hi64 = ((uint64_t)t + 1) * TB;
    const uint32_t hi = (uint32_t)(hi64 < q.len ? hi64 : q.len);
    const bool ok = (v.flags & 1u) ? E == v.entry : E >= hi;
    if (!ok) {
      if (lane == 0) {
        q.hints[t] = E;   // what this tile should have started from (>= hi: nothing starts here)
        atomicAdd(&q.result[2], 1u);
        q.hflag[1] = 1u;
      }
      // a frame that runs past this tile also covers every tile up to its end: tell them all now (tiles
      // inside one long value that mimics frames agree with each other and would otherwise be found
      // one per run)
      if (E >= hi) { const uint64_t last = (uint64_t)E / TB; for (uint64_t tt = (uint64_t)t + 1 + lane; tt < last && tt < q.ntiles; tt += 64) q.hints[tt] = E; }
    }
  }
  const uint64_t total_here = (uint64_t)N + v.n;
  if (total_here + 1 > q.offs_cap) fl |= 2u;
  if (fl && lane == 0) { atomicOr(&q.result[1], fl); q.hflag[1] = 1u; }
  if (t == q.ntiles - 1 && lane == 0) {
    q.result[0] = (uint32_t)total_here; q.hflag[0] = (uint32_t)total_here;
    if (!(fl & 2u)) q.offs[total_here] = (uint32_t)q.len;
  }
  if ((v.flags & 3u) != 1u || (fl & 2u)) return;   // nothing starts here, or the row overflowed, or the offsets do not fit
  const uint16_t* const row = q.rows + (size_t)t * CAP;
  const uint32_t lo = t * TB;
  for (uint32_t i = lane; i < v.n; i += 64) q.offs[N + i] = lo + row[i];   // consecutive lanes write consecutive entries
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

// Scratch (device, caller-owned), for ntiles = ceil(len / tile bytes): result block (64 bytes) | TileSum[ntiles] | TilePre[ntiles] |
// TilePre[ntiles / 256 + 1] | rows u16[ntiles][CAP]. Nothing in it needs initialising: every run writes what it reads. hints: u32[ntiles], 0xFF-filled (refilled by
// the caller after a run that set any). hflag: 2 u32 of pinned host memory.
size_t etlg_k_bounds_scratch_bytes(size_t ntiles) {
  return 64 + ntiles * sizeof(TileSum) + ntiles * sizeof(TilePre) + (((ntiles / RB + 2) * sizeof(TilePre) + 63) & ~(size_t)63) + ntiles * CAP * 2 + 64;
}

void etlg_k_launch_bounds(const uint8_t* in, uint64_t len, uint32_t* offs, uint32_t offs_cap, void* scratch, uint32_t* result, uint32_t* hints, uint32_t* hflag, int sequential, hipStream_t s) {
  BoundsParams q;
  q.in = in; q.len = len; q.offs = offs; q.offs_cap = offs_cap;
  q.ntiles = (uint32_t)((len + TB - 1) / TB);
  q.result = result ? result : (uint32_t*)scratch;   // (a batch decoded behind its scan keeps the result words in a block of its own, host.cpp)
  q.sum = (TileSum*)((uint8_t*)scratch + 64);
  q.pre = (TilePre*)(q.sum + q.ntiles);
  q.blk = q.pre + q.ntiles;
  q.rows = (uint16_t*)((uint8_t*)q.blk + ((((size_t)q.ntiles / RB + 2) * sizeof(TilePre) + 63) & ~(size_t)63));
  q.hints = hints; q.hflag = hflag;
  if (sequential) { hipLaunchKernelGGL(k_bounds_seq, dim3(1), dim3(64), 0, s, q); return; }
  hipLaunchKernelGGL(k_bounds_local, dim3(q.ntiles), dim3(64), 0, s, q);
  hipLaunchKernelGGL(k_bounds_resolve, dim3((q.ntiles + RB - 1) / RB), dim3(RB), 0, s, q);
  hipLaunchKernelGGL(k_bounds_write, dim3((q.ntiles + 3) / 4), dim3(256), 0, s, q);
}

}  // extern "C"
