// UTF-8 validation four bytes at a time, shared by the kernels and by a host-side unit test
// (tests/test_utf8_swar.py compiles this header with g++ and checks it exhaustively against a
// byte-serial restatement of core::str::from_utf8).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define ETLG_HD __host__ __device__ __forceinline__
#else
#define ETLG_HD static inline
#endif

namespace etlg {

// core::str::from_utf8 restated position-wise (strict RFC 3629): byte i must be a continuation
// byte exactly when one of the three bytes before it opens a sequence that reaches i, lead bytes
// C0, C1, F5.. never occur, and the second byte of E0 / ED / F0 / F4 sequences is range-limited.
// `prev` = the 4 bytes before `cur` (0 at the start of the text); bytes of `cur` past the end of
// the text are 0, which also catches a sequence cut off by the end; `at_end`: cur's last byte is
// the text's last byte.
ETLG_HD bool utf8_dword_bad(uint32_t prev, uint32_t cur, bool at_end) {
  // four positions at once: every predicate lives in bit 7 of its byte (other bits are garbage
  // until the final mask). bit k of a byte is brought to bit 7 by a left shift of 7 - k.
  if ((((cur & (cur << 1) & (cur << 2)) | (prev & (prev << 1) & (prev << 2))) & 0x80808080u) == 0) {
    // nothing in E0..FF here or just before (text of one- and two-byte characters, the common case): the only
    // sequences are C2..DF + one continuation byte, and the range rules of the longer ones cannot apply
    const uint32_t b6 = cur << 1;
    const uint32_t ge_c0 = cur & b6, is_cont = cur & ~b6;
    const uint32_t p1 = (prev >> 24) | (cur << 8);       // the byte before each position
    uint32_t bad = ((p1 & (p1 << 1)) ^ is_cont) | (ge_c0 & ~((cur << 2) | (cur << 3) | (cur << 4) | (cur << 5) | (cur << 6)));  // ... | C0, C1
    bad &= 0x80808080u;
    if (at_end) bad |= ge_c0 & 0x80000000u;
    return bad != 0;
  }
  const uint64_t W = prev | ((uint64_t)cur << 32);
  const uint32_t p1 = (uint32_t)(W >> 24), p2 = (uint32_t)(W >> 16), p3 = (uint32_t)(W >> 8);  // the 1st/2nd/3rd byte before
  const uint32_t b6 = cur << 1, b5 = cur << 2, b4 = cur << 3, b3 = cur << 4, b2 = cur << 5, b1 = cur << 6, b0 = cur << 7;
  const uint32_t ge_c0 = cur & b6;                       // 11xxxxxx
  const uint32_t is_cont = cur & ~b6;                    // 10xxxxxx
  const uint32_t must = (p1 & (p1 << 1)) | (p2 & (p2 << 1) & (p2 << 2)) | (p3 & (p3 << 1) & (p3 << 2) & (p3 << 3));
  uint32_t bad = must ^ is_cont;
  bad |= ge_c0 & ~(b5 | b4 | b3 | b2 | b1);              // C0, C1
  bad |= ge_c0 & b5 & b4 & (b3 | (b2 & (b1 | b0)));      // F5 .. FF
  auto eq = [](uint32_t v, uint32_t k) {                 // bit 7: byte == k
    const uint32_t x = v ^ (k * 0x01010101u);
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x);
  };
  bad |= eq(p1, 0xE0u) & ~b5;                            // E0 must be followed by A0..BF
  bad |= eq(p1, 0xEDu) & b5;                             // ED by 80..9F (no surrogates)
  bad |= eq(p1, 0xF0u) & ~(b5 | b4);                     // F0 by 90..BF
  bad |= eq(p1, 0xF4u) & (b5 | b4);                      // F4 by 80..8F
  bad &= 0x80808080u;
  if (at_end) bad |= (ge_c0 & 0x80000000u) | (cur & b6 & b5 & 0x00800000u) | (cur & b6 & b5 & b4 & 0x00008000u);
  return bad != 0;
}

}  // namespace etlg
