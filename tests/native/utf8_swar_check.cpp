// Host-side exhaustive check of etl_amd/csrc/utf8_swar.h against a byte-serial validator that
// follows core::str::from_utf8 (strict RFC 3629). Built and run by tests/test_utf8_swar.py.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "utf8_swar.h"

static bool serial_valid(const uint8_t* s, uint32_t n) {
  uint32_t i = 0;
  while (i < n) {
    const uint32_t c = s[i];
    if (c < 0x80) { i++; continue; }
    uint32_t need, lo = 0x80, hi = 0xBF;
    if (c >= 0xC2 && c <= 0xDF) need = 1;
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; if (c == 0xE0) lo = 0xA0; if (c == 0xED) hi = 0x9F; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; if (c == 0xF0) lo = 0x90; if (c == 0xF4) hi = 0x8F; }
    else return false;
    if (i + need >= n) return false;  // cut off by the end of the text
    if (s[i + 1] < lo || s[i + 1] > hi) return false;
    for (uint32_t k = 2; k <= need; k++) if ((s[i + k] & 0xC0) != 0x80) return false;
    i += need + 1;
  }
  return true;
}

// the way the kernels use it: dword w of the text with the dword before it, tail zero-masked
static bool swar_valid(const uint8_t* s, uint32_t n) {
  bool bad = false;
  for (uint32_t w = 0; w < (n + 3) / 4; w++) {
    const uint32_t rem = n - 4 * w;
    uint32_t x = 0, prev = 0;
    memcpy(&x, s + 4 * w, rem < 4 ? rem : 4);
    if (w) memcpy(&prev, s + 4 * w - 4, 4);
    if ((x | prev) & 0x80808080u) bad |= etlg::utf8_dword_bad(prev, x, rem == 4);
  }
  return !bad;
}

int main() {
  static const uint8_t B[] = {0x00, 0x41, 0x7f, 0x80, 0x8f, 0x90, 0x9f, 0xa0, 0xbf, 0xc0, 0xc1, 0xc2, 0xdf, 0xe0,
                              0xe1, 0xec, 0xed, 0xee, 0xef, 0xf0, 0xf1, 0xf3, 0xf4, 0xf5, 0xf7, 0xf8, 0xff};
  const int K = sizeof(B);
  uint8_t buf[24];
  unsigned long long cases = 0, mism = 0;
  for (int len = 1; len <= 4; len++) {
    long tot = 1;
    for (int i = 0; i < len; i++) tot *= K;
    for (long c = 0; c < tot; c++) {
      uint8_t seq[4];
      long t = c;
      for (int i = 0; i < len; i++) { seq[i] = B[t % K]; t /= K; }
      for (int p = 0; p < 5; p++) for (int q = 0; q < 3; q++) {
        uint32_t L = 0;
        for (int i = 0; i < p; i++) buf[L++] = 'a';
        for (int i = 0; i < len; i++) buf[L++] = seq[i];
        for (int i = 0; i < q; i++) buf[L++] = 'b';
        cases++;
        if (serial_valid(buf, L) != swar_valid(buf, L)) {
          if (mism < 8) { printf("mismatch:"); for (uint32_t i = 0; i < L; i++) printf(" %02x", buf[i]); printf("\n"); }
          mism++;
        }
      }
    }
  }
  srand(1);
  for (long it = 0; it < 2000000; it++) {
    const uint32_t L = rand() % 17;
    for (uint32_t i = 0; i < L; i++) buf[i] = (rand() & 3) ? B[rand() % K] : 'a';
    cases++;
    if (serial_valid(buf, L) != swar_valid(buf, L)) mism++;
  }
  printf("cases %llu mismatches %llu\n", cases, mism);
  return mism != 0;
}
