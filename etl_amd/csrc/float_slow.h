// float4 / float8 text -> IEEE bits when float_fast.h says "inconclusive": the exact fallback (round 6).
//
// Rust's `str::parse::<f32 / f64>` (core::num::dec2flt; the reference's call sites: crates/etl/src/postgres/codec/text.rs:52-59)
// is correctly rounded for EVERY input: when Clinger's path and Eisel-Lemire cannot decide it falls back to
// dec2flt::slow::parse_long_mantissa — the "simple decimal conversion" (N. Tao; Go's strconv/decimal.go): the
// mantissa as up to 768 decimal digits plus a "more non-zero digits were dropped" flag, shifted by powers of two
// until it sits in [1/2, 1), then shifted left by the mantissa width and rounded half-to-even. 768 digits are
// enough to separate any decimal from every half-way point of binary64, so the result is THE correctly rounded
// value — the same bits glibc strtod / strtof return, which is what the oracle compares against
// (tests/native/float_slow_check.cpp: millions of hard cases through a host build of this header).
//
// Used only by the finish pass (columns.hip: k_fin_fill), one thread per cell: the digit buffer lives in the
// thread's private memory, so no decode kernel pays for it.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define ETLG_FS __device__ __attribute__((noinline))
#else
#define ETLG_FS static
#endif

namespace etlg {

constexpr uint32_t kFsMaxDigits = 768;
struct FsDecimal {
  uint32_t num_digits;
  int32_t decimal_point;
  bool truncated;
  uint8_t digits[kFsMaxDigits];
};

// digits of (d * 2^shift) minus digits of d: found by a dry run of the multiplication (the carry that leaves the top)
ETLG_FS uint32_t fs_new_digits(const FsDecimal& d, uint32_t shift) {
  uint64_t n = 0;
  for (uint32_t r = d.num_digits; r != 0;) { r--; n += (uint64_t)d.digits[r] << shift; n /= 10; }
  uint32_t k = 0;
  while (n) { n /= 10; k++; }
  return k;
}
ETLG_FS void fs_trim(FsDecimal& d) {
  while (d.num_digits != 0 && d.digits[d.num_digits - 1] == 0) d.num_digits--;
}
ETLG_FS void fs_left_shift(FsDecimal& d, uint32_t shift) {   // shift <= 60
  if (d.num_digits == 0) return;
  const uint32_t nn = fs_new_digits(d, shift);
  uint32_t r = d.num_digits, w = d.num_digits + nn;
  uint64_t n = 0;
  while (r != 0) {
    r--; w--;
    n += (uint64_t)d.digits[r] << shift;
    const uint64_t q = n / 10, rem = n - 10 * q;
    if (w < kFsMaxDigits) d.digits[w] = (uint8_t)rem; else if (rem) d.truncated = true;
    n = q;
  }
  while (n) {
    w--;
    const uint64_t q = n / 10, rem = n - 10 * q;
    if (w < kFsMaxDigits) d.digits[w] = (uint8_t)rem; else if (rem) d.truncated = true;
    n = q;
  }
  d.num_digits += nn;
  if (d.num_digits > kFsMaxDigits) d.num_digits = kFsMaxDigits;
  d.decimal_point += (int32_t)nn;
  fs_trim(d);
}
ETLG_FS void fs_right_shift(FsDecimal& d, uint32_t shift) {  // shift <= 60
  uint32_t r = 0, w = 0;
  uint64_t n = 0;
  while ((n >> shift) == 0) {
    if (r < d.num_digits) n = 10 * n + d.digits[r++];
    else if (n == 0) return;
    else { while ((n >> shift) == 0) { n *= 10; r++; } break; }
  }
  d.decimal_point -= (int32_t)r - 1;
  if (d.decimal_point < -2047) { d.num_digits = 0; d.decimal_point = 0; d.truncated = false; return; }
  const uint64_t mask = (1ull << shift) - 1;
  while (r < d.num_digits) {
    const uint8_t nd = (uint8_t)(n >> shift);
    n = 10 * (n & mask) + d.digits[r++];
    d.digits[w++] = nd;
  }
  while (n) {
    const uint8_t nd = (uint8_t)(n >> shift);
    n = 10 * (n & mask);
    if (w < kFsMaxDigits) d.digits[w++] = nd; else if (nd) d.truncated = true;
  }
  d.num_digits = w;
  fs_trim(d);
}
ETLG_FS uint64_t fs_round(const FsDecimal& d) {
  if (d.num_digits == 0 || d.decimal_point < 0) return 0;
  if (d.decimal_point > 18) return ~0ull;
  const uint32_t dp = (uint32_t)d.decimal_point;
  uint64_t n = 0;
  for (uint32_t i = 0; i < dp; i++) { n *= 10; if (i < d.num_digits) n += d.digits[i]; }
  bool up = false;
  if (dp < d.num_digits) {
    up = d.digits[dp] >= 5;
    if (d.digits[dp] == 5 && dp + 1 == d.num_digits) up = d.truncated || (dp != 0 && (d.digits[dp - 1] & 1));   // exactly half: to even
  }
  return up ? n + 1 : n;
}

// The text is one parse_float_fast_t accepts as a finite decimal (float_fast.h said 1 = inconclusive): [+-] digits [. digits] [e[+-]digits].
template <class At>
ETLG_FS void fs_parse(At at, uint32_t n, FsDecimal& d, bool& neg) {
  d.num_digits = 0; d.decimal_point = 0; d.truncated = false;
  uint32_t i = 0;
  neg = false;
  if (n && (at(0) == '+' || at(0) == '-')) { neg = at(0) == '-'; i = 1; }
  uint32_t total = 0;          // significant digits seen (leading zeros excluded), kept or not
  int32_t point = 0;           // position of the decimal point relative to the first significant digit
  bool seen_point = false, leading = true;
  uint32_t last_nz = 0;        // total at the last non-zero digit: the digits behind it are trailing zeros
  for (; i < n; i++) {
    const uint32_t c = at(i);
    if (c == '.') { seen_point = true; continue; }
    const uint32_t v = c - '0';
    if (v > 9) break;
    if (leading && v == 0) { if (seen_point) point--; continue; }
    leading = false;
    if (total < kFsMaxDigits) d.digits[total] = (uint8_t)v; else if (v) d.truncated = true;
    total++;
    if (v) last_nz = total;
    if (!seen_point) point++;
  }
  d.num_digits = last_nz < kFsMaxDigits ? last_nz : kFsMaxDigits;
  fs_trim(d);
  d.decimal_point = point;
  if (i < n) {   // e / E
    i++;
    bool eneg = false;
    if (i < n && (at(i) == '+' || at(i) == '-')) { eneg = at(i) == '-'; i++; }
    int32_t ex = 0;
    for (; i < n; i++) { const uint32_t v = at(i) - '0'; if (v > 9) break; if (ex < 0x10000) ex = ex * 10 + (int32_t)v; }
    d.decimal_point += eneg ? -ex : ex;
  }
}

// Correctly rounded bits of the decimal text (sign included). is32: binary32 in the low word.
template <class At>
ETLG_FS uint64_t parse_float_exact_t(At at, uint32_t n, bool is32) {
  const int32_t mant_bits = is32 ? 23 : 52, min_exp = is32 ? -127 : -1023, inf_power = is32 ? 0xFF : 0x7FF;
  FsDecimal d;
  bool neg;
  fs_parse(at, n, d, neg);
  const uint64_t sign = neg ? (is32 ? 0x80000000ull : 0x8000000000000000ull) : 0ull;
  const uint64_t inf = (is32 ? 0x7F800000ull : 0x7FF0000000000000ull) | sign;
  auto get_shift = [](uint32_t k) -> uint32_t {
    const uint8_t powers[19] = {0, 3, 6, 9, 13, 16, 19, 23, 26, 29, 33, 36, 39, 43, 46, 49, 53, 56, 59};
    return k < 19 ? powers[k] : 60u;
  };
  if (d.num_digits == 0 || d.decimal_point < -324) return sign;
  if (d.decimal_point >= 310) return inf;
  int32_t exp2 = 0;
  while (d.decimal_point > 0) {
    const uint32_t shift = get_shift((uint32_t)d.decimal_point);
    fs_right_shift(d, shift);
    if (d.decimal_point < -2047) return sign;
    exp2 += (int32_t)shift;
  }
  while (d.decimal_point <= 0) {
    uint32_t shift;
    if (d.decimal_point == 0) {
      const uint8_t d0 = d.digits[0];
      if (d0 >= 5) break;
      shift = d0 < 2 ? 2u : 1u;
    } else shift = get_shift((uint32_t)(-d.decimal_point));
    fs_left_shift(d, shift);
    if (d.decimal_point > 2047) return inf;
    exp2 -= (int32_t)shift;
  }
  exp2 -= 1;   // the value is in [1/2, 1): its binary exponent
  while (min_exp + 1 > exp2) {
    uint32_t k = (uint32_t)((min_exp + 1) - exp2);
    if (k > 60) k = 60;
    fs_right_shift(d, k);
    exp2 += (int32_t)k;
  }
  if (exp2 - min_exp >= inf_power) return inf;
  fs_left_shift(d, (uint32_t)mant_bits + 1);
  uint64_t mant = fs_round(d);
  if (mant >= (1ull << (mant_bits + 1))) {   // the rounding carried into another bit
    fs_right_shift(d, 1);
    exp2 += 1;
    mant = fs_round(d);
    if (exp2 - min_exp >= inf_power) return inf;
  }
  int32_t power2 = exp2 - min_exp;
  if (mant < (1ull << mant_bits)) power2 -= 1;   // subnormal
  mant &= (1ull << mant_bits) - 1;
  return sign | ((uint64_t)(uint32_t)power2 << mant_bits) | mant;
}

}  // namespace etlg
