// Host-side check of etl_amd/csrc/float_slow.h (the exact decimal -> binary fallback of the finish pass) against glibc strtod / strtof,
// which are correctly rounded like Rust's dec2flt. Built and run by tests/test_float_fast.py. TEST INFRASTRUCTURE.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "float_fast.h"
#include "float_slow.h"

static unsigned long long cases = 0, mism = 0, inconclusive = 0;
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 7; rng_state ^= rng_state >> 9; return rng_state * 0x2545F4914F6CDD1Dull; }

static void check(const std::string& s) {
  for (int is32 = 0; is32 < 2; is32++) {
    uint64_t fast = 0;
    const int r = etlg::parse_float_fast_t([&](uint32_t i) { return (uint32_t)(unsigned char)s[i]; }, (uint32_t)s.size(), is32 != 0, fast);
    if (r == 2) { if (mism++ < 10) printf("generator produced a malformed text '%s'\n", s.c_str()); continue; }
    if (r == 1) inconclusive++;
    const uint64_t got = etlg::parse_float_exact_t([&](uint32_t i) { return (uint32_t)(unsigned char)s[i]; }, (uint32_t)s.size(), is32 != 0);
    uint64_t want = 0;
    if (is32) { const float f = strtof(s.c_str(), nullptr); uint32_t b; memcpy(&b, &f, 4); want = b; }
    else { const double f = strtod(s.c_str(), nullptr); memcpy(&want, &f, 8); }
    cases++;
    if (got != want) { if (mism++ < 10) printf("'%s' (%s): %llx, strtod says %llx\n", s.c_str(), is32 ? "f32" : "f64", (unsigned long long)got, (unsigned long long)want); }
    if (r == 0 && fast != want) { if (mism++ < 10) printf("fast path differs for '%s'\n", s.c_str()); }
  }
}

static std::string digits(int n) { std::string s; for (int i = 0; i < n; i++) s.push_back('0' + (int)(rnd() % 10)); return s; }

int main() {
  // (a) random decimals: 1..40 digits, a point anywhere, exponents over the whole range and beyond both ends
  for (int it = 0; it < 400000; it++) {
    std::string s;
    if (rnd() % 4 == 0) s.push_back(rnd() % 2 ? '-' : '+');
    const int n = 1 + (int)(rnd() % 40);
    std::string d = digits(n);
    if (rnd() % 2) d.insert(rnd() % (d.size() + 1), ".");
    if (d == ".") d = "0.";
    s += d;
    if (rnd() % 3) { char e[32]; snprintf(e, sizeof e, "%c%d", rnd() % 2 ? 'e' : 'E', (int)(rnd() % 700) - 350); s += e; }
    check(s);
  }
  // (b) exact half-way points between neighbouring floats / doubles (their full decimal expansions, up to ~770 digits), and the texts one
  //     unit above / below in the last place, with and without a long tail of zeros and a trailing non-zero digit (the `truncated` flag)
  for (int it = 0; it < 60000; it++) {
    char buf[1400];
    std::string mid;
    if (it % 2) {   // binary32: the midpoint is a double
      uint32_t b = (uint32_t)rnd() & 0x7FFFFFFFu;
      if (it % 7 == 0) b &= 0x007FFFFFu;                  // subnormals
      if ((b >> 23) == 0xFF) continue;
      float f; memcpy(&f, &b, 4);
      const float g = nextafterf(f, INFINITY);
      if (isinf(g)) continue;
      const double m = ((double)f + (double)g) / 2;
      snprintf(buf, sizeof buf, "%.200e", m);
    } else {        // binary64: the midpoint has 54 bits — a long double holds it
      uint64_t b = rnd() & 0x7FFFFFFFFFFFFFFFull;
      if (it % 6 == 0) b &= 0x000FFFFFFFFFFFFFull;        // subnormals
      if (((b >> 52) & 0x7FF) == 0x7FF) continue;
      double f; memcpy(&f, &b, 8);
      const double g = nextafter(f, INFINITY);
      if (isinf(g)) continue;
      const long double m = ((long double)f + (long double)g) / 2;
      snprintf(buf, sizeof buf, "%.1100Le", m);
    }
    mid = buf;
    // strip the zeros the format padded behind the exact expansion
    const size_t e = mid.find('e');
    std::string mant = mid.substr(0, e), ex = mid.substr(e);
    while (mant.size() > 2 && mant.back() == '0') mant.pop_back();
    check(mant + ex);
    check(mant + "0000000000000000000000001" + ex);          // just above the midpoint
    {  // just below: the last digit down by one (it is non-zero after the strip)
      std::string lo = mant; lo.back() = (char)(lo.back() - 1);
      check(lo + "9999999999" + ex);
    }
    if (mant.size() < 760) { std::string z = mant + std::string(800 - mant.size(), '0'); check(z + "1" + ex); check(z + ex); }   // beyond 768 digits: truncated
  }
  // (c) 17-25 random significant digits — where Eisel-Lemire is sometimes inconclusive
  for (int it = 0; it < 1500000; it++) {
    std::string s = std::to_string(1 + rnd() % 9) + "." + digits(16 + (int)(rnd() % 9));
    char e[32]; snprintf(e, sizeof e, "e%d", (int)(rnd() % 640) - 330); s += e;
    check(s);
  }
  // (d) edges
  const char* edge[] = {"0", "-0", "0.0", "0e999", "1e-400", "4.9e-324", "2.4703282292062327e-324", "2.4703282292062328e-324", "1.7976931348623157e308", "1.7976931348623158e308",
                        "1.7976931348623159e308", "1e309", "1e400", "1e-46", "7e-46", "1.4e-45", "3.4028235e38", "3.4028236e38", "3.402823466385288598117041834845169254401e38",
                        "0.000000000000000000000000000000000000000000001", "123456789012345678901234567890", "9007199254740993", "9007199254740992.5", "1e23", "8.5e-1",
                        "00000000000000000000000000000000000000001.5e0000000000000000000003", "1e99999", "1e-99999", ".5", "5."};
  for (const char* s : edge) check(s);
  printf("cases %llu, of them inconclusive for the fast path %llu, mismatches %llu\n", cases, inconclusive, mism);
  return mism ? 1 : 0;
}
