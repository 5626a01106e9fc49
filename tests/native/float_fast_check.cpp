// Host-side check of etl_amd/csrc/float_fast.h (the device's float4 / float8 text parser) against glibc
// strtod / strtof, which are correctly rounded like Rust's dec2flt. Built and run by tests/test_float_fast.py.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "float_fast.h"

// independent restatement of the dec2flt grammar (what is a float text at all)
static bool grammar(const std::string& s, bool& special) {
  special = false;
  size_t i = 0;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) i++;
  if (i >= s.size()) return false;
  std::string r = s.substr(i);
  std::string l;
  for (char c : r) l.push_back((c >= 'A' && c <= 'Z') ? c + 32 : c);
  if (l == "inf" || l == "infinity" || l == "nan") { special = true; return true; }
  size_t nd = 0;
  while (i < s.size() && s[i] >= '0' && s[i] <= '9') { i++; nd++; }
  if (i < s.size() && s[i] == '.') { i++; while (i < s.size() && s[i] >= '0' && s[i] <= '9') { i++; nd++; } }
  if (!nd) return false;
  if (i < s.size() && (s[i] == 'e' || s[i] == 'E')) {
    i++;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) i++;
    size_t ne = 0;
    while (i < s.size() && s[i] >= '0' && s[i] <= '9') { i++; ne++; }
    if (!ne) return false;
  }
  return i == s.size();
}

static unsigned long long cases = 0, values = 0, deferred = 0, bad = 0, mism = 0, swar_taken = 0, swar_left = 0;

static void check(const std::string& s) {
  for (int is32 = 0; is32 < 2; is32++) {
    uint64_t out = 0;
    const int r = etlg::parse_float_fast_t([&](uint32_t i) { return (uint32_t)(unsigned char)s[i]; }, (uint32_t)s.size(), is32 != 0, out);
    {  // the register front end (parse_float_swar) must agree with the loop on every text it takes: same return code, same bits;
       // the bytes behind the text are garbage, as they are on the device
      unsigned char raw[24];
      for (int k = 0; k < 24; k++) raw[k] = k < (int)s.size() ? (unsigned char)s[k] : (unsigned char)(rand() & 0xFF);
      uint64_t x[3]; memcpy(x, raw, 24);
      uint64_t out2 = 0;
      const int r2 = etlg::parse_float_swar(x[0], x[1], x[2], (uint32_t)s.size(), is32 != 0, out2);
      if (r2 == 3) swar_left++;
      else {
        swar_taken++;
        if (r2 != r || (r == 0 && out2 != out)) { if (mism++ < 10) printf("register front end differs for '%s' (%s): %d %llx vs %d %llx\n", s.c_str(), is32 ? "f32" : "f64", r2, (unsigned long long)out2, r, (unsigned long long)out); }
      }
    }
    bool special;
    const bool ok = grammar(s, special);
    cases++;
    if (r == 2) { bad++; if (ok) { if (mism++ < 10) printf("rejects a valid text: '%s'\n", s.c_str()); } continue; }
    if (!ok) { if (mism++ < 10) printf("accepts an invalid text: '%s' (r=%d)\n", s.c_str(), r); continue; }
    if (r == 1) { deferred++; continue; }
    values++;
    uint64_t want;
    if (is32) { float f = strtof(s.c_str(), nullptr); uint32_t b; memcpy(&b, &f, 4); want = b; }
    else { double d = strtod(s.c_str(), nullptr); memcpy(&want, &d, 8); }
    {
      std::string l;
      for (char c : s) l.push_back((c >= 'A' && c <= 'Z') ? c + 32 : c);
      if (special && l.find("nan") != std::string::npos) {
        // NaN: compare as "NaN with the same sign" (payload bits are not part of the value)
        const bool neg = s[0] == '-';
        want = is32 ? (0x7FC00000u | (neg ? 0x80000000u : 0u)) : (0x7FF8000000000000ull | (neg ? 0x8000000000000000ull : 0ull));
      }
    }
    if (out != want) { if (mism++ < 10) printf("wrong bits for '%s' (%s): got %llx want %llx\n", s.c_str(), is32 ? "f32" : "f64", (unsigned long long)out, (unsigned long long)want); }
  }
}

int main() {
  const char* fixed[] = {"0", "-0", "+0", "0.0", "-0.0", "0e0", "0e999999999", "-0.000e-5", "1", "-1", "1.5", "3.5", "0.1", "0.2", "0.3", "1e22", "1e23",
                         "1e-22", "1e-23", "9007199254740992", "9007199254740993", "9007199254740991", "12345678901234567890", "16777216", "16777217",
                         "33554434", "8388608.5", "8388609.5", "1.0000000596046448", "1.00000005960464477539", "3.4028235e38", "3.4028236e38", "1e-45",
                         "inf", "-inf", "Infinity", "-INFINITY", "+inf", "nan", "NaN", "-nan", "1.", ".5", "-.5e1", "+1.25E+2", "100000000000000000000000",
                         "", "+", "-", ".", "e5", "1e", "1e+", "1.2.3", "1 ", " 1", "0x10", "1_0", "infinit", "nane", "--1", "1e5.0", "1f", "in", "na", "i", "n"};
  for (const char* f : fixed) check(f);
  srand(7);
  for (long it = 0; it < 3000000; it++) {
    std::string s;
    const int r = rand();
    if (r % 16 == 0) s += "-"; else if (r % 16 == 1) s += "+";
    const int nd = 1 + rand() % 21;
    const int dot = (rand() % 3) ? rand() % (nd + 1) : -1;
    for (int i = 0; i < nd; i++) { if (i == dot) s += "."; s += (char)('0' + ((rand() % 4) ? rand() % 10 : 0)); }
    if (dot == nd) s += ".";
    if (rand() % 2) { s += (rand() % 2) ? "e" : "E"; if (rand() % 3 == 0) s += (rand() % 2) ? "-" : "+"; s += std::to_string(rand() % 45); }
    if (rand() % 50 == 0) s[rand() % s.size()] = "x _-+.e"[rand() % 7];  // sprinkle malformed ones
    check(s);
  }
  // wide exponents (overflow to inf, subnormals, underflow to zero) and long mantissas (truncation: w and w + 1)
  for (long it = 0; it < 2000000; it++) {
    std::string s;
    if (rand() % 8 == 0) s += "-";
    const int nd = 1 + rand() % 40;
    const int dot = (rand() % 2) ? rand() % (nd + 1) : -1;
    for (int i = 0; i < nd; i++) { if (i == dot) s += "."; s += (char)('0' + rand() % 10); }
    s += "e"; if (rand() % 2) s += "-";
    s += std::to_string(rand() % 420);
    check(s);
  }
  const char* edge[] = {"1e308", "1.7976931348623157e308", "1.7976931348623158e308", "1.7976931348623159e308", "1.8e308", "1e309", "2.2250738585072014e-308",
                        "2.2250738585072011e-308", "4.9406564584124654e-324", "2.4703282292062327e-324", "2.4703282292062328e-324", "1e-324", "1e-400",
                        "3.4028234663852886e38", "3.4028235677973366e38", "3.4028235677973367e38", "1.1754943508222875e-38", "1.4012984643248171e-45",
                        "7.0064923216240854e-46", "7.0064923216240862e-46", "1e-46", "8.5e-46", "0.000000000000000000000000000000000000000000001",
                        "9007199254740993", "9007199254740995", "18014398509481990", "123456789012345678901234567890", "0.1e1000", "1e-1000", "179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365510704342711559699508093042880177904174497791.9999999999999999999999999999999999999999999999999999999999999999999999"};
  for (const char* f : edge) check(f);
  // float midpoints: doubles that lie exactly between two floats must not be rounded twice
  for (uint32_t m = 0x3F800000u; m < 0x3F800000u + 4000; m++) {
    float a; memcpy(&a, &m, 4);
    uint32_t m2 = m + 1; float b2; memcpy(&b2, &m2, 4);
    const double mid = ((double)a + (double)b2) / 2;
    char buf[64];
    snprintf(buf, sizeof buf, "%.17g", mid); check(buf);
    snprintf(buf, sizeof buf, "%.9g", a); check(buf);
  }
  // texts as PostgreSQL prints float8 / float4 (shortest round-trip digits, exponents as e+NN / e-NN) and a dense sweep of short ones
  for (long it = 0; it < 1500000; it++) {
    char buf[64];
    uint64_t bits = ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ (uint64_t)rand();
    double d; memcpy(&d, &bits, 8);
    if (d != d) continue;
    if (it % 3 == 0) snprintf(buf, sizeof buf, "%.17g", d);
    else if (it % 3 == 1) snprintf(buf, sizeof buf, "%.9g", (double)(float)d);
    else snprintf(buf, sizeof buf, "%.*g", 1 + rand() % 17, d / (1 + rand() % 1000));
    check(buf);
  }
  const char* alpha = "0123456789.eE+-0000";
  for (long it = 0; it < 1500000; it++) {
    std::string s;
    const int len = 1 + rand() % 10;
    for (int i = 0; i < len; i++) s += alpha[rand() % 19];
    check(s);
  }
  printf("cases %llu values %llu deferred %llu malformed %llu mismatches %llu (register front end: took %llu, left %llu to the loop)\n", cases, values, deferred, bad, mism, swar_taken, swar_left);
  return mism != 0 || values < cases / 4 || swar_taken < cases / 2;
}
