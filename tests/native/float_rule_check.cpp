// Two independent statements of "which float texts the device decodes itself" must agree: the oracle's
// (oracle/oracle_codec.hpp float_device_rule: rounding of the two bracketing 19-digit decimals with glibc) and the device's
// (etl_amd/csrc/float_fast.h: Clinger + Eisel-Lemire). Built and run by tests/test_float_fast.py.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "float_fast.h"
#include "oracle_codec.hpp"

int main() {
  unsigned long long n = 0, mism = 0, deferred = 0, longm = 0;
  auto check = [&](const std::string& s) {
    for (int is32 = 0; is32 < 2; is32++) {
      uint64_t out = 0;
      const int dev = etlg::parse_float_fast_t([&](uint32_t i) { return (uint32_t)(unsigned char)s[i]; }, (uint32_t)s.size(), is32 != 0, out);
      const int orc = orc::float_device_rule(std::string_view(s), is32 != 0);
      n++;
      if (dev == 1) deferred++;
      if (dev != orc && mism++ < 10) printf("verdicts differ on '%s' (%s): device %d, oracle %d\n", s.c_str(), is32 ? "f32" : "f64", dev, orc);
    }
  };
  const char* fixed[] = {"0", "-0.000e-5", "1", "0.1", "1e22", "1e23", "9007199254740993", "12345678901234567890", "1.00000005960464477539",
                         "50537618.817359292015891086651596749e82", "107896223265412489690691363e88", "28879636596541978310003766487.741e-212",
                         "679604465747276.5742380775679666e23", "7137255.607280446341269933e14", "5693107746173304490483329377e264",
                         "1.7976931348623158e308", "2.4703282292062327e-324", "1e-400", "1e400", "99999999999999999999", "0.000000000000000000000000000001234567890123456789012",
                         "100000000000000000000000000000000000001", "1000000000000000000010000", "inf", "nan", "", "1e", "x"};
  for (const char* f : fixed) check(f);
  srand(11);
  for (long it = 0; it < 1500000; it++) {
    std::string s;
    const int r = rand();
    if (r % 16 == 0) s += "-"; else if (r % 16 == 1) s += "+";
    const int nd = 1 + rand() % 30;
    if (nd > 19) longm++;
    const int dot = (rand() % 3) ? rand() % (nd + 1) : -1;
    for (int i = 0; i < nd; i++) { if (i == dot) s += "."; s += (char)('0' + ((rand() % 4) ? rand() % 10 : 0)); }
    if (dot == nd) s += ".";
    if (rand() % 2) { s += (rand() % 2) ? "e" : "E"; if (rand() % 3 == 0) s += (rand() % 2) ? "-" : "+"; s += std::to_string(rand() % ((rand() % 8) ? 45 : 400)); }
    check(s);
  }
  // decimals right around float / double midpoints: 17-20 digit expansions of (m + 1/2) ulp with the tail perturbed
  for (long it = 0; it < 300000; it++) {
    const double base = ldexp(1.0 + (double)(rand() % (1 << 20)) / (1 << 20), (rand() % 80) - 40);
    char buf[128];
    snprintf(buf, sizeof buf, "%.*e", 17 + rand() % 14, nextafter(base, INFINITY) * 0.5 + base * 0.5);
    check(buf);
  }
  printf("texts x widths %llu, long mantissas %llu, deferred by the device %llu, mismatches %llu\n", n, longm, deferred, mism);
  return mism ? 1 : 0;
}
