"""json_valid / json_display (etl_amd/csrc/columns.hip) as plain host C++ under the sanitizers: the functions are cut out of the source,
compiled with g++ -fsanitize=address,undefined and with ROCm's clang++ -fsanitize=memory, and run over seeded random documents (the
generator of tests/test_gpu_json_display.py) from heap blocks of the exact size — a read past a text's end, undefined behaviour or a
read of an uninitialised value stops the run — with a byte-store writer; the output is compared with oracle/json_display.py.
usage: python tools/json_sanitizers.py [documents] [seed]      (round 5: 20 000 documents, seed 7: all three clean, outputs equal)"""
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import json_display as J                     # noqa: E402
from tests import test_gpu_json_display as T             # noqa: E402
from tests.golden import json_display_kats as K          # noqa: E402

MAIN = r'''
struct StrWrite { u8* p; void put(u8 b) { *p++ = b; } };
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "r");
  static char line[1 << 20];
  size_t n = 0, host = 0;
  while (fgets(line, sizeof line, f)) {
    size_t L = strlen(line); while (L && line[L - 1] == '\n') L--;
    const size_t tn = L / 2;
    u8* t = (u8*)malloc(tn ? tn : 1);
    auto hv = [](char c) { return c <= '9' ? c - '0' : c - 'a' + 10; };
    for (size_t i = 0; i < tn; i++) t[i] = (u8)(hv(line[2 * i]) * 16 + hv(line[2 * i + 1]));
    if (!json_valid(t, (uint32_t)tn)) { printf("INVALID\n"); free(t); continue; }
    JsCount c; const uint32_t e = json_display(c, t, (uint32_t)tn, false);
    if (e) { host++; printf("HOST\n"); free(t); continue; }
    u8* out = (u8*)malloc(c.n ? c.n : 1);
    StrWrite w{out};
    json_display(w, t, (uint32_t)tn, false);
    if ((size_t)(w.p - out) != c.n) { printf("LENGTH MISMATCH\n"); return 1; }
    for (uint32_t k = 0; k < c.n; k++) printf("%02x", out[k]);
    printf("\n");
    free(out); free(t);
    n++;
  }
  fprintf(stderr, "ok %zu host %zu\n", n, host);
  return 0;
}
'''


def main():
    ndocs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    src = open(os.path.join(ROOT, "etl_amd", "csrc", "columns.hip")).read()
    block = src[src.index("DEV int arr_hexv(uint32_t c)"):src.index("DEV uint32_t numeric_str_len(const u8* ent);")]
    d = tempfile.mkdtemp(prefix="jsan")
    with open(os.path.join(d, "j.cpp"), "w") as f:
        f.write("#include <cstdint>\n#include <cstdio>\n#include <cstring>\n#include <cstdlib>\n#define DEV inline\ntypedef uint8_t u8;\n" + block + MAIN)
    docs = [s for s, _ in K.PINNED + K.RESTATED]
    rng = random.Random(seed)
    while len(docs) < ndocs:
        docs.append(rng.choice(["", " ", "\n"]) + T._random_doc(rng) + rng.choice(["", " ", "\t\n"]))
    docs += ["[" * 16 + "]" * 16, "[" * 17 + "]" * 17, "{" + ",".join(f'"k{i}":{i}' for i in range(65)) + "}", '{"$serde_json::private::Number":"1"}']
    with open(os.path.join(d, "docs.hex"), "w") as f:
        f.write("\n".join(x.encode().hex() for x in docs) + "\n")
    want = "\n".join(J.display(x).hex() if J.device_limits_ok(x) else "HOST" for x in docs) + "\n"
    bad = 0
    for name, cmd in (("asan+ubsan", ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
                      ("msan", ["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O1", "-g", "-fsanitize=memory", "-fno-omit-frame-pointer"])):
        exe = os.path.join(d, name.replace("+", "_"))
        subprocess.check_call(cmd + [os.path.join(d, "j.cpp"), "-o", exe], stderr=subprocess.DEVNULL)
        r = subprocess.run([exe, os.path.join(d, "docs.hex")], capture_output=True, text=True)
        same = r.stdout == want
        print(name, "rc", r.returncode, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "", "outputs equal the oracle:", same)
        bad += r.returncode != 0 or not same
    return bad


if __name__ == "__main__":
    sys.exit(main())
