// ASAN/UBSAN fuzz driver for the two pack producers (host code only).
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Iclairs_to_amd/csrc \
//       clairs_to_amd/csrc/bam.cpp clairs_to_amd/csrc/pack.cpp tools/fuzz_producers.cpp -o /tmp/fuzz -lz -ldl -lpthread
//   /tmp/fuzz <dir with ok.bam, ok.bam.bai, ok.txt (tests/bamutil.py writes them)> <iterations>
// Last runs (round 2, after the tokeniser's single-pass row parser and the hashed indel-key scan): 3000 iterations + 600 with
// CTO_PACK_THREADS=4 under -fsanitize=address,undefined, no report.  Before that (mapped-file / CRC / run-wise column loop rework): 2500 iterations + 600 with CTO_PACK_THREADS=4 under
// -fsanitize=address,undefined; earlier: 3000 + 1500 iterations and 150 under -fsanitize=thread with CTO_PACK_THREADS=4
// (truncations, byte flips, insertions of BAM / BAI / mpileup text): no sanitizer report.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "clairsto_amd.h"
static std::vector<char> slurp(const char* p) { FILE* f = fopen(p, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<char> b(n); if (fread(b.data(), 1, n, f) != size_t(n)) abort(); fclose(f); return b; }
static void spit(const char* p, const std::vector<char>& b) { FILE* f = fopen(p, "wb"); fwrite(b.data(), 1, b.size(), f); fclose(f); }
int main(int argc, char** argv) {
    const std::string dir = argv[1];
    auto bam = slurp((dir + "/ok.bam").c_str()), bai = slurp((dir + "/ok.bam.bai").c_str()), txt = slurp((dir + "/ok.txt").c_str());
    std::string ref(5000, 'A');
    std::mt19937 rng(7);
    int ok = 0, err = 0;
    auto damage = [&](std::vector<char> b) {
        int mode = rng() % 3;
        if (mode == 0 && b.size() > 10) b.resize(1 + rng() % (b.size() - 1));
        else if (mode == 1) { int k = 1 + rng() % 8; for (int i = 0; i < k; ++i) b[rng() % b.size()] = char(rng()); }
        else { size_t i = rng() % b.size(); int k = 1 + rng() % 40; std::vector<char> ins(k); for (auto& c : ins) c = char(rng()); b.insert(b.begin() + i, ins.begin(), ins.end()); }
        return b;
    };
    const int N = atoi(argv[2]);
    for (int it = 0; it < N; ++it) {
        cto_pack* p = nullptr;
        int rc;
        if (it % 3 == 0) { spit((dir + "/f.bam").c_str(), damage(bam)); spit((dir + "/f.bam.bai").c_str(), bai);
            rc = cto_pack_from_bam((dir + "/f.bam").c_str(), nullptr, "chrA", 1, 5000, nullptr, 0, ref.data(), 1, ref.size(), 2316, 0, 8000, 60, &p); }
        else if (it % 3 == 1) { spit((dir + "/g.bai").c_str(), damage(bai));
            rc = cto_pack_from_bam((dir + "/ok.bam").c_str(), (dir + "/g.bai").c_str(), "chrA", 1, 5000, nullptr, 0, ref.data(), 1, ref.size(), 2316, 0, 8000, 60, &p); }
        else { auto t = damage(txt); rc = cto_pack_from_mpileup(t.data(), t.size(), ref.data(), 1, ref.size(), 60, &p); }
        if (rc == 0) { ++ok; cto_pack_view v; cto_pack_view_of(p, &v); cto_pack_free(p); } else ++err;
    }
    printf("fuzz: %d ok, %d errors, no sanitizer report\n", ok, err);
    return 0;
}
