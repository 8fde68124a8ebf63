// stdsort_adversary.cpp -- generator of tests/golden/stdsort_adversary_keys.npz: keys on which THIS C++ library's std::sort runs into its depth
// limit and finishes with heap sort.  McIlroy's adversary ("A Killer Adversary for Quicksort", Software P&E 29(4), 1999) is played against std::sort
// itself: every element starts as "gas", the comparator freezes a key only when two gas elements meet, and what it has frozen by the end is an input
// that reproduces the quadratic run.  ltm_pclsort.h must follow std::sort into the fallback on it (tests/test_abi.py).
//   g++ -O2 -std=c++17 stdsort_adversary.cpp -o stdsort_adversary && ./stdsort_adversary /tmp/adv
//   python -c "import numpy as np; np.savez_compressed('tests/golden/stdsort_adversary_keys.npz', **{f'n{n}': np.fromfile(f'/tmp/adv_{n}.bin', np.uint32) for n in (300, 2000, 20000)})"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../lt-mapper_amd/csrc/ltm_pclsort.h"

int main(int argc, char** argv)
{
    const std::string prefix = argc > 1 ? argv[1] : "/tmp/adv";
    for (size_t n : {300u, 2000u, 20000u, 120000u}) {
        const uint32_t gas = (uint32_t)(n - 1);
        std::vector<uint32_t> val(n, gas), ptr(n);
        uint32_t nsolid = 0;
        size_t candidate = 0;
        for (size_t i = 0; i < n; ++i) ptr[i] = (uint32_t)i;
        std::sort(ptr.begin(), ptr.end(), [&](uint32_t x, uint32_t y) {
            if (val[x] == gas && val[y] == gas) { if (x == candidate) val[x] = nsolid++; else val[y] = nsolid++; }
            if (val[x] == gas) candidate = x; else if (val[y] == gas) candidate = y;
            return val[x] < val[y];
        });
        std::vector<ltm_pclsort::Entry> a(n), b;
        for (size_t i = 0; i < n; ++i) a[i] = ltm_pclsort::Entry{val[i], (uint32_t)i};
        b = a;
        std::sort(a.begin(), a.end(), ltm_pclsort::Less());
        const unsigned long before = ltm_pclsort::heap_sort_fallbacks();
        ltm_pclsort::sort(b.data(), b.data() + n);
        std::printf("n=%zu: heap-sort fallbacks %lu, %s\n", n, ltm_pclsort::heap_sort_fallbacks() - before,
                    std::memcmp(a.data(), b.data(), n * sizeof(ltm_pclsort::Entry)) ? "MISMATCH" : "identical to std::sort");
        const std::string name = prefix + "_" + std::to_string(n) + ".bin";
        if (FILE* f = std::fopen(name.c_str(), "wb")) { std::fwrite(val.data(), 4, n, f); std::fclose(f); }
    }
    return 0;
}
