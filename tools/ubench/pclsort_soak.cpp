// pclsort_soak.cpp -- differential soak of lt-mapper_amd/csrc/ltm_pclsort.h against std::sort: random sizes (0 .. 300 k), key ranges from 2 values to 2^32,
// presorted / reversed / nearly sorted / interleaved-run inputs.  g++ -O2 -std=c++17 pclsort_soak.cpp -o pclsort_soak && ./pclsort_soak <seed> <cases>
#include "../../lt-mapper_amd/csrc/ltm_pclsort.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace ltm_pclsort;
int main(int argc, char** argv)
{
    std::mt19937_64 rng(argc > 1 ? atoll(argv[1]) : 1);
    const int reps = argc > 2 ? atoi(argv[2]) : 2000;
    long cases = 0; unsigned long long elems = 0;
    for (int rep = 0; rep < reps; ++rep) {
        const int kind = (int)(rng() % 8);
        size_t n = kind == 0 ? rng() % 600 : kind == 1 ? 200 + rng() % 1000 : kind == 2 ? 100000 + rng() % 200000 : rng() % 60000;
        const uint64_t ranges[8] = {2, 3, 17, n / 16 + 1, n / 2 + 1, n * 2 / 3 + 1, n * 8 + 1, 1ull << 32};
        const uint64_t range = ranges[rng() % 8];
        std::vector<Entry> v(n);
        for (size_t i = 0; i < n; ++i) v[i] = Entry{(uint32_t)(rng() % range), (uint32_t)i};
        const int shape = (int)(rng() % 6);
        if (shape == 1) std::sort(v.begin(), v.end(), Less());
        if (shape == 2) { std::sort(v.begin(), v.end(), Less()); std::reverse(v.begin(), v.end()); }
        if (shape == 3 && n > 8) { std::sort(v.begin(), v.end(), Less()); for (size_t i = 0; i < n / 20 + 1; ++i) std::swap(v[rng() % n].idx, v[rng() % n].idx); }
        if (shape == 4) for (size_t i = 0; i < n; ++i) v[i].idx = (uint32_t)((i % 64) * 1000 + i / 64);     // interleaved runs (a ring-ordered scan)
        for (size_t i = 0; i < n; ++i) v[i].cloud_point_index = (uint32_t)i;
        std::vector<Entry> a = v, b = v;
        std::sort(a.begin(), a.end(), Less());
        sort(b.data(), b.data() + n);
        ++cases; elems += n;
        if (n && std::memcmp(a.data(), b.data(), n * sizeof(Entry)) != 0) { std::printf("MISMATCH seed %s rep %d n=%zu range=%llu shape=%d\n", argc > 1 ? argv[1] : "1", rep, n, (unsigned long long)range, shape); return 1; }
    }
    std::printf("seed %s: %ld cases, %llu elements identical to std::sort (heap-sort fallbacks on this thread: %lu)\n", argc > 1 ? argv[1] : "1", cases, elems, heap_sort_fallbacks());
    return 0;
}
