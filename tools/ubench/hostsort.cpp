// hostsort.cpp -- what does the host side of ltm_voxel_grid_scanset's PCL-order path cost, and why is it bimodal (68 ... 198 ms for the
// same input on the 256-thread GPU box)?  No GPU involved: 500 keyframes x ~107 k (leaf index, point index) pairs, one std::sort per
// keyframe on a pool of threads, in the variants below.  Build: g++ -O3 -std=c++17 -pthread hostsort.cpp -o hostsort
//   A  per-thread std::vector scratch (what the library did), threads created per call
//   B  in place in the one shared buffer (no allocation inside the timed region)
//   C  B + every thread pinned to its own physical core (first hardware thread of each core, from sysfs)
//   P  B with ltm_pclsort::sort (the same permutation without std::sort's branch mispredictions, lt-mapper_amd/csrc/ltm_pclsort.h)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <set>
#include <string>
#include <thread>
#include <vector>
#include <pthread.h>
#include <sched.h>

#include "../../lt-mapper_amd/csrc/ltm_pclsort.h"

using Entry = ltm_pclsort::Entry;

static std::vector<int> physical_cores()
{
    std::vector<int> out;
    std::set<std::string> seen;
    const unsigned hw = std::thread::hardware_concurrency();
    for (unsigned c = 0; c < hw; ++c) {
        std::ifstream f("/sys/devices/system/cpu/cpu" + std::to_string(c) + "/topology/thread_siblings_list");
        std::string s;
        if (!f || !std::getline(f, s)) { out.push_back((int)c); continue; }
        if (seen.insert(s).second) out.push_back((int)c);
    }
    return out;
}

int main(int argc, char** argv)
{
    const size_t nk = argc > 1 ? (size_t)atoi(argv[1]) : 500, per = argc > 2 ? (size_t)atoi(argv[2]) : 107000;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    std::vector<size_t> off(nk + 1, 0);
    std::mt19937_64 rng(7);
    for (size_t k = 0; k < nk; ++k) off[k + 1] = off[k] + per + rng() % (per / 5);
    const size_t n = off[nk];
    std::vector<uint64_t> keys(n);
    for (size_t k = 0; k < nk; ++k) {
        const uint64_t range = (off[k + 1] - off[k]) * 2 / 3;             // ~1.5 points per leaf on average, like a scan under a 5 cm grid near the sensor
        for (size_t i = off[k]; i < off[k + 1]; ++i) keys[i] = rng() % range;
    }
    std::vector<uint64_t> hk(n);
    std::vector<uint32_t> hi(n), ref(n);
    const std::vector<int> cores = physical_cores();
    printf("keyframes %zu points %zu hardware threads %u physical cores %zu\n", nk, n, std::thread::hardware_concurrency(), cores.size());

    auto run = [&](char variant, size_t nt) {
        std::memcpy(hk.data(), keys.data(), n * 8);                        // "keys down": the main thread has touched the buffer last
        const auto t0 = std::chrono::steady_clock::now();
        std::atomic<size_t> next{0};
        auto work = [&](int slot) {
            if (variant == 'C' && slot > 0) {                                   // the calling thread stays unpinned (its children would inherit the mask)
                cpu_set_t set; CPU_ZERO(&set); CPU_SET(cores[(size_t)slot % cores.size()], &set);
                pthread_setaffinity_np(pthread_self(), sizeof set, &set);
            }
            std::vector<Entry> e;
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= nk) return;
                const size_t a = off[k], b = off[k + 1];
                if (variant == 'A') {
                    e.resize(b - a);
                    for (size_t i = a; i < b; ++i) e[i - a] = Entry{(uint32_t)hk[i], (uint32_t)i};
                    std::sort(e.begin(), e.end(), [](const Entry& x, const Entry& y) { return x.idx < y.idx; });
                    for (size_t i = a; i < b; ++i) hi[i] = e[i - a].cloud_point_index;
                } else {
                    Entry* s = reinterpret_cast<Entry*>(hk.data());
                    for (size_t i = a; i < b; ++i) s[i] = Entry{(uint32_t)hk[i], (uint32_t)i};
                    if (variant == 'P') ltm_pclsort::sort(s + a, s + b);
                    else std::sort(s + a, s + b, [](const Entry& x, const Entry& y) { return x.idx < y.idx; });
                    for (size_t i = a; i < b; ++i) hi[i] = s[i].cloud_point_index;
                }
            }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; ++t) pool.emplace_back(work, (int)t);
        work(0);
        for (std::thread& t : pool) t.join();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };

    run('A', 1 < cores.size() ? std::min<size_t>(cores.size(), 64) : 1);
    ref = hi;
    for (size_t nt : {8, 16, 24, 32, 48, 64, 128, 256}) {
        if (nt > 2 * std::thread::hardware_concurrency()) continue;
        for (char v : {'A', 'B', 'C', 'P'}) {
            printf("%c threads %3zu:", v, nt);
            for (int r = 0; r < reps; ++r) {
                const double ms = run(v, nt);
                printf(" %7.1f", ms);
                if (hi != ref) { printf(" ORDER DIFFERS\n"); return 1; }
            }
            printf(" ms\n");
            fflush(stdout);
        }
    }
    return 0;
}
