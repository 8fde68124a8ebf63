// comm_selftest.cpp -- CPU-only test of LocalComm (src/Comm.cpp): K rank threads exchange through the Comm interface with "device" memory that is
// plain host memory (the two C-ABI calls LocalComm makes, ltm_buffer_copy and ltm_last_error, are stubbed below; libltm_hip.so is not linked).
// Covers every exchange the sharded host uses -- label MAX all-reduce, all-gather of integers, all-gather-v, all-to-all-v -- the session groups of
// an even world (sub-communicators, pair swap of tables and bytes) and the release of every rank, whichever group's barrier it waits in, when one
// rank fails.  Built as lt-mapper_amd/host/comm_selftest, run by tests/test_host_cpp.py.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "removert/Comm.h"

extern "C" {
int ltm_buffer_copy(ltm_ctx*, void* dst, const void* src, size_t bytes, int) { if (bytes) std::memcpy(dst, src, bytes); return 0; }
const char* ltm_last_error(const ltm_ctx*) { return "stub"; }
}

using namespace ltremovert;

static std::atomic<int> fails{0};
#define CHECK(cond)                                                                              \
    do {                                                                                         \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++fails; }    \
    } while (0)

static void run_world(int world, bool expect_groups)
{
    std::vector<std::shared_ptr<Comm>> comms = makeLocalComms(world);
    CHECK((int)comms.size() == world);
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r)
        threads.emplace_back([&, r] {
            Comm& c = *comms[(size_t)r];
            CHECK(c.rank() == r && c.world() == world && std::string(c.backend()) == "local");
            // label union: byte i is set by rank i % world only
            std::vector<uint8_t> lab(1000, 0);
            for (size_t i = 0; i < lab.size(); ++i) lab[i] = (int)(i % (size_t)world) == r ? (uint8_t)(1 + i % 7) : 0;
            c.allReduceMaxU8(nullptr, lab.data(), lab.size());
            for (size_t i = 0; i < lab.size(); ++i) CHECK(lab[i] == (uint8_t)(1 + i % 7));
            std::vector<uint64_t> all;
            c.allGatherU64(nullptr, 100 + (uint64_t)r, all);
            for (int q = 0; q < world; ++q) CHECK(all[(size_t)q] == 100 + (uint64_t)q);
            // all-gather-v: rank r contributes r + 1 bytes of value r (rank 0's piece may be compared with an empty one: sizes differ per rank)
            std::vector<uint64_t> sizes;
            size_t total = 0;
            for (int q = 0; q < world; ++q) { sizes.push_back((uint64_t)(q == 1 ? 0 : q + 1)); total += sizes.back(); }
            std::vector<uint8_t> mine((size_t)sizes[(size_t)r], (uint8_t)r), got(total, 0xff);
            c.allGatherV(nullptr, mine.data(), mine.size(), got.data(), sizes);
            size_t at = 0;
            for (int q = 0; q < world; ++q) { for (uint64_t k = 0; k < sizes[(size_t)q]; ++k) CHECK(got[at + k] == (uint8_t)q); at += sizes[(size_t)q]; }
            // all-to-all-v: rank r sends (r + q) % 3 bytes of value 10 r + q to rank q
            std::vector<uint64_t> sb, rb;
            std::vector<uint8_t> sendbuf;
            for (int q = 0; q < world; ++q) { sb.push_back((uint64_t)((r + q) % 3)); sendbuf.insert(sendbuf.end(), (size_t)sb.back(), (uint8_t)(10 * r + q)); }
            size_t rtot = 0;
            for (int q = 0; q < world; ++q) { rb.push_back((uint64_t)((q + r) % 3)); rtot += rb.back(); }
            std::vector<uint8_t> recvbuf(rtot, 0xff);
            c.allToAllV(nullptr, sendbuf.data(), sb, recvbuf.data(), rb);
            at = 0;
            for (int q = 0; q < world; ++q) { for (uint64_t k = 0; k < rb[(size_t)q]; ++k) CHECK(recvbuf[at + k] == (uint8_t)(10 * q + r)); at += rb[(size_t)q]; }
            // session groups
            std::shared_ptr<Comm> g = c.sessionGroup();
            CHECK((g != nullptr) == expect_groups);
            if (g) {
                CHECK(g->world() == world / 2 && g->rank() == r / 2);
                std::vector<uint8_t> m(64, 0);
                m[(size_t)g->rank()] = (uint8_t)(1 + r % 2);                   // only my group's ranks write: the other group's value must not leak in
                g->allReduceMaxU8(nullptr, m.data(), m.size());
                for (int q = 0; q < world / 2; ++q) CHECK(m[(size_t)q] == (uint8_t)(1 + r % 2));
                CHECK(m[(size_t)(world / 2)] == 0);
                std::vector<uint64_t> ga;
                g->allGatherU64(nullptr, (uint64_t)r, ga);
                for (int q = 0; q < world / 2; ++q) CHECK(ga[(size_t)q] == (uint64_t)(2 * q + r % 2));
                // pair swap: tables, then bytes of different sizes in the two directions
                std::vector<uint64_t> tm = {(uint64_t)r, (uint64_t)(r + 5), 7}, tt;
                c.swapU64WithPeer(nullptr, tm, tt);
                CHECK(tt.size() == 3 && tt[0] == (uint64_t)(r ^ 1) && tt[1] == (uint64_t)((r ^ 1) + 5) && tt[2] == 7);
                std::vector<uint8_t> out((size_t)(3 + r), (uint8_t)(200 + r)), in((size_t)(3 + (r ^ 1)), 0);
                c.swapWithPeer(nullptr, out.data(), out.size(), in.data(), in.size());
                for (uint8_t v : in) CHECK(v == (uint8_t)(200 + (r ^ 1)));
                bool threw = false;                                            // a size the partner did not announce is an error on both sides
                try { c.swapWithPeer(nullptr, out.data(), out.size(), in.data(), in.size() + 1); } catch (const std::exception&) { threw = true; }
                CHECK(threw);
            } else {
                bool threw = false;
                try { std::vector<uint64_t> a = {1}, b; c.swapU64WithPeer(nullptr, a, b); } catch (const std::exception&) { threw = true; }
                CHECK(threw);
            }
        });
    for (auto& t : threads) t.join();
}

// one rank fails while the others wait in different barriers (world, central group, query group): everybody must come back with an exception
static void run_failure(int world)
{
    std::vector<std::shared_ptr<Comm>> comms = makeLocalComms(world);
    std::atomic<int> released{0};
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r)
        threads.emplace_back([&, r] {
            try {
                if (r == world - 1) { std::this_thread::sleep_for(std::chrono::milliseconds(50)); comms[(size_t)r]->abort(); return; }
                std::shared_ptr<Comm> g = comms[(size_t)r]->sessionGroup();
                if (g && r % 4 < 2) g->barrier();          // parks in its session group (the failing rank never arrives in the odd group)
                else comms[(size_t)r]->barrier();           // parks in the world
                if (g) { std::vector<uint64_t> a = {1}, b; comms[(size_t)r]->swapU64WithPeer(nullptr, a, b); }
            } catch (const std::exception&) { ++released; }
        });
    for (auto& t : threads) t.join();
    CHECK(released == world - 1);
}

int main()
{
    unsetenv("LTM_SESSION_GROUPS");
    for (int w : {1, 2, 3, 4, 8}) run_world(w, w >= 2 && w % 2 == 0);
    setenv("LTM_SESSION_GROUPS", "0", 1);
    run_world(4, false);
    unsetenv("LTM_SESSION_GROUPS");
    run_failure(4);
    run_failure(8);
    std::printf(fails ? "comm_selftest: %d FAILED\n" : "comm_selftest: all checks passed\n", fails.load());
    return fails ? 1 : 0;
}
