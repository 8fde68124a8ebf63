// Comm.cpp -- LocalComm: K logical ranks inside one process (see Comm.h).  Exchange goes device -> host staging -> device
// through the C ABI only, so it runs on any number of devices including one (the test box).
#include "removert/Comm.h"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

namespace ltremovert
{

namespace
{

struct LocalGroup
{
    int world;
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    bool failed = false;
    std::vector<std::vector<uint8_t>> stage;      // one staging buffer per rank
    std::vector<uint64_t> scalars;
    std::vector<std::vector<uint64_t>> a2a_sizes;  // allToAllV: every rank's send table
    // a failure anywhere must release every rank, whichever group's barrier it is parked in: the world group knows its session groups and they
    // know it
    std::vector<std::shared_ptr<LocalGroup>> subs;
    std::weak_ptr<LocalGroup> parent;      // (weak: the world group owns its session groups; a session-group endpoint may outlive the world endpoints)
    explicit LocalGroup(int w) : world(w), stage((size_t)w), scalars((size_t)w, 0), a2a_sizes((size_t)w) {}

    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        if (failed) throw std::runtime_error("LocalComm: another rank failed");
        const uint64_t gen = generation;
        if (++waiting == world) { waiting = 0; ++generation; cv.notify_all(); return; }
        cv.wait(lk, [&] { return generation != gen || failed; });
        if (failed) throw std::runtime_error("LocalComm: another rank failed");
    }
    void failHere()
    {
        std::lock_guard<std::mutex> lk(m);
        failed = true;
        cv.notify_all();
    }
    void fail()
    {
        const std::shared_ptr<LocalGroup> up = parent.lock();
        LocalGroup* top = up ? up.get() : this;
        top->failHere();
        for (auto& s : top->subs) s->failHere();
        failHere();      // (a session group whose world group is gone still releases its own ranks)
    }
};

void check(ltm_ctx* ctx, int rc, const char* what)
{
    if (rc != LTM_OK) throw std::runtime_error(std::string(what) + ": " + ltm_last_error(ctx));
}

class LocalComm : public Comm
{
    std::shared_ptr<LocalGroup> g_;
    int rank_;
    std::shared_ptr<Comm> session_group_;      // this rank's endpoint in its session group (world endpoints of an even world only)

public:
    LocalComm(std::shared_ptr<LocalGroup> g, int r) : g_(std::move(g)), rank_(r) {}
    void setSessionGroup(std::shared_ptr<Comm> c) { session_group_ = std::move(c); }
    std::shared_ptr<Comm> sessionGroup() override { return session_group_; }

    void swapU64WithPeer(ltm_ctx*, const std::vector<uint64_t>& mine, std::vector<uint64_t>& theirs) override
    {
        if (!session_group_) throw std::runtime_error("LocalComm::swapU64WithPeer: the world does not split into session groups");
        try {
            g_->a2a_sizes[(size_t)rank_] = mine;
            g_->barrier();
            theirs = g_->a2a_sizes[(size_t)(rank_ ^ 1)];
            g_->barrier();
            if (theirs.size() != mine.size()) throw std::runtime_error("LocalComm::swapU64WithPeer: the two sides' tables differ in length");
        } catch (...) { g_->fail(); throw; }
    }

    // every rank of the world takes part at the same point of the pipeline (all pairs swap together), so the world barrier orders the staging buffers
    void swapWithPeer(ltm_ctx* ctx, const void* send_dev, size_t send_bytes, void* recv_dev, size_t recv_bytes) override
    {
        if (!session_group_) throw std::runtime_error("LocalComm::swapWithPeer: the world does not split into session groups");
        try {
            std::vector<uint8_t>& mine = g_->stage[(size_t)rank_];
            mine.resize(send_bytes);
            if (send_bytes) check(ctx, ltm_buffer_copy(ctx, mine.data(), send_dev, send_bytes, 1), "ltm_buffer_copy d2h");
            g_->barrier();
            const std::vector<uint8_t>& theirs = g_->stage[(size_t)(rank_ ^ 1)];
            if (theirs.size() != recv_bytes) throw std::runtime_error("LocalComm::swapWithPeer: the partner sends a different size than announced");
            if (recv_bytes) check(ctx, ltm_buffer_copy(ctx, recv_dev, theirs.data(), recv_bytes, 0), "ltm_buffer_copy h2d");
            g_->barrier();
        } catch (...) { g_->fail(); throw; }
    }
    int rank() const override { return rank_; }
    int world() const override { return g_->world; }
    const char* backend() const override { return "local"; }
    void barrier() override { g_->barrier(); }
    void abort() override { g_->fail(); }

    void allReduceMaxU8(ltm_ctx* ctx, void* dev, size_t n) override
    {
        if (g_->world == 1 || n == 0) return;
        try {
            std::vector<uint8_t>& mine = g_->stage[(size_t)rank_];
            mine.resize(n);
            check(ctx, ltm_buffer_copy(ctx, mine.data(), dev, n, 1), "ltm_buffer_copy d2h");
            g_->barrier();
            // every rank reduces its own slice of the bytes into rank 0's buffer ...
            const size_t a = n * (size_t)rank_ / (size_t)g_->world, b = n * (size_t)(rank_ + 1) / (size_t)g_->world;
            uint8_t* out = g_->stage[0].data();
            for (int r = 1; r < g_->world; ++r) {
                const uint8_t* in = g_->stage[(size_t)r].data();
                for (size_t i = a; i < b; ++i) out[i] = std::max(out[i], in[i]);
            }
            g_->barrier();
            // ... and everybody takes the result
            check(ctx, ltm_buffer_copy(ctx, dev, out, n, 0), "ltm_buffer_copy h2d");
            g_->barrier();
        } catch (...) { g_->fail(); throw; }
    }

    void allGatherU64(ltm_ctx*, uint64_t mine, std::vector<uint64_t>& all) override
    {
        try {
            g_->scalars[(size_t)rank_] = mine;
            g_->barrier();
            all = g_->scalars;
            g_->barrier();
        } catch (...) { g_->fail(); throw; }
    }

    void allGatherV(ltm_ctx* ctx, const void* send_dev, size_t send_bytes, void* recv_dev, const std::vector<uint64_t>& bytes) override
    {
        try {
            if ((int)bytes.size() != g_->world || bytes[(size_t)rank_] != send_bytes) throw std::runtime_error("LocalComm::allGatherV: size table does not match");
            std::vector<uint8_t>& mine = g_->stage[(size_t)rank_];
            mine.resize(send_bytes);
            if (send_bytes) check(ctx, ltm_buffer_copy(ctx, mine.data(), send_dev, send_bytes, 1), "ltm_buffer_copy d2h");
            g_->barrier();
            size_t at = 0;
            for (int r = 0; r < g_->world; ++r) {
                if (bytes[(size_t)r]) check(ctx, ltm_buffer_copy(ctx, static_cast<uint8_t*>(recv_dev) + at, g_->stage[(size_t)r].data(), bytes[(size_t)r], 0), "ltm_buffer_copy h2d");
                at += bytes[(size_t)r];
            }
            g_->barrier();
        } catch (...) { g_->fail(); throw; }
    }

    void allToAllV(ltm_ctx* ctx, const void* send_dev, const std::vector<uint64_t>& send_bytes, void* recv_dev, const std::vector<uint64_t>& recv_bytes) override
    {
        try {
            if ((int)send_bytes.size() != g_->world || (int)recv_bytes.size() != g_->world) throw std::runtime_error("LocalComm::allToAllV: size tables do not match the world");
            size_t total = 0;
            for (uint64_t b : send_bytes) total += b;
            std::vector<uint8_t>& mine = g_->stage[(size_t)rank_];
            mine.resize(total);
            if (total) check(ctx, ltm_buffer_copy(ctx, mine.data(), send_dev, total, 1), "ltm_buffer_copy d2h");
            g_->a2a_sizes[(size_t)rank_] = send_bytes;
            g_->barrier();
            size_t at = 0;
            for (int r = 0; r < g_->world; ++r) {
                const std::vector<uint64_t>& theirs = g_->a2a_sizes[(size_t)r];
                if (theirs[(size_t)rank_] != recv_bytes[(size_t)r]) throw std::runtime_error("LocalComm::allToAllV: a sender and its receiver disagree on a piece size");
                size_t off = 0;
                for (int q = 0; q < rank_; ++q) off += theirs[(size_t)q];
                if (recv_bytes[(size_t)r]) check(ctx, ltm_buffer_copy(ctx, static_cast<uint8_t*>(recv_dev) + at, g_->stage[(size_t)r].data() + off, recv_bytes[(size_t)r], 0), "ltm_buffer_copy h2d");
                at += recv_bytes[(size_t)r];
            }
            g_->barrier();
        } catch (...) { g_->fail(); throw; }
    }
};

} // namespace

bool sessionGroupsEnabled(int world)
{
    const char* e = std::getenv("LTM_SESSION_GROUPS");
    return world >= 2 && world % 2 == 0 && !(e && std::string(e) == "0");
}

std::vector<std::shared_ptr<Comm>> makeLocalComms(int world)
{
    if (world < 1) throw std::runtime_error("makeLocalComms: world must be >= 1");
    auto g = std::make_shared<LocalGroup>(world);
    std::vector<std::shared_ptr<LocalComm>> ends;
    for (int r = 0; r < world; ++r) ends.push_back(std::make_shared<LocalComm>(g, r));
    if (sessionGroupsEnabled(world)) {
        for (int color = 0; color < 2; ++color) {      // even ranks: central session's group, odd ranks: query session's
            auto sg = std::make_shared<LocalGroup>(world / 2);
            sg->parent = g;
            g->subs.push_back(sg);
            for (int r = color; r < world; r += 2) ends[(size_t)r]->setSessionGroup(std::make_shared<LocalComm>(sg, r / 2));
        }
    }
    return std::vector<std::shared_ptr<Comm>>(ends.begin(), ends.end());
}

} // namespace ltremovert
