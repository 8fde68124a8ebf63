// Comm_rccl.cpp -- RcclComm: one rank per GPU of the node, collectives over xGMI (see Comm.h).  The only host file that talks to
// HIP / RCCL directly; everything else goes through the C ABI.
#include <cstdlib>
#include "removert/Comm.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <mutex>
#include <stdexcept>
#include <string>

namespace ltremovert
{

namespace
{

#define LTM_NCCL(expr)                                                                                           \
    do {                                                                                                         \
        ncclResult_t r__ = (expr);                                                                               \
        if (r__ != ncclSuccess) throw std::runtime_error(std::string(#expr) + ": " + ncclGetErrorString(r__));   \
    } while (0)
#define LTM_HIPRT(expr)                                                                                          \
    do {                                                                                                         \
        hipError_t e__ = (expr);                                                                                 \
        if (e__ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)

struct RcclGroup
{
    std::vector<ncclComm_t> comms;
    std::vector<int> devs;
    std::mutex mx;                 // abort() may come from any rank's thread
    // the session groups of a world group (its even / odd ranks): a rank that fails may have peers parked in a collective of EITHER of its groups, so an
    // abort of the world releases them too (only the world's communicators used to be aborted; LocalGroup::fail already did this)
    std::vector<std::shared_ptr<RcclGroup>> subs;
    // an aborted communicator is already released by ncclCommAbort: its slot is nulled there so that it is not destroyed twice
    ~RcclGroup() { for (ncclComm_t c : comms) if (c) ncclCommDestroy(c); }
    void abortAll()
    {
        {
            std::lock_guard<std::mutex> lk(mx);
            for (ncclComm_t& c : comms) if (c) { (void)ncclCommAbort(c); c = nullptr; }
        }
        for (const std::shared_ptr<RcclGroup>& s : subs) s->abortAll();
    }
};

class RcclComm : public Comm
{
    std::shared_ptr<RcclGroup> g_;
    int rank_;
    uint64_t* scratch_ = nullptr;      // world uint64 on the device for the size exchange
    uint64_t* swap_scratch_ = nullptr; // 2 x swap_cap_ uint64 for the table exchange with the partner rank
    size_t swap_cap_ = 0;
    std::shared_ptr<Comm> session_group_;

    ncclComm_t comm() const { return g_->comms[(size_t)rank_]; }
    static hipStream_t stream(ltm_ctx* ctx) { return static_cast<hipStream_t>(ltm_stream(ctx)); }

public:
    RcclComm(std::shared_ptr<RcclGroup> g, int r) : g_(std::move(g)), rank_(r) {}
    ~RcclComm() override
    {
        if (scratch_ || swap_scratch_) (void)hipSetDevice(g_->devs[(size_t)rank_]);
        if (scratch_) (void)hipFree(scratch_);
        if (swap_scratch_) (void)hipFree(swap_scratch_);
    }
    void setSessionGroup(std::shared_ptr<Comm> c) { session_group_ = std::move(c); }
    std::shared_ptr<Comm> sessionGroup() override { return session_group_; }

    // rank pairs (2i, 2i+1): one grouped send + receive per side; the pairs use different xGMI links and swap at the same time
    void swapU64WithPeer(ltm_ctx* ctx, const std::vector<uint64_t>& mine, std::vector<uint64_t>& theirs) override
    {
        if (!session_group_) throw std::runtime_error("RcclComm::swapU64WithPeer: the world does not split into session groups");
        const size_t n = mine.size();
        theirs.assign(n, 0);
        if (n == 0) return;
        LTM_HIPRT(hipSetDevice(g_->devs[(size_t)rank_]));
        if (swap_cap_ < n) {
            if (swap_scratch_) LTM_HIPRT(hipFree(swap_scratch_));
            swap_scratch_ = nullptr;
            LTM_HIPRT(hipMalloc(reinterpret_cast<void**>(&swap_scratch_), 2 * n * sizeof(uint64_t)));
            swap_cap_ = n;
        }
        const int peer = rank_ ^ 1;
        LTM_HIPRT(hipMemcpyAsync(swap_scratch_, mine.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, stream(ctx)));
        LTM_NCCL(ncclGroupStart());
        LTM_NCCL(ncclSend(swap_scratch_, n, ncclUint64, peer, comm(), stream(ctx)));
        LTM_NCCL(ncclRecv(swap_scratch_ + n, n, ncclUint64, peer, comm(), stream(ctx)));
        LTM_NCCL(ncclGroupEnd());
        LTM_HIPRT(hipMemcpyAsync(theirs.data(), swap_scratch_ + n, n * sizeof(uint64_t), hipMemcpyDeviceToHost, stream(ctx)));
        LTM_HIPRT(hipStreamSynchronize(stream(ctx)));
    }

    void swapWithPeer(ltm_ctx* ctx, const void* send_dev, size_t send_bytes, void* recv_dev, size_t recv_bytes) override
    {
        if (!session_group_) throw std::runtime_error("RcclComm::swapWithPeer: the world does not split into session groups");
        if (send_bytes == 0 && recv_bytes == 0) return;
        LTM_HIPRT(hipSetDevice(g_->devs[(size_t)rank_]));
        const int peer = rank_ ^ 1;
        LTM_NCCL(ncclGroupStart());
        if (send_bytes) LTM_NCCL(ncclSend(send_dev, send_bytes, ncclChar, peer, comm(), stream(ctx)));
        if (recv_bytes) LTM_NCCL(ncclRecv(recv_dev, recv_bytes, ncclChar, peer, comm(), stream(ctx)));
        LTM_NCCL(ncclGroupEnd());
    }
    int rank() const override { return rank_; }
    int world() const override { return (int)g_->comms.size(); }
    const char* backend() const override { return "rccl"; }

    // label union of a vote pass: one all-reduce of M bytes (6.8 MB at M = 6.8 M), on the stream that produced the labels
    void allReduceMaxU8(ltm_ctx* ctx, void* dev, size_t n) override
    {
        if (world() == 1 || n == 0) return;
        LTM_HIPRT(hipSetDevice(g_->devs[(size_t)rank_]));
        LTM_NCCL(ncclAllReduce(dev, dev, n, ncclUint8, ncclMax, comm(), stream(ctx)));
    }

    void allGatherU64(ltm_ctx* ctx, uint64_t mine, std::vector<uint64_t>& all) override
    {
        const int w = world();
        all.assign((size_t)w, mine);
        if (w == 1) return;
        LTM_HIPRT(hipSetDevice(g_->devs[(size_t)rank_]));
        if (!scratch_) LTM_HIPRT(hipMalloc(reinterpret_cast<void**>(&scratch_), sizeof(uint64_t) * (size_t)w));
        LTM_HIPRT(hipMemcpyAsync(scratch_ + rank_, &mine, sizeof mine, hipMemcpyHostToDevice, stream(ctx)));
        LTM_NCCL(ncclAllGather(scratch_ + rank_, scratch_, 1, ncclUint64, comm(), stream(ctx)));
        LTM_HIPRT(hipMemcpyAsync(all.data(), scratch_, sizeof(uint64_t) * (size_t)w, hipMemcpyDeviceToHost, stream(ctx)));
        LTM_HIPRT(hipStreamSynchronize(stream(ctx)));
    }

    // all-gather-v as one group of broadcasts (exact sizes, no padding): rank r's piece lands at its offset on every rank.
    // xGMI is point-to-point; the pieces are a few MB to ~100 MB, so this is latency / link bound and one group per stage.
    void allGatherV(ltm_ctx* ctx, const void* send_dev, size_t send_bytes, void* recv_dev, const std::vector<uint64_t>& bytes) override
    {
        const int w = world();
        if ((int)bytes.size() != w || bytes[(size_t)rank_] != send_bytes) throw std::runtime_error("RcclComm::allGatherV: size table does not match");
        LTM_HIPRT(hipSetDevice(g_->devs[(size_t)rank_]));
        size_t at = 0;
        LTM_NCCL(ncclGroupStart());
        for (int r = 0; r < w; ++r) {
            char* dst = static_cast<char*>(recv_dev) + at;
            if (bytes[(size_t)r]) LTM_NCCL(ncclBroadcast(r == rank_ ? send_dev : dst, dst, bytes[(size_t)r], ncclChar, r, comm(), stream(ctx)));
            at += bytes[(size_t)r];
        }
        LTM_NCCL(ncclGroupEnd());
    }

    // every point of a merge moves once, to the rank that owns its key range: grouped point-to-point sends and receives -- on xGMI's full mesh all
    // seven links of a GPU carry a piece at the same time, unlike the ring of an all-gather
    void allToAllV(ltm_ctx* ctx, const void* send_dev, const std::vector<uint64_t>& send_bytes, void* recv_dev, const std::vector<uint64_t>& recv_bytes) override
    {
        const int w = world();
        if ((int)send_bytes.size() != w || (int)recv_bytes.size() != w) throw std::runtime_error("RcclComm::allToAllV: size tables do not match the world");
        LTM_HIPRT(hipSetDevice(g_->devs[(size_t)rank_]));
        size_t so = 0, ro = 0, self_so = 0, self_ro = 0;
        LTM_NCCL(ncclGroupStart());
        for (int r = 0; r < w; ++r) {
            const char* src = static_cast<const char*>(send_dev) + so;
            char* dst = static_cast<char*>(recv_dev) + ro;
            if (r == rank_) { self_so = so; self_ro = ro; }
            else {
                if (send_bytes[(size_t)r]) LTM_NCCL(ncclSend(src, send_bytes[(size_t)r], ncclChar, r, comm(), stream(ctx)));
                if (recv_bytes[(size_t)r]) LTM_NCCL(ncclRecv(dst, recv_bytes[(size_t)r], ncclChar, r, comm(), stream(ctx)));
            }
            so += send_bytes[(size_t)r]; ro += recv_bytes[(size_t)r];
        }
        LTM_NCCL(ncclGroupEnd());
        if (send_bytes[(size_t)rank_] != recv_bytes[(size_t)rank_]) throw std::runtime_error("RcclComm::allToAllV: own piece sizes disagree");
        if (send_bytes[(size_t)rank_])
            LTM_HIPRT(hipMemcpyAsync(static_cast<char*>(recv_dev) + self_ro, static_cast<const char*>(send_dev) + self_so, send_bytes[(size_t)rank_], hipMemcpyDeviceToDevice, stream(ctx)));
    }

    void barrier() override {}      // every exchange is stream-ordered; the host threads never need to meet
    void abort() override { g_->abortAll(); }
};

} // namespace

std::vector<std::shared_ptr<Comm>> makeRcclComms(const std::vector<int>& devs)
{
    if (devs.empty()) throw std::runtime_error("makeRcclComms: no devices");
    auto g = std::make_shared<RcclGroup>();
    g->devs = devs;
    g->comms.assign(devs.size(), nullptr);
    LTM_NCCL(ncclCommInitAll(g->comms.data(), (int)devs.size(), devs.data()));
    std::vector<std::shared_ptr<RcclComm>> ends;
    for (size_t r = 0; r < devs.size(); ++r) ends.push_back(std::make_shared<RcclComm>(g, (int)r));
    // over RCCL the split is opt-in (LTM_SESSION_GROUPS=1): the two extra communicators and the pair swap have run under LocalComm and gloo only --
    // no multi-GPU hardware has executed them yet; tests/test_gpu_multi.py runs both settings where two devices exist
    const char* sg_env = std::getenv("LTM_SESSION_GROUPS");
    if (sg_env && std::string(sg_env) == "1" && sessionGroupsEnabled((int)devs.size())) {
        for (size_t color = 0; color < 2; ++color) {      // a communicator of their own for the even and for the odd ranks
            auto sg = std::make_shared<RcclGroup>();
            for (size_t r = color; r < devs.size(); r += 2) sg->devs.push_back(devs[r]);
            sg->comms.assign(sg->devs.size(), nullptr);
            LTM_NCCL(ncclCommInitAll(sg->comms.data(), (int)sg->devs.size(), sg->devs.data()));
            g->subs.push_back(sg);
            for (size_t r = color; r < devs.size(); r += 2) ends[r]->setSessionGroup(std::make_shared<RcclComm>(sg, (int)(r / 2)));
        }
    }
    return std::vector<std::shared_ptr<Comm>>(ends.begin(), ends.end());
}

} // namespace ltremovert
