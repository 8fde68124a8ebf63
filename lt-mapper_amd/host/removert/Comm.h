// Comm.h -- the exchange layer of the keyframe-sharded multi-GPU host (BASELINE.json north_star: "keyframes shard
// naturally across the 8 GPUs of one node with an RCCL all-gather over xGMI to assemble the final live/meta/delta maps").
//
// One host thread + one ltm_ctx per GPU (rank).  Session maps are replicated; every per-keyframe loop of the reference
// (Removerter.cpp:555 votes, Session.cpp:354 reprojection, Session.cpp:408,491 kNN) runs on this rank's contiguous block of
// keyframes, and the ranks exchange
//   * the M-byte label mask of a vote pass: element-wise MAX (RCCL has no bit-OR) = the std::set union of Removerter.cpp:589-590;
//   * per-rank pieces of clouds / scan sets of different sizes: an all-gather-v that concatenates them in rank order.
// Two back ends behind one interface:
//   LocalComm -- K logical ranks inside one process exchanging through host staging buffers and the C ABI's buffer copies; any
//                number of ranks may share one device (the 1-GPU test box: results must not depend on K);
//   RcclComm  -- ncclCommInitAll over the node's GPUs; every collective is enqueued on the rank's ltm_stream(), so it is ordered
//                with the kernels that produce and consume its buffers without host synchronisation.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

#include "ltm.h"

namespace ltremovert
{

class Comm
{
public:
    virtual ~Comm() {}
    virtual int rank() const = 0;
    virtual int world() const = 0;
    virtual const char* backend() const = 0;
    // in-place element-wise MAX of n bytes of device memory over all ranks (label union)
    virtual void allReduceMaxU8(ltm_ctx* ctx, void* dev, size_t n) = 0;
    // every rank contributes one host integer; all[r] = rank r's value
    virtual void allGatherU64(ltm_ctx* ctx, uint64_t mine, std::vector<uint64_t>& all) = 0;
    // concatenation in rank order of every rank's device array: recv_dev must hold sum(bytes[r]); bytes[rank()] == send_bytes
    virtual void allGatherV(ltm_ctx* ctx, const void* send_dev, size_t send_bytes, void* recv_dev, const std::vector<uint64_t>& bytes) = 0;
    // all-to-all-v of device buffers: send_bytes[r] consecutive bytes of send_dev (in rank order) go to rank r; what arrives from rank r lands
    // at recv_dev after the pieces of ranks 0..r-1 -- recv_bytes[r] bytes each (the caller has exchanged the sizes, e.g. with allGatherV)
    virtual void allToAllV(ltm_ctx* ctx, const void* send_dev, const std::vector<uint64_t>& send_bytes, void* recv_dev, const std::vector<uint64_t>& recv_bytes) = 0;
    virtual void barrier() = 0;
    // a rank died: release the others instead of letting them wait for an exchange that will never complete
    virtual void abort() {}

    // ---- session groups.  makeGlobalMap + Step 1 (Removerter.cpp:1584-1604) of the two sessions are independent: with an even world the even
    // ranks form the central session's group and the odd ranks the query session's, each sharding ITS session's keyframes world/2 ways, and rank
    // pairs (2i, 2i+1) swap the finished maps afterwards (same split as lt-mapper_amd/dist.py ShardedOps.session_groups).
    // This rank's endpoint in its group; null when the world does not split (odd or 1, or LTM_SESSION_GROUPS=0)
    virtual std::shared_ptr<Comm> sessionGroup() { return nullptr; }
    // exchange with rank() ^ 1: a table of host integers (same length on both sides), then device bytes (sizes agreed through the table)
    virtual void swapU64WithPeer(ltm_ctx*, const std::vector<uint64_t>&, std::vector<uint64_t>&) { throw std::runtime_error("Comm::swapU64WithPeer: this endpoint has no partner"); }
    virtual void swapWithPeer(ltm_ctx*, const void*, size_t, void*, size_t) { throw std::runtime_error("Comm::swapWithPeer: this endpoint has no partner"); }
};

// contiguous block of keyframes [kb, ke) of rank `rank` (same rule as lt-mapper_amd/dist.py shard_range)
inline void shardRange(size_t n, int rank, int world, size_t* kb, size_t* ke)
{
    *kb = n * (size_t)rank / (size_t)world;
    *ke = n * (size_t)(rank + 1) / (size_t)world;
}

// does a world of this size split into two session groups (even, >= 2, not switched off with LTM_SESSION_GROUPS=0)
bool sessionGroupsEnabled(int world);
// K endpoints of one in-process group; endpoint r is used by rank thread r only
std::vector<std::shared_ptr<Comm>> makeLocalComms(int world);
// K endpoints over devices devs[0..K) (ncclCommInitAll); throws if RCCL cannot initialise
std::vector<std::shared_ptr<Comm>> makeRcclComms(const std::vector<int>& devs);

} // namespace ltremovert
