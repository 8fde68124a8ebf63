// ltm_api_voxel.cpp -- C ABI: octreeDownsampling (utility.cpp:204-219), its batch / shard / key-range forms, and the loader's pcl::VoxelGrid (Session.cpp:284-289)
#include "ltm_internal.h"

namespace ltm_detail {

// --------------------------------------------------------------------------- voxel centroid
// PCL OctreePointCloud::defineBoundingBox() + getKeyBitSize() on an empty tree (octree_pointcloud.hpp);
// see DESIGN.md "voxel lattice".  Returns false if the depth does not fit 21 bits per axis.
bool octree_frame_from_bbox(const float mn[3], const float mx[3], float leaf, OctreeFrame* f)
{
    const float eps512 = FLT_EPSILON * 512.0f;
    const float minValue = FLT_EPSILON;
    double lo[3], hi[3];
    for (int d = 0; d < 3; ++d) { lo[d] = (double)mn[d]; hi[d] = (double)(float)(mx[d] + eps512); }
    const double res = (double)leaf;
    unsigned mk = 2;
    for (int d = 0; d < 3; ++d) mk = std::max(mk, (unsigned)std::ceil((hi[d] - lo[d] - minValue) / res));
    const unsigned depth = std::min(32u, (unsigned)std::ceil(std::log2((double)mk) - minValue));
    if (depth > 21) return false;
    const double side = (double)(1u << depth) * res;
    for (int d = 0; d < 3; ++d) {
        const double over = (side - (hi[d] - lo[d])) / 2.0;
        if (over > minValue) lo[d] -= over;
    }
    f->minx = lo[0]; f->miny = lo[1]; f->minz = lo[2]; f->res = res; f->depth = depth;
    return true;
}

// Which bits of the Morton code can the cloud's points tell apart?  (KeyCompress, ltm_kernels.h.)  Per axis the keys lie in
// [klo, khi] = the keys of the bounding box's corners (the key is monotone in the coordinate).  Bit L of an axis is a function of that
// axis' higher bits -- and therefore of more significant bits of the interleaved code -- iff the keys' prefixes k >> (L + 1) take at most
// two (consecutive) values and, within each, bit L is constant: one prefix: klo >> L == khi >> L; two: bit L of klo is 1 (the low
// prefix's keys run from klo to the end of its block: all in the upper half) and bit L of khi is 0.  Dropping such bits keeps both the
// order and the equality of codes, so the sorted sequence and the voxel boundaries are those of the full code.
KeyCompress key_compress_for(const float mn[3], const float mx[3], const OctreeFrame& f, bool enable)
{
    const unsigned depth = f.depth;
    uint64_t kept = 0;
    const double lo3[3] = {f.minx, f.miny, f.minz};
    for (int a = 0; a < 3; ++a) {
        const uint32_t klo = (uint32_t)(((double)mn[a] - lo3[a]) / f.res), khi = (uint32_t)(((double)mx[a] - lo3[a]) / f.res);
        for (unsigned L = 0; L < depth; ++L) {
            bool drop = false;
            if (enable && klo <= khi) {
                const uint64_t pl = (uint64_t)klo >> (L + 1), ph = (uint64_t)khi >> (L + 1);
                if (pl == ph) drop = (klo >> L) == (khi >> L);
                else if (ph == pl + 1) drop = ((klo >> L) & 1u) == 1u && ((khi >> L) & 1u) == 0u;
            }
            if (!drop) kept |= 1ull << (3 * L + (2 - a));        // x is the most significant bit of a level triple
        }
    }
    KeyCompress kc{};
    unsigned out = 0;
    for (unsigned b = 0; b < 3 * depth;) {
        if (!((kept >> b) & 1ull)) { ++b; continue; }
        unsigned e = b;
        while (e < 3 * depth && ((kept >> e) & 1ull)) ++e;
        if (kc.n_runs == kMaxKeyRuns) {          // cannot happen with <= 63 bits and runs separated by dropped bits of 3 axes, but stay safe: no compression
            KeyCompress id{};
            id.n_runs = 1; id.bits = 3 * depth; id.src[0] = 0; id.dst[0] = 0; id.mask[0] = (3 * depth >= 64) ? ~0ull : ((1ull << (3 * depth)) - 1);
            return id;
        }
        kc.src[kc.n_runs] = (unsigned char)b; kc.dst[kc.n_runs] = (unsigned char)out; kc.mask[kc.n_runs] = ((e - b) >= 64) ? ~0ull : ((1ull << (e - b)) - 1);
        ++kc.n_runs;
        out += e - b;
        b = e;
    }
    kc.bits = std::max(out, 1u);
    if (kc.n_runs == 0) { kc.n_runs = 1; kc.src[0] = 0; kc.dst[0] = 0; kc.mask[0] = 0; }      // a single voxel: every code equal
    return kc;
}

void bbox_of(ltm_ctx* c, const float4* pts, size_t n, float mn[3], float mx[3])
{
    DevBuf bb(c, 8 * sizeof(uint32_t));
    LTM_HIP(bbox_init(bb.as<uint32_t>(), c->stream));
    LTM_HIP(bbox_reduce(pts, n, bb.as<uint32_t>(), c->stream));
    uint32_t enc[8];
    d2h(c, enc, bb.p, sizeof enc);
    for (int d = 0; d < 3; ++d) { mn[d] = bbox_decode(enc[d]); mx[d] = bbox_decode(enc[3 + d]); }
}
bool same_frame(const OctreeFrame& a, const OctreeFrame& b)
{
    return a.minx == b.minx && a.miny == b.miny && a.minz == b.minz && a.res == b.res && a.depth == b.depth;
}
// head flags + scan + segment starts over sorted keys: the fused single-pass kernel (default) or the four-kernel form (LTM_VOXEL_FUSED_TAIL=0);
// `starts` gets one entry per segment (capacity n), the count goes to *count_dev
void voxel_segments(ltm_ctx* c, const uint64_t* keys2, size_t n, unsigned kshift, uint32_t* starts, uint32_t* count_dev)
{
    if (c->voxel_fused_tail) {
        const size_t tb = voxel_heads_starts_temp_bytes(n);
        DevBuf temp(c, tb);
        LTM_HIP(voxel_heads_starts(keys2, n, kshift, starts, temp.p, count_dev, c->stream));
        return;
    }
    DevBuf heads(c, n), pos(c, n * 4);
    LTM_HIP(head_flags(keys2, n, heads.as<uint8_t>(), c->stream, kshift));
    const size_t tb = scan_temp_bytes(n);
    DevBuf temp(c, tb);
    LTM_HIP(exclusive_scan_u8(heads.as<uint8_t>(), pos.as<uint32_t>(), n, temp.p, tb, c->stream));
    LTM_HIP(scan_total_to(heads.as<uint8_t>(), pos.as<uint32_t>(), n, count_dev, c->stream));
    LTM_HIP(segment_starts(heads.as<uint8_t>(), pos.as<uint32_t>(), n, starts, c->stream));
}

// voxel centroids of pts[0..n) into a freshly pooled array; returns count
// With n_shards > 1 only the voxels of shard `shard` are produced: the Morton key space is cut into n_shards contiguous
// ranges holding about n/n_shards points each (cut points on a 4096-bin histogram of the key prefix, so they are a pure
// function of the input), and the outputs of shards 0..n_shards-1 concatenated are exactly the unsharded output.
//
// Sort layout: when Morton bits + index bits fit one 64-bit word (always, for clouds the 32-bit index allows and octrees up
// to depth 10-13) the pair travels packed and the radix sort is keys-only over the Morton bits; otherwise key/index pairs.
size_t voxel_centroid_raw(ltm_ctx* c, const float4* pts, size_t n_in, float leaf, float4** out, uint32_t shard = 0, uint32_t n_shards = 1,
                          const OctreeFrame* cached = nullptr, OctreeFrame* frame_out = nullptr, const float* box_mn = nullptr, const float* box_mx = nullptr)
{
    *out = nullptr;
    if (n_in == 0) return 0;
    LTM_REQUIRE(leaf > 0.0f, "leaf size must be positive");
    LTM_REQUIRE(n_in < 0xffffffffull, "cloud too large for 32-bit point indices");
    ProfScope p(c, "voxel", (double)n_in, 64.0 * n_in);
    ++c->voxel_calls;
    float mn[3], mx[3];
    bool untouched = false;
    if (cached && c->voxel_identity && n_shards == 1) {
        DevBuf bb(c, 8 * sizeof(uint32_t));
        LTM_HIP(bbox_init(bb.as<uint32_t>(), c->stream));
        LTM_HIP(bbox_reduce_check(pts, n_in, *cached, bb.as<uint32_t>(), c->stream));
        uint32_t enc[8];
        d2h(c, enc, bb.p, sizeof enc);
        for (int d = 0; d < 3; ++d) { mn[d] = bbox_decode(enc[d]); mx[d] = bbox_decode(enc[3 + d]); }
        untouched = enc[6] == 0;
    } else if (box_mn && box_mx) {      // the bounding box of a LARGER cloud this one is a part of (key-range exchange, ltm_voxel_centroid_box)
        for (int d = 0; d < 3; ++d) { mn[d] = box_mn[d]; mx[d] = box_mx[d]; }
    } else bbox_of(c, pts, n_in, mn, mx);
    OctreeFrame f;
    if (!octree_frame_from_bbox(mn, mx, leaf, &f)) throw Err{LTM_E_UNSUPPORTED, "octree depth > 21 (extent / leaf too large)"};
    if (frame_out) *frame_out = f;
    if (untouched && same_frame(f, *cached)) {
        // every point alone in its voxel and already in octree order under the frame this very call would use: the grid is the identity
        float4* o = reinterpret_cast<float4*>(c->pool.alloc(n_in * sizeof(float4)));
        d2d(c, o, pts, n_in * sizeof(float4));
        ++c->voxel_identity_hits;
        *out = o;
        return n_in;
    }
    unsigned ib = 1;
    while (ib < 32 && ((size_t)1 << ib) < n_in) ++ib;
    // packed (code << index bits | index in one word, keys-only sort) whenever the COMPRESSED code fits beside the index: a 45 M-point
    // street map (depth 13, 26 index bits) misses 64 bits by one with the full code and fits easily without the undecidable bits
    const KeyCompress kc = key_compress_for(mn, mx, f, c->voxel_key_compress != 0);
    const bool packed = kc.bits + ib <= 64;
    const unsigned mbits = packed ? kc.bits : 3 * f.depth;        // code bits the sort has to look at
    const unsigned kshift = packed ? ib : 0;                       // Morton code = key >> kshift
    size_t n = n_in;
    DevBuf keys(c, n * 8), idx(c, packed ? 8 : n * 4);
    if (packed) LTM_HIP(morton_keys_packed(pts, n, f, kc, ib, keys.as<uint64_t>(), c->stream));
    else LTM_HIP(morton_keys(pts, n, f, keys.as<uint64_t>(), idx.as<uint32_t>(), c->stream));
    if (n_shards > 1) {
        const unsigned shift = mbits > 12 ? mbits - 12 : 0;
        std::vector<uint32_t> hist(kVoxelKeyBins);
        {
            DevBuf hd(c, kVoxelKeyBins * sizeof(uint32_t));
            LTM_HIP(key_histogram(keys.as<uint64_t>(), n, shift + kshift, hd.as<uint32_t>(), c->stream));
            d2h(c, hist.data(), hd.p, kVoxelKeyBins * sizeof(uint32_t));
        }
        // cut b (1..n_shards-1) = first bin whose preceding count reaches b*n/n_shards
        auto cut = [&](uint32_t b) -> uint64_t {
            if (b == 0) return 0;
            if (b >= n_shards) return kVoxelKeyBins;
            const uint64_t want = (uint64_t)n * b / n_shards;
            uint64_t cum = 0;
            for (uint64_t s = 0; s < (uint64_t)kVoxelKeyBins; ++s) {
                if (cum >= want) return s;
                cum += hist[s];
            }
            return kVoxelKeyBins;
        };
        auto bound = [&](uint64_t bin) { return bin >= (uint64_t)kVoxelKeyBins ? ~0ull : bin << (shift + kshift); };
        const uint64_t lo = bound(cut(shard));
        const uint64_t hi = (shard + 1 >= n_shards) ? ~0ull : bound(cut(shard + 1));
        DevBuf flags(c, n), pos(c, n * 4);
        LTM_HIP(key_range_flags(keys.as<uint64_t>(), n, lo, hi, flags.as<uint8_t>(), c->stream));
        const size_t tb = scan_temp_bytes(n);
        DevBuf temp(c, tb);
        LTM_HIP(exclusive_scan_u8(flags.as<uint8_t>(), pos.as<uint32_t>(), n, temp.p, tb, c->stream));
        const size_t nsel = scan_total_u8(c, flags.as<uint8_t>(), pos.as<uint32_t>(), n);
        if (nsel == 0) return 0;
        DevBuf ck(c, nsel * 8), ci(c, packed ? 8 : nsel * 4);
        if (packed) LTM_HIP(compact_keys(keys.as<uint64_t>(), flags.as<uint8_t>(), pos.as<uint32_t>(), n, ck.as<uint64_t>(), c->stream));
        else LTM_HIP(compact_pairs(keys.as<uint64_t>(), idx.as<uint32_t>(), flags.as<uint8_t>(), pos.as<uint32_t>(), n,
                                   ck.as<uint64_t>(), ci.as<uint32_t>(), c->stream));
        std::swap(keys.p, ck.p); std::swap(idx.p, ci.p);
        n = nsel;
    }
    DevBuf keys2(c, n * 8), idx2(c, packed ? 8 : n * 4);
    if (packed) {
        const size_t stb = sort_keys_temp_bytes(n);
        DevBuf stemp(c, stb);
        LTM_HIP(sort_keys_u64(keys.as<uint64_t>(), keys2.as<uint64_t>(), n, ib, ib + mbits, stemp.p, stb, c->stream));
    } else {
        const size_t stb = sort_temp_bytes(n);
        DevBuf stemp(c, stb);
        LTM_HIP(sort_pairs_u64(keys.as<uint64_t>(), keys2.as<uint64_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), n, mbits, stemp.p, stb, c->stream));
    }
    DevBuf starts(c, n * 4), cnt(c, 4);
    voxel_segments(c, keys2.as<uint64_t>(), n, kshift, starts.as<uint32_t>(), cnt.as<uint32_t>());
    uint32_t nvox32 = 0;
    d2h(c, &nvox32, cnt.p, 4);
    const size_t nvox = nvox32;
    float4* o = reinterpret_cast<float4*>(c->pool.alloc(nvox * sizeof(float4)));
    if (packed) LTM_HIP(voxel_centroids_packed(pts, keys2.as<uint64_t>(), ((uint64_t)1 << ib) - 1, starts.as<uint32_t>(), nvox, n, o, c->stream));
    else LTM_HIP(voxel_centroids(pts, idx2.as<uint32_t>(), starts.as<uint32_t>(), nvox, n, o, c->stream));
    *out = o;
    return nvox;
}

// Several independent voxel grids as one batch: the stages of every cloud are enqueued phase by phase and the host reads all bounding
// boxes in ONE copy and all voxel counts in ONE copy -- two host round trips (~70 us of idle GPU each) for the whole batch instead
// of two per cloud.  Results are exactly those of voxel_centroid_raw (same kernels, same order inside each cloud).
struct VoxelJob {
    const float4* pts; size_t n; float leaf;
    OctreeFrame f; unsigned ib = 1, kshift = 0, mbits = 0; bool packed = false;
    std::unique_ptr<DevBuf> keys, idx, keys2, idx2, starts;
    float4* out = nullptr; size_t nvox = 0;
    bool has_cached = false; OctreeFrame cached{};        // the frame the input was gridded under, if it still carries one
    bool identity = false;                                 // decided after the bounding-box round trip: output = copy of the input
};
void voxel_centroid_batch_impl(ltm_ctx* c, std::vector<VoxelJob>& jobs);
void voxel_centroid_batch(ltm_ctx* c, std::vector<VoxelJob>& jobs)
{
    try { voxel_centroid_batch_impl(c, jobs); }
    catch (...) {      // outputs already allocated for earlier jobs go back to the pool (the scratch buffers are RAII)
        for (VoxelJob& j : jobs) { if (j.out) c->pool.free(j.out); j.out = nullptr; j.nvox = 0; }
        throw;
    }
}
void voxel_centroid_batch_impl(ltm_ctx* c, std::vector<VoxelJob>& jobs)
{
    const size_t nj = jobs.size();
    if (nj == 0) return;
    double tot_pts = 0;
    for (VoxelJob& j : jobs) {
        LTM_REQUIRE(j.leaf > 0.0f || j.n == 0, "leaf size must be positive");
        LTM_REQUIRE(j.n < 0xffffffffull, "cloud too large for 32-bit point indices");
        tot_pts += (double)j.n;
    }
    ProfScope p(c, "voxel", tot_pts, 64.0 * tot_pts);
    // phase A: bounding boxes, one round trip
    DevBuf bb(c, nj * 8 * sizeof(uint32_t));
    for (size_t k = 0; k < nj; ++k) {
        LTM_HIP(bbox_init(bb.as<uint32_t>() + 8 * k, c->stream));
        if (jobs[k].has_cached && c->voxel_identity) LTM_HIP(bbox_reduce_check(jobs[k].pts, jobs[k].n, jobs[k].cached, bb.as<uint32_t>() + 8 * k, c->stream));
        else LTM_HIP(bbox_reduce(jobs[k].pts, jobs[k].n, bb.as<uint32_t>() + 8 * k, c->stream));
    }
    std::vector<uint32_t> enc(nj * 8);
    d2h(c, enc.data(), bb.p, enc.size() * 4);
    // phase B: keys, sort, head flags, scan; the counts go to one small device array
    DevBuf counts(c, nj * 4);
    for (size_t k = 0; k < nj; ++k) {
        VoxelJob& j = jobs[k];
        if (j.n == 0) { LTM_HIP(hipMemsetAsync(counts.as<uint32_t>() + k, 0, 4, c->stream)); continue; }
        ++c->voxel_calls;
        float mn[3], mx[3];
        for (int d = 0; d < 3; ++d) { mn[d] = bbox_decode(enc[8 * k + d]); mx[d] = bbox_decode(enc[8 * k + 3 + d]); }
        if (!octree_frame_from_bbox(mn, mx, j.leaf, &j.f)) throw Err{LTM_E_UNSUPPORTED, "octree depth > 21 (extent / leaf too large)"};
        if (j.has_cached && c->voxel_identity && enc[8 * k + 6] == 0 && same_frame(j.f, j.cached)) {      // see voxel_centroid_raw
            j.identity = true;
            j.out = reinterpret_cast<float4*>(c->pool.alloc(j.n * sizeof(float4)));
            d2d(c, j.out, j.pts, j.n * sizeof(float4));
            LTM_HIP(fill_u32(counts.as<uint32_t>() + k, (uint32_t)j.n, 1, c->stream));
            ++c->voxel_identity_hits;
            continue;
        }
        while (j.ib < 32 && ((size_t)1 << j.ib) < j.n) ++j.ib;
        const KeyCompress kc = key_compress_for(mn, mx, j.f, c->voxel_key_compress != 0);
        j.packed = kc.bits + j.ib <= 64;
        j.mbits = j.packed ? kc.bits : 3 * j.f.depth;
        j.kshift = j.packed ? j.ib : 0;
        const size_t n = j.n;
        j.keys.reset(new DevBuf(c, n * 8)); j.idx.reset(new DevBuf(c, j.packed ? 8 : n * 4));
        j.keys2.reset(new DevBuf(c, n * 8)); j.idx2.reset(new DevBuf(c, j.packed ? 8 : n * 4));
        if (j.packed) {
            LTM_HIP(morton_keys_packed(j.pts, n, j.f, kc, j.ib, j.keys->as<uint64_t>(), c->stream));
            const size_t stb = sort_keys_temp_bytes(n);
            DevBuf stemp(c, stb);
            LTM_HIP(sort_keys_u64(j.keys->as<uint64_t>(), j.keys2->as<uint64_t>(), n, j.ib, j.ib + j.mbits, stemp.p, stb, c->stream));
        } else {
            LTM_HIP(morton_keys(j.pts, n, j.f, j.keys->as<uint64_t>(), j.idx->as<uint32_t>(), c->stream));
            const size_t stb = sort_temp_bytes(n);
            DevBuf stemp(c, stb);
            LTM_HIP(sort_pairs_u64(j.keys->as<uint64_t>(), j.keys2->as<uint64_t>(), j.idx->as<uint32_t>(), j.idx2->as<uint32_t>(), n, j.mbits, stemp.p, stb, c->stream));
        }
        j.keys.reset(); j.idx.reset();      // stream-ordered pool: reusable by the next job's buffers
        j.starts.reset(new DevBuf(c, n * 4));
        voxel_segments(c, j.keys2->as<uint64_t>(), n, j.kshift, j.starts->as<uint32_t>(), counts.as<uint32_t>() + k);
    }
    std::vector<uint32_t> nv(nj);
    d2h(c, nv.data(), counts.p, nj * 4);
    // phase C: centroids
    for (size_t k = 0; k < nj; ++k) {
        VoxelJob& j = jobs[k];
        j.nvox = nv[k];
        if (j.n == 0 || j.identity) continue;
        j.out = reinterpret_cast<float4*>(c->pool.alloc(j.nvox * sizeof(float4)));
        if (j.packed) LTM_HIP(voxel_centroids_packed(j.pts, j.keys2->as<uint64_t>(), ((uint64_t)1 << j.ib) - 1, j.starts->as<uint32_t>(), j.nvox, j.n, j.out, c->stream));
        else LTM_HIP(voxel_centroids(j.pts, j.idx2->as<uint32_t>(), j.starts->as<uint32_t>(), j.nvox, j.n, j.out, c->stream));
        j.keys2.reset(); j.idx2.reset(); j.starts.reset();
    }
}


} // namespace ltm_detail

// =========================================================================================== C ABI
extern "C" {

int ltm_voxel_centroid(ltm_ctx* c, ltm_cloud hin, float leaf, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        const Cloud in = get_cloud(c, hin);
        float4* d = nullptr;
        OctreeFrame f{};
        const bool have = in.vf_ok && in.vleaf == leaf;
        const size_t nv = voxel_centroid_raw(c, in.d, in.n, leaf, &d, 0, 1, have ? &in.vf : nullptr, &f);
        if (!d) d = reinterpret_cast<float4*>(c->pool.alloc(sizeof(float4)));
        *out = new_cloud(c, d, nv);
        if (in.n) set_frame(c, *out, f, leaf, true);
    });
}

int ltm_voxel_centroid_batch(ltm_ctx* c, size_t n, const ltm_cloud* in, const float* leaf, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE((in && leaf && out) || n == 0, "null argument");
        std::vector<VoxelJob> jobs(n);
        for (size_t k = 0; k < n; ++k) {
            const Cloud cl = get_cloud(c, in[k]);
            jobs[k].pts = cl.d; jobs[k].n = cl.n; jobs[k].leaf = leaf[k];
            jobs[k].has_cached = cl.vf_ok && cl.vleaf == leaf[k] && cl.n > 0;
            jobs[k].cached = cl.vf;
        }
        voxel_centroid_batch(c, jobs);
        size_t done = 0;
        try {
            for (; done < n; ++done) {
                float4* d = jobs[done].out ? jobs[done].out : reinterpret_cast<float4*>(c->pool.alloc(sizeof(float4)));
                jobs[done].out = nullptr;
                out[done] = new_cloud(c, d, jobs[done].nvox);
                if (jobs[done].n) set_frame(c, out[done], jobs[done].f, jobs[done].leaf, true);
            }
        } catch (...) {      // handles made so far stay valid for the caller to free; the rest of the outputs go back to the pool
            for (size_t k = done; k < n; ++k) if (jobs[k].out) c->pool.free(jobs[k].out);
            throw;
        }
    });
}

int ltm_voxel_centroid_shard(ltm_ctx* c, ltm_cloud hin, float leaf, uint32_t shard, uint32_t n_shards, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        LTM_REQUIRE(n_shards >= 1 && n_shards <= 4096 && shard < n_shards, "shard index out of range");
        const Cloud in = get_cloud(c, hin);
        float4* d = nullptr;
        const size_t nv = voxel_centroid_raw(c, in.d, in.n, leaf, &d, shard, n_shards);
        if (!d) d = reinterpret_cast<float4*>(c->pool.alloc(sizeof(float4)));
        *out = new_cloud(c, d, nv);
    });
}

// ---- key-range exchange (multi-GPU voxel grid of a cloud whose POINTS are spread over the ranks)
static void box_check(const float* mn, const float* mx)
{
    LTM_REQUIRE(mn && mx, "null bounding box");
    for (int d = 0; d < 3; ++d) LTM_REQUIRE(std::isfinite(mn[d]) && std::isfinite(mx[d]) && mn[d] <= mx[d], "bounding box must be finite and ordered");
}

// packed keys of `in` under the frame of the box (mn, mx): the compressed Morton code starts at bit `ib` of every key, `shift` brings its top
// <= 12 bits down to the histogram bin
// The compressed code without its `drop` lowest bits (a PREFIX of the code: points of one voxel still share it, order is kept up to ties)
static KeyCompress key_compress_drop_low(const KeyCompress& kc, unsigned drop)
{
    if (!drop) return kc;
    KeyCompress o{};
    for (int r = 0; r < kc.n_runs; ++r) {
        unsigned len = 0;
        while (len < 64 && ((kc.mask[r] >> len) & 1ull)) ++len;
        const unsigned lo = kc.dst[r], hi = lo + len;            // the run fills output bits [lo, hi)
        if (hi <= drop) continue;
        const unsigned cut = lo < drop ? drop - lo : 0;           // bits of the run that fall below the cut
        o.src[o.n_runs] = (unsigned char)(kc.src[r] + cut);
        o.dst[o.n_runs] = (unsigned char)(lo + cut - drop);
        o.mask[o.n_runs] = kc.mask[r] >> cut;
        ++o.n_runs;
    }
    o.bits = kc.bits > drop ? kc.bits - drop : 1;
    if (o.n_runs == 0) { o.n_runs = 1; o.src[0] = 0; o.dst[0] = 0; o.mask[0] = 0; }
    return o;
}

static void box_keys(ltm_ctx* c, const Cloud& in, const float* mn, const float* mx, float leaf, DevBuf& keys, unsigned* ib_out, unsigned* shift_out)
{
    OctreeFrame f;
    if (!octree_frame_from_bbox(mn, mx, leaf, &f)) throw Err{LTM_E_UNSUPPORTED, "octree depth > 21 (extent / leaf too large)"};
    unsigned ib = 1;
    while (ib < 32 && ((size_t)1 << ib) < in.n) ++ib;
    // These keys only ROUTE points (4096-bin histogram of the top 12 code bits, then a range split): whole voxels must stay together, which any prefix of
    // the code guarantees.  So the code is cut to what fits beside a 32-bit index -- a function of the SHARED box alone: every rank of the exchange takes the
    // same decision whatever its local point count (a rank-local `ib` could let one rank throw while the others entered the collective)
    KeyCompress kc = key_compress_for(mn, mx, f, true);
    if (kc.bits + 32 > 64) kc = key_compress_drop_low(kc, kc.bits + 32 - 64);
    LTM_HIP(morton_keys_packed(in.d, in.n, f, kc, ib, keys.as<uint64_t>(), c->stream));
    *ib_out = ib;
    *shift_out = ib + (kc.bits > 12 ? kc.bits - 12 : 0);
}

int ltm_cloud_bbox(ltm_ctx* c, ltm_cloud hin, float* mn, float* mx)
{
    return guarded(c, [&] {
        LTM_REQUIRE(mn && mx, "null argument");
        const Cloud in = get_cloud(c, hin);
        if (in.n == 0) { for (int d = 0; d < 3; ++d) { mn[d] = INFINITY; mx[d] = -INFINITY; } return; }
        bbox_of(c, in.d, in.n, mn, mx);
    });
}

int ltm_voxel_key_histogram(ltm_ctx* c, ltm_cloud hin, const float* mn, const float* mx, float leaf, uint32_t* hist)
{
    return guarded(c, [&] {
        LTM_REQUIRE(hist, "null argument");
        LTM_REQUIRE(leaf > 0.0f, "leaf size must be positive");
        const Cloud in = get_cloud(c, hin);
        std::memset(hist, 0, kVoxelKeyBins * sizeof(uint32_t));
        if (in.n == 0) return;
        box_check(mn, mx);
        LTM_REQUIRE(in.n < 0xffffffffull, "cloud too large for 32-bit point indices");
        DevBuf keys(c, in.n * 8), hd(c, kVoxelKeyBins * sizeof(uint32_t));
        unsigned ib, shift;
        box_keys(c, in, mn, mx, leaf, keys, &ib, &shift);
        LTM_HIP(key_histogram(keys.as<uint64_t>(), in.n, shift, hd.as<uint32_t>(), c->stream));
        d2h(c, hist, hd.p, kVoxelKeyBins * sizeof(uint32_t));
    });
}

int ltm_voxel_key_split(ltm_ctx* c, ltm_cloud hin, const float* mn, const float* mx, float leaf, uint32_t n_parts, const uint32_t* cut_bins, ltm_cloud* parts)
{
    return guarded(c, [&] {
        LTM_REQUIRE(parts && cut_bins, "null argument");
        LTM_REQUIRE(n_parts >= 1 && n_parts <= 4096, "part count out of range");
        LTM_REQUIRE(cut_bins[0] == 0 && cut_bins[n_parts] == (uint32_t)kVoxelKeyBins, "cuts must run from bin 0 to the number of bins");
        for (uint32_t r = 0; r < n_parts; ++r) LTM_REQUIRE(cut_bins[r] <= cut_bins[r + 1], "cuts must not decrease");
        const Cloud in = get_cloud(c, hin);
        for (uint32_t r = 0; r < n_parts; ++r) parts[r] = 0;
        if (in.n == 0) { float4* d; for (uint32_t r = 0; r < n_parts; ++r) parts[r] = alloc_cloud(c, 0, &d); return; }
        box_check(mn, mx);
        LTM_REQUIRE(leaf > 0.0f, "leaf size must be positive");
        LTM_REQUIRE(in.n < 0xffffffffull, "cloud too large for 32-bit point indices");
        DevBuf keys(c, in.n * 8), flags(c, in.n);
        unsigned ib, shift;
        box_keys(c, in, mn, mx, leaf, keys, &ib, &shift);
        for (uint32_t r = 0; r < n_parts; ++r) {
            // part r = the points whose histogram bin lies in [cut[r], cut[r+1]), in input order (bins are prefixes of the code: no voxel straddles a cut)
            const uint64_t lo = (uint64_t)cut_bins[r] << shift;
            const uint64_t hi = cut_bins[r + 1] >= (uint32_t)kVoxelKeyBins ? ~0ull : (uint64_t)cut_bins[r + 1] << shift;
            if (cut_bins[r] == cut_bins[r + 1]) { float4* d; parts[r] = alloc_cloud(c, 0, &d); continue; }
            LTM_HIP(key_range_flags(keys.as<uint64_t>(), in.n, lo, hi, flags.as<uint8_t>(), c->stream));
            ltm_cloud rest_unused = 0;
            (void)rest_unused;
            do_partition(c, in, flags.as<uint8_t>(), nullptr, &parts[r]);
            c->clouds[parts[r]].vf_ok = false;         // a part of a merged cloud was never gridded
        }
    });
}

int ltm_voxel_centroid_box(ltm_ctx* c, ltm_cloud hin, const float* mn, const float* mx, float leaf, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        const Cloud in = get_cloud(c, hin);
        float4* d = nullptr;
        size_t nv = 0;
        if (in.n) {
            box_check(mn, mx);
            nv = voxel_centroid_raw(c, in.d, in.n, leaf, &d, 0, 1, nullptr, nullptr, mn, mx);
        }
        if (!d) d = reinterpret_cast<float4*>(c->pool.alloc(sizeof(float4)));
        *out = new_cloud(c, d, nv);
    });
}

int ltm_voxel_centroid_scanset(ltm_ctx* c, ltm_scanset hin, float leaf, ltm_scanset* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        LTM_REQUIRE(leaf > 0.0f, "leaf size must be positive");
        const ScanSet& s = get_ss(c, hin);
        const size_t nk = s.nkf();
        const size_t n = s.n_pts;
        LTM_REQUIRE(n < 0xffffffffull, "scan set too large for 32-bit point indices");
        std::vector<uint64_t> off(nk + 1, 0);
        if (n == 0 || nk == 0) {
            *out = new_scanset(c, reinterpret_cast<float4*>(c->pool.alloc(sizeof(float4))), std::move(off));
            return;
        }
        ProfScope ps(c, "voxel_scanset", (double)n, 64.0 * n);
        // every keyframe gets its own octree frame (bounding box re-derived per cloud, as octreeDownsampling does)
        DevBuf bb(c, nk * 6 * sizeof(uint32_t));
        LTM_HIP(bbox_reduce_seg(s.d, s.off_dev, nk, n, bb.as<uint32_t>(), c->stream));
        std::vector<uint32_t> enc(nk * 6);
        d2h(c, enc.data(), bb.p, enc.size() * 4);
        std::vector<OctreeFrame> frames(nk);
        unsigned dmax = 1;
        for (size_t k = 0; k < nk; ++k) {
            frames[k] = OctreeFrame{0, 0, 0, (double)leaf, 1};
            if (s.off[k + 1] == s.off[k]) continue;
            float mn[3], mx[3];
            for (int d = 0; d < 3; ++d) { mn[d] = bbox_decode(enc[6 * k + d]); mx[d] = bbox_decode(enc[6 * k + 3 + d]); }
            if (!octree_frame_from_bbox(mn, mx, leaf, &frames[k])) throw Err{LTM_E_UNSUPPORTED, "octree depth > 21 (extent / leaf too large)"};
            dmax = std::max(dmax, frames[k].depth);
        }
        unsigned kf_bits = 1;
        while ((1ull << kf_bits) < nk) ++kf_bits;
        const unsigned shift = 3 * dmax;
        if (shift + kf_bits > 64) throw Err{LTM_E_UNSUPPORTED, "scan set: keyframe id + Morton code exceed 64 key bits"};
        DevBuf fdev(c, nk * sizeof(OctreeFrame));
        h2d(c, fdev.p, frames.data(), nk * sizeof(OctreeFrame));
        DevBuf keys(c, n * 8), keys2(c, n * 8), idx(c, n * 4), idx2(c, n * 4);
        LTM_HIP(morton_keys_seg(s.d, s.off_dev, nk, n, fdev.as<OctreeFrame>(), shift, keys.as<uint64_t>(), idx.as<uint32_t>(), c->stream));
        const size_t stb = sort_temp_bytes(n);
        {
            DevBuf stemp(c, stb);
            LTM_HIP(sort_pairs_u64(keys.as<uint64_t>(), keys2.as<uint64_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), n, shift + kf_bits, stemp.p, stb, c->stream));
        }
        DevBuf heads(c, n), pos(c, n * 4);
        LTM_HIP(head_flags(keys2.as<uint64_t>(), n, heads.as<uint8_t>(), c->stream));
        const size_t tb = scan_temp_bytes(n);
        DevBuf temp(c, tb);
        LTM_HIP(exclusive_scan_u8(heads.as<uint8_t>(), pos.as<uint32_t>(), n, temp.p, tb, c->stream));
        const size_t nvox = scan_total_u8(c, heads.as<uint8_t>(), pos.as<uint32_t>(), n);
        // the sort key leads with the keyframe id, so keyframe k still occupies sorted positions [off[k], off[k+1])
        DevBuf bout(c, (nk + 1) * 4);
        LTM_HIP(gather_u32(pos.as<uint32_t>(), s.off_dev, nk + 1, n, (uint32_t)nvox, bout.as<uint32_t>(), c->stream));
        std::vector<uint32_t> b(nk + 1);
        d2h(c, b.data(), bout.p, (nk + 1) * 4);
        for (size_t k = 0; k <= nk; ++k) off[k] = b[k];
        DevBuf starts(c, nvox * 4);
        LTM_HIP(segment_starts(heads.as<uint8_t>(), pos.as<uint32_t>(), n, starts.as<uint32_t>(), c->stream));
        float4* o = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(nvox, 1) * sizeof(float4)));
        LTM_HIP(voxel_centroids(s.d, idx2.as<uint32_t>(), starts.as<uint32_t>(), nvox, n, o, c->stream));
        *out = new_scanset(c, o, std::move(off));
    });
}

// The loader's per-scan pcl::VoxelGrid on a device-resident scan set, in two halves so that the caller can keep the GPU busy while the host threads
// reproduce std::sort's order (round 5, VERDICT r4 item 6: on the lifelong cascade the hand-over was 199 of 1054 ms per step with the GPU idle for most of it):
//   begin: bounding boxes, frames, keys on the device; the keys travel to a pinned buffer on the COPY stream and a coordinator thread sorts them keyframe
//          by keyframe as soon as they are there.  Returns at once: what the caller enqueues next on the context runs beside the transfer and the sort.
//   end:   waits for the order, sends it up, gathers, segments and averages.
// ltm_voxel_grid_scanset = begin + end back to back.
struct ltm_vgs {
    ltm_scanset in = 0;
    float leaf = 0.0f;
    size_t nk = 0, n = 0;
    unsigned kf_bits = 1;
    bool pcl_order = true, trivial = false;
    std::vector<VoxelGridFrame> frames;
    std::unique_ptr<DevBuf> fdev, keys, idx;
    ltm_pclsort::Entry* he = nullptr;      // pinned: the keys as they arrive, then the sorted (leaf index, point index) pairs
    uint32_t* hi = nullptr;               // pinned: the point order
    hipEvent_t ev_keys = nullptr;
    std::thread coordinator;
    std::atomic<bool> failed{false};
    std::chrono::steady_clock::time_point t_begin, t_sorted;
};

namespace {
void vgs_release(ltm_ctx* c, ltm_vgs* v)
{
    if (v->coordinator.joinable()) v->coordinator.join();
    if (v->ev_keys) { (void)hipEventDestroy(v->ev_keys); v->ev_keys = nullptr; }
    if (v->he) { pinned_free(c, v->he); v->he = nullptr; }
    if (v->hi) { pinned_free(c, v->hi); v->hi = nullptr; }
    delete v;
}
void vgs_begin(ltm_ctx* c, ltm_scanset hin, float leaf, ltm_vgs** ticket)
{
    LTM_REQUIRE(ticket, "null argument");
    LTM_REQUIRE(leaf > 0.0f, "leaf size must be positive");
    const ScanSet& s = get_ss(c, hin);
    // every way out of this function before the ticket is handed over goes through ONE cleanup: the copy stream drained (a transfer may still read the
    // keys), the coordinator joined, pinned buffers and the event released
    struct Guard {
        ltm_ctx* c; ltm_vgs* p;
        ~Guard() { if (p) { if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream); vgs_release(c, p); } }
        ltm_vgs* operator->() const { return p; }
        ltm_vgs* get() const { return p; }
        ltm_vgs* release() { ltm_vgs* r = p; p = nullptr; return r; }
    } v{c, new ltm_vgs};
    v->in = hin; v->leaf = leaf; v->nk = s.nkf(); v->n = s.n_pts;
    v->t_begin = v->t_sorted = std::chrono::steady_clock::now();
    const size_t nk = v->nk, n = v->n;
    LTM_REQUIRE(n < 0xffffffffull, "scan set too large for 32-bit point indices");
    if (n == 0 || nk == 0) { v->trivial = true; c->vgs_open.push_back(v.get()); *ticket = v.release(); return; }
    ProfScope ps(c, "voxel_grid_scanset", (double)n, 64.0 * n);
    DevBuf bb(c, nk * 6 * sizeof(uint32_t));
    LTM_HIP(bbox_reduce_seg(s.d, s.off_dev, nk, n, bb.as<uint32_t>(), c->stream));
    std::vector<uint32_t> enc(nk * 6);
    d2h(c, enc.data(), bb.p, enc.size() * 4);
    // pcl::VoxelGrid::applyFilter (PCL 1.10 voxel_grid.hpp): inverse leaf size in float, the "leaf size is too small"
    // test on int64 cell counts, min_b / div_b from floor(min * inv), floor(max * inv)
    const float inv = 1.0f / leaf;
    v->frames.resize(nk);
    for (size_t k = 0; k < nk; ++k) {
        VoxelGridFrame& f = v->frames[k];
        f.inv = inv; f.passthrough = 1;
        for (int d = 0; d < 3; ++d) { f.min_b[d] = 0; f.div_b[d] = 1; }
        if (s.off[k + 1] == s.off[k]) continue;
        float mn[3], mx[3];
        int64_t cells = 1;
        for (int d = 0; d < 3; ++d) {
            mn[d] = bbox_decode(enc[6 * k + d]); mx[d] = bbox_decode(enc[6 * k + 3 + d]);
            const float ext = (mx[d] - mn[d]) * inv;
            cells *= (int64_t)ext + 1;
        }
        if (cells > (int64_t)INT32_MAX) continue;           // output = input (the common case for a raw 0.05 m scan)
        f.passthrough = 0;
        for (int d = 0; d < 3; ++d) {
            f.min_b[d] = (int)std::floor(mn[d] * inv);
            f.div_b[d] = (int)std::floor(mx[d] * inv) - f.min_b[d] + 1;
        }
    }
    while ((1ull << v->kf_bits) < nk) ++v->kf_bits;
    v->fdev.reset(new DevBuf(c, nk * sizeof(VoxelGridFrame)));
    h2d(c, v->fdev->p, v->frames.data(), nk * sizeof(VoxelGridFrame));
    v->keys.reset(new DevBuf(c, n * 8));
    v->idx.reset(new DevBuf(c, n * 4));
    LTM_HIP(voxelgrid_keys_seg(s.d, s.off_dev, nk, n, v->fdev->as<VoxelGridFrame>(), v->keys->as<uint64_t>(), v->idx->as<uint32_t>(), c->stream));
    // PCL groups the points of a leaf with std::sort on the LEAF INDEX ONLY (cloud_point_index_idx::operator<): the order of the float
    // sums inside a voxel is whatever that (unstable) sort leaves, and a voxel with three or more points rounds differently in a
    // different order.  Round 4 measured it against the reference's own sources compiled with stand-in headers (oracle/_ref): an
    // input-order sum changes the last bit of ~0.05 % of the loaded points.  To hand over what the reference would re-load, the
    // default makes the SAME std::sort call on the host, one keyframe per task (the permutation is a function of the key sequence
    // alone): keys down (8 B / point), point order up (4 B / point), everything else stays on the device.  LTM_VOXELGRID_ORDER=input
    // keeps the whole grid on the device with a stable radix sort (input order inside a voxel; faster, not bit-identical to PCL).
    const char* order_env = std::getenv("LTM_VOXELGRID_ORDER");
    v->pcl_order = !(order_env && std::strcmp(order_env, "input") == 0);
    if (v->pcl_order) {
        // The pinned buffer receives the 64-bit keys (keyframe id << 32 | leaf index) and is read as the (leaf index, point index) pairs PCL
        // sorts: on this little-endian host a key's low word IS the pair's first member, and the high word -- the keyframe id, which the
        // keyframe-by-keyframe sort does not need -- is overwritten with the point index.  ltm_pclsort::sort performs std::sort's element moves
        // without its branch mispredictions (ltm_pclsort.h: checked against std::sort itself).
        using Entry = ltm_pclsort::Entry;
        static_assert(sizeof(Entry) == sizeof(uint64_t) && offsetof(Entry, idx) == 0 && offsetof(Entry, cloud_point_index) == 4, "a pair overlays a key");
        v->he = static_cast<Entry*>(pinned_alloc(c, n * 8));
        v->hi = static_cast<uint32_t*>(pinned_alloc(c, n * 4));
        // keys down on the copy stream, behind the kernel that makes them: the compute stream is free for whatever the caller enqueues next
        hipEvent_t made = nullptr;
        LTM_HIP(hipEventCreateWithFlags(&made, hipEventDisableTiming));
        hipError_t e = hipEventRecord(made, c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(copy_stream(c), made, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(v->he, v->keys->p, n * 8, hipMemcpyDeviceToHost, copy_stream(c));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&v->ev_keys, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(v->ev_keys, copy_stream(c));
        (void)hipEventDestroy(made);
        LTM_HIP(e);
        // one keyframe per task.  A scans_updated set of 500 keyframes x 107-134 k points is ~1 s of host CPU time with ltm_pclsort (2.5 s
        // with std::sort): 20 ms on 64 threads of the GPU box (profiles/r4_hostsort_pclsort_vs_stdsort.txt); LTM_VOXELGRID_THREADS
        // overrides the cap of 64
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const char* tenv = getenv("LTM_VOXELGRID_THREADS");
        const size_t cap = tenv && atoi(tenv) > 0 ? (size_t)atoi(tenv) : 64;
        const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw, cap), nk));
        ltm_vgs* vp = v.get();
        const ScanSet* sp = &s;          // (the input scan set stays alive until the ticket is ended: the caller's contract)
        const int device = c->device;
        vp->coordinator = std::thread([vp, sp, nt, device] {
            // the only wait of the whole order: for the keys.  One thread waits; the workers never touch the runtime
            if (hipSetDevice(device) != hipSuccess || hipEventSynchronize(vp->ev_keys) != hipSuccess) { vp->failed = true; return; }
            std::atomic<size_t> next{0};
            auto work = [&] {
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= vp->nk) return;
                    const size_t a = sp->off[k], b = sp->off[k + 1];
                    Entry* he = vp->he; uint32_t* hi = vp->hi;
                    if (vp->frames[k].passthrough) { for (size_t i = a; i < b; ++i) hi[i] = (uint32_t)i; continue; }
                    for (size_t i = a; i < b; ++i) he[i].cloud_point_index = (uint32_t)i;
                    ltm_pclsort::sort(he + a, he + b);
                    for (size_t i = a; i < b; ++i) hi[i] = he[i].cloud_point_index;
                }
            };
            std::vector<std::thread> pool;
            try { for (size_t t = 1; t < nt; ++t) pool.emplace_back(work); }
            catch (...) {}      // fewer threads than asked for: the ones that started (and this one) still finish every keyframe
            try { work(); } catch (...) { vp->failed = true; }
            for (std::thread& t : pool) t.join();
            vp->t_sorted = std::chrono::steady_clock::now();
        });
    }
    c->vgs_open.push_back(v.get());      // the context joins and releases whatever is still open when it is destroyed
    *ticket = v.release();
}
void vgs_end(ltm_ctx* c, ltm_vgs* v, ltm_scanset* out)
{
    LTM_REQUIRE(out, "null argument");
    const size_t nk = v->nk, n = v->n;
    std::vector<uint64_t> off(nk + 1, 0);
    if (v->trivial) { *out = new_scanset(c, reinterpret_cast<float4*>(c->pool.alloc(sizeof(float4))), std::move(off)); return; }
    const ScanSet& s = get_ss(c, v->in);
    LTM_REQUIRE(s.nkf() == nk && s.n_pts == n, "the scan set changed between begin and end");
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    DevBuf keys2(c, n * 8), idx2(c, n * 4);
    ProfScope ps(c, "voxel_grid_scanset", 0.0, 0.0, -1.0, false);      // the second half of the launch counted by _begin: order upload / sort, gather, segments, centroids
    if (v->pcl_order) {
        const auto t_wait = std::chrono::steady_clock::now();
        v->coordinator.join();
        if (v->failed) throw Err{LTM_E_DEVICE, "voxel_grid_scanset: the host order of the keys failed (keys transfer or worker threads)"};
        const auto t_joined = std::chrono::steady_clock::now();
        LTM_HIP(hipMemcpyAsync(idx2.p, v->hi, n * 4, hipMemcpyHostToDevice, c->stream));
        LTM_HIP(gather_u64_by_u32(v->keys->as<uint64_t>(), idx2.as<uint32_t>(), n, keys2.as<uint64_t>(), c->stream));
        sync(c);
        if (getenv("LTM_VOXELGRID_TIMING"))
            fprintf(stderr, "[ltm] voxel_grid_scanset (PCL order): %zu points, %zu keyframes: begin -> order ready %.1f ms (keys down on the copy stream + the sort on host threads), "
                            "the caller waited %.1f ms of it in end, order up + gather %.1f ms\n",
                    n, nk, ms(v->t_begin, v->t_sorted), ms(t_wait, t_joined), ms(t_joined, std::chrono::steady_clock::now()));
    } else {
        const size_t stb = sort_temp_bytes(n);
        DevBuf stemp(c, stb);
        LTM_HIP(sort_pairs_u64(v->keys->as<uint64_t>(), keys2.as<uint64_t>(), v->idx->as<uint32_t>(), idx2.as<uint32_t>(), n, 32 + v->kf_bits, stemp.p, stb, c->stream));
    }
    DevBuf heads(c, n), pos(c, n * 4);
    LTM_HIP(head_flags(keys2.as<uint64_t>(), n, heads.as<uint8_t>(), c->stream));
    const size_t tb = scan_temp_bytes(n);
    DevBuf temp(c, tb);
    LTM_HIP(exclusive_scan_u8(heads.as<uint8_t>(), pos.as<uint32_t>(), n, temp.p, tb, c->stream));
    const size_t nvox = scan_total_u8(c, heads.as<uint8_t>(), pos.as<uint32_t>(), n);
    // the keys lead with the keyframe id (and the host order works keyframe by keyframe), so keyframe k still occupies positions [off[k], off[k+1])
    DevBuf bout(c, (nk + 1) * 4);
    LTM_HIP(gather_u32(pos.as<uint32_t>(), s.off_dev, nk + 1, n, (uint32_t)nvox, bout.as<uint32_t>(), c->stream));
    std::vector<uint32_t> b(nk + 1);
    d2h(c, b.data(), bout.p, (nk + 1) * 4);
    for (size_t k = 0; k <= nk; ++k) off[k] = b[k];
    DevBuf starts(c, nvox * 4);
    LTM_HIP(segment_starts(heads.as<uint8_t>(), pos.as<uint32_t>(), n, starts.as<uint32_t>(), c->stream));
    float4* o = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(nvox, 1) * sizeof(float4)));
    LTM_HIP(voxelgrid_centroids(s.d, keys2.as<uint64_t>(), idx2.as<uint32_t>(), starts.as<uint32_t>(), v->fdev->as<VoxelGridFrame>(), nvox, n, o, c->stream));
    *out = new_scanset(c, o, std::move(off));
}
} // namespace

int ltm_voxel_grid_scanset_begin(ltm_ctx* c, ltm_scanset hin, float leaf, ltm_vgs** ticket)
{
    if (ticket) *ticket = nullptr;
    return guarded(c, [&] { vgs_begin(c, hin, leaf, ticket); });
}

int ltm_voxel_grid_scanset_end(ltm_ctx* c, ltm_vgs* ticket, ltm_scanset* out)
{
    if (!ticket) return LTM_E_INVALID;
    const int rc = guarded(c, [&] { vgs_end(c, ticket, out); });
    if (c) {      // the ticket is consumed either way (its buffers may still be read by queued work)
        std::lock_guard<std::recursive_mutex> lk(c->mx);
        (void)hipStreamSynchronize(c->stream);
        c->vgs_open.erase(std::remove(c->vgs_open.begin(), c->vgs_open.end(), ticket), c->vgs_open.end());
        vgs_release(c, ticket);
    }
    return rc;
}

int ltm_voxel_grid_scanset(ltm_ctx* c, ltm_scanset hin, float leaf, ltm_scanset* out)
{
    ltm_vgs* t = nullptr;
    const int rc = ltm_voxel_grid_scanset_begin(c, hin, leaf, &t);
    if (rc != LTM_OK) return rc;
    return ltm_voxel_grid_scanset_end(c, t, out);
}

int ltm_debug_voxel_key_bits(const float* mn3, const float* mx3, float leaf, uint64_t* kept_mask, unsigned* depth, double* frame_min3)
{
    if (!mn3 || !mx3 || !kept_mask || !(leaf > 0.0f)) return LTM_E_INVALID;
    OctreeFrame f;
    if (!octree_frame_from_bbox(mn3, mx3, leaf, &f)) return LTM_E_UNSUPPORTED;
    const KeyCompress kc = key_compress_for(mn3, mx3, f, true);
    uint64_t m = 0;
    for (int r = 0; r < kc.n_runs; ++r) m |= kc.mask[r] << kc.src[r];
    *kept_mask = m;
    if (depth) *depth = f.depth;
    if (frame_min3) { frame_min3[0] = f.minx; frame_min3[1] = f.miny; frame_min3[2] = f.minz; }
    return (int)kc.bits;
}

int ltm_debug_voxel_stats(ltm_ctx* c, uint64_t* grids, uint64_t* identity_hits, int reset)
{
    return guarded(c, [&] {
        if (grids) *grids = c->voxel_calls;
        if (identity_hits) *identity_hits = c->voxel_identity_hits;
        if (reset) c->voxel_calls = c->voxel_identity_hits = 0;
    });
}

} // extern "C"

void ltm_detail::vgs_release_all(ltm_ctx* c)      // ltm_destroy: tickets nobody ended (an exception between _begin and _end on the host side)
{
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (ltm_vgs* t : c->vgs_open) vgs_release(c, t);
    c->vgs_open.clear();
}

