// ltm_api_knn.cpp -- C ABI: inter-session kNN change detection and the pre-clean (Session.cpp:393-533, 537-642)
#include "ltm_internal.h"

namespace ltm_detail {

// ------------------------------------------------------------------------------------ kNN
struct KnnIndex {
    ltm_ctx* c;
    float4* sorted = nullptr; HashEntry* table = nullptr; uint32_t mask = 0; KnnGrid g{}; float cell2_lo = 0; size_t Mt = 0;
    void* buckets = nullptr; uint32_t n_buckets = 0;      // phase-1 table of the two-phase query (k <= 4), see ltm_k_knn.hip
    void* bitmap = nullptr; uint32_t bitmap_mask = 0;     // sparse occupancy bitmap of the grid (phase 2 skips empty cells)
    explicit KnnIndex(ltm_ctx* c_) : c(c_) {}
    ~KnnIndex() { c->pool.free(sorted); c->pool.free(table); c->pool.free(buckets); c->pool.free(bitmap); }
    void build(const Cloud& target, int k, float thr)
    {
        Mt = target.n;
        LTM_REQUIRE(k >= 1 && k <= 16, "k must be in [1,16]");
        LTM_REQUIRE(thr > 0.0f, "kNN threshold must be positive");
        LTM_REQUIRE(Mt < 0xffffffffull, "target too large");
        if (Mt == 0) return;
        ProfScope p(c, "knn_build", (double)Mt, 20.0 * Mt);
        sorted = reinterpret_cast<float4*>(c->pool.alloc(Mt * sizeof(float4)));
        if (Mt <= 64) { d2d(c, sorted, target.d, Mt * sizeof(float4)); return; }   // brute force inside the query kernel
        float mn[3], mx[3];
        bbox_of(c, target.d, Mt, mn, mx);
        // cell edge: every neighbour with d^2 < k*thr must fall in the 27-cell block (margin 1e-3, floor 1e-4 m)
        double cell = std::sqrt((double)k * (double)thr) * (1.0 + 1e-3);
        const double ext = std::max({(double)mx[0] - mn[0], (double)mx[1] - mn[1], (double)mx[2] - mn[2], 1e-3});
        cell = std::max(cell, ext / 1.0e6);   // keeps every axis below 2^20 cells (id < 2^62)
        g.ox = (double)mn[0] - cell; g.oy = (double)mn[1] - cell; g.oz = (double)mn[2] - cell;
        g.inv_cell = 1.0 / cell;
        g.nx = (long long)std::floor(((double)mx[0] - g.ox) * g.inv_cell) + 2;
        g.ny = (long long)std::floor(((double)mx[1] - g.oy) * g.inv_cell) + 2;
        g.nz = (long long)std::floor(((double)mx[2] - g.oz) * g.inv_cell) + 2;
        cell2_lo = (float)(cell * cell * (1.0 - 1e-5));
        const double ncells = (double)g.nx * (double)g.ny * (double)g.nz;
        unsigned bits = 1;
        while (bits < 64 && std::ldexp(1.0, (int)bits) < ncells) ++bits;
        DevBuf keys(c, Mt * 8), keys2(c, Mt * 8), idx(c, Mt * 4), idx2(c, Mt * 4);
        LTM_HIP(cell_keys(target.d, Mt, g, keys.as<uint64_t>(), idx.as<uint32_t>(), c->stream));
        const size_t stb = sort_temp_bytes(Mt);
        {
            DevBuf stemp(c, stb);
            LTM_HIP(sort_pairs_u64(keys.as<uint64_t>(), keys2.as<uint64_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), Mt, bits, stemp.p, stb, c->stream));
        }
        LTM_HIP(gather_points(target.d, idx2.as<uint32_t>(), Mt, sorted, c->stream));
        DevBuf heads(c, Mt), pos(c, Mt * 4);
        LTM_HIP(head_flags(keys2.as<uint64_t>(), Mt, heads.as<uint8_t>(), c->stream));
        const size_t tb = scan_temp_bytes(Mt);
        DevBuf temp(c, tb);
        LTM_HIP(exclusive_scan_u8(heads.as<uint8_t>(), pos.as<uint32_t>(), Mt, temp.p, tb, c->stream));
        const size_t ncell = scan_total_u8(c, heads.as<uint8_t>(), pos.as<uint32_t>(), Mt);
        DevBuf starts(c, ncell * 4);
        LTM_HIP(segment_starts(heads.as<uint8_t>(), pos.as<uint32_t>(), Mt, starts.as<uint32_t>(), c->stream));
        size_t tsize = 1024;
        while (tsize < 2 * ncell) tsize <<= 1;
        mask = (uint32_t)(tsize - 1);
        table = reinterpret_cast<HashEntry*>(c->pool.alloc(tsize * sizeof(HashEntry)));
        LTM_HIP(fill_u64(reinterpret_cast<uint64_t*>(table), ~0ull, tsize * 2, c->stream));
        LTM_HIP(hash_build(keys2.as<uint64_t>(), starts.as<uint32_t>(), ncell, Mt, table, mask, c->stream));
        if (k <= 4 && c->knn_two_phase) {
            // 64-byte buckets at a load factor of ~0.55: with two candidate places ~97 % of the cells get one (the others are served by phase 2)
            n_buckets = (uint32_t)std::min<size_t>(std::max<size_t>(1024, ncell + ncell * 4 / 5), 0x7fffffffu);
            buckets = c->pool.alloc((size_t)n_buckets * 64);
            LTM_HIP(hipMemsetAsync(buckets, 0xff, (size_t)n_buckets * 64, c->stream));
            LTM_HIP(knn_bucket_build(sorted, keys2.as<uint64_t>(), starts.as<uint32_t>(), ncell, Mt, g, buckets, n_buckets, c->stream));
            if (c->knn_two_phase != 2) {      // LTM_KNN_FAST=2: no occupancy bitmap (A/B)
                size_t words = 1024;
                while (words < ncell / 4 && words < ((size_t)1 << 30)) words <<= 1;      // a surface fills ~16 of a block's 64 cells: ~4 words per occupied block
                bitmap_mask = (uint32_t)(words - 1);
                bitmap = c->pool.alloc(words * 8);
                LTM_HIP(hipMemsetAsync(bitmap, 0, words * 8, c->stream));
                LTM_HIP(knn_bitmap_build(keys2.as<uint64_t>(), starts.as<uint32_t>(), ncell, g, bitmap, bitmap_mask, c->stream));
            }
        }
    }
};

// split pts[0..n) by flag (1 -> first output) keeping order; per-keyframe offsets from `bounds` (n_b+1 point positions)
// offsets_dev[kf0 + j] - first are the same boundaries on the device (the scan set's own offset table)
void split_by_flag(ltm_ctx* c, const float4* pts, const uint8_t* flag, size_t n, const std::vector<uint64_t>& bounds, const uint64_t* offsets_dev,
                   size_t kf0, uint64_t first, float4** d_set, std::vector<uint64_t>* off_set, float4** d_unset, std::vector<uint64_t>* off_unset)
{
    const size_t nb = bounds.size() - 1;
    off_set->assign(nb + 1, 0); off_unset->assign(nb + 1, 0);
    *d_set = nullptr; *d_unset = nullptr;
    size_t nset = 0;
    if (n) {
        DevBuf pos(c, n * 4);
        const size_t tb = scan_temp_bytes(n);
        DevBuf temp(c, tb);
        LTM_HIP(exclusive_scan_u8(flag, pos.as<uint32_t>(), n, temp.p, tb, c->stream));
        DevBuf bout(c, (nb + 1) * 4);      // per-keyframe boundaries and the total in one small array: one host round trip
        LTM_HIP(flag_bounds(pos.as<uint32_t>(), flag, n, offsets_dev, kf0, first, nb, bout.as<uint32_t>(), c->stream));
        std::vector<uint32_t> b(nb + 1);
        d2h(c, b.data(), bout.p, (nb + 1) * 4);
        nset = b[nb];
        for (size_t j = 0; j <= nb; ++j) { (*off_set)[j] = b[j]; (*off_unset)[j] = bounds[j] - b[j]; }
        *d_set = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(nset, 1) * sizeof(float4)));
        *d_unset = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(n - nset, 1) * sizeof(float4)));
        LTM_HIP(partition_scatter(pts, flag, pos.as<uint32_t>(), n, *d_unset, *d_set, c->stream));
    } else {
        *d_set = reinterpret_cast<float4*>(c->pool.alloc(sizeof(float4)));
        *d_unset = reinterpret_cast<float4*>(c->pool.alloc(sizeof(float4)));
    }
}


} // namespace ltm_detail

// =========================================================================================== C ABI
extern "C" {

// ------------------------------------------------------------------------------- stages
int ltm_preclean(ltm_ctx* c, ltm_scanset hin, float radius, ltm_scanset* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        const ScanSet& s = get_ss(c, hin);
        DevBuf drop(c, std::max<size_t>(s.n_pts, 1));
        LTM_HIP(preclean_flags(s.d, s.n_pts, radius, drop.as<uint8_t>(), c->stream));
        float4 *d_drop, *d_keep;
        std::vector<uint64_t> off_drop, off_keep;
        split_by_flag(c, s.d, drop.as<uint8_t>(), s.n_pts, s.off, s.off_dev, 0, 0, &d_drop, &off_drop, &d_keep, &off_keep);
        c->pool.free(d_drop);
        *out = new_scanset(c, d_keep, std::move(off_keep));
    });
}

int ltm_knn_partition(ltm_ctx* c, ltm_cloud htarget, ltm_scanset hs, ltm_poses hp, size_t kf_begin, size_t kf_end, int k, float thr,
                      ltm_scanset* coexist, ltm_scanset* diff)
{
    return guarded(c, [&] {
        const Cloud target = get_cloud(c, htarget);
        const ScanSet& s = get_ss(c, hs);
        const Poses& p = get_poses(c, hp);
        LTM_REQUIRE(s.nkf() == p.n, "scan set and poses have different keyframe counts");
        LTM_REQUIRE(kf_begin <= kf_end && kf_end <= p.n, "keyframe range out of bounds");
        KnnIndex index(c);
        index.build(target, k, thr);
        const uint64_t first = s.off[kf_begin], n = s.off[kf_end] - first;
        DevBuf flag(c, std::max<size_t>(n, 1)), local(c, std::max<size_t>(n, 1) * 16);
        uint64_t longest = 0;
        for (size_t kk = kf_begin; kk < kf_end; ++kk) longest = std::max<uint64_t>(longest, s.off[kk + 1] - s.off[kk]);
        if (index.buckets && n && n < 0xffffffffull) {
            unsigned ibits = 0;
            const unsigned kbits = knn_sorted_queue_bits(index.g, n, &ibits);      // 0: the keys do not fit, phase 2 walks the queue in scan order
            // the global point of every undecided query, written by phase 1 (only those entries are ever read)
            std::unique_ptr<DevBuf> gpu;
            if (kbits) gpu.reset(new DevBuf(c, n * sizeof(float4)));
            float4* gp = gpu ? gpu->as<float4>() : nullptr;
            {
                ProfScope ps(c, "knn_query", (double)n, (double)n * (16.0 + 16.0 * k + 1.0));
                LTM_HIP(knn_two_phase_fast(s.d, s.off_dev, kf_begin, kf_end, first, n, longest, p.pose_dev, p.inv_dev, c->B2L, c->b2l_identity, index.g, index.buckets,
                                           index.n_buckets, k, thr, flag.as<uint8_t>(), local.as<float4>(), c->stream, gp));
            }
            if (kbits) {
                // phase 2 on a queue sorted by cell (round 4): one host round trip for the undecided count (four kNN stages per step)
                DevBuf pos(c, n * 4), count(c, 4);
                const size_t tb = std::max(scan_temp_bytes(n), sort_keys_temp_bytes(n));
                DevBuf temp(c, tb);
                ProfScope ps(c, "knn_query_p2", 0.0, 0.0);
                DevBuf q1(c, n * 8);
                LTM_HIP(knn_two_phase_compact_keyed(s.d, s.off_dev, kf_begin, kf_end, first, n, p.pose_dev, c->B2L, c->b2l_identity, index.g, ibits, flag.as<uint8_t>(),
                                                    pos.as<uint32_t>(), q1.as<uint64_t>(), count.as<uint32_t>(), temp.p, tb, c->stream, gp));
                uint32_t und = 0;
                d2h(c, &und, count.p, 4);
                if (c->prof_on) {      // the exact search's own floor: every undecided query is read again and must see its k neighbours (SURVEY 8d's per-query bytes)
                    ProfClass& pc = c->prof[(size_t)prof_class(c, "knn_query_p2")];
                    pc.units += (double)und; pc.bytes += (double)und * (16.0 + 16.0 * k + 1.0); pc.bytes_c += (double)und * (16.0 + 16.0 * k + 1.0);
                }
                if (und) {
                    DevBuf q2(c, (size_t)und * 8);
                    LTM_HIP(knn_two_phase_exact_sorted(s.d, s.off_dev, kf_begin, kf_end, first, p.pose_dev, c->B2L, c->b2l_identity, index.sorted, index.Mt, index.g, index.table,
                                                       index.mask, index.bitmap, index.bitmap_mask, k, thr, index.cell2_lo, flag.as<uint8_t>(), q1.as<uint64_t>(), q2.as<uint64_t>(),
                                                       und, ibits, kbits, temp.p, tb, c->stream, gp));
                }
                if (c->knn_stats_on) { c->knn_undecided += und; c->knn_queries += n; }
            } else {
            DevBuf pos(c, n * 4), queue(c, n * 4), count(c, 4);
            const size_t tb = scan_temp_bytes(n);
            DevBuf temp(c, tb);
            {
                ProfScope ps(c, "knn_query_p2", 0.0, 0.0);      // compaction + exact search of the undecided queries: its bytes are part of knn_query's algorithmic figure
                LTM_HIP(knn_two_phase_exact(s.d, s.off_dev, kf_begin, kf_end, first, n, p.pose_dev, c->B2L, c->b2l_identity, index.sorted, index.Mt, index.g, index.table,
                                            index.mask, index.bitmap, index.bitmap_mask, k, thr, index.cell2_lo, flag.as<uint8_t>(), pos.as<uint32_t>(), queue.as<uint32_t>(), count.as<uint32_t>(), temp.p, tb,
                                            c->stream));
            }
            if (c->knn_stats_on) {
                uint32_t und = 0;
                d2h(c, &und, count.p, 4);
                c->knn_undecided += und; c->knn_queries += n;
            }
            }
        } else {
            ProfScope ps(c, "knn_query", (double)n, (double)n * (16.0 + 16.0 * k + 1.0));
            LTM_HIP(knn_query_scans(s.d, s.off_dev, kf_begin, kf_end, first, n, longest, p.pose_dev, p.inv_dev, c->B2L, c->b2l_identity, index.sorted,
                                    index.Mt, index.g, index.table, index.mask, k, thr, index.cell2_lo, flag.as<uint8_t>(), local.as<float4>(), c->stream));
        }
        std::vector<uint64_t> bounds(kf_end - kf_begin + 1);
        for (size_t j = 0; j < bounds.size(); ++j) bounds[j] = s.off[kf_begin + j] - first;
        float4 *d_co, *d_di;
        std::vector<uint64_t> off_co, off_di;
        split_by_flag(c, local.as<float4>(), flag.as<uint8_t>(), n, bounds, s.off_dev, kf_begin, first, &d_co, &off_co, &d_di, &off_di);
        if (coexist) *coexist = new_scanset(c, d_co, std::move(off_co)); else c->pool.free(d_co);
        if (diff) *diff = new_scanset(c, d_di, std::move(off_di)); else c->pool.free(d_di);
    });
}

int ltm_knn_split_cloud(ltm_ctx* c, ltm_cloud htarget, ltm_cloud hquery, int k, float thr, ltm_cloud* near, ltm_cloud* far)
{
    return guarded(c, [&] {
        const Cloud target = get_cloud(c, htarget);
        const Cloud query = get_cloud(c, hquery);
        KnnIndex index(c);
        index.build(target, k, thr);
        DevBuf flag(c, std::max<size_t>(query.n, 1));
        {
            ProfScope ps(c, "knn_query", (double)query.n, (double)query.n * (16.0 + 16.0 * k + 1.0));
            LTM_HIP(knn_query_cloud(query.d, query.n, index.sorted, index.Mt, index.g, index.table, index.mask, k, thr, index.cell2_lo,
                                    flag.as<uint8_t>(), c->stream));
        }
        do_partition(c, query, flag.as<uint8_t>(), far, near);
    });
}

} // extern "C"
