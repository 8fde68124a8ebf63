// ltm_api_vote.cpp -- C ABI: scan / map range images, the remove / revert visibility vote, partition, reprojection, RViz images (Removerter.cpp:109-156, 381-593, 675-946; utility.cpp:64-142)
#include "ltm_internal.h"

namespace ltm_detail {

// utility.cpp:222-236 resetRimgSize
Geom geom_for(const ltm_ctx* c, float alpha)
{
    Geom g;
    g.vfov = c->cfg.vfov; g.hfov = c->cfg.hfov;
    g.rows = (int)roundf(c->cfg.vfov * alpha);
    g.cols = (int)roundf(c->cfg.hfov * alpha);
    g.fast = c->fast_math;
    // Error budget of the bounded-error projection in ANGLE: elevation polynomial 6e-7 rad fitted / 1.8e-6 generic, azimuth 4e-7,
    // transform 5e-7, v_rsq 1e-7, plus the reference's own float roundings of the degree / pixel arithmetic (~6e-7 rad equivalent).
    // In pixels the error is proportional to the resolution, and so is the band; never below cull_eps_floor (1e-3 px).
    // ltm_debug_cull_check validates it on the device (tests: 1e8 points incl. points placed on pixel boundaries of every resolution).
    const float ppd = std::max((float)g.rows / g.vfov, (float)g.cols / g.hfov);      // pixels per degree = alpha
    // tools/eps_sweep.py (profiles/r2_cull_eps_sweep*.json): with the band switched down the first exact pixels are missed at
    // 1e-4 * ppd with the generic elevation polynomial and at 5e-5 * ppd with the fitted one, at every resolution; the shipped band
    // is six times that.
    const float scale = c->cull_eps_scale > 0.0f ? c->cull_eps_scale : (c->el_fit ? 3.0e-4f : 6.0e-4f);
    g.cull_eps_px = std::max(c->cull_eps_floor, scale * ppd);
    // elevation of the bounded-error projection (see Geom): the clamp sits one pixel outside the image, at most 2 deg (the fitted range)
    g.el_fit = c->el_fit;
    for (int i = 0; i < 4; ++i) g.el_c[i] = c->el_c[i];
    const double out_deg = std::min(2.0, (double)g.vfov / std::max(g.rows, 1));
    g.el_tclamp = (float)std::tan((0.5 * (double)g.vfov + out_deg) * (3.14159265358979323846 / 180.0));
    return g;
}

// count of set labels given the exclusive scan `pos` of `labels` (n > 0)
size_t scan_total_u8(ltm_ctx* c, const uint8_t* labels, const uint32_t* pos, size_t n)
{
    uint32_t last_pos = 0; uint8_t last = 0;
    if (unsigned char* sp = static_cast<unsigned char*>(small_scratch(c))) {      // both words into the pinned scratch, one wait
        LTM_HIP(hipMemcpyAsync(sp, pos + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
        LTM_HIP(hipMemcpyAsync(sp + 8, labels + (n - 1), 1, hipMemcpyDeviceToHost, c->stream));
        sync(c);
        memcpy(&last_pos, sp, 4); last = sp[8];
    } else {
        LTM_HIP(hipMemcpyAsync(&last_pos, pos + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
        LTM_HIP(hipMemcpyAsync(&last, labels + (n - 1), 1, hipMemcpyDeviceToHost, c->stream));
        sync(c);
    }
    return (size_t)last_pos + (last ? 1 : 0);
}

// --------------------------------------------------------------------------------- vote
void scan_cache_drop(ltm_ctx* c, uint64_t ss_handle)
{
    for (size_t i = 0; i < c->scan_cache.size();) {
        if (ss_handle == 0 || c->scan_cache[i].ss == ss_handle) { c->pool.free(c->scan_cache[i].buf); c->pool.free(c->scan_cache[i].smax); c->pool.free(c->scan_cache[i].qbound); c->scan_cache.erase(c->scan_cache.begin() + i); }
        else ++i;
    }
}

// returns the finished scan range images of keyframes [kb, kb+nb) (cached or freshly computed and then cached)
// qbound_thr >= 0 also returns (in *qbound_out) the squared-range bound image of the range-culled vote kernel for that threshold
const uint32_t* scan_images(ltm_ctx* c, uint64_t ss_handle, const ScanSet& ss, size_t kb, size_t nb, const Geom& g, const uint32_t** smax_out,
                            float qbound_thr = -1.0f, const float** qbound_out = nullptr)
{
    const size_t npx = (size_t)g.rows * g.cols;
    auto with_qbound = [&](ScanImgEntry& e) {
        if (qbound_thr < 0.0f || !qbound_out) return;
        if (!e.qbound || e.q_thr != qbound_thr) {
            if (!e.qbound) { e.qbound = reinterpret_cast<float*>(c->pool.alloc(nb * npx * sizeof(float))); e.bytes += nb * npx * sizeof(float); }
            ProfScope p(c, "vote_scan", 0.0, (double)(nb * npx) * 8);
            LTM_HIP(scan_qbound(e.buf, nb * npx, qbound_thr, e.qbound, c->stream));
            e.q_thr = qbound_thr;
        }
        *qbound_out = e.qbound;
    };
    for (ScanImgEntry& e : c->scan_cache)
        if (e.ss == ss_handle && e.rows == g.rows && e.cols == g.cols && e.kb == kb && e.nb == nb) { e.stamp = ++c->scan_cache_stamp; *smax_out = e.smax; with_qbound(e); return e.buf; }
    const size_t bytes = nb * npx * sizeof(uint32_t);
    size_t held = 0;
    for (const ScanImgEntry& e : c->scan_cache) held += e.bytes;
    while (!c->scan_cache.empty() && held + bytes > c->scan_cache_cap) {      // evict least recently used
        size_t lru = 0;
        for (size_t i = 1; i < c->scan_cache.size(); ++i) if (c->scan_cache[i].stamp < c->scan_cache[lru].stamp) lru = i;
        held -= c->scan_cache[lru].bytes;
        c->pool.free(c->scan_cache[lru].buf);
        c->pool.free(c->scan_cache[lru].smax);
        c->pool.free(c->scan_cache[lru].qbound);
        c->scan_cache.erase(c->scan_cache.begin() + lru);
    }
    uint32_t* buf = reinterpret_cast<uint32_t*>(c->pool.alloc(bytes));
    uint32_t* smax = reinterpret_cast<uint32_t*>(c->pool.alloc(nb * sizeof(uint32_t)));
    const uint64_t first = ss.off[kb], npts = ss.off[kb + nb] - first;
    {
        ProfScope p(c, "vote_scan", (double)npts, (double)npts * 16 + (double)(nb * npx) * 4);
        LTM_HIP(fill_u32(buf, kNoPointBits, nb * npx, c->stream));
        LTM_HIP(hipMemsetAsync(smax, 0, nb * sizeof(uint32_t), c->stream));
        uint64_t longest = 0;
        for (size_t k = kb; k < kb + nb; ++k) longest = std::max<uint64_t>(longest, ss.off[k + 1] - ss.off[k]);
        LTM_HIP(scan_range_images(ss.d, ss.off_dev, kb, nb, first, npts, longest, g, buf, smax, c->stream));
    }
    c->scan_cache.push_back(ScanImgEntry{ss_handle, g.rows, g.cols, kb, nb, buf, smax, bytes, ++c->scan_cache_stamp, nullptr, -1.0f});
    *smax_out = smax;
    with_qbound(c->scan_cache.back());
    return buf;
}

// the finished scan images of keyframes [kb, kb+nb) for SEVERAL shapes, the missing ones computed in one pass over the scans (k_scan_rimg_multi) and cached
void scan_images_prepare(ltm_ctx* c, uint64_t ss_handle, const ScanSet& ss, size_t kb, size_t nb, const std::vector<Geom>& geoms)
{
    std::vector<Geom> todo;
    for (const Geom& g : geoms) {
        bool have = false;
        for (ScanImgEntry& e : c->scan_cache)
            if (e.ss == ss_handle && e.rows == g.rows && e.cols == g.cols && e.kb == kb && e.nb == nb) { e.stamp = ++c->scan_cache_stamp; have = true; }
        for (const Geom& t : todo) have = have || (t.rows == g.rows && t.cols == g.cols);
        if (!have && g.rows > 0 && g.cols > 0) todo.push_back(g);
    }
    if (todo.empty() || !nb) return;
    const uint64_t first = ss.off[kb], npts = ss.off[kb + nb] - first;
    uint64_t longest = 0;
    for (size_t k = kb; k < kb + nb; ++k) longest = std::max<uint64_t>(longest, ss.off[k + 1] - ss.off[k]);
    {
        // The one-pass kernel pays while a scan has about as many points as the smallest shape has pixels (71 x 513 at the reference's 50 degree field of view:
        // 1.3 points per pixel for the lot's os1-64 scans; cascade vote_scan 71 -> 64 ms per step with every session on it).  With several points per pixel its six
        // atomics per point pile up on the same few lines while the point's neighbours do the same: hdl-64e on the street, 3.0 per pixel of the whole image and twice
        // that where the sensor's 27 degrees actually fall, 73 -> 113 ms per step (street 3-res 1057 -> 1100 ms); the re-gridded scans a cascade hands over (3.3) sit
        // on the same side of the line and give up 5 of their 790 ms.  Above `scan_multi_max_density` (2.5) the votes compute their images one shape at a time, as
        // before round 6 (identical images either way).
        size_t min_px = (size_t)todo[0].rows * todo[0].cols;
        for (const Geom& t : todo) min_px = std::min(min_px, (size_t)t.rows * t.cols);
        const double density = (double)(npts / std::max<size_t>(nb, 1)) / (double)std::max<size_t>(min_px, 1);
        if (density > c->scan_multi_max_density) return;
    }
    for (size_t at = 0; at < todo.size(); at += kMaxScanShapes) {
        const int n = (int)std::min<size_t>(kMaxScanShapes, todo.size() - at);
        int rows[kMaxScanShapes], cols[kMaxScanShapes];
        uint32_t *imgs[kMaxScanShapes], *smax[kMaxScanShapes];
        size_t need = 0;
        for (int j = 0; j < n; ++j) { rows[j] = todo[at + j].rows; cols[j] = todo[at + j].cols; need += nb * (size_t)rows[j] * cols[j] * sizeof(uint32_t); }
        size_t held = 0;
        for (const ScanImgEntry& e : c->scan_cache) held += e.bytes;
        while (!c->scan_cache.empty() && held + need > c->scan_cache_cap) {      // evict least recently used (the shapes asked for were stamped above)
            size_t lru = 0;
            for (size_t i = 1; i < c->scan_cache.size(); ++i) if (c->scan_cache[i].stamp < c->scan_cache[lru].stamp) lru = i;
            held -= c->scan_cache[lru].bytes;
            c->pool.free(c->scan_cache[lru].buf); c->pool.free(c->scan_cache[lru].smax); c->pool.free(c->scan_cache[lru].qbound);
            c->scan_cache.erase(c->scan_cache.begin() + lru);
        }
        double px_all = 0;
        for (int j = 0; j < n; ++j) {
            const size_t npx = (size_t)rows[j] * cols[j];
            imgs[j] = reinterpret_cast<uint32_t*>(c->pool.alloc(nb * npx * sizeof(uint32_t)));
            smax[j] = reinterpret_cast<uint32_t*>(c->pool.alloc(nb * sizeof(uint32_t)));
            c->scan_cache.push_back(ScanImgEntry{ss_handle, rows[j], cols[j], kb, nb, imgs[j], smax[j], nb * npx * sizeof(uint32_t), ++c->scan_cache_stamp, nullptr, -1.0f});
            px_all += (double)(nb * npx);
        }
        // algorithmic bytes as SURVEY 8d counts them: every shape is one scan2RangeImg of every scan (16 B per point read, 4 B per pixel written)
        ProfScope p(c, "vote_scan", (double)npts * n, (double)npts * 16 * n + px_all * 4, (double)npts * 16 + px_all * 4);
        for (int j = 0; j < n; ++j) {
            LTM_HIP(fill_u32(imgs[j], kNoPointBits, nb * (size_t)rows[j] * cols[j], c->stream));
            LTM_HIP(hipMemsetAsync(smax[j], 0, nb * sizeof(uint32_t), c->stream));
        }
        LTM_HIP(scan_range_images_multi(ss.d, ss.off_dev, kb, nb, npts, longest, todo[at], n, rows, cols, imgs, smax, c->stream));
    }
}

// first use of an image shape by this context: is the bounded-error projection inside its bounds for it?  (see ltm_ctx::cull_geom_ok)
bool cull_geometry_ok(ltm_ctx* c, const Geom& g, const Poses& ps, size_t kf)
{
    if (!c->cull_selfcheck) return true;
    const std::pair<int, int> key(g.rows, g.cols);
    auto it = c->cull_geom_ok.find(key);
    if (it != c->cull_geom_ok.end()) return it->second;
    const size_t n = (size_t)1 << 20;
    DevBuf pts(c, n * 12), bad(c, 8);
    LTM_HIP(hipMemsetAsync(bad.p, 0, 8, c->stream));
    LTM_HIP(cull_probe_points(g, n, nullptr, pts.as<float>(), c->stream));
    LTM_HIP(cull_check(pts.as<float>(), n, nullptr, &c->B2L, c->b2l_identity, nullptr, g, bad.as<unsigned long long>(), c->stream));
    if (ps.approx_dev && kf < ps.n) {      // the same directions seen through a real keyframe pose: exact transform vs A (p - c)
        const HostMat34 pose = to34(&ps.pose[16 * kf]), inv = to34(&ps.inv[16 * kf]);
        LTM_HIP(cull_probe_points(g, n, &pose, pts.as<float>(), c->stream));
        LTM_HIP(cull_check(pts.as<float>(), n, &inv, &c->B2L, c->b2l_identity, ps.approx_dev + 16 * kf, g, bad.as<unsigned long long>(), c->stream));
    }
    unsigned long long v = 0;
    d2h(c, &v, bad.p, 8);
    const bool ok = v == 0;
    ++c->cull_geoms_checked;
    if (!ok) {
        ++c->cull_geoms_failed;
        fprintf(stderr, "[ltm] the bounded-error projection left its validated bounds for the %d x %d range image (%llu of %zu probe points): exact kernels for this shape\n",
                g.rows, g.cols, v, 2 * n);
    }
    c->cull_geom_ok[key] = ok;
    return ok;
}

// Exact arg-min images of keyframes [kb, kb + nb) (reprojection, ND votes, RViz images): on a large map behind an occlusion cull
// (ltm_k_projection.hip: near pairs first, a coarse maximum of the partial image, far pairs that nearer returns cover completely are dropped).
// The image is bit-identical to the plain launch; below `occlusion_min_pairs` (tile, keyframe) pairs the plain launch is used.
void exact_map_images(ltm_ctx* c, const Cloud& map, const Poses& ps, size_t kb, size_t nb, const Geom& g, uint64_t* img)
{
    if (!map.n || !nb) return;
    const size_t n_tiles = (map.n + 4095) / 4096, n_pairs = n_tiles * nb;
    KernelOpts ko = c->kopts;
    if (ko.map_kernel_variant >= 2 && ps.approx_dev && !cull_geometry_ok(c, g, ps, kb)) ko.map_kernel_variant = 1;      // the pre-filter uses the bounded-error projection
    // (the list-driven launch exists for the block-local arg-min kernel only: LTM_MAP_KERNEL=0/1, the A/B baselines, take the plain launch)
    const bool occl = c->occlusion_cull && ko.map_kernel_variant >= 2 && ps.approx_dev && n_pairs >= c->occlusion_min_pairs && n_pairs < 0xffffffffull;
    if (!occl) {
        HeavyScope hs(c, n_pairs);
        LTM_HIP(map_range_images(map.d, map.n, ps.inv_dev, ps.approx_dev, kb, nb, c->B2L, c->b2l_identity, g, img, hs.stream(), ko));
        hs.done();
        return;
    }
    const size_t rbs = (size_t)g.rows, cbs = ((size_t)g.cols + 7) / 8;
    // scratch of the cull lives with the context (grown on demand): the stage runs dozens of times per step with the same sizes, and
    // taking it from the pool every time changes which blocks the stages around it find there
    const size_t tbytes = scan_temp_bytes(n_pairs);
    const size_t dwords = (rbs + 31) / 32;         // dirty-row bitmap of the incremental coarse maximum
    const bool subtiles = c->occlusion_subtile != 0;      // second look at the 1024-point quarters of a live tile (round 6)
    const size_t need = n_tiles * 24 + n_pairs * (1 + 1 + 4 + 4) + 64 + nb * rbs * cbs * 4 + nb * dwords * 4 + tbytes + (subtiles ? n_tiles * 96 + n_pairs + 64 : 0) + 12 * 256;
    if (c->occl_scratch_bytes < need) {
        if (c->occl_scratch) { sync(c); c->pool.free(c->occl_scratch); c->occl_scratch = nullptr; c->occl_scratch_bytes = 0; }   // (the alloc below may throw)
        c->occl_scratch = c->pool.alloc(need + need / 4);
        c->occl_scratch_bytes = need + need / 4;
    }
    char* base = static_cast<char*>(c->occl_scratch);
    auto carve = [&](size_t bytes) { char* p = base; base += (bytes + 255) & ~(size_t)255; return p; };
    float* tb = reinterpret_cast<float*>(carve(n_tiles * 24));
    uint8_t* done = reinterpret_cast<uint8_t*>(carve(n_pairs));
    uint8_t* flags = reinterpret_cast<uint8_t*>(carve(n_pairs));
    uint32_t* pos = reinterpret_cast<uint32_t*>(carve(n_pairs * 4));
    uint32_t* list = reinterpret_cast<uint32_t*>(carve(n_pairs * 4));
    uint32_t* count = reinterpret_cast<uint32_t*>(carve(64));
    uint32_t* cmax = reinterpret_cast<uint32_t*>(carve(nb * rbs * cbs * 4));
    uint32_t* dirty = reinterpret_cast<uint32_t*>(carve(nb * dwords * 4));
    void* temp = carve(tbytes);
    float* tbq = subtiles ? reinterpret_cast<float*>(carve(n_tiles * 96)) : nullptr;
    uint8_t* submask = subtiles ? reinterpret_cast<uint8_t*>(carve(n_pairs)) : nullptr;
    unsigned long long* sub_stats = subtiles ? reinterpret_cast<unsigned long long*>(carve(64)) : nullptr;
    LTM_HIP(tile_bounds(map.d, map.n, tb, c->stream));
    if (subtiles) {
        LTM_HIP(subtile_bounds(map.d, map.n, tbq, c->stream));
        LTM_HIP(hipMemsetAsync(sub_stats, 0, 16, c->stream));
    }
    LTM_HIP(hipMemsetAsync(done, 0, n_pairs, c->stream));
    if (dirty) {      // rows no projection has touched yet hold empty pixels: their coarse maximum is the empty range (10000 m, utility.h:93)
        LTM_HIP(hipMemsetAsync(dirty, 0, nb * dwords * 4, c->stream));
        LTM_HIP(fill_u32(cmax, 0x461c4000u, nb * rbs * cbs, c->stream));
    }
    float r_lo = 0.0f, r_hi = c->occlusion_r_near;
    size_t n_done = 0, n_proj = 0;
    for (int shell = 0; shell < 12; ++shell) {
        const bool last = shell == 11 || r_hi > 1.0e4f;
        if (last) r_hi = 3.0e38f;
        LTM_HIP(occlusion_shell_pairs(ps.approx_dev, kb, nb, tb, n_tiles, g, r_lo, r_hi, img, shell > 0, cmax, done, flags, pos, list, count, temp, tbytes, c->stream, dirty,
                                      tbq, submask, c->occlusion_stats_on ? sub_stats : nullptr));
        uint32_t n_live = 0;
        d2h(c, &n_live, count, 4);
        {
            HeavyScope hs(c, n_live);
            LTM_HIP(map_range_images_pairs(map.d, map.n, ps.inv_dev, ps.approx_dev, kb, nb, c->B2L, c->b2l_identity, g, img, list, n_live, hs.stream(), ko, submask));
            hs.done();
        }
        n_proj += n_live;
        if (shell == 0) c->occl_near += n_live;
        if (last) break;
        r_lo = r_hi; r_hi *= 2.0f;
    }
    (void)n_done;
    if (subtiles && c->occlusion_stats_on) {
        unsigned long long st[2] = {0, 0};
        d2h(c, st, sub_stats, 16);
        c->occl_quarters += st[0]; c->occl_quarters_live += st[1];
    }
    c->occl_pairs += n_pairs; c->occl_far_live += n_proj;
}

void do_vote(ltm_ctx* c, const Cloud& map, uint64_t ss_handle, const ScanSet& ss, const Poses& ps, size_t kf_begin, size_t kf_end, float alpha, float thr,
             int mode, uint8_t* labels_dev)
{
    LTM_REQUIRE(ss.nkf() == ps.n, "scan set and poses have different keyframe counts");
    LTM_REQUIRE(kf_begin <= kf_end && kf_end <= ps.n, "keyframe range out of bounds");
    LTM_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (scan-map) or 1 (map-scan)");
    LTM_REQUIRE(map.n < 0xffffffffull, "map too large for 32-bit point indices");
    if (kf_begin == kf_end || map.n == 0) return;
    const Geom g = geom_for(c, alpha);
    LTM_REQUIRE(g.rows > 0 && g.cols > 0, "empty range image");
    const size_t npx = (size_t)g.rows * g.cols;
    const size_t KB = std::min(c->kf_batch, kf_end - kf_begin);
    DevBuf map_img(c, KB * npx * sizeof(uint64_t));
    const size_t n_tiles = (map.n + 4095) / 4096;
    DevBuf tb(c, n_tiles * 6 * sizeof(float));
    if (mode == 0) LTM_HIP(tile_bounds(map.d, map.n, tb.as<float>(), c->stream));
    for (size_t kb = kf_begin; kb < kf_end; kb += KB) {
        const size_t nb = std::min(KB, kf_end - kb);
        const uint32_t* smax = nullptr;
        const float* qbound = nullptr;
        const bool cull = mode == 0 && (c->kopts.vote_cull != 0) && ps.approx_dev && cull_geometry_ok(c, g, ps, kb);
        const uint32_t* scan_img = scan_images(c, ss_handle, ss, kb, nb, g, &smax, cull ? thr : -1.0f, &qbound);
        {
            ProfScope p(c, "vote_fill", (double)(nb * npx), (double)(nb * npx * 8));
            LTM_HIP(fill_u64(map_img.as<uint64_t>(), (uint64_t)kNoPointBits << 32, nb * npx, c->stream));
        }
        {
            // class name = kernel: k_vote_map_cull for mode 0 (when enabled), k_map_rimg_blockmin otherwise
            // algorithmic bytes = map tiles read + images written.  Tiles that the whole-tile range cull drops are never read, so
            // (measurement only, when profiling is on) they are counted by the same predicate and left out.
            // compulsory bytes of the launch as designed: the map once, per keyframe the range|index image written (8 B) and, culled form, the bound image read (4 B)
            double pts = (double)map.n * nb, bytes = 16.0 * pts + (double)nb * 8.0 * npx, bytes_c = 16.0 * map.n + (double)nb * (cull ? 12.0 : 8.0) * npx;
            const bool count_live = cull && c->prof_on && (c->kopts.tile_cull != 0) && smax && c->pending_live.size() < (size_t)kLiveSlots;
            if (count_live) {      // no host round trip here: the count is read when the profile is collected
                if (!c->live_counts) LTM_HIP(hipMalloc(reinterpret_cast<void**>(&c->live_counts), sizeof(unsigned long long) * kLiveSlots));
                const int slot = (int)c->pending_live.size();
                LTM_HIP(hipMemsetAsync(c->live_counts + slot, 0, sizeof(unsigned long long), c->stream));
                LTM_HIP(count_live_tiles(ps.approx_dev, kb, nb, tb.as<float>(), n_tiles, smax, thr, c->live_counts + slot, c->stream));
                c->pending_live.push_back(PendingLive{prof_class(c, "vote_map_cull"), slot, pts, (double)nb * 8.0 * npx, (double)map.n, (double)nb * 12.0 * npx});
                pts = 0.0; bytes = 0.0; bytes_c = 0.0;      // added by prof_collect
            }
            ProfScope p(c, cull ? "vote_map_cull" : "vote_map_exact", pts, bytes, bytes_c);
            if (cull) {
                HeavyScope hs(c, n_tiles * nb);
                LTM_HIP(vote_map_range_images(map.d, map.n, ps.inv_dev, ps.approx_dev, kb, nb, c->B2L, c->b2l_identity, g, qbound, tb.as<float>(), smax, thr, mode,
                                              map_img.as<uint64_t>(), hs.stream(), c->kopts));
                hs.done();
            } else exact_map_images(c, map, ps, kb, nb, g, map_img.as<uint64_t>());
        }
        {
            ProfScope p(c, "vote_compare", (double)(nb * npx), (double)(nb * npx) * 12 + (double)nb * map.n / 8.0);
            LTM_HIP(compare_and_flag(scan_img, map_img.as<uint64_t>(), nb * npx, thr, mode, labels_dev, c->stream));
        }
    }
}

void do_partition(ltm_ctx* c, const Cloud& map, const uint8_t* labels, ltm_cloud* kept, ltm_cloud* flagged)
{
    const size_t n = map.n;
    if (n == 0) {
        float4* d;
        if (kept) *kept = alloc_cloud(c, 0, &d);
        if (flagged) *flagged = alloc_cloud(c, 0, &d);
        return;
    }
    DevBuf pos(c, n * sizeof(uint32_t));
    const size_t tb = scan_temp_bytes(n);
    DevBuf temp(c, tb);
    ProfScope p(c, "partition", (double)n, (double)n * (16 + 1 + 4 + 4 + 16));
    LTM_HIP(exclusive_scan_u8(labels, pos.as<uint32_t>(), n, temp.p, tb, c->stream));
    const size_t nf = scan_total_u8(c, labels, pos.as<uint32_t>(), n);
    float4 *dk = nullptr, *df = nullptr;
    ltm_cloud hk = 0, hf = 0;
    if (kept) hk = alloc_cloud(c, n - nf, &dk);
    if (flagged) hf = alloc_cloud(c, nf, &df);
    LTM_HIP(partition_scatter(map.d, labels, pos.as<uint32_t>(), n, dk, df, c->stream));
    if (kept) { *kept = hk; inherit_frame(c, hk, map); }
    if (flagged) { *flagged = hf; inherit_frame(c, hf, map); }
}


} // namespace ltm_detail

// =========================================================================================== C ABI
extern "C" {

int ltm_scanset_prepare_range_images(ltm_ctx* c, ltm_scanset hs, size_t kf_begin, size_t kf_end, const float* alphas, size_t n_alphas)
{
    return guarded(c, [&] {
        LTM_REQUIRE(alphas || n_alphas == 0, "null argument");
        const ScanSet& ss = get_ss(c, hs);
        LTM_REQUIRE(kf_begin <= kf_end && kf_end <= ss.nkf(), "keyframe range out of bounds");
        if (kf_begin == kf_end || n_alphas == 0) return;
        std::vector<Geom> geoms;
        for (size_t i = 0; i < n_alphas; ++i) geoms.push_back(geom_for(c, alphas[i]));
        const size_t KB = std::min(c->kf_batch, kf_end - kf_begin);      // the batches ltm_visibility_vote will ask for
        for (size_t kb = kf_begin; kb < kf_end; kb += KB) scan_images_prepare(c, hs, ss, kb, std::min(KB, kf_end - kb), geoms);
    });
}

int ltm_visibility_vote(ltm_ctx* c, ltm_cloud hmap, ltm_scanset hs, ltm_poses hp, size_t kf_begin, size_t kf_end, float alpha,
                        float thr, int mode, uint8_t* labels_dev)
{
    return guarded(c, [&] {
        LTM_REQUIRE(labels_dev, "null labels buffer");
        do_vote(c, get_cloud(c, hmap), hs, get_ss(c, hs), get_poses(c, hp), kf_begin, kf_end, alpha, thr, mode, labels_dev);
        sync(c);   // the caller may hand labels_dev to a collective on another stream
    });
}

int ltm_partition_by_labels(ltm_ctx* c, ltm_cloud hmap, const uint8_t* labels_dev, ltm_cloud* kept, ltm_cloud* flagged)
{
    return guarded(c, [&] {
        LTM_REQUIRE(labels_dev || get_cloud(c, hmap).n == 0, "null labels buffer");
        const Cloud map = get_cloud(c, hmap);
        do_partition(c, map, labels_dev, kept, flagged);
        sync(c);   // labels_dev is the caller's (e.g. a torch tensor that may be recycled on another stream as soon as we return)
    });
}

int ltm_visibility_partition(ltm_ctx* c, ltm_cloud hmap, ltm_scanset hs, ltm_poses hp, float alpha, float thr, int mode,
                             ltm_cloud* kept, ltm_cloud* flagged, uint8_t* host_labels)
{
    return guarded(c, [&] {
        const Cloud map = get_cloud(c, hmap);
        const Poses& p = get_poses(c, hp);
        DevBuf labels(c, std::max<size_t>(map.n, 1));
        LTM_HIP(hipMemsetAsync(labels.p, 0, std::max<size_t>(map.n, 1), c->stream));
        do_vote(c, map, hs, get_ss(c, hs), p, 0, p.n, alpha, thr, mode, labels.as<uint8_t>());
        if (host_labels) d2h(c, host_labels, labels.p, map.n);
        do_partition(c, map, labels.as<uint8_t>(), kept, flagged);
    });
}

int ltm_reproject(ltm_ctx* c, ltm_cloud hmap, ltm_poses hp, size_t kf_begin, size_t kf_end, float alpha, ltm_scanset* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        const Cloud map = get_cloud(c, hmap);
        const Poses& p = get_poses(c, hp);
        LTM_REQUIRE(kf_begin <= kf_end && kf_end <= p.n, "keyframe range out of bounds");
        LTM_REQUIRE(map.n < 0xffffffffull, "map too large for 32-bit point indices");
        const Geom g = geom_for(c, alpha);
        LTM_REQUIRE(g.rows > 0 && g.cols > 0, "empty range image");
        const size_t npx = (size_t)g.rows * g.cols;
        const size_t nk = kf_end - kf_begin;
        std::vector<uint64_t> off(nk + 1, 0);
        struct Piece { float4* d; size_t n; };
        std::vector<Piece> pieces;
        if (nk && map.n) {
            const size_t KB = std::min(c->kf_batch, nk);
            DevBuf img(c, KB * npx * 8), pos(c, KB * npx * 4);
            const size_t tb = scan_temp_bytes(KB * npx);
            DevBuf temp(c, tb), bout(c, (KB + 1) * 4);
            for (size_t kb = kf_begin; kb < kf_end; kb += KB) {
                const size_t nb = std::min(KB, kf_end - kb);
                LTM_HIP(fill_u64(img.as<uint64_t>(), (uint64_t)kNoPointBits << 32, nb * npx, c->stream));
                {
                    ProfScope ps(c, "reproject_map", (double)map.n * nb, (double)nb * (16.0 * map.n + 8.0 * npx), 16.0 * map.n + (double)nb * 8.0 * npx);
                    exact_map_images(c, map, p, kb, nb, g, img.as<uint64_t>());
                }
                ProfScope ps(c, "reproject_gather", (double)(nb * npx), (double)(nb * npx) * 12);
                LTM_HIP(exclusive_scan_img_valid(img.as<uint64_t>(), pos.as<uint32_t>(), nb * npx, temp.p, tb, c->stream));
                // per-keyframe boundaries = scan value at each image start, the total = the scan past the end: one small array, one round trip
                LTM_HIP(image_bounds(pos.as<uint32_t>(), img.as<uint64_t>(), npx, nb, bout.as<uint32_t>(), c->stream));
                std::vector<uint32_t> b(nb + 1);
                d2h(c, b.data(), bout.p, (nb + 1) * 4);
                const size_t total = b[nb];
                const uint64_t base = off[kb - kf_begin];
                for (size_t j = 1; j <= nb; ++j) off[kb - kf_begin + j] = base + b[j];
                float4* d = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(total, 1) * 16));
                LTM_HIP(reproject_gather(img.as<uint64_t>(), pos.as<uint32_t>(), npx, nb, map.d, p.inv_dev, kb, c->B2L, c->b2l_identity, d, c->stream));
                pieces.push_back(Piece{d, total});
            }
        }
        float4* d = nullptr;
        if (pieces.size() == 1) d = pieces[0].d;
        else {
            d = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(off[nk], 1) * 16));
            size_t at = 0;
            for (Piece& pc : pieces) { d2d(c, d + at, pc.d, pc.n * 16); at += pc.n; }
            sync(c);
            for (Piece& pc : pieces) c->pool.free(pc.d);
        }
        *out = new_scanset(c, d, std::move(off));
    });
}

// ------------------------------------------------------------------------- debug / parity
int ltm_debug_range_image(ltm_ctx* c, ltm_cloud h, const double* T1, const double* T2, float alpha, float* rimg, int32_t* ptidx)
{
    return guarded(c, [&] {
        LTM_REQUIRE(rimg, "null argument");
        const Cloud cl = get_cloud(c, h);
        const Geom g = geom_for(c, alpha);
        const size_t npx = (size_t)g.rows * g.cols;
        DevBuf img(c, npx * 8), r(c, npx * 4), ix(c, npx * 4);
        LTM_HIP(fill_u64(img.as<uint64_t>(), (uint64_t)kNoPointBits << 32, npx, c->stream));
        HostMat34 a, b;
        if (T1) a = to34(T1);
        if (T2) b = to34(T2);
        LTM_HIP(single_range_image(cl.d, cl.n, T1 ? &a : nullptr, T2 ? &b : nullptr, g, img.as<uint64_t>(), c->stream));
        LTM_HIP(decode_image(img.as<uint64_t>(), npx, r.as<float>(), ix.as<int32_t>(), c->stream));
        d2h(c, rimg, r.p, npx * 4);
        if (ptidx) d2h(c, ptidx, ix.p, npx * 4);
    });
}

// cv::COLORMAP_JET as OpenCV builds it: 256 float samples of the piecewise-linear jet ramps, times 255, round half to even
static void jet_lut_bgr(uint8_t* lut)
{
    for (int i = 0; i < 256; ++i) {
        const double x = (double)i / 255.0;
        const double bgr[3] = {std::min(4.0 * x + 0.5, 2.5 - 4.0 * x), std::min(4.0 * x - 0.5, 3.5 - 4.0 * x), std::min(4.0 * x - 1.5, 4.5 - 4.0 * x)};
        for (int ch = 0; ch < 3; ++ch) {
            const float sample = (float)std::min(1.0, std::max(0.0, bgr[ch]));
            lut[3 * i + ch] = (uint8_t)std::min(255l, std::max(0l, std::lrint((double)(sample * 255.0f))));
        }
    }
}

int ltm_debug_viz_images(ltm_ctx* c, ltm_cloud hmap, ltm_scanset hscans, ltm_poses hposes, size_t kf, float alpha, int mode,
                         float range_min, float range_max, float diff_min, float diff_max,
                         uint8_t* scan_bgr, uint8_t* map_bgr, uint8_t* diff_bgr, uint8_t* ptidx_bgr)
{
    return guarded(c, [&] {
        LTM_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (scan-map) or 1 (map-scan)");
        LTM_REQUIRE(range_max != range_min && diff_max != diff_min, "empty colour axis");
        const Cloud map = get_cloud(c, hmap);
        const ScanSet& ss = get_ss(c, hscans);
        const Poses& ps = get_poses(c, hposes);
        LTM_REQUIRE(ss.nkf() == ps.n && kf < ps.n, "keyframe out of range");
        LTM_REQUIRE(map.n < 0x7fffffffull, "map too large for an int32 index image");
        const Geom g = geom_for(c, alpha);
        const size_t npx = (size_t)g.rows * g.cols;
        const uint32_t* smax = nullptr;
        const uint32_t* scan_img = scan_images(c, hscans, ss, kf, 1, g, &smax);          // scan2RangeImg
        DevBuf img(c, npx * 8), mr(c, npx * 4), mi(c, npx * 4), df(c, npx * 4), out(c, npx * 3), lutd(c, 768);
        LTM_HIP(fill_u64(img.as<uint64_t>(), (uint64_t)kNoPointBits << 32, npx, c->stream));
        if (map.n)                                                                         // transformGlobalMapToLocal + map2RangeImg (exact image)
            LTM_HIP(map_range_images(map.d, map.n, ps.inv_dev, ps.approx_dev, kf, 1, c->B2L, c->b2l_identity, g, img.as<uint64_t>(), c->stream, c->kopts));
        LTM_HIP(decode_image(img.as<uint64_t>(), npx, mr.as<float>(), mi.as<int32_t>(), c->stream));
        uint8_t lut[768];
        jet_lut_bgr(lut);
        h2d(c, lutd.p, lut, sizeof lut);
        // cv::MatExpr folds 255 * (src - min) / (max - min) into src * a + b with a = 255 * (1/(max-min)), b = -255*min * (1/(max-min))
        auto axis = [](float lo, float hi, double* a, double* b) {
            const double inv = 1.0 / (double)(float)(hi - lo);
            *a = 255.0 * inv; *b = -((double)lo * 255.0) * inv;
        };
        double a, b;
        auto emit_f32 = [&](const float* src, uint8_t* host) {
            if (!host) return;
            LTM_HIP(viz_colormap_f32(src, npx, (float)a, (float)b, lutd.as<uint8_t>(), out.as<uint8_t>(), c->stream));
            d2h(c, host, out.p, npx * 3);
        };
        axis(range_min, range_max, &a, &b);
        emit_f32(reinterpret_cast<const float*>(scan_img), scan_bgr);
        emit_f32(mr.as<float>(), map_bgr);
        if (diff_bgr) {
            LTM_HIP(viz_diff(scan_img, mr.as<float>(), npx, mode, df.as<float>(), c->stream));
            axis(diff_min, diff_max, &a, &b);
            emit_f32(df.as<float>(), diff_bgr);
        }
        if (ptidx_bgr) {
            LTM_REQUIRE(map.n > 0, "index image of an empty map has no colour axis");
            axis(0.0f, (float)map.n, &a, &b);                                              // Removerter.cpp:583
            LTM_HIP(viz_colormap_i32(mi.as<int32_t>(), npx, a, b, lutd.as<uint8_t>(), out.as<uint8_t>(), c->stream));
            d2h(c, ptidx_bgr, out.p, npx * 3);
        }
    });
}

int ltm_debug_project(ltm_ctx* c, const float* xyz, size_t n, float alpha, float* sph, int32_t* rc)
{
    return guarded(c, [&] {
        LTM_REQUIRE((xyz && sph && rc) || n == 0, "null argument");
        if (!n) return;
        const Geom g = geom_for(c, alpha);
        DevBuf in(c, n * 12), o1(c, n * 12), o2(c, n * 8);
        h2d(c, in.p, xyz, n * 12);
        LTM_HIP(debug_project(in.as<float>(), n, g, o1.as<float>(), o2.as<int32_t>(), c->stream));
        d2h(c, sph, o1.p, n * 12);
        d2h(c, rc, o2.p, n * 8);
    });
}

int ltm_debug_cull_check(ltm_ctx* c, const float* xyz, size_t n, const double* inv_pose16, float alpha, uint64_t* violations)
{
    return guarded(c, [&] {
        LTM_REQUIRE((xyz && violations) || n == 0, "null argument");
        if (violations) *violations = 0;
        if (!n) return;
        const Geom g = geom_for(c, alpha);
        DevBuf in(c, n * 12), bad(c, 8), apd(c, 64);
        h2d(c, in.p, xyz, n * 12);
        LTM_HIP(hipMemsetAsync(bad.p, 0, 8, c->stream));
        HostMat34 T{};
        if (inv_pose16) {
            float ap[16];
            double b2l16[16] = {0};
            memcpy(b2l16, c->B2L.m, 12 * sizeof(double)); b2l16[15] = 1.0;
            approx_pose(b2l16, inv_pose16, ap);
            h2d(c, apd.p, ap, 64);
            T = to34(inv_pose16);
        }
        LTM_HIP(cull_check(in.as<float>(), n, inv_pose16 ? &T : nullptr, &c->B2L, c->b2l_identity, apd.as<float>(), g, bad.as<unsigned long long>(), c->stream));
        unsigned long long v = 0;
        d2h(c, &v, bad.p, 8);
        *violations = v;
    });
}

int ltm_debug_cull_validation(ltm_ctx* c, uint64_t* shapes_checked, uint64_t* shapes_failed)
{
    return guarded(c, [&] { if (shapes_checked) *shapes_checked = c->cull_geoms_checked; if (shapes_failed) *shapes_failed = c->cull_geoms_failed; });
}

int ltm_debug_cull_stats(ltm_ctx* c, uint64_t* survivors, uint64_t* points, int reset)
{
    return guarded(c, [&] {
        unsigned long long v[2] = {0, 0};
        LTM_HIP(cull_stats(v, reset, c->stream, c->kopts.stats_blockmin));
        if (survivors) *survivors = v[0];
        if (points) *points = v[1];
    });
}

int ltm_debug_occlusion_stats(ltm_ctx* c, uint64_t* pairs, uint64_t* first_shell, uint64_t* projected, int reset)
{
    return guarded(c, [&] {
        if (pairs) *pairs = c->occl_pairs;
        if (first_shell) *first_shell = c->occl_near;
        if (projected) *projected = c->occl_far_live;
        if (reset) { c->occl_pairs = 0; c->occl_near = 0; c->occl_far_live = 0; }
    });
}

} // extern "C"
