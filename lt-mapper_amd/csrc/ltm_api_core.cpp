// ltm_api_core.cpp -- C ABI of libltm_hip.so (include/ltm.h): context, lanes, clouds, scan sets, poses, pipelined upload, asynchronous fetch, profile
#include "ltm_internal.h"

namespace ltm_detail {

// General 4x4 inverse in double with the operation order of Eigen 3.3.7's Matrix4d::inverse() in an SSE2 build -- what the reference
// calls for every pose (Session.cpp:109-110) and for the extrinsic (RosParamServer.cpp:29-30); restated from knowledge of its
// structure (PARITY UNPINNED): the 16 doubles of the column-major matrix are read in memory order as the four 2x2
// blocks A B / C D of N = M^T, the inverse is assembled from the adjugate products A#B and D#C ("divide and conquer" over the
// blocks), det = |A||D| + |B||C| - trace(A#B D#C), every product and sum rounded on its own (no FMA).  m, inv: row-major.
bool inverse4x4(const double* m, double* inv)
{
    double A[2][2], B[2][2], C[2][2], D[2][2];
    for (int r = 0; r < 2; ++r)
        for (int k = 0; k < 2; ++k) {        // N(r, k) = m(k, r)
            A[r][k] = m[4 * k + r]; B[r][k] = m[4 * (k + 2) + r];
            C[r][k] = m[4 * k + r + 2]; D[r][k] = m[4 * (k + 2) + r + 2];
        }
    const double dA = A[0][0] * A[1][1] - A[0][1] * A[1][0], dB = B[0][0] * B[1][1] - B[0][1] * B[1][0];
    const double dC = C[0][0] * C[1][1] - C[0][1] * C[1][0], dD = D[0][0] * D[1][1] - D[0][1] * D[1][0];
    double AB[2][2], DC[2][2];
    for (int j = 0; j < 2; ++j) {
        AB[0][j] = B[0][j] * A[1][1] - B[1][j] * A[0][1]; AB[1][j] = B[1][j] * A[0][0] - B[0][j] * A[1][0];
        DC[0][j] = C[0][j] * D[1][1] - C[1][j] * D[0][1]; DC[1][j] = C[1][j] * D[0][0] - C[0][j] * D[1][0];
    }
    const double tr = (AB[0][0] * DC[0][0] + AB[1][0] * DC[0][1]) + (AB[0][1] * DC[1][0] + AB[1][1] * DC[1][1]);
    double iA[2][2], iB[2][2], iC[2][2], iD[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            iD[i][j] = D[i][j] * dA - (AB[0][j] * C[i][0] + AB[1][j] * C[i][1]);
            iA[i][j] = A[i][j] * dD - (DC[0][j] * B[i][0] + DC[1][j] * B[i][1]);
        }
    for (int i = 0; i < 2; ++i) {
        iB[i][0] = D[i][0] * AB[1][1] - D[i][1] * AB[1][0]; iB[i][1] = D[i][1] * AB[0][0] - D[i][0] * AB[0][1];
        iC[i][0] = A[i][0] * DC[1][1] - A[i][1] * DC[1][0]; iC[i][1] = A[i][1] * DC[0][0] - A[i][0] * DC[0][1];
    }
    const double det = (dA * dD + dB * dC) - tr;
    if (det == 0.0 || det != det) return false;      // Eigen would return inf / NaN entries; a singular pose is an error here
    const double rd = 1.0 / det;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) { iB[i][j] = C[i][j] * dB - iB[i][j]; iC[i][j] = B[i][j] * dC - iC[i][j]; }
    double R[4][4];      // inverse of N, row-major: the blocks' adjugates times +-1/det
    auto put = [&](const double X[2][2], int r, int c) {
        R[r][c] = X[1][1] * rd; R[r][c + 1] = X[0][1] * -rd; R[r + 1][c] = X[1][0] * -rd; R[r + 1][c + 1] = X[0][0] * rd;
    };
    put(iA, 0, 0); put(iB, 0, 2); put(iC, 2, 0); put(iD, 2, 2);
    for (int r = 0; r < 4; ++r) for (int k = 0; k < 4; ++k) inv[4 * r + k] = R[k][r];
    return true;
}

// Bounded-error form of "base2lidar * inverse pose" for the cull test: p_local ~= A (p - c), c = sensor position in
// the map frame as a float-float pair.  out[16] = {A row-major, c_hi, c_lo, ok}.
void approx_pose(const double* b2l16, const double* inv16, float* out)
{
    double T[12];
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 4; ++k) {
            double v = (k == 3) ? b2l16[4 * r + 3] : 0.0;
            for (int j = 0; j < 3; ++j) v += b2l16[4 * r + j] * inv16[4 * j + k];
            T[4 * r + k] = v;
        }
    const double a = T[0], b = T[1], c3 = T[2], d = T[4], e = T[5], f = T[6], g = T[8], h = T[9], i = T[10];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c3 * (d * h - e * g);
    for (int k = 0; k < 16; ++k) out[k] = 0.0f;
    if (!(std::fabs(det) > 1e-12) || !std::isfinite(det)) return;          // ok stays 0: every point takes the exact path
    // The exact path rounds to float BETWEEN the inverse pose and base->lidar (utility.cpp:70-71), so its error relative to the
    // range grows with |lever arm| / range; the bounds ltm_debug_cull_check validates (3e-3 px, 3e-6 r) assume a sensor mounted
    // within a few metres of the pose base.  A larger extrinsic translation sends every point down the exact path.
    if (std::sqrt(b2l16[3] * b2l16[3] + b2l16[7] * b2l16[7] + b2l16[11] * b2l16[11]) > 10.0) return;
    const double inv[9] = {(e * i - f * h) / det, (c3 * h - b * i) / det, (b * f - c3 * e) / det,
                           (f * g - d * i) / det, (a * i - c3 * g) / det, (c3 * d - a * f) / det,
                           (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
    const double t[3] = {T[3], T[7], T[11]};
    double c_lo[3];
    for (int r = 0; r < 3; ++r) {
        const double cr = -(inv[3 * r] * t[0] + inv[3 * r + 1] * t[1] + inv[3 * r + 2] * t[2]);
        out[9 + r] = (float)cr;
        c_lo[r] = cr - (double)out[9 + r];
        for (int k = 0; k < 3; ++k) out[3 * r + k] = (float)T[4 * r + k];
    }
    // A (p - c) = A (p - c_hi) - A c_lo: the second term is a per-keyframe constant, folded into the first FMA of each row
    for (int r = 0; r < 3; ++r)
        out[12 + r] = (float)-((double)out[3 * r] * c_lo[0] + (double)out[3 * r + 1] * c_lo[1] + (double)out[3 * r + 2] * c_lo[2]);
    // out[15] doubles as a lower bound of the smallest singular value of A (Gershgorin on A^T A, rounded down): the tile
    // range cull needs |A v| >= smin |v|.  Poses from 6-significant-digit text are rotations up to ~1e-6.
    double gmin = 1e300;
    for (int a2 = 0; a2 < 3; ++a2) {
        double diag = 0, off = 0;
        for (int b2 = 0; b2 < 3; ++b2) {
            double g2 = 0;
            for (int r = 0; r < 3; ++r) g2 += T[4 * r + a2] * T[4 * r + b2];
            if (a2 == b2) diag = g2; else off += std::fabs(g2);
        }
        gmin = std::min(gmin, diag - off);
    }
    double smin = gmin > 0.25 ? std::sqrt(gmin) * (1.0 - 1e-6) : 0.0;
    // The occlusion cull of the exact-image kernel (sphere_rect) treats the pose as RIGID -- a tile's bounding sphere keeps its radius in
    // the sensor frame -- and gates on this value being > 0.999.  A lower bound alone does not exclude a scale or shear > 1 (the reference
    // accepts any 4x4 pose): bound the largest singular value too (Gershgorin, upper end) and, if it can exceed 1.001, report at
    // most 0.99 -- still a valid lower bound for the tile range cull, but "not rigid" for the occlusion cull.
    double gmax = 0.0;
    for (int a2 = 0; a2 < 3; ++a2) {
        double row = 0;
        for (int b2 = 0; b2 < 3; ++b2) {
            double g2 = 0;
            for (int r = 0; r < 3; ++r) g2 += T[4 * r + a2] * T[4 * r + b2];
            row += std::fabs(g2);
        }
        gmax = std::max(gmax, row);
    }
    if (!(std::sqrt(gmax) * (1.0 + 1e-6) < 1.001)) smin = std::min(smin, 0.99);
    out[15] = smin > 0.5 ? (float)std::nextafter((float)smin, 0.0f) : 1.0e-30f;   // tiny = usable transform, no tile cull
}

// Degree-3 polynomial in u = t^2 with atan(t) ~ t * p(u) on t in [0, tmax], minimising the largest ANGLE error |t p(t^2) - atan t|
// (Lawson's iteratively re-weighted least squares on a grid: 4 unknowns, converges to the minimax fit), coefficients rounded to
// binary32.  Returns the largest error [rad] of the binary32 Horner evaluation the kernels use (fmaf = v_fma_f32), scanned on a
// dense grid against atan in double.
static double fit_elevation_poly(double tmax, float c_out[4])
{
    const int N = 4000;
    std::vector<double> t(N), y(N), w(N, 1.0 / N), B(4 * (size_t)N);
    for (int i = 0; i < N; ++i) {
        t[i] = tmax * (i + 1) / N;
        y[i] = std::atan(t[i]);
        const double u = (t[i] / tmax) * (t[i] / tmax);        // scaled so that the normal equations stay well conditioned
        double b = t[i];
        for (int k = 0; k < 4; ++k) { B[4 * (size_t)i + k] = b; b *= u; }
    }
    double c[4] = {1.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < 200; ++it) {
        double M[4][5] = {};
        for (int i = 0; i < N; ++i)
            for (int r = 0; r < 4; ++r) {
                const double wb = w[i] * B[4 * (size_t)i + r];
                for (int k = 0; k < 4; ++k) M[r][k] += wb * B[4 * (size_t)i + k];
                M[r][4] += wb * y[i];
            }
        for (int col = 0; col < 4; ++col) {                      // Gauss-Jordan with partial pivoting
            int piv = col;
            for (int r = col + 1; r < 4; ++r) if (std::fabs(M[r][col]) > std::fabs(M[piv][col])) piv = r;
            for (int k = 0; k < 5; ++k) std::swap(M[col][k], M[piv][k]);
            if (M[col][col] == 0.0) return 1.0;
            for (int r = 0; r < 4; ++r) {
                if (r == col) continue;
                const double f = M[r][col] / M[col][col];
                for (int k = col; k < 5; ++k) M[r][k] -= f * M[col][k];
            }
        }
        for (int k = 0; k < 4; ++k) c[k] = M[k][4] / M[k][k];
        double sum = 0.0;
        for (int i = 0; i < N; ++i) {
            double p = 0.0;
            for (int k = 0; k < 4; ++k) p += c[k] * B[4 * (size_t)i + k];
            w[i] *= std::fabs(p - y[i]);
            sum += w[i];
        }
        if (!(sum > 0.0)) break;
        for (int i = 0; i < N; ++i) w[i] /= sum;
    }
    double scale = 1.0;
    for (int k = 0; k < 4; ++k) { c_out[k] = (float)(c[k] * scale); scale /= tmax * tmax; }
    double worst = 0.0;
    const int G = 200000;
    for (int i = 0; i <= G; ++i) {
        const float tf = (float)(tmax * i / G), uu = tf * tf;
        const float pf = tf * std::fmaf(std::fmaf(std::fmaf(c_out[3], uu, c_out[2]), uu, c_out[1]), uu, c_out[0]);
        worst = std::max(worst, std::fabs((double)pf - std::atan((double)tf)));
    }
    return worst;
}

// the fitted polynomial is used when the field of view clamps everything steeper than vfov/2 + 2 deg <= 45 deg and the fit is at least
// as good as the generic polynomial on [0, 1] needs to be for the error budget of geom_for (1.8e-6 rad there; 1e-6 asked here)
int elevation_fit_for(float vfov, float c4[4], double* err)
{
    c4[0] = 1.0f; c4[1] = c4[2] = c4[3] = 0.0f;
    *err = 0.0;
    if (!(vfov > 0.0f) || 0.5 * (double)vfov + 2.0 > 45.0) return 0;
    *err = fit_elevation_poly(std::tan((0.5 * (double)vfov + 2.0) * (3.14159265358979323846 / 180.0)), c4);
    return *err <= 1.0e-6 ? 1 : 0;
}

void pack_from_host(const void* src, size_t n, size_t stride, std::vector<float>& out)
{
    out.resize(n * 4);
    const unsigned char* s = static_cast<const unsigned char*>(src);
    const size_t ioff = (stride >= 32) ? 16 : 12;   // pcl::PointXYZI keeps intensity in its second 16-byte lane
    for (size_t i = 0; i < n; ++i) {
        memcpy(&out[4 * i], s + i * stride, 12);
        memcpy(&out[4 * i + 3], s + i * stride + ioff, 4);
    }
}
void unpack_to_host(const float* packed, size_t n, size_t stride, void* dst)
{
    unsigned char* d = static_cast<unsigned char*>(dst);
    const float one = 1.0f;
    for (size_t i = 0; i < n; ++i) {
        unsigned char* p = d + i * stride;
        if (stride >= 32) {
            memset(p, 0, 32);
            memcpy(p, &packed[4 * i], 12); memcpy(p + 12, &one, 4); memcpy(p + 16, &packed[4 * i + 3], 4);
        } else {
            memcpy(p, &packed[4 * i], 16);
        }
    }
}



} // namespace ltm_detail

// =========================================================================================== C ABI
extern "C" {

int ltm_abi_version(void) { return LTM_ABI_VERSION; }

int ltm_create(const ltm_config* cfg, ltm_ctx** out)
{
    if (!cfg || !out) return LTM_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LTM_E_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return LTM_E_INVALID;
    ltm_ctx* c = new (std::nothrow) ltm_ctx();
    if (!c) return LTM_E_NOMEM;
    c->cfg = *cfg;
    c->device = cfg->device;
    if (cfg->max_kf_batch > 0) c->kf_batch = (size_t)cfg->max_kf_batch;
    double b2l[16];
    if (!(cfg->vfov > 0.0f) || !(cfg->hfov > 0.0f) || !inverse4x4(cfg->lidar2base, b2l)) { delete c; return LTM_E_INVALID; }
    c->l2b_identity = mat_is_identity(cfg->lidar2base);
    if (c->l2b_identity) memcpy(b2l, cfg->lidar2base, sizeof b2l);   // the inverse of I is exactly I
    c->b2l_identity = mat_is_identity(b2l);
    c->L2B = to34(cfg->lidar2base); c->B2L = to34(b2l);
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return LTM_E_DEVICE;
    }
    c->pool.stream = c->stream;
    // A/B switches and diagnostics of the projection kernels: part of THIS context (KernelOpts, ltm_kernels.h) -- round 4 kept them in process-wide
    // statics that every ltm_create rewrote, a data race by the letter for `ltm_run --gpus K` (K threads, K contexts)
    auto env_int = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
    c->kopts.map_kernel_variant = env_int("LTM_MAP_KERNEL", 2);
    c->kopts.vote_cull = env_int("LTM_VOTE_CULL", 1);
    c->kopts.tile_cull = env_int("LTM_TILE_CULL", 1);
    c->kopts.stats_blockmin = env_int("LTM_STATS_BLOCKMIN", 0);
    // Exhaustive (2^32 inputs, a few ms) device check of the fast rad2deg / divide-by-FOV forms for THIS context's
    // constants; they are enabled only if they reproduce the exact IEEE results for every input.
    {
        unsigned long long* d = nullptr;
        bool ok = hipMalloc(&d, 3 * sizeof(unsigned long long)) == hipSuccess && hipMemsetAsync(d, 0, 24, c->stream) == hipSuccess &&
                  selfcheck_fast_math(cfg->vfov, cfg->hfov, d, c->stream) == hipSuccess &&
                  hipMemcpyAsync(c->selfcheck, d, 24, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                  hipStreamSynchronize(c->stream) == hipSuccess;
        if (d) (void)hipFree(d);
        if (!ok) { (void)hipStreamDestroy(c->stream); delete c; return LTM_E_DEVICE; }
        c->fast_math = (c->selfcheck[0] == 0 && c->selfcheck[1] == 0 && c->selfcheck[2] == 0) ? 1 : 0;
        if (const char* v = getenv("LTM_KNN_FAST")) c->knn_two_phase = atoi(v);
        if (const char* v = getenv("LTM_VOXEL_KEYBITS")) c->voxel_key_compress = atoi(v);
        if (const char* v = getenv("LTM_VOXEL_FUSED_TAIL")) c->voxel_fused_tail = atoi(v);
        if (const char* v = getenv("LTM_VOXEL_IDENTITY")) c->voxel_identity = atoi(v);
        if (const char* v = getenv("LTM_OCCLUSION")) c->occlusion_cull = atoi(v);
        if (getenv("LTM_OCCLUSION_STATS")) c->occlusion_stats_on = 1;
        if (const char* v = getenv("LTM_OCCLUSION_MIN_PAIRS")) c->occlusion_min_pairs = (size_t)atoll(v);
        if (const char* v = getenv("LTM_OCCLUSION_RNEAR")) {      // a non-positive first shell would select no pair in any shell; NaN / inf fall back to the default
            const float r = (float)atof(v);
            c->occlusion_r_near = std::isfinite(r) ? std::max(1.0f, r) : 60.0f;
        }
        if (const char* v = getenv("LTM_KNN_STATS")) c->knn_stats_on = atoi(v);
        if (const char* v = getenv("LTM_CULL_SELFCHECK")) c->cull_selfcheck = atoi(v);
        if (const char* v = getenv("LTM_CULL_EPS_SCALE")) c->cull_eps_scale = (float)atof(v);
        if (const char* v = getenv("LTM_CULL_EPS_FLOOR")) c->cull_eps_floor = (float)atof(v);
        c->el_fit = elevation_fit_for(c->cfg.vfov, c->el_c, &c->el_fit_err);
        if (const char* v = getenv("LTM_HEAVY_CHAIN")) c->heavy_chain_on = atoi(v);
        if (const char* v = getenv("LTM_HEAVY_MIN_BLOCKS")) c->heavy_min_blocks = (size_t)atoll(v);
    }
    c->heavy = std::make_shared<HeavyChain>();
    *out = c;
    return LTM_OK;
}

void ltm_destroy(ltm_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    vgs_release_all(c);      // before the pinned blocks and the pool go: a coordinator thread of an abandoned ticket reads and writes them
    for (Pending& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (auto& kv : c->uploads) { for (int b = 0; b < 2; ++b) if (kv.second.ev[b]) (void)hipEventDestroy(kv.second.ev[b]); c->pool.free(kv.second.d); }
    for (PinnedBlock& b : c->pinned) (void)hipHostFree(b.p);
    if (c->scratch_pinned) (void)hipHostFree(c->scratch_pinned);
    if (c->live_counts) (void)hipFree(c->live_counts);
    if (getenv("LTM_OCCLUSION_STATS") && c->occl_pairs)
        fprintf(stderr, "[ltm] occlusion cull of the exact-image kernel: %llu (tile, keyframe) pairs, %.2f %% in the first shell, %.2f %% projected in all, %.2f %% dropped\n",
                (unsigned long long)c->occl_pairs, 100.0 * c->occl_near / c->occl_pairs, 100.0 * c->occl_far_live / c->occl_pairs,
                100.0 * (c->occl_pairs - c->occl_far_live) / c->occl_pairs);
    if (getenv("LTM_OCCLUSION_STATS") && c->occl_quarters)
        fprintf(stderr, "[ltm] occlusion cull, second look: %llu 1024-point quarters in the pairs left alive, %.2f %% of them projected\n",
                (unsigned long long)c->occl_quarters, 100.0 * c->occl_quarters_live / c->occl_quarters);
    if (c->knn_stats_on && c->knn_queries)
        fprintf(stderr, "[ltm] kNN two-phase: %llu of %llu scan queries left undecided by the bucket test (%.2f %%)\n", (unsigned long long)c->knn_undecided,
                (unsigned long long)c->knn_queries, 100.0 * (double)c->knn_undecided / (double)c->knn_queries);
    if (getenv("LTM_POOL_STATS"))
        fprintf(stderr, "[ltm] device pool: %zu hipMalloc calls, %.1f MB held, %.1f ms inside hipMalloc; pinned host blocks: %zu, %.1f MB, %.1f ms inside hipHostMalloc\n",
                c->pool.n_malloc, c->pool.bytes_total / 1048576.0, 1e3 * c->pool.malloc_s, c->pinned.size(), c->pinned_bytes / 1048576.0, 1e3 * c->pinned_s);
    c->pool.release_all();
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* ltm_last_error(const ltm_ctx* c) { return c ? c->err.c_str() : "null context"; }

int ltm_synchronize(ltm_ctx* c) { return guarded(c, [&] { sync(c); }); }

int ltm_clear_caches(ltm_ctx* c) { return guarded(c, [&] { sync(c); scan_cache_drop(c, 0); }); }

void* ltm_stream(ltm_ctx* c) { return c ? (void*)c->stream : nullptr; }

void ltm_rimg_size(float vfov, float hfov, float alpha, int* rows, int* cols)
{
    if (rows) *rows = (int)roundf(vfov * alpha);
    if (cols) *cols = (int)roundf(hfov * alpha);
}

// ------------------------------------------------------------------------------- clouds
int ltm_cloud_upload(ltm_ctx* c, const void* pts, size_t n, size_t stride, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out && (pts || n == 0), "null argument");
        LTM_REQUIRE(stride == 16 || stride >= 32 || n == 0, "stride must be 16 (packed) or >= 32 (pcl::PointXYZI)");
        float4* d;
        const ltm_cloud h = alloc_cloud(c, n, &d);
        if (n) {
            if (stride == 16) h2d(c, d, pts, n * 16);
            else { std::vector<float> tmp; pack_from_host(pts, n, stride, tmp); h2d(c, d, tmp.data(), n * 16); }
        }
        *out = h;
    });
}

int ltm_cloud_from_device(ltm_ctx* c, const void* dev, size_t n, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out && (dev || n == 0), "null argument");
        float4* d;
        const ltm_cloud h = alloc_cloud(c, n, &d);
        d2d(c, d, dev, n * 16);
        sync(c);
        *out = h;
    });
}

int ltm_cloud_size(ltm_ctx* c, ltm_cloud h, size_t* n)
{
    return guarded(c, [&] { LTM_REQUIRE(n, "null argument"); *n = get_cloud(c, h).n; });
}

int ltm_cloud_download(ltm_ctx* c, ltm_cloud h, void* dst, size_t cap, size_t stride)
{
    return guarded(c, [&] {
        const Cloud& cl = get_cloud(c, h);
        LTM_REQUIRE(dst || cl.n == 0, "null destination");
        LTM_REQUIRE(cap >= cl.n, "destination too small");
        LTM_REQUIRE(stride == 16 || stride >= 32, "stride must be 16 or >= 32");
        if (!cl.n) return;
        if (stride == 16) d2h(c, dst, cl.d, cl.n * 16);
        else { std::vector<float> tmp(cl.n * 4); d2h(c, tmp.data(), cl.d, cl.n * 16); unpack_to_host(tmp.data(), cl.n, stride, dst); }
    });
}

int ltm_cloud_device_ptr(ltm_ctx* c, ltm_cloud h, const void** p)
{
    return guarded(c, [&] { LTM_REQUIRE(p, "null argument"); *p = get_cloud(c, h).d; });
}

int ltm_cloud_clone(ltm_ctx* c, ltm_cloud h, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        const Cloud src = get_cloud(c, h);
        float4* d;
        const ltm_cloud nh = alloc_cloud(c, src.n, &d);
        d2d(c, d, src.d, src.n * 16);
        inherit_frame(c, nh, src);
        *out = nh;
    });
}

int ltm_cloud_concat(ltm_ctx* c, const ltm_cloud* in, size_t n, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out && (in || n == 0), "null argument");
        size_t tot = 0;
        for (size_t i = 0; i < n; ++i) tot += get_cloud(c, in[i]).n;
        float4* d;
        const ltm_cloud h = alloc_cloud(c, tot, &d);
        size_t at = 0;
        for (size_t i = 0; i < n; ++i) { const Cloud& s = get_cloud(c, in[i]); d2d(c, d + at, s.d, s.n * 16); at += s.n; }
        *out = h;
    });
}

int ltm_cloud_transform(ltm_ctx* c, ltm_cloud hin, const double* T1, const double* T2, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        const Cloud in = get_cloud(c, hin);
        float4* d;
        const ltm_cloud h = alloc_cloud(c, in.n, &d);
        HostMat34 a, b;
        if (T1) a = to34(T1);
        if (T2) b = to34(T2);
        if (!T1 && !T2) d2d(c, d, in.d, in.n * 16);
        else LTM_HIP(transform_cloud(in.d, in.n, T1 ? &a : nullptr, T2 ? &b : nullptr, d, c->stream));
        *out = h;
    });
}

int ltm_cloud_select(ltm_ctx* c, ltm_cloud hin, const int32_t* idx_host, size_t n_idx, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out && (idx_host || n_idx == 0), "null argument");
        const Cloud in = get_cloud(c, hin);
        for (size_t j = 0; j < n_idx; ++j) LTM_REQUIRE(idx_host[j] >= 0 && (size_t)idx_host[j] < in.n, "point index out of range");
        float4* d;
        const ltm_cloud h = alloc_cloud(c, n_idx, &d);
        if (n_idx) {
            DevBuf idx(c, n_idx * sizeof(uint32_t));
            h2d(c, idx.p, idx_host, n_idx * sizeof(uint32_t));     // non-negative int32 == uint32
            LTM_HIP(gather_points(in.d, idx.as<uint32_t>(), n_idx, d, c->stream));
            sync(c);
        }
        *out = h;
    });
}

int ltm_scanset_keyframe(ltm_ctx* c, ltm_scanset hs, size_t kf, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        const ScanSet& s = get_ss(c, hs);
        LTM_REQUIRE(kf < s.nkf(), "keyframe out of range");
        const size_t a = s.off[kf], n = s.off[kf + 1] - a;
        float4* d;
        const ltm_cloud h = alloc_cloud(c, n, &d);
        if (n) d2d(c, d, s.d + a, n * 16);
        *out = h;
    });
}

int ltm_cloud_alloc(ltm_ctx* c, size_t n, ltm_cloud* out)
{
    return guarded(c, [&] { LTM_REQUIRE(out, "null argument"); float4* d; *out = alloc_cloud(c, n, &d); });
}

int ltm_buffer_alloc(ltm_ctx* c, size_t bytes, void** dev)
{
    return guarded(c, [&] { LTM_REQUIRE(dev, "null argument"); *dev = c->pool.alloc(bytes); });
}

int ltm_buffer_free(ltm_ctx* c, void* dev)
{
    return guarded(c, [&] { sync(c); c->pool.free(dev); });      // the caller may have used it on another stream: drain ours before recycling
}

int ltm_buffer_fill(ltm_ctx* c, void* dev, int byte_value, size_t bytes)
{
    return guarded(c, [&] { LTM_REQUIRE(dev || bytes == 0, "null buffer"); if (bytes) LTM_HIP(hipMemsetAsync(dev, byte_value, bytes, c->stream)); });
}

int ltm_buffer_copy(ltm_ctx* c, void* dst, const void* src, size_t bytes, int kind)
{
    return guarded(c, [&] {
        LTM_REQUIRE((dst && src) || bytes == 0, "null buffer");
        LTM_REQUIRE(kind >= 0 && kind <= 2, "kind must be 0 (h2d), 1 (d2h) or 2 (d2d)");
        if (!bytes) return;
        if (kind == 0) h2d(c, dst, src, bytes);
        else if (kind == 1) d2h(c, dst, src, bytes);
        else { d2d(c, dst, src, bytes); sync(c); }
    });
}

int ltm_cloud_free(ltm_ctx* c, ltm_cloud h)
{
    return guarded(c, [&] { Cloud& cl = get_cloud(c, h); if (!cl.borrowed) c->pool.free(cl.d); c->clouds.erase(h); });
}

// ---------------------------------------------------------------------------- scan sets
static void check_offsets(const uint64_t* off, size_t n_kf)
{
    LTM_REQUIRE(off, "null offsets");
    LTM_REQUIRE(off[0] == 0, "offsets[0] must be 0");
    for (size_t i = 0; i < n_kf; ++i) LTM_REQUIRE(off[i] <= off[i + 1], "offsets must be non-decreasing");
}

int ltm_scanset_upload(ltm_ctx* c, const void* pts, size_t stride, const uint64_t* off, size_t n_kf, ltm_scanset* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        check_offsets(off, n_kf);
        const size_t n = off[n_kf];
        LTM_REQUIRE(pts || n == 0, "null points");
        LTM_REQUIRE(stride == 16 || stride >= 32 || n == 0, "stride must be 16 or >= 32");
        float4* d = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(n, 1) * 16));
        if (n) {
            if (stride == 16) h2d(c, d, pts, n * 16);
            else { std::vector<float> tmp; pack_from_host(pts, n, stride, tmp); h2d(c, d, tmp.data(), n * 16); }
        }
        *out = new_scanset(c, d, std::vector<uint64_t>(off, off + n_kf + 1));
    });
}

int ltm_scanset_from_device(ltm_ctx* c, const void* dev, const uint64_t* off, size_t n_kf, ltm_scanset* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        check_offsets(off, n_kf);
        const size_t n = off[n_kf];
        LTM_REQUIRE(dev || n == 0, "null points");
        float4* d = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(n, 1) * 16));
        d2d(c, d, dev, n * 16);
        sync(c);
        *out = new_scanset(c, d, std::vector<uint64_t>(off, off + n_kf + 1));
    });
}

int ltm_scanset_info(ltm_ctx* c, ltm_scanset h, size_t* n_kf, size_t* n_pts)
{
    return guarded(c, [&] { const ScanSet& s = get_ss(c, h); if (n_kf) *n_kf = s.nkf(); if (n_pts) *n_pts = s.n_pts; });
}

int ltm_scanset_offsets(ltm_ctx* c, ltm_scanset h, uint64_t* off)
{
    return guarded(c, [&] { LTM_REQUIRE(off, "null argument"); const ScanSet& s = get_ss(c, h); memcpy(off, s.off.data(), s.off.size() * 8); });
}

int ltm_scanset_download(ltm_ctx* c, ltm_scanset h, void* dst, size_t cap, size_t stride)
{
    return guarded(c, [&] {
        const ScanSet& s = get_ss(c, h);
        LTM_REQUIRE(dst || s.n_pts == 0, "null destination");
        LTM_REQUIRE(cap >= s.n_pts, "destination too small");
        LTM_REQUIRE(stride == 16 || stride >= 32, "stride must be 16 or >= 32");
        if (!s.n_pts) return;
        if (stride == 16) d2h(c, dst, s.d, s.n_pts * 16);
        else { std::vector<float> tmp(s.n_pts * 4); d2h(c, tmp.data(), s.d, s.n_pts * 16); unpack_to_host(tmp.data(), s.n_pts, stride, dst); }
    });
}

int ltm_scanset_device_ptr(ltm_ctx* c, ltm_scanset h, const void** p)
{
    return guarded(c, [&] { LTM_REQUIRE(p, "null argument"); *p = get_ss(c, h).d; });
}

int ltm_scanset_as_cloud(ltm_ctx* c, ltm_scanset h, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        const ScanSet& s = get_ss(c, h);
        float4* d;
        const ltm_cloud nh = alloc_cloud(c, s.n_pts, &d);
        d2d(c, d, s.d, s.n_pts * 16);
        *out = nh;
    });
}

int ltm_scanset_concat(ltm_ctx* c, const ltm_scanset* in, size_t n, ltm_scanset* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out && (in || n == 0), "null argument");
        std::vector<uint64_t> off(1, 0);
        size_t tot = 0;
        for (size_t i = 0; i < n; ++i) tot += get_ss(c, in[i]).n_pts;
        float4* d = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(tot, 1) * 16));
        size_t at = 0;
        for (size_t i = 0; i < n; ++i) {
            const ScanSet& s = get_ss(c, in[i]);
            d2d(c, d + at, s.d, s.n_pts * 16);
            for (size_t k = 1; k < s.off.size(); ++k) off.push_back(at + s.off[k]);
            at += s.n_pts;
        }
        *out = new_scanset(c, d, std::move(off));
    });
}

int ltm_scanset_zip_concat(ltm_ctx* c, ltm_scanset ha, ltm_scanset hb, ltm_scanset hc, ltm_scanset* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        const ScanSet& A = get_ss(c, ha);
        const ScanSet& B = get_ss(c, hb);
        const ScanSet* C = hc ? &get_ss(c, hc) : nullptr;
        const size_t nk = A.nkf();
        LTM_REQUIRE(B.nkf() == nk && (!C || C->nkf() == nk), "scan sets have different keyframe counts");
        std::vector<uint64_t> off(nk + 1, 0);
        for (size_t k = 0; k < nk; ++k)
            off[k + 1] = off[k] + (A.off[k + 1] - A.off[k]) + (B.off[k + 1] - B.off[k]) + (C ? C->off[k + 1] - C->off[k] : 0);
        const size_t tot = off[nk];
        float4* d = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(tot, 1) * 16));
        const ltm_scanset h = new_scanset(c, d, std::move(off));      // uploads the result offsets
        const ScanSet& O = get_ss(c, h);
        // a missing third operand is given zero-length segments by pointing it at B with B's own start offsets twice:
        // (oc[k+1]-oc[k]) is never read for it because j < na + nb always holds; pass B's arrays to keep pointers valid
        LTM_HIP(zip_concat(A.d, A.off_dev, B.d, B.off_dev, C ? C->d : B.d, C ? C->off_dev : B.off_dev, O.off_dev, nk, tot, d, c->stream));
        *out = h;
    });
}

int ltm_scanset_alloc(ltm_ctx* c, const uint64_t* off, size_t n_kf, ltm_scanset* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        check_offsets(off, n_kf);
        float4* d = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(off[n_kf], 1) * 16));
        *out = new_scanset(c, d, std::vector<uint64_t>(off, off + n_kf + 1));
    });
}

int ltm_scanset_free(ltm_ctx* c, ltm_scanset h)
{
    return guarded(c, [&] { ScanSet& s = get_ss(c, h); scan_cache_drop(c, h); if (!s.borrowed) { c->pool.free(s.d); c->pool.free(s.off_dev); } c->scansets.erase(h); });
}

// ------------------------------------------------------------------ pipelined upload / async fetch
namespace {
// what has been gathered in the current staging buffer goes to the device array behind the points already sent; the other buffer becomes current
void upload_flush(ltm_ctx* c, UploadState& u)
{
    if (!u.fill) return;
    const int b = u.next;
    if (!u.ev[b]) LTM_HIP(hipEventCreateWithFlags(&u.ev[b], hipEventDisableTiming));
    LTM_HIP(hipMemcpyAsync(u.d + u.flushed, u.stage[b], u.fill, hipMemcpyHostToDevice, copy_stream(c)));
    LTM_HIP(hipEventRecord(u.ev[b], copy_stream(c)));
    u.busy[b] = true;
    u.flushed += u.fill / 16;
    u.fill = 0;
    u.next ^= 1;
}
} // namespace

int ltm_scanset_upload_begin(ltm_ctx* c, size_t capacity_points, ltm_upload* up)
{
    return guarded(c, [&] {
        LTM_REQUIRE(up, "null argument");
        UploadState u;
        u.cap = capacity_points;
        u.d = reinterpret_cast<float4*>(c->pool.alloc(std::max<size_t>(capacity_points, 1) * 16));
        try { copy_after_compute(c); }      // the block may be recycled: kernels already queued on the compute stream may still use it
        catch (...) { c->pool.free(u.d); throw; }
        const uint64_t h = c->next_handle++;
        c->uploads[h] = std::move(u);
        *up = h;
    });
}

int ltm_scanset_upload_chunk(ltm_ctx* c, ltm_upload up, const void* pts, size_t stride, const uint64_t* kf_sizes, size_t n_kf)
{
    return guarded(c, [&] {
        auto it = c->uploads.find(up);
        LTM_REQUIRE(it != c->uploads.end(), "invalid upload handle");
        UploadState& u = it->second;
        LTM_REQUIRE(kf_sizes || n_kf == 0, "null keyframe sizes");
        size_t n = 0;
        for (size_t k = 0; k < n_kf; ++k) n += kf_sizes[k];
        LTM_REQUIRE(pts || n == 0, "null points");
        LTM_REQUIRE(stride == 16 || stride >= 32 || n == 0, "stride must be 16 or >= 32");
        LTM_REQUIRE(u.n + n <= u.cap, "upload exceeds the announced capacity");
        if (n) {
            static constexpr size_t kStage = (size_t)16 << 20;
            if (u.fill && u.fill + n * 16 > u.stage_sz[u.next]) upload_flush(c, u);      // does not fit behind what is gathered: send that first
            const int b = u.next;
            if (u.busy[b]) { LTM_HIP(hipEventSynchronize(u.ev[b])); u.busy[b] = false; }      // its previous DMA must have drained
            const size_t want = std::max(kStage, n * 16);
            if (u.stage_sz[b] < want) {
                if (u.stage[b]) pinned_free(c, u.stage[b]);
                u.stage[b] = nullptr; u.stage_sz[b] = 0;
                u.stage[b] = pinned_alloc(c, want);
                u.stage_sz[b] = want;
            }
            unsigned char* dst = static_cast<unsigned char*>(u.stage[b]) + u.fill;
            if (stride == 16) memcpy(dst, pts, n * 16);
            else {
                const unsigned char* s = static_cast<const unsigned char*>(pts);
                float* o = reinterpret_cast<float*>(dst);
                for (size_t i = 0; i < n; ++i) { memcpy(o + 4 * i, s + i * stride, 12); memcpy(o + 4 * i + 3, s + i * stride + 16, 4); }
            }
            u.fill += n * 16;
            if (u.fill >= u.stage_sz[b]) upload_flush(c, u);
        }
        for (size_t k = 0; k < n_kf; ++k) { u.n += kf_sizes[k]; u.off.push_back(u.n); }
    });
}

int ltm_scanset_upload_end(ltm_ctx* c, ltm_upload up, ltm_scanset* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        auto it = c->uploads.find(up);
        LTM_REQUIRE(it != c->uploads.end(), "invalid upload handle");
        upload_flush(c, it->second);
        UploadState u = std::move(it->second);
        c->uploads.erase(it);
        LTM_HIP(hipStreamSynchronize(copy_stream(c)));
        for (int b = 0; b < 2; ++b) { if (u.ev[b]) (void)hipEventDestroy(u.ev[b]); if (u.stage[b]) pinned_free(c, u.stage[b]); }
        *out = new_scanset(c, u.d, std::move(u.off));
    });
}

// copier thread of a context: takes the chunked tickets in the order they were begun; for each, waits until the compute stream has
// reached the fetch point, then moves the planned chunks one after the other through free ring slots (D2H into pinned memory on
// its own stream) and hands them to the ticket's consumers.  The context thread never waits for any of this.
static void ring_worker(FetchRing* r)
{
    (void)hipSetDevice(r->device);
    hipStream_t stream = nullptr;
    const bool have_stream = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess;
    for (;;) {
        ltm_fetch* t = nullptr;
        {
            std::unique_lock<std::mutex> lk(r->mx);
            r->cv_jobs.wait(lk, [&] { return r->stop || !r->jobs.empty(); });
            if (r->jobs.empty()) break;
            t = r->jobs.front();
            r->jobs.pop_front();
        }
        int rc = (have_stream && hipEventSynchronize(t->done) == hipSuccess) ? LTM_OK : LTM_E_DEVICE;
        for (size_t i = 0; i < t->plan.size() && rc == LTM_OK; ++i) {
            FetchChunk ch = t->plan[i];
            void* slot = nullptr;
            {
                std::unique_lock<std::mutex> lk(r->mx);
                r->cv_free.wait(lk, [&] { return r->stop || !r->free_slots.empty(); });
                if (r->free_slots.empty()) { rc = LTM_E_DEVICE; break; }       // shut down under us
                slot = r->free_slots.back();
                r->free_slots.pop_back();
            }
            if (ch.n_points && (hipMemcpyAsync(slot, t->src + ch.first_point, ch.n_points * sizeof(float4), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                                hipStreamSynchronize(stream) != hipSuccess)) {
                std::lock_guard<std::mutex> lk(r->mx);
                r->free_slots.push_back(slot);
                rc = LTM_E_DEVICE;
                break;
            }
            ch.host = slot;
            std::lock_guard<std::mutex> lk(t->mx);
            t->avail.push_back(ch);
            t->cv.notify_one();
        }
        std::lock_guard<std::mutex> lk(t->mx);      // notify under the lock: ltm_fetch_release may delete the ticket right after
        t->error = rc;
        t->produced_all = true;
        t->cv.notify_all();
    }
    if (have_stream) (void)hipStreamDestroy(stream);
}

static FetchRing* ensure_ring(ltm_ctx* c)
{
    std::lock_guard<std::mutex> fam(c->heavy->ring_mx);
    if (c->heavy->ring) return c->heavy->ring;
    std::unique_ptr<FetchRing> r(new FetchRing());
    r->device = c->device;
    size_t mb = 8, n_slots = 8;      // 64 MB pinned once (~15 ms; round 4's 8 x 32 MB cost 45 ms of page-locking inside the first fetch, i.e. inside makeGlobalMap's map write)
    if (const char* v = getenv("LTM_FETCH_CHUNK_MB")) mb = (size_t)std::max(1, atoi(v));
    if (const char* v = getenv("LTM_FETCH_SLOTS")) n_slots = (size_t)std::max(2, atoi(v));
    r->slot_bytes = mb << 20;
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n_slots; ++i) {
        void* p = nullptr;
        if (hipHostMalloc(&p, r->slot_bytes, hipHostMallocDefault) != hipSuccess) {
            for (void* q : r->slots) (void)hipHostFree(q);
            throw Err{LTM_E_NOMEM, "hipHostMalloc of a fetch staging chunk failed"};
        }
        r->slots.push_back(p);
    }
    c->pinned_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c->pinned_bytes += n_slots * r->slot_bytes;
    r->free_slots = r->slots;
    r->worker = std::thread(ring_worker, r.get());
    c->heavy->ring = r.release();
    return c->heavy->ring;
}

void destroy_ring(FetchRing* r)      // with the last context of the family (declared in ltm_internal.h: ~HeavyChain calls it)
{
    if (!r) return;
    { std::lock_guard<std::mutex> lk(r->mx); r->stop = true; }
    r->cv_jobs.notify_all();
    r->cv_free.notify_all();
    if (r->worker.joinable()) r->worker.join();
    (void)hipSetDevice(r->device);
    for (void* p : r->slots) (void)hipHostFree(p);
    delete r;
}

static void fetch_chunks_begin(ltm_ctx* c, const float4* src, size_t n, std::vector<uint64_t> off, ltm_fetch** out)
{
    FetchRing* r = ensure_ring(c);
    const size_t cap = r->slot_bytes / sizeof(float4);
    std::unique_ptr<ltm_fetch> t(new ltm_fetch());
    t->chunked = true; t->src = src; t->ring = r;
    t->n_points = n; t->bytes = n * 16; t->off = std::move(off); t->device = c->device;
    if (t->off.empty()) {
        for (size_t first = 0; first < n; first += cap) t->plan.push_back(FetchChunk{nullptr, first, std::min(cap, n - first), 0, 0});
    } else {      // whole keyframes per chunk, so that every chunk can be written out on its own
        const size_t n_kf = t->off.size() - 1;
        for (size_t a = 0; a < n_kf;) {
            size_t b = a + 1;
            LTM_REQUIRE(t->off[b] - t->off[a] <= cap, "a keyframe does not fit a fetch staging chunk (LTM_FETCH_CHUNK_MB)");
            while (b < n_kf && t->off[b + 1] - t->off[a] <= cap) ++b;
            t->plan.push_back(FetchChunk{nullptr, (size_t)t->off[a], (size_t)(t->off[b] - t->off[a]), a, b - a});
            a = b;
        }
    }
    LTM_HIP(hipEventCreateWithFlags(&t->done, hipEventDisableTiming));
    if (hipEventRecord(t->done, c->stream) != hipSuccess) {        // the source is final once the compute stream gets here
        (void)hipEventDestroy(t->done);
        throw Err{LTM_E_DEVICE, "hipEventRecord failed for a chunked fetch"};
    }
    ltm_fetch* raw = t.release();
    { std::lock_guard<std::mutex> lk(r->mx); r->jobs.push_back(raw); }
    r->cv_jobs.notify_one();
    *out = raw;
}

static void fetch_begin(ltm_ctx* c, const float4* src, size_t n, std::vector<uint64_t> off, ltm_fetch** out)
{
    std::unique_ptr<ltm_fetch> t(new ltm_fetch());
    t->n_points = n; t->bytes = n * 16; t->off = std::move(off); t->device = c->device;
    t->host = pinned_alloc(c, t->bytes);
    try {
        LTM_HIP(hipEventCreateWithFlags(&t->done, hipEventDisableTiming));
        copy_after_compute(c);
        if (n) LTM_HIP(hipMemcpyAsync(t->host, src, t->bytes, hipMemcpyDeviceToHost, copy_stream(c)));
        LTM_HIP(hipEventRecord(t->done, copy_stream(c)));
    } catch (...) {      // the ticket dies with the unique_ptr: hand the pinned block back and drop the event
        if (t->done) { (void)hipStreamSynchronize(c->copy_stream); (void)hipEventDestroy(t->done); }
        pinned_free(c, t->host);
        throw;
    }
    *out = t.release();
}

int ltm_cloud_fetch_begin(ltm_ctx* c, ltm_cloud h, ltm_fetch** out)
{
    return guarded(c, [&] { LTM_REQUIRE(out, "null argument"); const Cloud cl = get_cloud(c, h); fetch_begin(c, cl.d, cl.n, {}, out); });
}

int ltm_scanset_fetch_begin(ltm_ctx* c, ltm_scanset h, ltm_fetch** out)
{
    return guarded(c, [&] { LTM_REQUIRE(out, "null argument"); const ScanSet& s = get_ss(c, h); fetch_begin(c, s.d, s.n_pts, s.off, out); });
}

int ltm_cloud_fetch_chunks_begin(ltm_ctx* c, ltm_cloud h, ltm_fetch** out)
{
    return guarded(c, [&] { LTM_REQUIRE(out, "null argument"); const Cloud cl = get_cloud(c, h); fetch_chunks_begin(c, cl.d, cl.n, {}, out); });
}

int ltm_scanset_fetch_chunks_begin(ltm_ctx* c, ltm_scanset h, ltm_fetch** out)
{
    return guarded(c, [&] { LTM_REQUIRE(out, "null argument"); const ScanSet& s = get_ss(c, h); fetch_chunks_begin(c, s.d, s.n_pts, s.off, out); });
}

int ltm_fetch_info(ltm_fetch* t, size_t* n_points, const uint64_t** offsets, size_t* n_kf)
{
    if (!t) return LTM_E_INVALID;
    if (n_points) *n_points = t->n_points;
    if (offsets) *offsets = t->off.empty() ? nullptr : t->off.data();
    if (n_kf) *n_kf = t->off.empty() ? 0 : t->off.size() - 1;
    return LTM_OK;
}

int ltm_fetch_next_chunk(ltm_fetch* t, const void** host_xyzi, size_t* first_point, size_t* n_points, size_t* first_kf, size_t* n_kf)
{
    if (!t || !t->chunked || !host_xyzi) return LTM_E_INVALID;
    std::unique_lock<std::mutex> lk(t->mx);
    t->cv.wait(lk, [&] { return !t->avail.empty() || t->produced_all; });
    if (t->avail.empty()) return t->error != LTM_OK ? t->error : 0;
    const FetchChunk ch = t->avail.front();
    t->avail.pop_front();
    *host_xyzi = ch.host;
    if (first_point) *first_point = ch.first_point;
    if (n_points) *n_points = ch.n_points;
    if (first_kf) *first_kf = ch.first_kf;
    if (n_kf) *n_kf = ch.n_kf;
    return 1;
}

int ltm_fetch_chunk_done(ltm_fetch* t, const void* host_xyzi)
{
    if (!t || !t->chunked || !host_xyzi) return LTM_E_INVALID;
    FetchRing* r = t->ring;
    { std::lock_guard<std::mutex> lk(r->mx); r->free_slots.push_back(const_cast<void*>(host_xyzi)); }
    r->cv_free.notify_one();
    return LTM_OK;
}

int ltm_fetch_wait(ltm_fetch* t, const void** host_xyzi, size_t* n_points, const uint64_t** offsets, size_t* n_kf)
{
    if (!t || t->chunked) return LTM_E_INVALID;
    if (hipEventSynchronize(t->done) != hipSuccess) return LTM_E_DEVICE;      // thread-safe: touches only this ticket
    if (host_xyzi) *host_xyzi = t->host;
    if (n_points) *n_points = t->n_points;
    if (offsets) *offsets = t->off.empty() ? nullptr : t->off.data();
    if (n_kf) *n_kf = t->off.empty() ? 0 : t->off.size() - 1;
    return LTM_OK;
}

int ltm_fetch_release(ltm_ctx* c, ltm_fetch* t)       // any thread: touches the ticket and, under its mutex, the pinned-block list
{
    if (!c || !t) return LTM_E_INVALID;
    if (t->chunked) {     // the copier thread must be through with the ticket; chunks nobody consumed go back to the ring
        {
            std::unique_lock<std::mutex> lk(t->mx);
            t->cv.wait(lk, [&] { return t->produced_all; });
        }
        {
            std::lock_guard<std::mutex> lk(t->ring->mx);
            for (const FetchChunk& ch : t->avail) t->ring->free_slots.push_back(ch.host);
        }
        t->ring->cv_free.notify_all();
        (void)hipEventDestroy(t->done);
        delete t;
        return LTM_OK;
    }
    (void)hipEventSynchronize(t->done);
    (void)hipEventDestroy(t->done);
    pinned_free(c, t->host);
    delete t;
    return LTM_OK;
}

// -------------------------------------------------------------------------------- poses
int ltm_poses_create(ltm_ctx* c, size_t n, const double* poses, const double* inv, ltm_poses* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out && (poses || n == 0), "null argument");
        Poses p;
        p.n = n;
        p.pose.assign(poses, poses + 16 * n);
        if (inv) p.inv.assign(inv, inv + 16 * n);
        else {
            p.inv.resize(16 * n);
            for (size_t i = 0; i < n; ++i) LTM_REQUIRE(inverse4x4(&p.pose[16 * i], &p.inv[16 * i]), "singular pose");
        }
        std::vector<double> a(12 * std::max<size_t>(n, 1)), b(12 * std::max<size_t>(n, 1));
        for (size_t i = 0; i < n; ++i) { memcpy(&a[12 * i], &p.pose[16 * i], 96); memcpy(&b[12 * i], &p.inv[16 * i], 96); }
        p.pose_dev = reinterpret_cast<double*>(c->pool.alloc(a.size() * 8));
        p.inv_dev = reinterpret_cast<double*>(c->pool.alloc(b.size() * 8));
        h2d(c, p.pose_dev, a.data(), a.size() * 8);
        h2d(c, p.inv_dev, b.data(), b.size() * 8);
        std::vector<float> ap(16 * std::max<size_t>(n, 1), 0.0f);
        double b2l16[16] = {0};
        memcpy(b2l16, c->B2L.m, 12 * sizeof(double)); b2l16[15] = 1.0;
        for (size_t i = 0; i < n; ++i) approx_pose(b2l16, &p.inv[16 * i], &ap[16 * i]);
        p.approx_dev = reinterpret_cast<float*>(c->pool.alloc(ap.size() * 4));
        h2d(c, p.approx_dev, ap.data(), ap.size() * 4);
        const uint64_t h = c->next_handle++;
        c->poses[h] = std::move(p);
        *out = h;
    });
}

int ltm_inverse4x4(const double* m16, double* inv16)
{
    if (!m16 || !inv16) return LTM_E_INVALID;
    return inverse4x4(m16, inv16) ? LTM_OK : LTM_E_INVALID;
}

int ltm_poses_free(ltm_ctx* c, ltm_poses h)
{
    return guarded(c, [&] { Poses& p = get_poses(c, h); c->pool.free(p.pose_dev); c->pool.free(p.inv_dev); c->pool.free(p.approx_dev); c->poses.erase(h); });
}

int ltm_merge_to_global(ltm_ctx* c, ltm_scanset hs, ltm_poses hp, ltm_cloud* out)
{
    return guarded(c, [&] {
        LTM_REQUIRE(out, "null argument");
        const ScanSet& s = get_ss(c, hs);
        const Poses& p = get_poses(c, hp);
        LTM_REQUIRE(s.nkf() == p.n, "scan set and poses have different keyframe counts");
        float4* d;
        const ltm_cloud h = alloc_cloud(c, s.n_pts, &d);
        {
            ProfScope ps(c, "merge", (double)s.n_pts, 32.0 * s.n_pts);
            LTM_HIP(transform_scans(s.d, s.off_dev, s.nkf(), s.n_pts, c->L2B, c->l2b_identity, p.pose_dev, d, c->stream));
        }
        *out = h;
    });
}

// ------------------------------------------------------------------------------- lanes
struct ltm_event { hipEvent_t e = nullptr; int device = 0; };

int ltm_lane_create(ltm_ctx* parent, ltm_ctx** out)
{
    if (!parent || !out) return LTM_E_INVALID;
    *out = nullptr;
    ltm_ctx* c = new (std::nothrow) ltm_ctx();
    if (!c) return LTM_E_NOMEM;
    {
        std::lock_guard<std::recursive_mutex> lk(parent->mx);
        c->cfg = parent->cfg; c->device = parent->device; c->L2B = parent->L2B; c->B2L = parent->B2L;
        c->l2b_identity = parent->l2b_identity; c->b2l_identity = parent->b2l_identity; c->kf_batch = parent->kf_batch; c->kopts = parent->kopts;
        c->fast_math = parent->fast_math;      // the exhaustive create-time self-check is a property of (device, vfov, hfov): not repeated
        for (int i = 0; i < 3; ++i) c->selfcheck[i] = parent->selfcheck[i];
        c->scan_cache_cap = parent->scan_cache_cap; c->occlusion_cull = parent->occlusion_cull;
        c->occlusion_min_pairs = parent->occlusion_min_pairs; c->occlusion_r_near = parent->occlusion_r_near;
        c->occlusion_subtile = parent->occlusion_subtile; c->occlusion_stats_on = parent->occlusion_stats_on;
        c->voxel_key_compress = parent->voxel_key_compress; c->voxel_fused_tail = parent->voxel_fused_tail; c->voxel_identity = parent->voxel_identity;
        c->knn_two_phase = parent->knn_two_phase; c->knn_stats_on = parent->knn_stats_on;
        c->cull_eps_scale = parent->cull_eps_scale; c->cull_eps_floor = parent->cull_eps_floor;
        c->cull_selfcheck = parent->cull_selfcheck; c->cull_geom_ok = parent->cull_geom_ok;      // shapes the parent has checked already (same device, field of view, extrinsic)
        c->el_fit = parent->el_fit; for (int i = 0; i < 4; ++i) c->el_c[i] = parent->el_c[i]; c->el_fit_err = parent->el_fit_err;
    }
    if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return LTM_E_DEVICE;
    }
    c->pool.stream = c->stream;
    {      // the parent joins the family: from now on its heavy launches are chained with the lane's
        std::lock_guard<std::recursive_mutex> lk(parent->mx);
        c->heavy = parent->heavy;
        c->heavy_chain_on = parent->heavy_chain_on; c->heavy_min_blocks = parent->heavy_min_blocks;
        c->in_lane_family = parent->in_lane_family = true;
    }
    *out = c;
    return LTM_OK;
}

int ltm_lane_fence(ltm_ctx* from, ltm_ctx* to) { return guarded2(from, to, [&] { stream_after(from, to); }); }

int ltm_event_record(ltm_ctx* c, ltm_event** ev)
{
    if (ev) *ev = nullptr;
    return guarded(c, [&] {
        LTM_REQUIRE(ev, "null argument");
        std::unique_ptr<ltm_event> t(new ltm_event());
        t->device = c->device;
        LTM_HIP(hipEventCreateWithFlags(&t->e, hipEventDisableTiming));
        const hipError_t rc = hipEventRecord(t->e, c->stream);
        if (rc != hipSuccess) { (void)hipEventDestroy(t->e); LTM_HIP(rc); }
        *ev = t.release();
    });
}

int ltm_event_wait(ltm_ctx* c, ltm_event* ev)
{
    return guarded(c, [&] {
        LTM_REQUIRE(ev && ev->e, "null event");
        LTM_REQUIRE(ev->device == c->device, "event of another device");
        LTM_HIP(hipStreamWaitEvent(c->stream, ev->e, 0));
    });
}

void ltm_event_destroy(ltm_event* ev)
{
    if (!ev) return;
    if (ev->e) { (void)hipSetDevice(ev->device); (void)hipEventDestroy(ev->e); }
    delete ev;
}

static int cloud_pass(ltm_ctx* from, ltm_cloud h, ltm_ctx* to, ltm_cloud* out, bool give)
{
    return guarded2(from, to, [&] {
        LTM_REQUIRE(out, "null argument");
        LTM_REQUIRE(from != to, "lend / give need two contexts");
        const Cloud src = get_cloud(from, h);
        LTM_REQUIRE(!give || !src.borrowed, "a borrowed cloud cannot be given away");
        stream_after(from, to);
        const ltm_cloud nh = new_cloud(to, src.d, src.n);
        Cloud& dst = to->clouds[nh];
        dst.vf_ok = src.vf_ok; dst.vf = src.vf; dst.vleaf = src.vleaf;
        if (give) {
            if (!from->pool.move_to(src.d, to->pool)) { to->clouds.erase(nh); throw Err{LTM_E_INVALID, "cloud memory is not owned by this context's pool"}; }
            from->clouds.erase(h);
        } else dst.borrowed = true;
        *out = nh;
    });
}

int ltm_cloud_lend(ltm_ctx* from, ltm_cloud h, ltm_ctx* to, ltm_cloud* out) { return cloud_pass(from, h, to, out, false); }

int ltm_cloud_give(ltm_ctx* from, ltm_cloud h, ltm_ctx* to, ltm_cloud* out) { return cloud_pass(from, h, to, out, true); }

static int scanset_pass(ltm_ctx* from, ltm_scanset h, ltm_ctx* to, ltm_scanset* out, bool give)
{
    return guarded2(from, to, [&] {
        LTM_REQUIRE(out, "null argument");
        LTM_REQUIRE(from != to, "lend / give need two contexts");
        ScanSet& src = get_ss(from, h);
        LTM_REQUIRE(!give || !src.borrowed, "a borrowed scan set cannot be given away");
        stream_after(from, to);
        ScanSet dst;
        dst.d = src.d; dst.n_pts = src.n_pts; dst.off = src.off; dst.off_dev = src.off_dev; dst.borrowed = !give;
        if (give) {
            LTM_REQUIRE(from->pool.owns(src.d) && from->pool.owns(src.off_dev), "scan set memory is not owned by this context's pool");
            scan_cache_drop(from, h);
            from->pool.move_to(src.d, to->pool); from->pool.move_to(src.off_dev, to->pool);
            from->scansets.erase(h);
        }
        const uint64_t nh = to->next_handle++;
        to->scansets[nh] = std::move(dst);
        *out = nh;
    });
}

int ltm_scanset_lend(ltm_ctx* from, ltm_scanset h, ltm_ctx* to, ltm_scanset* out) { return scanset_pass(from, h, to, out, false); }

int ltm_scanset_give(ltm_ctx* from, ltm_scanset h, ltm_ctx* to, ltm_scanset* out) { return scanset_pass(from, h, to, out, true); }

// Host arithmetic only (no device, no context): the order ltm_voxel_grid_scanset's PCL-order path gives the points of one keyframe, through
// ltm_pclsort::sort (use_std_sort == 0) or through the C++ library's std::sort (!= 0) -- tests/test_abi.py requires the two to agree.
int ltm_debug_pcl_sort_order(const uint32_t* leaf_idx, size_t n, uint32_t* order_out, int use_std_sort, uint32_t* heap_sort_fallbacks)
{
    if ((!leaf_idx || !order_out) && n) return LTM_E_INVALID;
    try {
        std::vector<ltm_pclsort::Entry> e(n);
        for (size_t i = 0; i < n; ++i) e[i] = ltm_pclsort::Entry{leaf_idx[i], (uint32_t)i};
        const unsigned long before = ltm_pclsort::heap_sort_fallbacks();
        if (use_std_sort) std::sort(e.begin(), e.end(), ltm_pclsort::Less());
        else ltm_pclsort::sort(e.data(), e.data() + n);
        if (heap_sort_fallbacks) *heap_sort_fallbacks = (uint32_t)(ltm_pclsort::heap_sort_fallbacks() - before);
        for (size_t i = 0; i < n; ++i) order_out[i] = e[i].cloud_point_index;
    } catch (...) { return LTM_E_NOMEM; }
    return LTM_OK;
}

int ltm_debug_elevation_fit(float vfov_deg, float* c4, double* max_err_rad)
{
    if (!c4 || !max_err_rad || !(vfov_deg > 0.0f) || !(vfov_deg < 180.0f)) return LTM_E_INVALID;
    try { return elevation_fit_for(vfov_deg, c4, max_err_rad); } catch (...) { return LTM_E_NOMEM; }
}

int ltm_debug_selfcheck(ltm_ctx* c, uint64_t* mismatches3, int* fast_math_enabled)
{
    return guarded(c, [&] {
        if (mismatches3) for (int i = 0; i < 3; ++i) mismatches3[i] = c->selfcheck[i];
        if (fast_math_enabled) *fast_math_enabled = c->fast_math;
    });
}

// ---------------------------------------------------------------------------- profiling
int ltm_profile_enable(ltm_ctx* c, int on) { return guarded(c, [&] { if (!on) prof_collect(c); c->prof_on = on != 0; }); }

int ltm_profile_reset(ltm_ctx* c)
{
    return guarded(c, [&] { prof_collect(c); for (ProfClass& p : c->prof) p = ProfClass(); });
}

int ltm_profile_read(ltm_ctx* c, const char** names, double* ms, uint64_t* launches, double* units, double* bytes, int cap)
{
    if (!c) return LTM_E_INVALID;
    const int rc = guarded(c, [&] { prof_collect(c); });
    if (rc != LTM_OK) return rc;
    const int n = (int)c->prof.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = c->prof_names[i].c_str();
        if (ms) ms[i] = c->prof[i].ms;
        if (launches) launches[i] = c->prof[i].launches;
        if (units) units[i] = c->prof[i].units;
        if (bytes) bytes[i] = c->prof[i].bytes;
    }
    return n;
}

int ltm_profile_read_compulsory(ltm_ctx* c, double* bytes_c, int cap)
{
    if (!c) return LTM_E_INVALID;
    const int rc = guarded(c, [&] { prof_collect(c); });
    if (rc != LTM_OK) return rc;
    const int n = (int)c->prof.size();
    for (int i = 0; i < n && i < cap; ++i) if (bytes_c) bytes_c[i] = c->prof[i].bytes_c;
    return n;
}

} // extern "C"
