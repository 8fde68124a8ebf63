/*
 * ltm_oracle.cpp -- TEST INFRASTRUCTURE (see ltm_oracle.h).  CPU restatement of the
 * LT-removert / LT-map hot path of gisbi-kim/lt-mapper.  Not shipped, not linked by the
 * product, never on the measured path (except as bench.py's separately reported
 * cpu_baseline).  Build: g++ -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fopenmp.
 *
 * Numerics contract: the reference is a generic x86-64 Release
 * build (ltremovert/CMakeLists.txt:4-6,87): SSE2 scalar float/double, no FMA.
 */
#include "ltm_oracle.h"
#include "oracle_math.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Pt { float x, y, z, i; };
typedef std::vector<Pt> Cloud;

const float kFlagNoPOINT = 10000.0f;        /* utility.h:93 */
const float kValidDiffUpperBound = 200.0f;  /* utility.h:94 */
const float kReprojectionAlpha = 3.0f;      /* Session.h:13 */

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

/* utility.cpp:53-56 : float rad2deg(float) { return radians * 180.0 / M_PI; }  (double math, float result) */
inline float rad2deg(float r) { return (float)((double)r * 180.0 / M_PI); }

/* Sensitivity experiments only (tools/numerics_sensitivity.py): with g_atan_perturb_ppm != 0 that fraction of the atan2f results is
 * moved by one ulp up or down, chosen by a hash of the arguments -- "what if the reference's libm rounded differently".  Off by
 * default; nothing else in the oracle reads it. */
unsigned g_atan_perturb_ppm = 0;
uint64_t g_atan_perturb_seed = 0;
inline float atan2f_ref(float y, float x)
{
    const float a = om_atan2f(y, x);
    if (__builtin_expect(g_atan_perturb_ppm == 0, 1)) return a;
    uint32_t by, bx;
    memcpy(&by, &y, 4); memcpy(&bx, &x, 4);
    uint64_t h = (((uint64_t)by << 32) | bx) + g_atan_perturb_seed + 0x9e3779b97f4a7c15ull;      /* splitmix64 */
    h = (h ^ (h >> 30)) * 0xbf58476d1ce4e5b9ull; h = (h ^ (h >> 27)) * 0x94d049bb133111ebull; h ^= h >> 31;
    if ((h >> 8) % 1000000u >= g_atan_perturb_ppm) return a;
    return nextafterf(a, (h & 1) ? INFINITY : -INFINITY);
}

struct Sph { float az, el, r; };
/* utility.cpp:38-51 */
inline Sph cart2sph(float x, float y, float z)
{
    Sph s;
    s.az = atan2f_ref(y, x);
    s.el = atan2f_ref(z, sqrtf(x * x + y * y));
    s.r = sqrtf(x * x + y * y + z * z);
    return s;
}

/* utility.cpp:222-236 */
inline void rimg_size(float vfov, float hfov, float alpha, int* R, int* C)
{
    *R = (int)roundf(vfov * alpha);
    *C = (int)roundf(hfov * alpha);
}

/* utility.cpp:114-125 (identical text at Removerter.cpp:130-139) */
inline void pixel_of(const Sph& s, float vfov, float hfov, int R, int C, int* row, int* col)
{
    float fr = roundf((float)R * (1 - (rad2deg(s.el) + (vfov / 2.0f)) / (vfov - 0.0f)));
    float fc = roundf((float)C * ((rad2deg(s.az) + (hfov / 2.0f)) / (hfov - 0.0f)));
    /* std::min(std::max(v, lo), hi) with std:: semantics: max(a,b) = (a<b)?b:a ; min(a,b) = (b<a)?b:a */
    float lo = 0.0f, hr = (float)(R - 1), hc = (float)(C - 1);
    float r1 = (fr < lo) ? lo : fr;  r1 = (hr < r1) ? hr : r1;
    float c1 = (fc < lo) ? lo : fc;  c1 = (hc < c1) ? hc : c1;
    *row = (int)r1;
    *col = (int)c1;
}

/* PCL 1.10 pcl::transformPointCloud<PointXYZI,double>(in,out,Matrix4d): generic (non-AVX)
 * detail::Transformer<double>::se3 -- each output = (float)(m0*x + m1*y + m2*z + m3) evaluated
 * left-to-right in double.  Call sites utility.cpp:70-71,164-165,184-185,198-199. */
inline Pt xform(const double* T, const Pt& p)
{
    const double x = p.x, y = p.y, z = p.z;
    Pt o;
    o.x = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
    o.y = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
    o.z = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
    o.i = p.i;
    return o;
}

void transform_cloud(const double* T, const Cloud& in, Cloud& out)
{
    out.resize(in.size());
    for (size_t i = 0; i < in.size(); ++i) out[i] = xform(T, in[i]);
}

/* utility.cpp:64-72 transformGlobalMapToLocal: T^-1 then base2lidar, float store in between */
inline Pt global_to_local_pt(const double* Tinv, const double* B2L, const Pt& p) { return xform(B2L, xform(Tinv, p)); }

/* utility.cpp:92-142 map2RangeImg (serial semantics: strict <, lowest index wins ties) */
void map2rimg(const Pt* pts, size_t n, const double* T1, const double* T2, float vfov, float hfov, int R, int C,
              float* rimg, int32_t* ptidx)
{
    const size_t npx = (size_t)R * C;
    for (size_t i = 0; i < npx; ++i) rimg[i] = kFlagNoPOINT;
    if (ptidx) for (size_t i = 0; i < npx; ++i) ptidx[i] = 0;
    for (size_t i = 0; i < n; ++i) {
        Pt p = pts[i];
        if (T1) p = xform(T1, p);
        if (T2) p = xform(T2, p);
        Sph s = cart2sph(p.x, p.y, p.z);
        int row, col;
        pixel_of(s, vfov, hfov, R, C, &row, &col);
        const size_t px = (size_t)row * C + col;
        if (s.r < rimg[px]) {
            rimg[px] = s.r;
            if (ptidx) ptidx[px] = (int32_t)i;
        }
    }
}

bool is_identity(const double* T)
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            if (T[r * 4 + c] != (r == c ? 1.0 : 0.0)) return false;
    return true;
}

/* Removerter.cpp:381-413 + the per-scan body of :429-593 for one keyframe.
 * mode 0: diff = scan - map (:572 / :458); mode 1: diff = map - scan (:515). */
void vote_one_kf(const Pt* map, size_t M, const Pt* scan, size_t S, const double* Tinv, const double* B2L,
                 float vfov, float hfov, int R, int C, float thr, int mode,
                 std::vector<float>& scan_rimg, std::vector<float>& map_rimg, std::vector<int32_t>& map_idx,
                 uint8_t* labels)
{
    const size_t npx = (size_t)R * C;
    scan_rimg.resize(npx); map_rimg.resize(npx); map_idx.resize(npx);
    map2rimg(scan, S, nullptr, nullptr, vfov, hfov, R, C, scan_rimg.data(), nullptr);       /* scan2RangeImg */
    map2rimg(map, M, Tinv, B2L, vfov, hfov, R, C, map_rimg.data(), map_idx.data());
    if (M == 0) return;
    for (size_t px = 0; px < npx; ++px) {
        const float diff = (mode == 0) ? (scan_rimg[px] - map_rimg[px]) : (map_rimg[px] - scan_rimg[px]);
        if (diff < kValidDiffUpperBound && diff > thr) labels[map_idx[px]] = 1;
    }
}

/* ---------------------------------------------------------------------------------------
 * Voxel centroid: PCL 1.10 pcl::octree::OctreePointCloudVoxelCentroid as driven by
 * utility.cpp:204-219 (setInputCloud, defineBoundingBox, addPointsFromInputCloud,
 * getVoxelCentroids).  Restated from the published behaviour of octree_pointcloud.hpp:
 *   defineBoundingBox():   min = (double)min_pt ; max = (double)(max_pt + FLT_EPSILON*512) [float add]
 *   getKeyBitSize():       max_key = ceil((max-min-FLT_EPSILON)/res); depth = ceil(log2(max(max_key,2)) - FLT_EPSILON)
 *                          side = (1<<depth)*res; empty tree => box centred: over=(side-(max-min))/2, if over>FLT_EPSILON
 *   genOctreeKeyforPoint:  key = (unsigned)(((double)p - min)/res)
 *   leaf container:        float sums of x,y,z,intensity in input order, / (float)count
 *   getVoxelCentroids:     DFS, child index = (xbit<<2)|(ybit<<1)|zbit  => Morton order, x most significant
 * "parity unpinned" (no PCL source in the container).
 * ------------------------------------------------------------------------------------- */
struct OctreeFrame { double minx, miny, minz; double res; unsigned depth; bool ok; };

OctreeFrame octree_frame_of_box(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float leaf);
OctreeFrame octree_frame(const Pt* pts, size_t n, float leaf)
{
    OctreeFrame f; f.ok = false; f.res = (double)leaf; f.depth = 0; f.minx = f.miny = f.minz = 0;
    if (n == 0) return f;
    float mnx = FLT_MAX, mny = FLT_MAX, mnz = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX, mxz = -FLT_MAX;
    for (size_t i = 0; i < n; ++i) {
        mnx = std::min(mnx, pts[i].x); mny = std::min(mny, pts[i].y); mnz = std::min(mnz, pts[i].z);
        mxx = std::max(mxx, pts[i].x); mxy = std::max(mxy, pts[i].y); mxz = std::max(mxz, pts[i].z);
    }
    return octree_frame_of_box(mnx, mny, mnz, mxx, mxy, mxz, leaf);
}
/* the frame PCL derives from a bounding box (getMinMax3D result): used directly by the multi-rank tests, where the box is that of a larger
 * cloud the points are a part of (key-range exchange) */
OctreeFrame octree_frame_of_box(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float leaf)
{
    OctreeFrame f; f.ok = false; f.res = (double)leaf; f.depth = 0; f.minx = f.miny = f.minz = 0;
    const float eps512 = FLT_EPSILON * 512.0f;
    double min_x = mnx, min_y = mny, min_z = mnz;
    double max_x = (float)(mxx + eps512), max_y = (float)(mxy + eps512), max_z = (float)(mxz + eps512);
    const float minValue = FLT_EPSILON;
    const double res = f.res;
    unsigned kx = (unsigned)std::ceil((max_x - min_x - minValue) / res);
    unsigned ky = (unsigned)std::ceil((max_y - min_y - minValue) / res);
    unsigned kz = (unsigned)std::ceil((max_z - min_z - minValue) / res);
    unsigned max_voxels = std::max(std::max(std::max(kx, ky), kz), 2u);
    unsigned depth = std::max(std::min(32u, (unsigned)std::ceil(std::log2((double)max_voxels) - minValue)), 0u);
    if (depth > 21) return f; /* 3*depth must fit the 64-bit Morton code used below (104 km at 0.05 m) */
    const double side = (double)(1u << depth) * res;
    double ox = (side - (max_x - min_x)) / 2.0, oy = (side - (max_y - min_y)) / 2.0, oz = (side - (max_z - min_z)) / 2.0;
    if (ox > minValue) { min_x -= ox; }
    if (oy > minValue) { min_y -= oy; }
    if (oz > minValue) { min_z -= oz; }
    f.minx = min_x; f.miny = min_y; f.minz = min_z; f.depth = depth; f.ok = true;
    return f;
}

inline uint64_t morton_xyz(unsigned kx, unsigned ky, unsigned kz, unsigned depth)
{
    uint64_t m = 0;
    for (int b = (int)depth - 1; b >= 0; --b)
        m = (m << 3) | (uint64_t)((((kx >> b) & 1u) << 2) | (((ky >> b) & 1u) << 1) | ((kz >> b) & 1u));
    return m;
}

void voxel_centroid(const Cloud& in, float leaf, Cloud& out, const float* box = nullptr)
{
    const size_t n = in.size();
    Cloud res;
    if (n == 0) { out.swap(res); return; }
    OctreeFrame f = box ? octree_frame_of_box(box[0], box[1], box[2], box[3], box[4], box[5], leaf) : octree_frame(in.data(), n, leaf);
    if (!f.ok) { fprintf(stderr, "[oracle] voxel_centroid: octree depth > 21 unsupported\n"); out.swap(res); return; }
    std::vector<std::pair<uint64_t, uint32_t>> keyed(n);
    for (size_t i = 0; i < n; ++i) {
        unsigned kx = (unsigned)(((double)in[i].x - f.minx) / f.res);
        unsigned ky = (unsigned)(((double)in[i].y - f.miny) / f.res);
        unsigned kz = (unsigned)(((double)in[i].z - f.minz) / f.res);
        keyed[i] = std::make_pair(morton_xyz(kx, ky, kz, f.depth), (uint32_t)i);
    }
    std::sort(keyed.begin(), keyed.end()); /* (key, original index): input order inside a voxel */
    res.reserve(n / 2 + 16);
    size_t a = 0;
    while (a < n) {
        size_t b = a;
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        while (b < n && keyed[b].first == keyed[a].first) {
            const Pt& p = in[keyed[b].second];
            sx += p.x; sy += p.y; sz += p.z; si += p.i;
            ++b;
        }
        const float cnt = (float)(b - a);
        Pt c; c.x = sx / cnt; c.y = sy / cnt; c.z = sz / cnt; c.i = si / cnt;
        res.push_back(c);
        a = b;
    }
    out.swap(res);
}

/* ---------------------------------------------------------------------------------------
 * Exact k-NN.  Stands in for pcl::KdTreeFLANN -> FLANN 1.9.1 KDTreeSingleIndex<L2_Simple<float>>
 * (Session.cpp:404,457,489 build; :471,592,627 query).  Only the k smallest squared distances
 * matter to the caller; they are computed as FLANN's L2_Simple does: float, ((dx*dx)+dy*dy)+dz*dz.
 * The tree below is an independent median-split kd-tree whose pruning is conservative, so the
 * returned multiset of distances is exact.  "parity unpinned".
 * ------------------------------------------------------------------------------------- */
inline float sqdist_l2simple(const Pt& q, const Pt& t)
{
    float dx = q.x - t.x, dy = q.y - t.y, dz = q.z - t.z;
    float r = dx * dx;
    r += dy * dy;
    r += dz * dz;
    return r;
}

struct KdTree {
    struct Node { int left, right; int dim; float split; int lo, hi; };
    std::vector<Node> nodes;
    std::vector<Pt> pts; /* reordered */
    static const int kLeaf = 15;

    void build(const Pt* p, size_t n)
    {
        pts.assign(p, p + n);
        nodes.clear();
        if (n) { nodes.reserve(2 * n / kLeaf + 8); build_rec(0, (int)n); }
    }
    int build_rec(int lo, int hi)
    {
        int id = (int)nodes.size();
        nodes.push_back(Node{-1, -1, -1, 0.f, lo, hi});
        if (hi - lo <= kLeaf) return id;
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = lo; i < hi; ++i) {
            const float c[3] = {pts[i].x, pts[i].y, pts[i].z};
            for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], c[d]); mx[d] = std::max(mx[d], c[d]); }
        }
        int dim = 0;
        if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
        if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        if (!(mx[dim] > mn[dim])) return id; /* all identical: keep as (big) leaf */
        int mid = (lo + hi) / 2;
        auto key = [dim](const Pt& a) { return dim == 0 ? a.x : (dim == 1 ? a.y : a.z); };
        std::nth_element(pts.begin() + lo, pts.begin() + mid, pts.begin() + hi,
                         [&](const Pt& a, const Pt& b) { return key(a) < key(b); });
        float split = key(pts[mid]);
        int l = build_rec(lo, mid);
        int r = build_rec(mid, hi);
        nodes[id].left = l; nodes[id].right = r; nodes[id].dim = dim; nodes[id].split = split;
        return id;
    }
    /* best[0..k) ascending; cnt = number filled */
    void search(const Pt& q, int k, float* best, int& cnt) const
    {
        cnt = 0;
        if (!nodes.empty()) search_rec(0, q, k, best, cnt);
    }
    static inline void push(float d, int k, float* best, int& cnt)
    {
        if (cnt == k && !(d < best[k - 1])) return;
        int j = (cnt < k) ? cnt++ : k - 1;
        while (j > 0 && best[j - 1] > d) { best[j] = best[j - 1]; --j; }
        best[j] = d;
    }
    void search_rec(int id, const Pt& q, int k, float* best, int& cnt) const
    {
        const Node& nd = nodes[id];
        if (nd.left < 0) {
            for (int i = nd.lo; i < nd.hi; ++i) push(sqdist_l2simple(q, pts[i]), k, best, cnt);
            return;
        }
        const float qc = nd.dim == 0 ? q.x : (nd.dim == 1 ? q.y : q.z);
        const int nearc = (qc < nd.split) ? nd.left : nd.right;
        const int farc = (qc < nd.split) ? nd.right : nd.left;
        search_rec(nearc, q, k, best, cnt);
        const double d = (double)qc - (double)nd.split;
        /* conservative: float d^2 of any far-side point >= d*d*(1-2^-20) */
        if (cnt < k || d * d * (1.0 - 1e-6) < (double)best[k - 1]) search_rec(farc, q, k, best, cnt);
    }
};

struct Knn {
    KdTree tree;
    const Pt* raw = nullptr; size_t n = 0; bool use_tree = true;
    void build(const Pt* p, size_t n_, bool use_tree_) { raw = p; n = n_; use_tree = use_tree_; if (use_tree) tree.build(p, n_); }
    /* Session.cpp:588-599 / :625-634 / :466-479 : returns the "coexist"/"near" predicate */
    bool near(const Pt& q, int k_param, float thr) const
    {
        /* pcl::KdTreeFLANN::nearestKSearch clamps k to the number of points */
        if (n == 0) return false; /* reference: undefined (PCL refuses an empty tree); defined here as "far" */
        int k = (int)std::min<size_t>((size_t)k_param, n);
        float best[64]; int cnt = 0;
        if (k > 64) k = 64;
        if (k > 0) {
            if (use_tree) tree.search(q, k, best, cnt);
            else for (size_t i = 0; i < n; ++i) KdTree::push(sqdist_l2simple(q, raw[i]), k, best, cnt);
        }
        /* float sum = accumulate(begin,end,0.0) : double accumulation in ascending order, then to float */
        double acc = 0.0;
        for (int j = 0; j < cnt; ++j) acc += (double)best[j];
        const float sum = (float)acc;
        const float avg = sum / (float)k_param;
        return std::fabs(avg) < thr;
    }
};

/* utility.cpp:160-168 local2global(scan, pose, M): transform by M then by pose */
inline Pt local2global_pt(const double* pose, const double* first, const Pt& p) { return xform(pose, xform(first, p)); }

/* -------------------------------------------------------------------------------------- */
struct Sess {
    std::vector<Cloud> scans;                 /* keyframe_scans_ (after load + pre-clean) */
    std::vector<double> poses, inv;           /* 16 doubles per keyframe */
    size_t nkf() const { return scans.size(); }
    const double* pose(size_t i) const { return &poses[16 * i]; }
    const double* ipose(size_t i) const { return &inv[16 * i]; }
    Cloud map_orig, map_curr, map_static, map_dynamic;
    std::vector<Cloud> scans_static_projected, scans_dynamic, knn_coexist, knn_diff;
    std::vector<Cloud> scans_updated, scans_updated_strong, scans_pd, scans_strong_pd, scans_strong_nd, scans_weak_nd;
    Cloud nd, nd_strong, nd_weak, pd, pd_orig, pd_strong, pd_weak, updated, updated_strong;
};

struct Ctx {
    orc_params P;
    double L2B[16], B2L[16];
    std::map<std::string, double> t;
    std::vector<std::string> t_order;
    void tick(const std::string& name, double dt) { if (!t.count(name)) t_order.push_back(name); t[name] += dt; }
};

void vote_labels(const Ctx& c, const Cloud& map, const std::vector<Cloud>& scans, const Sess& src,
                 float alpha, float thr, int mode, std::vector<uint8_t>& labels)
{
    int R, C;
    rimg_size(c.P.vfov, c.P.hfov, alpha, &R, &C);
    labels.assign(map.size(), 0);
    const int stride = std::max(1, c.P.kf_sample_stride);
    const long nk = (long)scans.size();
#pragma omp parallel num_threads(std::max(1, c.P.threads))
    {
        std::vector<float> a, b; std::vector<int32_t> ix;
        std::vector<uint8_t> local(map.size(), 0);
#pragma omp for schedule(dynamic, 1)
        for (long kf = 0; kf < nk; kf += stride)
            vote_one_kf(map.data(), map.size(), scans[kf].data(), scans[kf].size(), src.ipose(kf), c.B2L,
                        c.P.vfov, c.P.hfov, R, C, thr, mode, a, b, ix, local.data());
#pragma omp critical
        for (size_t i = 0; i < labels.size(); ++i) labels[i] |= local[i];
    }
}

/* Removerter.cpp:801-828 (+ :675-687 complement, :933-946 ExtractIndices): ascending-index gather */
void partition(const Cloud& map, const std::vector<uint8_t>& labels, Cloud& kept, Cloud& flagged)
{
    Cloud k, f;
    for (size_t i = 0; i < map.size(); ++i) (labels[i] ? f : k).push_back(map[i]);
    kept.swap(k); flagged.swap(f);
}

void append(Cloud& a, const Cloud& b) { a.insert(a.end(), b.begin(), b.end()); }

/* Removerter.cpp:882-905 */
void remove_once(Ctx& c, Sess& tgt, const Sess& src, float res)
{
    double t0 = now_s();
    std::vector<uint8_t> lab; Cloud st, dy;
    vote_labels(c, tgt.map_curr, src.scans, src, res, 0.1f, 0, lab);
    partition(tgt.map_curr, lab, st, dy);
    double t1 = now_s();
    voxel_centroid(st, 0.05f, tgt.map_static);
    tgt.map_curr = tgt.map_static;
    append(tgt.map_dynamic, dy);
    voxel_centroid(tgt.map_dynamic, 0.05f, tgt.map_dynamic);
    c.tick("vote_large", t1 - t0); c.tick("voxel", now_s() - t1);
}

/* Removerter.cpp:908-931 */
void revert_once(Ctx& c, Sess& tgt, const Sess& src, float res)
{
    double t0 = now_s();
    std::vector<uint8_t> lab; Cloud st, dy;
    vote_labels(c, tgt.map_curr, src.scans, src, res, 0.1f, 0, lab);
    partition(tgt.map_curr, lab, st, dy);
    double t1 = now_s();
    voxel_centroid(dy, 0.05f, tgt.map_dynamic);
    tgt.map_curr = tgt.map_dynamic;
    append(tgt.map_static, st);
    voxel_centroid(tgt.map_static, 0.05f, tgt.map_static);
    c.tick("vote_small", t1 - t0); c.tick("voxel", now_s() - t1);
}

/* Removerter.cpp:1378-1393 */
void self_removert(Ctx& c, Sess& s)
{
    for (int r = 0; r < c.P.n_res; ++r) {
        const float res = c.P.res_list[r];
        for (int i = 0; i < c.P.repeat; ++i) {              /* `i < _repeat`, :1381 */
            remove_once(c, s, s, res);
            s.map_curr = s.map_dynamic;                     /* resetCurrrentMapAsDynamic :714-737 */
            revert_once(c, s, s, (float)(0.95 * res));      /* :1385 double product narrowed to the float parameter */
            s.map_curr = s.map_static;                      /* resetCurrrentMapAsStatic */
            remove_once(c, s, s, res);
        }
    }
}

/* utility.cpp:170-192 */
void merge_to_global(const Ctx& c, const std::vector<Cloud>& scans, const Sess& s, Cloud& out)
{
    Cloud r;
    for (size_t kf = 0; kf < scans.size(); ++kf)
        for (const Pt& p : scans[kf]) r.push_back(xform(s.pose(kf), xform(c.L2B, p)));
    out.swap(r);
}

/* Session.cpp:348-360 + utility.cpp:74-89 */
void reproject_kf(const Ctx& c, const Cloud& map, const double* Tinv, int R, int C,
                  std::vector<float>& rimg, std::vector<int32_t>& idx, Cloud& out)
{
    const size_t npx = (size_t)R * C;
    rimg.resize(npx); idx.resize(npx);
    map2rimg(map.data(), map.size(), Tinv, c.B2L, c.P.vfov, c.P.hfov, R, C, rimg.data(), idx.data());
    out.clear();
    for (size_t px = 0; px < npx; ++px) {
        if (idx[px] == 0) continue; /* utility.cpp:82 : 0 doubles as "no point" (quirk Q3) */
        out.push_back(global_to_local_pt(Tinv, c.B2L, map[idx[px]]));
    }
}

void reproject(Ctx& c, const Cloud& map, const Sess& s, std::vector<Cloud>& out, const char* tag)
{
    double t0 = now_s();
    int R, C;
    rimg_size(c.P.vfov, c.P.hfov, kReprojectionAlpha, &R, &C);
    out.assign(s.nkf(), Cloud());
    const int stride = std::max(1, c.P.kf_sample_stride);
    const long nk = (long)s.nkf();
#pragma omp parallel num_threads(std::max(1, c.P.threads))
    {
        std::vector<float> rimg; std::vector<int32_t> idx;
#pragma omp for schedule(dynamic, 1)
        for (long kf = 0; kf < nk; kf += stride) reproject_kf(c, map, s.ipose(kf), R, C, rimg, idx, out[kf]);
    }
    c.tick(tag, now_s() - t0);
}

/* Session.cpp:537-607 (LD) and :610-642 (HD): identical arithmetic on different scan sets */
void knn_partition(Ctx& c, const Cloud& target, const std::vector<Cloud>& scans, const Sess& s,
                   std::vector<Cloud>* coexist, std::vector<Cloud>* diff, bool use_tree = true)
{
    double t0 = now_s();
    Knn knn; knn.build(target.data(), target.size(), use_tree);
    double t1 = now_s();
    if (coexist) coexist->assign(scans.size(), Cloud());
    if (diff) diff->assign(scans.size(), Cloud());
    const int stride = std::max(1, c.P.kf_sample_stride);
    const long nk = (long)scans.size();
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, c.P.threads))
    for (long kf = 0; kf < nk; kf += stride) {
        Cloud co, di;
        for (const Pt& p : scans[kf]) {
            /* :545 / :618 pass kSE3MatExtrinsicPoseBasetoLiDAR where LiDAR->base is expected (quirk Q7) */
            const Pt g = local2global_pt(s.pose(kf), c.B2L, p);
            const Pt l = global_to_local_pt(s.ipose(kf), c.B2L, g);
            if (knn.near(g, c.P.k, c.P.knn_thr)) co.push_back(l); else di.push_back(l);
        }
        if (coexist) (*coexist)[kf].swap(co);
        if (diff) (*diff)[kf].swap(di);
    }
    c.tick("knn_build", t1 - t0); c.tick("knn_query", now_s() - t1);
}

struct Run {
    Ctx c; Sess C, Q;
    std::map<std::string, Cloud> clouds;
    std::map<std::string, std::vector<Cloud>> scansets;
    /* flattened scanset storage for the C ABI */
    std::map<std::string, std::pair<Cloud, std::vector<uint64_t>>> flat;
};

void vd(Cloud& c) { voxel_centroid(c, 0.05f, c); }

/* Removerter.cpp:831-854 (ND) and :856-880 (PD) */
void remove_once_ldmap(Ctx& c, Cloud& map, Cloud& strong, Cloud& weak, const Sess& src, float res, int mode)
{
    double t0 = now_s();
    std::vector<uint8_t> lab; Cloud st, dy;
    vote_labels(c, map, src.scans_static_projected, src, res, 0.1f, mode, lab);
    partition(map, lab, st, dy);
    double t1 = now_s();
    voxel_centroid(st, 0.05f, strong);
    map = strong;
    append(weak, dy);
    vd(weak);
    c.tick("vote_small", t1 - t0); c.tick("voxel", now_s() - t1);
}

void load_session(Sess& s, const float* scans, const uint64_t* off, size_t nkf, const double* poses, const double* inv)
{
    s.scans.resize(nkf);
    for (size_t k = 0; k < nkf; ++k) {
        const Pt* b = reinterpret_cast<const Pt*>(scans) + off[k];
        s.scans[k].assign(b, b + (off[k + 1] - off[k]));
    }
    s.poses.assign(poses, poses + 16 * nkf);
    s.inv.assign(inv, inv + 16 * nkf);
}

/* Eigen::Matrix4d::inverse() (Session.cpp:109-110, RosParamServer.cpp:29-30) as Eigen 3.3.7 evaluates it in the reference's
 * SSE2 build (Eigen/src/LU/arch/Inverse_SSE.h, double specialisation; restated from knowledge of its structure -- PARITY
 * UNPINNED): the column-major matrix is read in memory order as four 2x2 blocks A B / C D of N = M^T; with X# the adjugate,
 *   AB = A#B, DC = D#C, det = |A||D| + |B||C| - trace(AB DC),
 *   inverse blocks = (A|D| - B DC)#, (C|B| - D AB#)#, (B|C| - A DC#)#, (D|A| - C AB)#, each times +-1/det,
 * every product and sum a separately rounded double operation in the order written (SSE2 has no FMA).  Row-major in/out. */
int inverse4x4(const double* m, double* inv)
{
    double A[2][2], B[2][2], C[2][2], D[2][2];
    for (int r = 0; r < 2; ++r)
        for (int k = 0; k < 2; ++k) {
            A[r][k] = m[4 * k + r]; B[r][k] = m[4 * (k + 2) + r];
            C[r][k] = m[4 * k + r + 2]; D[r][k] = m[4 * (k + 2) + r + 2];
        }
    const double dA = A[0][0] * A[1][1] - A[0][1] * A[1][0], dB = B[0][0] * B[1][1] - B[0][1] * B[1][0];
    const double dC = C[0][0] * C[1][1] - C[0][1] * C[1][0], dD = D[0][0] * D[1][1] - D[0][1] * D[1][0];
    double AB[2][2], DC[2][2], iA[2][2], iB[2][2], iC[2][2], iD[2][2];
    for (int j = 0; j < 2; ++j) {
        AB[0][j] = B[0][j] * A[1][1] - B[1][j] * A[0][1]; AB[1][j] = B[1][j] * A[0][0] - B[0][j] * A[1][0];
        DC[0][j] = C[0][j] * D[1][1] - C[1][j] * D[0][1]; DC[1][j] = C[1][j] * D[0][0] - C[0][j] * D[1][0];
    }
    const double tr = (AB[0][0] * DC[0][0] + AB[1][0] * DC[0][1]) + (AB[0][1] * DC[1][0] + AB[1][1] * DC[1][1]);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            const double cab = AB[0][j] * C[i][0] + AB[1][j] * C[i][1];
            const double bdc = DC[0][j] * B[i][0] + DC[1][j] * B[i][1];
            iD[i][j] = D[i][j] * dA - cab;
            iA[i][j] = A[i][j] * dD - bdc;
        }
    for (int i = 0; i < 2; ++i) {
        iB[i][0] = D[i][0] * AB[1][1] - D[i][1] * AB[1][0]; iB[i][1] = D[i][1] * AB[0][0] - D[i][0] * AB[0][1];
        iC[i][0] = A[i][0] * DC[1][1] - A[i][1] * DC[1][0]; iC[i][1] = A[i][1] * DC[0][0] - A[i][0] * DC[0][1];
    }
    const double d1 = dA * dD, d2 = dB * dC;
    const double det = (d1 + d2) - tr;
    if (det == 0.0 || det != det) return -1;
    const double rd = 1.0 / det, nrd = -rd;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) { iB[i][j] = C[i][j] * dB - iB[i][j]; iC[i][j] = B[i][j] * dC - iC[i][j]; }
    const double (*blk[4])[2] = {iA, iB, iC, iD};
    const int at[4][2] = {{0, 0}, {0, 2}, {2, 0}, {2, 2}};
    for (int q = 0; q < 4; ++q) {
        const double (*X)[2] = blk[q];
        const int r = at[q][0], c = at[q][1];       /* (r, c) of N^-1 is element (c, r) of M^-1 */
        inv[4 * c + r] = X[1][1] * rd; inv[4 * (c + 1) + r] = X[0][1] * nrd;
        inv[4 * c + r + 1] = X[1][0] * nrd; inv[4 * (c + 1) + r + 1] = X[0][0] * rd;
    }
    return 0;
}

/* the same inverse by cofactor expansion (what rounds 1-2 of this oracle used) and by Gauss-Jordan elimination with partial
 * pivoting: only for tests/test_numerics_sensitivity.py, which measures what the last bits of the inverse can move */
int inverse4x4_cofactor(const double* m, double* inv)
{
    double a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
    if (det == 0.0) return -1;
    det = 1.0 / det;
    for (int i = 0; i < 16; ++i) inv[i] = a[i] * det;
    return 0;
}

int inverse4x4_gauss_jordan(const double* m, double* inv)
{
    double a[4][8];
    for (int r = 0; r < 4; ++r) for (int k = 0; k < 4; ++k) { a[r][k] = m[4 * r + k]; a[r][4 + k] = (r == k) ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) return -1;
        if (piv != col) for (int k = 0; k < 8; ++k) std::swap(a[piv][k], a[col][k]);
        const double d = a[col][col];
        for (int k = 0; k < 8; ++k) a[col][k] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
            if (f != 0.0) for (int k = 0; k < 8; ++k) a[r][k] -= f * a[col][k];
        }
    }
    for (int r = 0; r < 4; ++r) for (int k = 0; k < 4; ++k) inv[4 * r + k] = a[r][4 + k];
    return 0;
}

void flatten(Run& r, const std::string& name, const std::vector<Cloud>& ss)
{
    auto& f = r.flat[name];
    f.first.clear(); f.second.assign(1, 0);
    for (const Cloud& c : ss) { append(f.first, c); f.second.push_back(f.first.size()); }
}

} // namespace

/* ========================================== C ABI ========================================== */
extern "C" {

float orc_atan2f(float y, float x) { return om_atan2f(y, x); }
void orc_atan2f_array(const float* y, const float* x, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = om_atan2f(y[i], x[i]); }
float orc_rad2deg(float rad) { return rad2deg(rad); }
void orc_cart2sph(const float* p, float* o) { Sph s = cart2sph(p[0], p[1], p[2]); o[0] = s.az; o[1] = s.el; o[2] = s.r; }
void orc_rimg_size(float vfov, float hfov, float alpha, int* R, int* C) { rimg_size(vfov, hfov, alpha, R, C); }
void orc_pixel(const float* p, float vfov, float hfov, int R, int C, int* row, int* col, float* range)
{
    Sph s = cart2sph(p[0], p[1], p[2]);
    pixel_of(s, vfov, hfov, R, C, row, col);
    if (range) *range = s.r;
}

void orc_transform(const double* T, const float* in, float* out, size_t n)
{
    const Pt* a = reinterpret_cast<const Pt*>(in); Pt* b = reinterpret_cast<Pt*>(out);
    for (size_t i = 0; i < n; ++i) b[i] = xform(T, a[i]);
}
int orc_inverse4x4(const double* m, double* inv) { return inverse4x4(m, inv); }
void orc_set_atan2f_perturbation(unsigned ppm, uint64_t seed) { g_atan_perturb_ppm = ppm; g_atan_perturb_seed = seed; }
int orc_inverse4x4_variant(const double* m, double* inv, int variant)
{
    return variant == 1 ? inverse4x4_cofactor(m, inv) : variant == 2 ? inverse4x4_gauss_jordan(m, inv) : inverse4x4(m, inv);
}

void orc_range_image(const float* pts, size_t n, const double* T1, const double* T2, float vfov, float hfov,
                     int R, int C, float* rimg, int32_t* ptidx)
{
    map2rimg(reinterpret_cast<const Pt*>(pts), n, T1, T2, vfov, hfov, R, C, rimg, ptidx);
}

void orc_vote_labels(const float* map, size_t M, const float* scans, const uint64_t* off, size_t n_kf,
                     const double* inv_poses, const double* b2l, float vfov, float hfov, float alpha, float thr,
                     int mode, size_t kf_begin, size_t kf_end, int threads, uint8_t* labels)
{
    (void)n_kf;
    int R, C; rimg_size(vfov, hfov, alpha, &R, &C);
    const Pt* mp = reinterpret_cast<const Pt*>(map);
    const Pt* sp = reinterpret_cast<const Pt*>(scans);
#pragma omp parallel num_threads(std::max(1, threads))
    {
        std::vector<float> a, b; std::vector<int32_t> ix; std::vector<uint8_t> local(M, 0);
#pragma omp for schedule(dynamic, 1)
        for (long kf = (long)kf_begin; kf < (long)kf_end; ++kf)
            vote_one_kf(mp, M, sp + off[kf], off[kf + 1] - off[kf], inv_poses + 16 * kf, b2l, vfov, hfov, R, C, thr, mode,
                        a, b, ix, local.data());
#pragma omp critical
        for (size_t i = 0; i < M; ++i) labels[i] |= local[i];
    }
}

/* multi-rank tests only: the voxel grid of `pts` under the octree frame of the box mn_mx[6] (of a larger cloud the points are part of), and the
 * points' Morton keys under that frame */
size_t orc_voxel_centroid_box(const float* pts, size_t n, const float* mn_mx, float leaf, float* out, size_t cap)
{
    Cloud in(reinterpret_cast<const Pt*>(pts), reinterpret_cast<const Pt*>(pts) + n), o;
    voxel_centroid(in, leaf, o, mn_mx);
    for (size_t i = 0; i < o.size() && i < cap && out; ++i) reinterpret_cast<Pt*>(out)[i] = o[i];
    return o.size();
}
int orc_voxel_keys_box(const float* pts, size_t n, const float* mn_mx, float leaf, uint64_t* keys)
{
    const OctreeFrame f = octree_frame_of_box(mn_mx[0], mn_mx[1], mn_mx[2], mn_mx[3], mn_mx[4], mn_mx[5], leaf);
    if (!f.ok) return -1;
    const Pt* p = reinterpret_cast<const Pt*>(pts);
    for (size_t i = 0; i < n; ++i)
        keys[i] = morton_xyz((unsigned)(((double)p[i].x - f.minx) / f.res), (unsigned)(((double)p[i].y - f.miny) / f.res), (unsigned)(((double)p[i].z - f.minz) / f.res), f.depth);
    return (int)f.depth;
}

size_t orc_voxel_centroid(const float* pts, size_t n, float leaf, float* out, size_t cap)
{
    Cloud in(reinterpret_cast<const Pt*>(pts), reinterpret_cast<const Pt*>(pts) + n), o;
    voxel_centroid(in, leaf, o);
    if (out) memcpy(out, o.data(), std::min(cap, o.size()) * sizeof(Pt));
    return o.size();
}

size_t orc_reproject(const float* map, size_t M, const double* inv_poses, const double* b2l, float vfov, float hfov,
                     float alpha, size_t kf_begin, size_t kf_end, int threads, float* out, size_t cap, uint64_t* out_offsets)
{
    Ctx c; memset(&c.P, 0, sizeof c.P); c.P.vfov = vfov; c.P.hfov = hfov; memcpy(c.B2L, b2l, sizeof c.B2L);
    Cloud mp(reinterpret_cast<const Pt*>(map), reinterpret_cast<const Pt*>(map) + M);
    int R, C; rimg_size(vfov, hfov, alpha, &R, &C);
    const long nk = (long)(kf_end - kf_begin);
    std::vector<Cloud> res(nk);
#pragma omp parallel num_threads(std::max(1, threads))
    {
        std::vector<float> rimg; std::vector<int32_t> idx;
#pragma omp for schedule(dynamic, 1)
        for (long j = 0; j < nk; ++j) reproject_kf(c, mp, inv_poses + 16 * (kf_begin + j), R, C, rimg, idx, res[j]);
    }
    size_t tot = 0;
    out_offsets[0] = 0;
    for (long j = 0; j < nk; ++j) {
        if (out) for (const Pt& p : res[j]) { if (tot < cap) reinterpret_cast<Pt*>(out)[tot] = p; ++tot; }
        else tot += res[j].size();
        out_offsets[j + 1] = tot;
    }
    return tot;
}

void orc_knn_labels(const float* target, size_t Mt, const float* scans, const uint64_t* off, size_t n_kf,
                    const double* poses, const double* inv_poses, const double* b2l, int k, float thr,
                    size_t kf_begin, size_t kf_end, int threads, int use_kdtree, uint8_t* coexist, float* local_out)
{
    (void)n_kf;
    Knn knn; knn.build(reinterpret_cast<const Pt*>(target), Mt, use_kdtree != 0);
    const Pt* sp = reinterpret_cast<const Pt*>(scans);
    Pt* lo = reinterpret_cast<Pt*>(local_out);
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, threads))
    for (long kf = (long)kf_begin; kf < (long)kf_end; ++kf)
        for (uint64_t i = off[kf]; i < off[kf + 1]; ++i) {
            const Pt g = local2global_pt(poses + 16 * kf, b2l, sp[i]);
            if (lo) lo[i] = global_to_local_pt(inv_poses + 16 * kf, b2l, g);
            coexist[i] = knn.near(g, k, thr) ? 1 : 0;
        }
}

void orc_knn_split(const float* target, size_t Mt, const float* query, size_t Q, int k, float thr, int use_kdtree, uint8_t* near)
{
    Knn knn; knn.build(reinterpret_cast<const Pt*>(target), Mt, use_kdtree != 0);
    const Pt* q = reinterpret_cast<const Pt*>(query);
    for (size_t i = 0; i < Q; ++i) near[i] = knn.near(q[i], k, thr) ? 1 : 0;
}

void orc_merge_to_global(const float* scans, const uint64_t* off, size_t n_kf, const double* poses, const double* l2b, float* out)
{
    const Pt* sp = reinterpret_cast<const Pt*>(scans); Pt* o = reinterpret_cast<Pt*>(out);
    for (size_t kf = 0; kf < n_kf; ++kf)
        for (uint64_t i = off[kf]; i < off[kf + 1]; ++i) o[i] = xform(poses + 16 * kf, xform(l2b, sp[i]));
}

/* Session.cpp:506-533: drop iff range < radius & z < 0.5 & -0.5 < z */
size_t orc_preclean(const float* pts, size_t n, float radius, float* out)
{
    const Pt* p = reinterpret_cast<const Pt*>(pts); Pt* o = reinterpret_cast<Pt*>(out);
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        const float r = cart2sph(p[i].x, p[i].y, p[i].z).r;
        if ((r < radius) & (p[i].z < 0.5f) & (-0.5f < p[i].z)) continue;
        if (o) o[m] = p[i];
        ++m;
    }
    return m;
}

/* pcl::VoxelGrid<PointXYZI>::applyFilter as the loader uses it (Session.cpp:284-289; PCL 1.10 voxel_grid.hpp, restated from
 * its published behaviour -- PARITY UNPINNED): inverse leaf size in float; getMinMax3D; the "leaf size is too small" test
 * (dx*dy*dz > INT32_MAX with d = (int64)((max-min)*inv_leaf) + 1) returns the INPUT unchanged -- the common case for a raw
 * 0.05 m scan; otherwise min_b / div_b from floor(min*inv), floor(max*inv), leaf index ijk0 + ijk1*div0 + ijk2*div0*div1 with
 * ijk = (int)(floor(x*inv) - (float)min_b), points grouped by std::sort on the LEAF INDEX ONLY (PCL's cloud_point_index_idx::operator<):
 * the order inside a voxel is what the C++ library's std::sort leaves, and the same call is made here -- round 4: with an input-order
 * (stable) sort the oracle differed from the reference's own sources compiled against stand-in headers (oracle/_ref) in the last bit
 * of ~0.05 % of the loaded os1-64 points; with this call it is bit-identical.  orc_set_voxel_grid_stable(1) selects input order, which
 * is what the device form of the cascade hand-over (ltm_voxel_grid_scanset) sums in.  Float sums (CentroidPoint accumulators),
 * divided by the count, output in ascending leaf index.  min_points_per_voxel = 0, all fields downsampled.  Returns the output
 * count (out may be NULL). */
int g_voxel_grid_stable = 0;
void orc_set_voxel_grid_stable(int stable) { g_voxel_grid_stable = stable; }
size_t orc_voxel_grid(const float* pts, size_t n, float leaf, float* out, size_t cap)
{
    const Pt* p = reinterpret_cast<const Pt*>(pts); Pt* o = reinterpret_cast<Pt*>(out);
    if (n == 0) return 0;
    const float inv = 1.0f / leaf;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n; ++i) {
        const float c[3] = {p[i].x, p[i].y, p[i].z};
        for (int d = 0; d < 3; ++d) { if (c[d] < mn[d]) mn[d] = c[d]; if (c[d] > mx[d]) mx[d] = c[d]; }
    }
    int64_t dd[3];
    for (int d = 0; d < 3; ++d) dd[d] = (int64_t)((mx[d] - mn[d]) * inv) + 1;
    if (dd[0] * dd[1] * dd[2] > (int64_t)INT32_MAX) {
        for (size_t i = 0; i < n && i < cap && o; ++i) o[i] = p[i];
        return n;
    }
    int min_b[3], div_b[3];
    for (int d = 0; d < 3; ++d) {
        min_b[d] = (int)std::floor(mn[d] * inv);
        div_b[d] = (int)std::floor(mx[d] * inv) - min_b[d] + 1;
    }
    std::vector<std::pair<unsigned, unsigned>> iv(n);
    for (size_t i = 0; i < n; ++i) {
        const int i0 = (int)(std::floor(p[i].x * inv) - (float)min_b[0]);
        const int i1 = (int)(std::floor(p[i].y * inv) - (float)min_b[1]);
        const int i2 = (int)(std::floor(p[i].z * inv) - (float)min_b[2]);
        iv[i] = {(unsigned)(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]), (unsigned)i};
    }
    const auto by_leaf = [](const std::pair<unsigned, unsigned>& a, const std::pair<unsigned, unsigned>& b) { return a.first < b.first; };
    if (g_voxel_grid_stable) std::stable_sort(iv.begin(), iv.end(), by_leaf);
    else std::sort(iv.begin(), iv.end(), by_leaf);      /* PCL: std::sort with cloud_point_index_idx::operator< (leaf index only) */
    size_t m = 0;
    for (size_t a = 0; a < n;) {
        size_t b = a;
        float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
        while (b < n && iv[b].first == iv[a].first) {
            const Pt& q = p[iv[b].second];
            sx = sx + q.x; sy = sy + q.y; sz = sz + q.z; si = si + q.i;
            ++b;
        }
        const float cnt = (float)(b - a);
        if (o && m < cap) o[m] = Pt{sx / cnt, sy / cnt, sz / cnt, si / cnt};
        ++m;
        a = b;
    }
    return m;
}

/* Removerter::run() Steps 0(makeGlobalMap only)..3, Removerter.cpp:1653-1678 */
orc_run* orc_pipeline_run(const orc_params* p,
                          const float* c_scans, const uint64_t* c_off, size_t c_nkf, const double* c_poses, const double* c_inv,
                          const float* q_scans, const uint64_t* q_off, size_t q_nkf, const double* q_poses, const double* q_inv)
{
    Run* r = new Run();
    Ctx& c = r->c;
    c.P = *p;
    memcpy(c.L2B, p->lidar2base, sizeof c.L2B);
    if (inverse4x4(c.L2B, c.B2L) != 0) { delete r; return nullptr; }
    if (is_identity(c.L2B)) memcpy(c.B2L, c.L2B, sizeof c.B2L);
    Sess& C = r->C; Sess& Q = r->Q;
    load_session(C, c_scans, c_off, c_nkf, c_poses, c_inv);
    load_session(Q, q_scans, q_off, q_nkf, q_poses, q_inv);

    /* Step 0: makeGlobalMap, Removerter.cpp:213-252 (+ Session.cpp:186-202) */
    double t0 = now_s();
    for (Sess* s : {&C, &Q}) {
        merge_to_global(c, s->scans, *s, s->map_orig);
        voxel_centroid(s->map_orig, c.P.voxel, s->map_curr);
    }
    r->clouds["OriginalNoisyCentralMapGlobal"] = C.map_curr;
    r->clouds["OriginalNoisyQueryMapGlobal"] = Q.map_curr;
    c.tick("step0_make_global_map", now_s() - t0);
    const double t_steps = now_s();

    /* Step 1: removeHighDynamicPoints, Removerter.cpp:1580-1604 */
    if (c.P.use_self_removert && c.P.n_res > 0) { self_removert(c, C); self_removert(c, Q); }
    else { remove_once(c, C, C, 2.5f); remove_once(c, Q, Q, 2.5f); }
    r->clouds["central_map_static"] = C.map_static; r->clouds["central_map_dynamic"] = C.map_dynamic;
    r->clouds["query_map_static"] = Q.map_static;   r->clouds["query_map_dynamic"] = Q.map_dynamic;
    if (!c.P.skip_hd_knn) {
        for (Sess* s : {&C, &Q}) {
            knn_partition(c, s->map_static, s->scans, *s, nullptr, &s->scans_dynamic);   /* Session.cpp:487-504 */
            double t1 = now_s();
            Cloud hd; merge_to_global(c, s->scans_dynamic, *s, hd); vd(hd);
            r->clouds[s == &C ? "central_sess_high_dyn" : "query_sess_high_dyn"] = hd;
            c.tick("voxel", now_s() - t1);
        }
    }
    /* parseStaticScansViaProjection, Removerter.cpp:1534-1545 / Session.cpp:305-309 */
    reproject(c, C.map_curr, C, C.scans_static_projected, "reproject_large");
    reproject(c, Q.map_curr, Q, Q.scans_static_projected, "reproject_large");

    /* Step 2: detectLowDynamicPoints, Removerter.cpp:1413-1481 */
    {   /* Session.cpp:401 : a 0.4 m octree of the target for a disabled ICP -- executed, result unused (quirk Q11) */
        double t1 = now_s(); Cloud unused;
        voxel_centroid(Q.map_static, 0.4f, unused); voxel_centroid(C.map_static, 0.4f, unused);
        c.tick("voxel", now_s() - t1);
    }
    knn_partition(c, Q.map_static, C.scans_static_projected, C, &C.knn_coexist, &C.knn_diff);
    knn_partition(c, C.map_static, Q.scans_static_projected, Q, &Q.knn_coexist, &Q.knn_diff);

    double t1 = now_s();
    merge_to_global(c, C.knn_diff, C, C.nd); vd(C.nd);                                   /* constructGlobalNDMap :430-435 */
    c.tick("voxel", now_s() - t1);
    for (int i = 0; i < 3; ++i) remove_once_ldmap(c, C.nd, C.nd_strong, C.nd_weak, Q, 2.5f, 1);  /* filterStrongND :1403-1411 */
    t1 = now_s();
    if (!C.nd_strong.empty()) {                                                          /* Session.cpp:452-484 */
        Knn knn; knn.build(C.nd_strong.data(), C.nd_strong.size(), true);
        Cloud add, nw;
        for (const Pt& pt : C.nd_weak) (knn.near(pt, 2, 1.0f) ? add : nw).push_back(pt);
        append(C.nd_strong, add);
        C.nd_weak.swap(nw);
    }
    c.tick("knn_weak_strong", now_s() - t1);

    t1 = now_s();
    merge_to_global(c, Q.knn_diff, Q, Q.pd); vd(Q.pd); Q.pd_orig = Q.pd;                 /* constructGlobalPDMap :437-445 */
    c.tick("voxel", now_s() - t1);
    for (int i = 0; i < 3; ++i) remove_once_ldmap(c, Q.pd, Q.pd_strong, Q.pd_weak, C, 2.5f, 0);  /* filterStrongPD :1395-1401 */
    C.pd = Q.pd; C.pd_orig = Q.pd_orig; C.pd_strong = Q.pd_strong;                       /* :1435-1437 */

    t1 = now_s();
    {   /* :1443-1480 debug/visualisation maps (several of them mutate state that Step 3 reads) */
        Cloud m;
        merge_to_global(c, Q.knn_coexist, Q, m); vd(m); r->clouds["union_map_queryside"] = m;
        merge_to_global(c, C.knn_coexist, C, m); vd(m); r->clouds["union_map_centralside"] = m;
        merge_to_global(c, Q.knn_diff, Q, m); vd(m); r->clouds["pd_map"] = m;
        merge_to_global(c, C.knn_diff, C, m); vd(m); r->clouds["nd_map"] = m;
        if (!C.nd_strong.empty()) { vd(C.nd_strong); r->clouds["strong_nd_map"] = C.nd_strong; }
        vd(C.nd_weak); r->clouds["weak_nd_map"] = C.nd_weak;
        vd(Q.pd_strong); r->clouds["strong_pd_map"] = Q.pd_strong;
        vd(Q.pd_weak); r->clouds["weak_pd_map"] = Q.pd_weak;
    }
    /* Step 3: updateCurrentMap, Removerter.cpp:1483-1524 */
    {
        Cloud uq, uc;
        merge_to_global(c, Q.knn_coexist, Q, uq); vd(uq);
        merge_to_global(c, C.knn_coexist, C, uc); vd(uc);
        Cloud up = uq; append(up, uc);
        append(up, C.nd_weak);
        Cloud ups = up; append(ups, C.pd_strong); vd(ups);
        append(up, C.pd_orig); vd(up);
        C.updated = up; C.updated_strong = ups;
        r->clouds["updated_map"] = up; r->clouds["updated_map_strong"] = ups;
    }
    c.tick("voxel", now_s() - t1);
    /* parseUpdatedStaticScansViaProjection :1551-1562, parseLDScansViaProjection :1564-1577 */
    reproject(c, C.updated, C, C.scans_updated, "reproject_large");
    reproject(c, C.updated_strong, C, C.scans_updated_strong, "reproject_large");
    reproject(c, C.pd_orig, C, C.scans_pd, "reproject_small");
    reproject(c, C.pd_strong, C, C.scans_strong_pd, "reproject_small");
    reproject(c, C.nd_weak, C, C.scans_weak_nd, "reproject_small");
    reproject(c, C.nd_strong, C, C.scans_strong_nd, "reproject_small");
    /* updateScansScanwise, Session.cpp:362-380 */
    t1 = now_s();
    for (size_t i = 0; i < C.scans_updated.size(); ++i) {
        Cloud f = C.scans_updated[i];
        append(f, C.scans_weak_nd[i]); append(f, C.scans_pd[i]);
        vd(f);
        C.scans_updated[i] = f;
    }
    c.tick("voxel_scanwise", now_s() - t1);
    c.tick("steps_1_to_3_total", now_s() - t_steps);

    r->scansets["scans_updated"] = C.scans_updated;
    r->scansets["scans_updated_strong"] = C.scans_updated_strong;
    r->scansets["scans_pd"] = C.scans_pd;
    r->scansets["scans_pd_strong"] = C.scans_strong_pd;
    r->scansets["scans_nd_strong"] = C.scans_strong_nd;
    r->scansets["scans_nd_weak"] = C.scans_weak_nd;
    r->scansets["central_static_projected"] = C.scans_static_projected;
    r->scansets["query_static_projected"] = Q.scans_static_projected;
    r->scansets["central_knn_coexist"] = C.knn_coexist; r->scansets["central_knn_diff"] = C.knn_diff;
    r->scansets["query_knn_coexist"] = Q.knn_coexist;   r->scansets["query_knn_diff"] = Q.knn_diff;
    for (auto& kv : r->scansets) flatten(*r, kv.first, kv.second);
    return reinterpret_cast<orc_run*>(r);
}

int orc_run_cloud(const orc_run* rr, const char* name, const float** pts, size_t* n)
{
    const Run* r = reinterpret_cast<const Run*>(rr);
    auto it = r->clouds.find(name);
    if (it == r->clouds.end()) return -1;
    *pts = reinterpret_cast<const float*>(it->second.data()); *n = it->second.size();
    return 0;
}

int orc_run_scanset(const orc_run* rr, const char* name, const float** pts, const uint64_t** offsets, size_t* n_kf)
{
    const Run* r = reinterpret_cast<const Run*>(rr);
    auto it = r->flat.find(name);
    if (it == r->flat.end()) return -1;
    *pts = reinterpret_cast<const float*>(it->second.first.data());
    *offsets = it->second.second.data();
    *n_kf = it->second.second.size() - 1;
    return 0;
}

int orc_run_timings(const orc_run* rr, const char** names, double* secs, int cap)
{
    const Run* r = reinterpret_cast<const Run*>(rr);
    int n = 0;
    for (const std::string& k : r->c.t_order) {
        if (n < cap) { names[n] = k.c_str(); secs[n] = r->c.t.at(k); }
        ++n;
    }
    return n;
}

void orc_run_free(orc_run* r) { delete reinterpret_cast<Run*>(r); }

} // extern "C"

/* ------------------------------------------------------------------ RViz images (visualisation only) */
extern "C" void orc_jet_lut(uint8_t* lut)
{
    /* OpenCV's Jet base map: 256 samples of r(x) = clamp(min(4x - 1.5, -4x + 4.5)), g(x) = clamp(min(4x - 0.5, -4x + 3.5)),
       b(x) = clamp(min(4x + 0.5, -4x + 2.5)) at x = i/255, stored as float, then convertTo(CV_8U, 255.) */
    for (int i = 0; i < 256; ++i) {
        const double x = (double)i / 255.0;
        const double ch[3] = {std::min(4.0 * x + 0.5, -4.0 * x + 2.5), std::min(4.0 * x - 0.5, -4.0 * x + 3.5),
                              std::min(4.0 * x - 1.5, -4.0 * x + 4.5)};                       /* B, G, R */
        for (int c = 0; c < 3; ++c) {
            const float lit = (float)std::min(1.0, std::max(0.0, ch[c]));
            const float v = lit * 255.0f;
            const long q = std::lrint((double)v);               /* round half to even, as cv::saturate_cast<uchar> */
            lut[3 * i + c] = (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
        }
    }
}

static inline void orc_axis(float cmin, float cmax, double* a, double* b)
{
    const double inv = 1.0 / (double)(float)(cmax - cmin);    /* cv::MatExpr: e / s == e * (1./s) */
    *a = 255.0 * inv;
    *b = -((double)cmin * 255.0) * inv;
}

extern "C" void orc_colormap_f32(const float* src, size_t n, float cmin, float cmax, uint8_t* bgr)
{
    uint8_t lut[768];
    orc_jet_lut(lut);
    double a, b;
    orc_axis(cmin, cmax, &a, &b);
    const float fa = (float)a, fb = (float)b;
    for (size_t i = 0; i < n; ++i) {
        const float v = src[i] * fa + fb;                    /* float image: float arithmetic, no FMA */
        long q = std::isnan(v) ? 0 : (v <= -1.f ? 0 : v >= 256.f ? 255 : std::lrint((double)v));
        if (q < 0) q = 0;
        if (q > 255) q = 255;
        std::memcpy(bgr + 3 * i, lut + 3 * q, 3);
    }
}

extern "C" void orc_colormap_i32(const int32_t* src, size_t n, float cmin, float cmax, uint8_t* bgr)
{
    uint8_t lut[768];
    orc_jet_lut(lut);
    double a, b;
    orc_axis(cmin, cmax, &a, &b);
    for (size_t i = 0; i < n; ++i) {
        const double v = (double)src[i] * a + b;             /* int32 image: double arithmetic, rounded to int first */
        long q = v <= -1.0 ? 0 : v >= 256.0 ? 255 : std::lrint(v);
        if (q < 0) q = 0;
        if (q > 255) q = 255;
        std::memcpy(bgr + 3 * i, lut + 3 * q, 3);
    }
}
