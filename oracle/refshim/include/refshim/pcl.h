/*
 * TEST INFRASTRUCTURE -- stand-in for the part of PCL 1.10 (+ FLANN 1.9.1 behind pcl::KdTreeFLANN) that the unmodified reference
 * sources call, so that /root/reference/ltremovert/src/{utility,Session,Removerter,RosParamServer}.cpp compile and run here.
 * PCL is NOT in /root/reference (un-vendored; versions pinned only by docker/Dockerfile:1): every routine below is a restatement
 * of the library's published behaviour written from knowledge of its sources -- PARITY UNPINNED, like the same leaves in
 * oracle/ltm_oracle.cpp -- but written independently of the oracle and deliberately literal (a pointer octree, an index sort
 * with PCL's comparator, a plain kd-tree), so that "oracle == reference-compiled code over these leaves" is a two-derivation
 * statement for the leaves and a reference-compiled statement for everything the reference itself computes.
 * Original code; nothing copied from PCL, FLANN or the reference.
 */
#ifndef REFSHIM_PCL_H
#define REFSHIM_PCL_H

#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "refshim/eigen.h"
#include "refshim/ros.h"

namespace pcl {

/* pcl::PointXYZI: 32 bytes, xyz + padding word (1.0) then intensity + 3 padding words, 16-byte aligned */
struct alignas(16) PointXYZI {
    union { float data[4]; struct { float x, y, z; }; };
    union { struct { float intensity; }; float data_c[4]; };
    PointXYZI() { x = y = z = 0.0f; data[3] = 1.0f; intensity = 0.0f; data_c[1] = data_c[2] = data_c[3] = 0.0f; }
};

inline bool isFinite(const PointXYZI& p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); }

template <class PointT> class PointCloud {
public:
    typedef boost::shared_ptr<PointCloud<PointT>> Ptr;
    typedef boost::shared_ptr<const PointCloud<PointT>> ConstPtr;

    std_msgs::Header header;
    std::vector<PointT> points;
    uint32_t width = 0, height = 0;
    bool is_dense = true;

    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void clear() { points.clear(); width = 0; height = 0; }
    void push_back(const PointT& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
    PointT& operator[](size_t i) { return points[i]; }
    const PointT& operator[](size_t i) const { return points[i]; }
    typename std::vector<PointT>::iterator begin() { return points.begin(); }
    typename std::vector<PointT>::iterator end() { return points.end(); }

    /* PointCloud::operator+= : append, result unorganised */
    PointCloud& operator+=(const PointCloud& rhs)
    {
        const size_t n = points.size();
        points.resize(n + rhs.points.size());               /* (rhs may be *this) */
        for (size_t i = n; i < points.size(); ++i) points[i] = rhs.points[i - n];
        width = (uint32_t)points.size(); height = 1;
        is_dense = is_dense && rhs.is_dense;
        return *this;
    }
};

namespace console {
enum VERBOSITY_LEVEL { L_ALWAYS, L_ERROR, L_WARN, L_INFO, L_DEBUG, L_VERBOSE };
inline void setVerbosityLevel(VERBOSITY_LEVEL) {}
}

/* pcl::getMinMax3D(cloud, min_pt, max_pt) on a dense cloud: float component-wise min / max */
template <class PointT> inline void getMinMax3D(const PointCloud<PointT>& c, float mn[3], float mx[3])
{
    mn[0] = mn[1] = mn[2] = FLT_MAX;
    mx[0] = mx[1] = mx[2] = -FLT_MAX;
    for (const PointT& p : c.points) {
        if (!c.is_dense && !isFinite(p)) continue;
        mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
        mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
    }
}

/* pcl::transformPointCloud(in, out, Eigen::Matrix<double,4,4>) (utility.cpp:70-71,164-165,184-185,198-199; Session.cpp:196-197):
 * the generic Transformer<double>::se3 -- every coordinate (float)(t(r,0)*x + t(r,1)*y + t(r,2)*z + t(r,3)), the products and sums in
 * double from left to right; the remaining fields are copied; in == out is allowed. */
template <class PointT>
inline void transformPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, const Eigen::Matrix4d& t, bool = true)
{
    if (&in != &out) {
        out.header = in.header; out.is_dense = in.is_dense; out.width = in.width; out.height = in.height;
        out.points.assign(in.points.begin(), in.points.end());
    }
    for (size_t i = 0; i < out.points.size(); ++i) {
        if (!in.is_dense && !isFinite(in.points[i])) continue;
        const double x = in.points[i].x, y = in.points[i].y, z = in.points[i].z;
        PointT& o = out.points[i];
        o.x = static_cast<float>(t(0, 0) * x + t(0, 1) * y + t(0, 2) * z + t(0, 3));
        o.y = static_cast<float>(t(1, 0) * x + t(1, 1) * y + t(1, 2) * z + t(1, 3));
        o.z = static_cast<float>(t(2, 0) * x + t(2, 1) * y + t(2, 2) * z + t(2, 3));
        o.data[3] = 1.0f;
    }
}

/* pcl::ExtractIndices<PointT> with setNegative(false): the indexed points in the given order (Removerter.cpp:933-946) */
template <class PointT> class ExtractIndices {
public:
    void setInputCloud(const typename PointCloud<PointT>::Ptr& c) { in_ = c; }
    void setIndices(const boost::shared_ptr<std::vector<int>>& i) { idx_ = i; }
    void setNegative(bool n) { negative_ = n; }
    void filter(PointCloud<PointT>& out)
    {
        PointCloud<PointT> tmp;
        tmp.header = in_->header; tmp.is_dense = in_->is_dense;
        if (!negative_) {
            tmp.points.reserve(idx_->size());
            for (int i : *idx_) tmp.points.push_back(in_->points[(size_t)i]);
        } else {
            std::vector<char> drop(in_->points.size(), 0);
            for (int i : *idx_) drop[(size_t)i] = 1;
            for (size_t i = 0; i < in_->points.size(); ++i) if (!drop[i]) tmp.points.push_back(in_->points[i]);
        }
        tmp.width = (uint32_t)tmp.points.size(); tmp.height = 1;
        out = tmp;
    }
private:
    typename PointCloud<PointT>::Ptr in_;
    boost::shared_ptr<std::vector<int>> idx_;
    bool negative_ = false;
};

/* pcl::VoxelGrid<PointT>::applyFilter as Session::loadKeyframes uses it (Session.cpp:284-289): all fields downsampled, no minimum
 * point count.  inverse leaf size in float; "leaf size too small" (dx*dy*dz > INT32_MAX) returns the input unchanged; otherwise
 * (leaf index, point index) pairs sorted with std::sort on the LEAF INDEX ONLY -- PCL's cloud_point_index_idx::operator< -- so the
 * order of the points inside a voxel is whatever the C++ library's std::sort leaves, exactly as in the reference binary; float sums
 * (CentroidPoint accumulators) in that order, divided by (float)count; output in ascending leaf index. */
template <class PointT> class VoxelGrid {
public:
    void setLeafSize(float lx, float ly, float lz)
    {
        leaf_[0] = lx; leaf_[1] = ly; leaf_[2] = lz;
        for (int d = 0; d < 3; ++d) inv_[d] = 1.0f / leaf_[d];
    }
    void setInputCloud(const typename PointCloud<PointT>::Ptr& c) { in_ = c; }
    void filter(PointCloud<PointT>& out)
    {
        const PointCloud<PointT>& in = *in_;
        out.height = 1; out.is_dense = true;
        if (in.points.empty()) { out.points.clear(); out.width = 0; return; }
        float mn[3], mx[3];
        getMinMax3D(in, mn, mx);
        const int64_t dx = static_cast<int64_t>((mx[0] - mn[0]) * inv_[0]) + 1;
        const int64_t dy = static_cast<int64_t>((mx[1] - mn[1]) * inv_[1]) + 1;
        const int64_t dz = static_cast<int64_t>((mx[2] - mn[2]) * inv_[2]) + 1;
        if (dx * dy * dz > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) { out = in; return; }
        int min_b[3], max_b[3], div_b[3];
        for (int d = 0; d < 3; ++d) {
            min_b[d] = static_cast<int>(std::floor(mn[d] * inv_[d]));
            max_b[d] = static_cast<int>(std::floor(mx[d] * inv_[d]));
            div_b[d] = max_b[d] - min_b[d] + 1;
        }
        const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
        struct Entry {
            unsigned idx, cloud_point_index;
            bool operator<(const Entry& o) const { return idx < o.idx; }
        };
        std::vector<Entry> iv;
        iv.reserve(in.points.size());
        for (size_t i = 0; i < in.points.size(); ++i) {
            const PointT& p = in.points[i];
            if (!in.is_dense && !isFinite(p)) continue;
            const int i0 = static_cast<int>(std::floor(p.x * inv_[0]) - static_cast<float>(min_b[0]));
            const int i1 = static_cast<int>(std::floor(p.y * inv_[1]) - static_cast<float>(min_b[1]));
            const int i2 = static_cast<int>(std::floor(p.z * inv_[2]) - static_cast<float>(min_b[2]));
            iv.push_back(Entry{static_cast<unsigned>(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), static_cast<unsigned>(i)});
        }
        std::sort(iv.begin(), iv.end(), std::less<Entry>());
        std::vector<PointT> res;
        for (size_t a = 0; a < iv.size();) {
            size_t b = a;
            float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
            while (b < iv.size() && iv[b].idx == iv[a].idx) {
                const PointT& q = in.points[iv[b].cloud_point_index];
                sx += q.x; sy += q.y; sz += q.z; si += q.intensity;
                ++b;
            }
            const float n = static_cast<float>(b - a);
            PointT c;
            c.x = sx / n; c.y = sy / n; c.z = sz / n; c.intensity = si / n;
            res.push_back(c);
            a = b;
        }
        out.points.swap(res);
        out.width = (uint32_t)out.points.size();
    }
private:
    float leaf_[3] = {0, 0, 0}, inv_[3] = {0, 0, 0};
    typename PointCloud<PointT>::Ptr in_;
};

/* pcl::KdTreeFLANN<PointT>::nearestKSearch: the exact k nearest neighbours, squared distances ascending, FLANN's L2_Simple<float>
 * (((dx*dx) + dy*dy) + dz*dz in float); k is clamped to the number of points.  A plain median-split kd-tree with exact pruning
 * stands in for FLANN's KDTreeSingleIndex (leaf 15, eps 0): the result SET of an exact search does not depend on the tree. */
template <class PointT> class KdTreeFLANN {
public:
    typedef boost::shared_ptr<KdTreeFLANN<PointT>> Ptr;

    void setInputCloud(const typename PointCloud<PointT>::Ptr& cloud)
    {
        cloud_ = cloud;
        const size_t n = cloud ? cloud->points.size() : 0;
        xyz_.resize(3 * n);
        perm_.resize(n);
        for (size_t i = 0; i < n; ++i) {
            xyz_[3 * i] = cloud->points[i].x; xyz_[3 * i + 1] = cloud->points[i].y; xyz_[3 * i + 2] = cloud->points[i].z;
            perm_[i] = (int)i;
        }
        nodes_.clear();
        if (n) build(0, n);
    }

    int nearestKSearch(const PointT& q, int k, std::vector<int>& idx, std::vector<float>& sqd) const
    {
        if (perm_.empty()) throw std::runtime_error("refshim KdTreeFLANN: search in an empty tree (PCL refuses to build one; the reference would crash here)");
        if (k > (int)perm_.size()) k = (int)perm_.size();
        idx.assign((size_t)k, -1);
        sqd.assign((size_t)k, FLT_MAX);
        if (k == 0) return 0;
        const float qv[3] = {q.x, q.y, q.z};
        int found = 0;
        search(0, qv, k, idx.data(), sqd.data(), found);
        return k;
    }

private:
    struct Node { int axis; float split; size_t lo, hi; int left, right; };   /* axis < 0: leaf over perm_[lo, hi) */

    int build(size_t lo, size_t hi)
    {
        const int me = (int)nodes_.size();
        nodes_.push_back(Node{-1, 0.0f, lo, hi, -1, -1});
        if (hi - lo <= 15) return me;
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (size_t i = lo; i < hi; ++i)
            for (int d = 0; d < 3; ++d) {
                const float v = xyz_[3 * (size_t)perm_[i] + d];
                mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v);
            }
        int ax = 0;
        for (int d = 1; d < 3; ++d) if (mx[d] - mn[d] > mx[ax] - mn[ax]) ax = d;
        if (!(mx[ax] > mn[ax])) return me;                                        /* all points identical: keep as one leaf */
        const size_t mid = lo + (hi - lo) / 2;
        std::nth_element(perm_.begin() + lo, perm_.begin() + mid, perm_.begin() + hi,
                         [&](int a, int b) { return xyz_[3 * (size_t)a + ax] < xyz_[3 * (size_t)b + ax]; });
        const float split = xyz_[3 * (size_t)perm_[mid] + ax];
        const int l = build(lo, mid), r = build(mid, hi);
        nodes_[me].axis = ax; nodes_[me].split = split; nodes_[me].left = l; nodes_[me].right = r;
        return me;
    }

    void search(int ni, const float* q, int k, int* idx, float* sqd, int& found) const
    {
        const Node& nd = nodes_[ni];
        if (nd.axis < 0) {
            for (size_t i = nd.lo; i < nd.hi; ++i) {
                const int pi = perm_[i];
                const float* p = &xyz_[3 * (size_t)pi];
                float d = 0.0f;
                for (int c = 0; c < 3; ++c) { const float diff = q[c] - p[c]; d += diff * diff; }     /* L2_Simple */
                if (found < k || d < sqd[k - 1]) {
                    int j = found < k ? found++ : k - 1;
                    while (j > 0 && sqd[j - 1] > d) { sqd[j] = sqd[j - 1]; idx[j] = idx[j - 1]; --j; }
                    sqd[j] = d; idx[j] = pi;
                }
            }
            return;
        }
        const float diff = q[nd.axis] - nd.split;
        const int near = diff < 0.0f ? nd.left : nd.right, far = diff < 0.0f ? nd.right : nd.left;
        search(near, q, k, idx, sqd, found);
        /* everything on the far side is at least |diff| away along this axis; float rounding is monotone, so its computed squared
           distance is >= diff*diff: visit unless that already cannot beat the current k-th best */
        if (found < k || diff * diff < sqd[k - 1]) search(far, q, k, idx, sqd, found);
    }

    typename PointCloud<PointT>::Ptr cloud_;
    std::vector<float> xyz_;
    std::vector<int> perm_;
    std::vector<Node> nodes_;
};

/* pcl::IterativeClosestPoint: only behind `useICPrefinement{false}` (Session.cpp:551) and as unused members -- setters store nothing,
 * align() is never reached */
template <class S, class T> class IterativeClosestPoint {
public:
    void setMaxCorrespondenceDistance(double) {}
    void setMaximumIterations(int) {}
    void setTransformationEpsilon(double) {}
    void setEuclideanFitnessEpsilon(double) {}
    void setRANSACIterations(int) {}
    void setInputTarget(const typename PointCloud<T>::Ptr&) {}
    void setInputSource(const typename PointCloud<S>::Ptr&) {}
    void align(PointCloud<S>&) { throw std::runtime_error("refshim: ICP is not provided (the reference never runs it: useICPrefinement is false)"); }
    Eigen::Matrix4d getFinalTransformation() const { return Eigen::Matrix4d::Identity(); }
    double getFitnessScore() const { return 0.0; }
};

template <class PointT> inline void toROSMsg(const PointCloud<PointT>&, sensor_msgs::PointCloud2&) {}

namespace octree {

/* pcl::octree::OctreePointCloudVoxelCentroid<PointT> as utility.cpp:204-219 drives it (setInputCloud, defineBoundingBox,
 * addPointsFromInputCloud, getVoxelCentroids): a literal pointer octree.
 *  - defineBoundingBox(): float getMinMax3D; max + 512*FLT_EPSILON added IN FLOAT, then everything in double;
 *  - getKeyBitSize(): max_key = ceil((max - min - FLT_EPSILON) / res) per axis, max_voxels = max(keys, 2),
 *    depth = ceil(log2(max_voxels) - FLT_EPSILON), side = (1 << depth) * res, and -- the tree being empty -- the box is centred:
 *    oversize = (side - (max - min)) / 2, applied to both ends where oversize > FLT_EPSILON;
 *  - key = (unsigned)(((double)p - min) / res) per axis; descent from depth mask 1 << (depth-1), child = x<<2 | y<<1 | z;
 *  - leaf container: ++count, sum += point in float in insertion order; centroid = sum / (float)count;
 *  - getVoxelCentroids: depth-first, children 0..7. */
template <class PointT> class OctreePointCloudVoxelCentroid {
public:
    typedef std::vector<PointT, Eigen::aligned_allocator<PointT>> AlignedPointTVector;

    explicit OctreePointCloudVoxelCentroid(double resolution) : res_(resolution) {}
    ~OctreePointCloudVoxelCentroid() { destroy(root_); }
    OctreePointCloudVoxelCentroid(const OctreePointCloudVoxelCentroid&) = delete;
    OctreePointCloudVoxelCentroid& operator=(const OctreePointCloudVoxelCentroid&) = delete;

    void setInputCloud(const typename PointCloud<PointT>::Ptr& c) { in_ = c; }

    void defineBoundingBox()
    {
        float mn[3], mx[3];
        getMinMax3D(*in_, mn, mx);
        const float min_value = std::numeric_limits<float>::epsilon() * 512.0f;
        for (int d = 0; d < 3; ++d) { min_[d] = mn[d]; max_[d] = mx[d] + min_value; }
        for (int d = 0; d < 3; ++d) { const double lo = std::min(min_[d], max_[d]), hi = std::max(min_[d], max_[d]); min_[d] = lo; max_[d] = hi; }
        const float eps = std::numeric_limits<float>::epsilon();
        unsigned max_key[3];
        for (int d = 0; d < 3; ++d) max_key[d] = static_cast<unsigned>(std::ceil((max_[d] - min_[d] - eps) / res_));
        const unsigned max_voxels = std::max(std::max(std::max(max_key[0], max_key[1]), max_key[2]), 2u);
        depth_ = std::max(std::min(32u, static_cast<unsigned>(std::ceil(std::log2(max_voxels) - eps))), 0u);
        const double side = static_cast<double>(1 << depth_) * res_;
        for (int d = 0; d < 3; ++d) {
            const double over = (side - (max_[d] - min_[d])) / 2.0;
            if (over > eps) { min_[d] -= over; max_[d] += over; }
        }
        defined_ = true;
    }

    void addPointsFromInputCloud()
    {
        if (!defined_) defineBoundingBox();
        if (!root_) root_ = new Branch();
        const unsigned top = depth_ ? 1u << (depth_ - 1) : 0u;
        for (const PointT& p : in_->points) {
            if (!isFinite(p)) continue;
            const unsigned k[3] = {static_cast<unsigned>((p.x - min_[0]) / res_), static_cast<unsigned>((p.y - min_[1]) / res_),
                                   static_cast<unsigned>((p.z - min_[2]) / res_)};
            Branch* b = root_;
            unsigned mask = top;
            for (;;) {
                const int child = ((!!(k[0] & mask)) << 2) | ((!!(k[1] & mask)) << 1) | (!!(k[2] & mask));
                if (mask > 1) {
                    if (!b->child[child]) b->child[child] = new Branch();
                    b = static_cast<Branch*>(b->child[child]);
                    mask >>= 1;
                } else {
                    if (!b->child[child]) { Leaf* l = new Leaf(); b->child[child] = l; b->leaf_mask |= 1u << child; }
                    Leaf* l = static_cast<Leaf*>(b->child[child]);
                    ++l->count;
                    l->sum.x += p.x; l->sum.y += p.y; l->sum.z += p.z; l->sum.intensity += p.intensity;
                    break;
                }
            }
        }
    }

    size_t getVoxelCentroids(AlignedPointTVector& out) const
    {
        out.clear();
        if (root_) collect(root_, out);
        return out.size();
    }

private:
    struct Node { virtual ~Node() {} };
    struct Branch : Node { Node* child[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; unsigned leaf_mask = 0; };
    struct Leaf : Node { unsigned count = 0; PointT sum; };

    void collect(const Branch* b, AlignedPointTVector& out) const
    {
        for (int c = 0; c < 8; ++c) {
            if (!b->child[c]) continue;
            if (b->leaf_mask & (1u << c)) {
                const Leaf* l = static_cast<const Leaf*>(b->child[c]);
                PointT cen = l->sum;
                const float n = static_cast<float>(l->count);
                cen.x /= n; cen.y /= n; cen.z /= n; cen.intensity /= n;
                out.push_back(cen);
            } else collect(static_cast<const Branch*>(b->child[c]), out);
        }
    }
    void destroy(Branch* b)
    {
        if (!b) return;
        for (int c = 0; c < 8; ++c) {
            if (!b->child[c]) continue;
            if (b->leaf_mask & (1u << c)) delete b->child[c]; else destroy(static_cast<Branch*>(b->child[c]));
        }
        delete b;
    }

    double res_;
    typename PointCloud<PointT>::Ptr in_;
    double min_[3] = {0, 0, 0}, max_[3] = {0, 0, 0};
    unsigned depth_ = 0;
    bool defined_ = false;
    Branch* root_ = nullptr;
};

} // namespace octree

namespace io {

/* every cloud that goes through savePCDFileBinary is also kept here (path -> XYZI floats) when refshim::capture_saves is on, so an
 * in-process driver (ref_capi.cpp) can read the reference's outputs without touching the disk */
struct SaveSink {
    bool capture = false, write_files = true;
    std::map<std::string, std::vector<float>> clouds;
    std::map<std::string, std::pair<uint32_t, uint32_t>> shapes;   /* width, height */
};
inline SaveSink& save_sink() { static SaveSink s; return s; }

/* pcl::io::savePCDFileBinary<PointXYZI>: the v0.7 header PCL writes for (x y z intensity) float fields, then 16 bytes per point.
 * (PCL throws pcl::IOException on an EMPTY cloud -- "Input point cloud has no data!" -- which ends the reference process at the first
 * keyframe without, e.g., strong ND points (Removerter.cpp:1645); the stand-in writes a header with POINTS 0 instead and counts it.) */
inline int& empty_saves() { static int n = 0; return n; }
template <class PointT> inline int savePCDFileBinary(const std::string& path, const PointCloud<PointT>& c)
{
    SaveSink& s = save_sink();
    if (c.points.empty()) ++empty_saves();
    if (s.capture) {
        std::vector<float>& v = s.clouds[path];
        v.resize(4 * c.points.size());
        for (size_t i = 0; i < c.points.size(); ++i) { v[4 * i] = c.points[i].x; v[4 * i + 1] = c.points[i].y; v[4 * i + 2] = c.points[i].z; v[4 * i + 3] = c.points[i].intensity; }
        s.shapes[path] = std::make_pair(c.width, c.height);
    }
    if (!s.write_files) return 0;
    std::ofstream f(path, std::ios::binary);
    if (!f) return -1;
    char hdr[512];
    const int n = std::snprintf(hdr, sizeof hdr,
        "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
        "WIDTH %u\nHEIGHT %u\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n", c.width, c.height, c.points.size());
    f.write(hdr, n);
    for (const PointT& p : c.points) { const float v[4] = {p.x, p.y, p.z, p.intensity}; f.write(reinterpret_cast<const char*>(v), 16); }
    return f ? 0 : -1;
}

/* pcl::io::loadPCDFile<PointXYZI>: v0.7 files with float x y z intensity fields, DATA ascii or binary (what the tests and the
 * synthetic sessions use; binary_compressed is not provided here) */
template <class PointT> inline int loadPCDFile(const std::string& path, PointCloud<PointT>& c)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return -1;
    std::string line, data;
    std::vector<std::string> fields;
    size_t n = 0; uint32_t w = 0, h = 1;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string key; ss >> key;
        if (key == "FIELDS") { std::string t; while (ss >> t) fields.push_back(t); }
        else if (key == "WIDTH") ss >> w;
        else if (key == "HEIGHT") ss >> h;
        else if (key == "POINTS") ss >> n;
        else if (key == "DATA") { ss >> data; break; }
    }
    int ix = -1, iy = -1, iz = -1, ii = -1;
    for (size_t k = 0; k < fields.size(); ++k) {
        if (fields[k] == "x") ix = (int)k; else if (fields[k] == "y") iy = (int)k; else if (fields[k] == "z") iz = (int)k; else if (fields[k] == "intensity") ii = (int)k;
    }
    if (ix < 0 || iy < 0 || iz < 0) return -1;
    const size_t nf = fields.size();
    c.points.assign(n, PointT());
    c.width = w; c.height = h; c.is_dense = true;
    std::vector<float> row(nf);
    for (size_t i = 0; i < n; ++i) {
        if (data == "binary") f.read(reinterpret_cast<char*>(row.data()), 4 * nf);
        else if (data == "ascii") { for (size_t k = 0; k < nf; ++k) { std::string t; f >> t; row[k] = std::strtof(t.c_str(), nullptr); } }
        else return -1;
        if (!f) return -1;
        c.points[i].x = row[ix]; c.points[i].y = row[iy]; c.points[i].z = row[iz];
        if (ii >= 0) c.points[i].intensity = row[ii];
        if (!isFinite(c.points[i])) c.is_dense = false;
    }
    return 0;
}

} // namespace io
} // namespace pcl

#endif
