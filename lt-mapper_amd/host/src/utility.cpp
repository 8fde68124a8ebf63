// utility.cpp -- host-side helpers of the MI355X build of `removert` (mirror of ltremovert/src/utility.cpp without
// ROS/PCL/OpenCV).  File formats only; no point arithmetic of the hot path happens here.
#include "ltm_pclsort.h"
#include "removert/utility.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace fs = std::filesystem;

namespace ltremovert
{

std::vector<double> splitPoseLine(const std::string& _str_line, char _delimiter)
{
    std::vector<double> parsed;
    std::stringstream ss(_str_line);
    std::string temp;
    while (getline(ss, temp, _delimiter)) {
        if (temp.empty()) continue;          // tolerate repeated blanks (std::stod would throw in the reference)
        parsed.push_back(std::stod(temp));
    }
    return parsed;
}

std::pair<int, int> resetRimgSize(const std::pair<float, float> _fov, const float _resize_ratio)
{
    int rows = 0, cols = 0;
    ltm_rimg_size(_fov.first, _fov.second, _resize_ratio, &rows, &cols);
    return {rows, cols};
}

// Eigen::Matrix4d::inverse() of Session.cpp:109-110 / RosParamServer.cpp:29-30: the library's restatement of Eigen 3.3.7's SSE2
// kernel (ltm_inverse4x4) -- the one inverse host, library and oracle agree on bit for bit
bool inverse4x4(const double* m, double* inv) { return ltm_inverse4x4(m, inv) == LTM_OK; }

void parallelFor(size_t n, const std::function<void(size_t)>& f, unsigned threads)
{
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    threads = (unsigned)std::min<size_t>(threads, n);
    if (threads <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex m;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            try {
                for (size_t i = next++; i < n; i = next++) f(i);
            } catch (...) {
                std::lock_guard<std::mutex> g(m);
                if (!err) err = std::current_exception();
                next = n;
            }
        });
    for (auto& th : pool) th.join();
    if (err) std::rethrow_exception(err);
}

void fsmkdir(const std::string& _path)
{
    if (!fs::is_directory(_path) || !fs::exists(_path)) fs::create_directories(_path);
}

std::vector<std::string> listDirectorySorted(const std::string& dir, std::vector<std::string>* names)
{
    std::vector<std::string> paths;
    if (names) names->clear();
    for (auto& e : fs::directory_iterator(dir)) {        // Session.cpp:87-92
        paths.emplace_back(e.path().string());
        if (names) names->emplace_back(e.path().filename().string());
    }
    std::sort(paths.begin(), paths.end());
    if (names) std::sort(names->begin(), names->end());
    return paths;
}

void ltmCheck(ltm_ctx* ctx, int rc, const char* what)
{
    if (rc == LTM_OK) return;
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + (ctx ? ltm_last_error(ctx) : "no context"));
}

// ------------------------------------------------------------------------------------------ PCD
namespace {

struct Field { std::string name; int size = 4; char type = 'F'; int count = 1; int offset = 0; };

} // namespace

// LZF decompression (the codec PCL's binary_compressed PCD files use)
bool lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len)
{
    const unsigned char* ip = in; const unsigned char* const in_end = in + in_len;
    unsigned char* op = out; unsigned char* const out_end = out + out_len;
    while (ip < in_end) {
        unsigned ctrl = *ip++;
        if (ctrl < 32) {                       // literal run of ctrl+1 bytes
            ++ctrl;
            if (op + ctrl > out_end || ip + ctrl > in_end) return false;
            memcpy(op, ip, ctrl); op += ctrl; ip += ctrl;
        } else {                               // back reference
            unsigned len = ctrl >> 5;
            if (ip >= in_end) return false;
            if (len == 7) { len += *ip++; if (ip >= in_end) return false; }
            const unsigned char* ref = op - ((ctrl & 0x1f) << 8) - 1 - *ip++;
            len += 2;
            if (ref < out || op + len > out_end) return false;
            for (unsigned i = 0; i < len; ++i) op[i] = ref[i];   // may overlap
            op += len;
        }
    }
    return op == out_end;
}

namespace {

float read_as_float(const unsigned char* p, const Field& f)
{
    switch (f.type) {
    case 'F': if (f.size == 4) { float v; memcpy(&v, p, 4); return v; } else { double v; memcpy(&v, p, 8); return (float)v; }
    case 'U': if (f.size == 1) return (float)*p; if (f.size == 2) { uint16_t v; memcpy(&v, p, 2); return (float)v; } { uint32_t v; memcpy(&v, p, 4); return (float)v; }
    case 'I': if (f.size == 1) return (float)*(const int8_t*)p; if (f.size == 2) { int16_t v; memcpy(&v, p, 2); return (float)v; } { int32_t v; memcpy(&v, p, 4); return (float)v; }
    }
    return 0.0f;
}

} // namespace

bool loadPCDFile(const std::string& path, Cloud& out, std::string* err)
{
    auto fail = [&](const std::string& m) { if (err) *err = path + ": " + m; return false; };
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail("cannot open");
    std::vector<Field> fields;
    size_t n_points = 0, width = 0, height = 1;
    std::string data_kind, line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::stringstream ss(line);
        std::string key; ss >> key;
        if (key == "FIELDS") { std::string n; while (ss >> n) { Field fd; fd.name = n; fields.push_back(fd); } }
        else if (key == "SIZE") { for (auto& fd : fields) ss >> fd.size; }
        else if (key == "TYPE") { for (auto& fd : fields) ss >> fd.type; }
        else if (key == "COUNT") { for (auto& fd : fields) ss >> fd.count; }
        else if (key == "WIDTH") ss >> width;
        else if (key == "HEIGHT") ss >> height;
        else if (key == "POINTS") ss >> n_points;
        else if (key == "DATA") { ss >> data_kind; break; }
    }
    if (fields.empty() || data_kind.empty()) return fail("malformed header");
    if (n_points == 0) n_points = width * height;
    int stride = 0;
    for (auto& fd : fields) { fd.offset = stride; stride += fd.size * fd.count; }
    int ix = -1, iy = -1, iz = -1, ii = -1;
    for (size_t k = 0; k < fields.size(); ++k) {
        if (fields[k].name == "x") ix = (int)k; else if (fields[k].name == "y") iy = (int)k;
        else if (fields[k].name == "z") iz = (int)k; else if (fields[k].name == "intensity") ii = (int)k;
    }
    if (ix < 0 || iy < 0 || iz < 0) return fail("no x/y/z fields");
    out.assign(n_points, PointType{0, 0, 0, 0});
    if (data_kind == "ascii") {
        for (size_t p = 0; p < n_points; ++p) {
            if (!std::getline(f, line)) return fail("truncated ascii data");
            std::stringstream ss(line);
            for (size_t k = 0; k < fields.size(); ++k)
                for (int c = 0; c < fields[k].count; ++c) {
                    double v; ss >> v;
                    if (c) continue;
                    if ((int)k == ix) out[p].x = (float)v; else if ((int)k == iy) out[p].y = (float)v;
                    else if ((int)k == iz) out[p].z = (float)v; else if ((int)k == ii) out[p].intensity = (float)v;
                }
        }
        return true;
    }
    if (data_kind == "binary" && stride == 16 && fields.size() == 4 && ix == 0 && iy == 1 && iz == 2 && ii == 3 &&
        std::all_of(fields.begin(), fields.end(), [](const Field& fd) { return fd.type == 'F' && fd.size == 4 && fd.count == 1; })) {
        // the usual scan file (what pcl::io::savePCDFileBinary writes for PointXYZI, padding stripped): the records ARE PointType -- read in place
        // (round 4 read into a byte vector and decoded field by field: two more passes over every scan on the four loader threads, Step 0 of files -> files)
        f.read(reinterpret_cast<char*>(out.data()), (std::streamsize)(n_points * sizeof(PointType)));
        if ((size_t)f.gcount() != n_points * sizeof(PointType)) return fail("truncated binary data");
        return true;
    }
    std::vector<unsigned char> raw((size_t)stride * n_points);
    if (data_kind == "binary") {
        f.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)raw.size());
        if ((size_t)f.gcount() != raw.size()) return fail("truncated binary data");
        for (size_t p = 0; p < n_points; ++p) {
            const unsigned char* b = raw.data() + p * stride;
            out[p].x = read_as_float(b + fields[ix].offset, fields[ix]);
            out[p].y = read_as_float(b + fields[iy].offset, fields[iy]);
            out[p].z = read_as_float(b + fields[iz].offset, fields[iz]);
            if (ii >= 0) out[p].intensity = read_as_float(b + fields[ii].offset, fields[ii]);
        }
        return true;
    }
    if (data_kind == "binary_compressed") {
        uint32_t csize = 0, usize = 0;
        f.read(reinterpret_cast<char*>(&csize), 4); f.read(reinterpret_cast<char*>(&usize), 4);
        if (!f || usize != raw.size()) return fail("bad compressed sizes");
        std::vector<unsigned char> comp(csize);
        f.read(reinterpret_cast<char*>(comp.data()), csize);
        if ((size_t)f.gcount() != csize) return fail("truncated compressed data");
        if (!lzf_decompress(comp.data(), csize, raw.data(), usize)) return fail("LZF decode failed");
        // compressed payload is structure-of-arrays: all of field 0, then all of field 1, ...
        size_t base = 0;
        for (size_t k = 0; k < fields.size(); ++k) {
            const size_t fsz = (size_t)fields[k].size * fields[k].count;
            if ((int)k == ix || (int)k == iy || (int)k == iz || (int)k == ii)
                for (size_t p = 0; p < n_points; ++p) {
                    const float v = read_as_float(raw.data() + base + p * fsz, fields[k]);
                    if ((int)k == ix) out[p].x = v; else if ((int)k == iy) out[p].y = v; else if ((int)k == iz) out[p].z = v; else out[p].intensity = v;
                }
            base += fsz * n_points;
        }
        return true;
    }
    return fail("unsupported DATA kind " + data_kind);
}

bool savePCDFileBinary(const std::string& path, const Cloud& cloud, bool octree_layout, std::string* err)
{
    return savePCDFileBinary(path, cloud.data(), cloud.size(), octree_layout, err);
}

bool readPCDPointCount(const std::string& path, size_t* n_points, std::string* err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { if (err) *err = path + ": cannot open"; return false; }
    size_t n = 0, width = 0, height = 1;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::stringstream ss(line);
        std::string key; ss >> key;
        if (key == "WIDTH") ss >> width;
        else if (key == "HEIGHT") ss >> height;
        else if (key == "POINTS") ss >> n;
        else if (key == "DATA") { *n_points = n ? n : width * height; return true; }
    }
    if (err) *err = path + ": malformed header";
    return false;
}

bool openPCDFileBinary(const std::string& path, size_t n, bool octree_layout, std::ofstream* f, std::string* err)
{
    f->open(path, std::ios::binary | std::ios::trunc);
    if (!*f) { if (err) *err = path + ": cannot open for writing"; return false; }
    const size_t width = octree_layout ? 1 : n, height = octree_layout ? n : 1;
    std::ostringstream h;   // the header pcl::PCDWriter::generateHeader emits for PointXYZI (padding fields stripped)
    h << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
      << "WIDTH " << width << "\nHEIGHT " << height << "\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA binary\n";
    const std::string hs = h.str();
    f->write(hs.data(), (std::streamsize)hs.size());
    if (!*f) { if (err) *err = path + ": write failed"; return false; }
    return true;
}

bool savePCDFileBinary(const std::string& path, const PointType* pts, size_t n, bool octree_layout, std::string* err)
{
    std::ofstream f;
    if (!openPCDFileBinary(path, n, octree_layout, &f, err)) return false;
    if (n) f.write(reinterpret_cast<const char*>(pts), (std::streamsize)(n * sizeof(PointType)));
    if (!f) { if (err) *err = path + ": write failed"; return false; }
    return true;
}

bool& logQuiet() { thread_local bool quiet = false; return quiet; }

struct AsyncWriter::Impl
{
    std::mutex m;
    std::condition_variable cv_work, cv_idle;
    std::deque<std::function<void()>> q;
    size_t running = 0;
    bool stop = false;
    std::exception_ptr err;
    std::vector<std::thread> pool;
};
AsyncWriter::AsyncWriter(unsigned threads) : impl_(new Impl)
{
    for (unsigned t = 0; t < std::max(1u, threads); ++t)
        impl_->pool.emplace_back([this] {
            Impl& I = *impl_;
            for (;;) {
                std::function<void()> task;
                {
                    std::unique_lock<std::mutex> lk(I.m);
                    I.cv_work.wait(lk, [&] { return I.stop || !I.q.empty(); });
                    if (I.q.empty()) return;
                    task = std::move(I.q.front());
                    I.q.pop_front();
                    ++I.running;
                }
                try { task(); } catch (...) { std::lock_guard<std::mutex> g(I.m); if (!I.err) I.err = std::current_exception(); }
                {
                    std::lock_guard<std::mutex> g(I.m);
                    --I.running;
                    if (I.q.empty() && I.running == 0) I.cv_idle.notify_all();
                }
            }
        });
}
AsyncWriter::~AsyncWriter()
{
    { std::lock_guard<std::mutex> g(impl_->m); impl_->stop = true; }
    impl_->cv_work.notify_all();
    for (auto& t : impl_->pool) t.join();
    delete impl_;
}
void AsyncWriter::submit(std::function<void()> task)
{
    { std::lock_guard<std::mutex> g(impl_->m); impl_->q.push_back(std::move(task)); }
    impl_->cv_work.notify_one();
}
void AsyncWriter::drain()
{
    std::unique_lock<std::mutex> lk(impl_->m);
    impl_->cv_idle.wait(lk, [&] { return impl_->q.empty() && impl_->running == 0; });
    if (impl_->err) { std::exception_ptr e = impl_->err; impl_->err = nullptr; std::rethrow_exception(e); }
}

// pcl::VoxelGrid::applyFilter (PCL 1.10, from its published behaviour): inverse leaf in float, bounding box from
// getMinMax3D, dx*dy*dz > INT32_MAX => "Leaf size is too small" and output = input; else centroids ordered by linear
// voxel index.  PCL sorts its (voxel, point) pairs with std::sort on the VOXEL INDEX ONLY (cloud_point_index_idx::operator<), so the
// float summation order inside a voxel is whatever the C++ library's std::sort leaves; the same call with the same comparator on the
// same sequence is made here, which reproduces it (round 4: the reference's own sources compiled against stand-in headers,
// oracle/_ref, differ from an input-order sum in the last bit of ~0.05 % of the loaded points of an os1-64 scan).  PCL itself is not
// in /root/reference: "parity unpinned" for the rest of the routine.
void voxelGridFilter(const Cloud& in, float leaf, Cloud& out)
{
    if (in.empty()) { out.clear(); return; }
    const float inv = 1.0f / leaf;
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float mx[3] = {-mn[0], -mn[1], -mn[2]};
    for (const PointType& p : in) {
        const float c[3] = {p.x, p.y, p.z};
        for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], c[d]); mx[d] = std::max(mx[d], c[d]); }
    }
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)std::numeric_limits<int32_t>::max()) { out = in; return; }      // "leaf size is too small": output = input
    int minb[3], divb[3];
    for (int d = 0; d < 3; ++d) {
        minb[d] = (int)std::floor(mn[d] * inv);
        divb[d] = (int)std::floor(mx[d] * inv) - minb[d] + 1;
    }
    std::vector<ltm_pclsort::Entry> keyed(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
        const int i0 = (int)(std::floor(in[i].x * inv) - (float)minb[0]), i1 = (int)(std::floor(in[i].y * inv) - (float)minb[1]),
                  i2 = (int)(std::floor(in[i].z * inv) - (float)minb[2]);
        keyed[i] = ltm_pclsort::Entry{(uint32_t)(i0 + i1 * divb[0] + i2 * divb[0] * divb[1]), (uint32_t)i};
    }
    // std::sort's permutation for the leaf-index-only comparator, without its branch mispredictions (csrc/ltm_pclsort.h, checked against
    // std::sort itself in tests/test_abi.py; the library's device hand-over uses the same routine)
    ltm_pclsort::sort(keyed.data(), keyed.data() + keyed.size());
    Cloud res;
    size_t a = 0;
    while (a < keyed.size()) {
        size_t b = a;
        float sx = 0, sy = 0, sz = 0, si = 0;
        while (b < keyed.size() && keyed[b].idx == keyed[a].idx) {
            const PointType& p = in[keyed[b].cloud_point_index];
            sx += p.x; sy += p.y; sz += p.z; si += p.intensity;
            ++b;
        }
        const float n = (float)(b - a);
        res.push_back(PointType{sx / n, sy / n, sz / n, si / n});
        a = b;
    }
    out.swap(res);
}

// same filter; when PCL's overflow early-out applies (the usual case for a raw scan at 0.05 m) the input buffer
// becomes the output without a copy
void voxelGridFilter(Cloud&& in, float leaf, Cloud& out)
{
    if (in.empty()) { out.clear(); return; }
    const float inv = 1.0f / leaf;
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float mx[3] = {-mn[0], -mn[1], -mn[2]};
    for (const PointType& p : in) {
        const float c[3] = {p.x, p.y, p.z};
        for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], c[d]); mx[d] = std::max(mx[d], c[d]); }
    }
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)std::numeric_limits<int32_t>::max()) { out = std::move(in); return; }
    voxelGridFilter(static_cast<const Cloud&>(in), leaf, out);
}

} // namespace ltremovert
