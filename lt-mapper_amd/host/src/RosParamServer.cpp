#include "removert/RosParamServer.h"
#include <cstdlib>

#include <algorithm>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace {
std::string g_param_file;

std::string trim(const std::string& s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
std::string strip_comment(const std::string& line)
{
    bool q = false; char qc = 0;
    for (size_t i = 0; i < line.size(); ++i) {
        const char c = line[i];
        if (q) { if (c == qc) q = false; }
        else if (c == '"' || c == '\'') { q = true; qc = c; }
        else if (c == '#') return line.substr(0, i);
    }
    return line;
}
std::string unquote(std::string v)
{
    v = trim(v);
    if (v.size() >= 2 && (v.front() == '"' || v.front() == '\'') && v.back() == v.front()) v = v.substr(1, v.size() - 2);
    return v;
}
template <class T> std::vector<T> parse_list(const std::string& v)
{
    std::vector<T> out;
    std::string s = v;
    std::replace(s.begin(), s.end(), '[', ' '); std::replace(s.begin(), s.end(), ']', ' '); std::replace(s.begin(), s.end(), ',', ' ');
    std::stringstream ss(s);
    double d;
    while (ss >> d) out.push_back((T)d);
    return out;
}
bool parse_bool(const std::string& v, bool dflt)
{
    std::string s = unquote(v);
    std::transform(s.begin(), s.end(), s.begin(), ::tolower);
    if (s == "true" || s == "1" || s == "yes" || s == "on") return true;
    if (s == "false" || s == "0" || s == "no" || s == "off") return false;
    return dflt;
}
} // namespace

void RosParamServer::setParamFile(const std::string& path) { g_param_file = path; }
std::string RosParamServer::paramFile() { return g_param_file; }

// YAML subset: a top-level `ns:` mapping whose values are scalars, quoted strings or flow lists (possibly multi-line)
std::map<std::string, std::string> RosParamServer::readYamlNamespace(const std::string& path, const std::string& ns)
{
    std::map<std::string, std::string> kv;
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open parameter file " + path);
    std::string line, key, val;
    bool in_ns = false;
    int depth = 0;
    auto flush = [&] { if (!key.empty()) kv[key] = trim(val); key.clear(); val.clear(); };
    while (std::getline(f, line)) {
        line = strip_comment(line);
        if (trim(line).empty()) continue;
        const size_t indent = line.find_first_not_of(' ');
        if (depth > 0) {                                  // continuation of a multi-line flow list
            val += " " + trim(line);
            for (char c : line) { if (c == '[') ++depth; else if (c == ']') --depth; }
            if (depth == 0) flush();
            continue;
        }
        if (indent == 0) { flush(); in_ns = trim(line) == ns + ":"; continue; }
        if (!in_ns) continue;
        const size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        flush();
        key = trim(line.substr(0, colon));
        val = trim(line.substr(colon + 1));
        for (char c : val) { if (c == '[') ++depth; else if (c == ']') --depth; }
        if (depth == 0) flush();
    }
    flush();
    return kv;
}

RosParamServer::RosParamServer()
{
    std::map<std::string, std::string> p;
    if (!g_param_file.empty()) p = readYamlNamespace(g_param_file, "removert");
    auto has = [&](const char* k) { return p.count(k) != 0; };
    auto getf = [&](const char* k, float d) { return has(k) ? std::stof(unquote(p[k])) : d; };
    auto geti = [&](const char* k, int d) { return has(k) ? std::stoi(unquote(p[k])) : d; };
    auto getb = [&](const char* k, bool d) { return has(k) ? parse_bool(p[k], d) : d; };
    auto gets = [&](const char* k, const char* d) { return has(k) ? unquote(p[k]) : std::string(d); };

    // defaults: RosParamServer.cpp:7-59
    isScanFileKITTIFormat_ = getb("isScanFileKITTIFormat", true);
    rimg_color_min_ = getf("rimg_color_min", 0.0f);
    rimg_color_max_ = getf("rimg_color_max", 10.0f);
    kRangeColorAxis = {rimg_color_min_, rimg_color_max_};
    kRangeColorAxisForDiff = {0.0f, 0.5f};
    kVFOV = getf("sequence_vfov", 50.0f);
    kHFOV = getf("sequence_hfov", 360.0f);
    kFOV = {kVFOV, kHFOV};
    if (has("remove_resolution_list")) remove_resolution_list_ = parse_list<float>(p["remove_resolution_list"]);
    if (has("revert_resolution_list")) revert_resolution_list_ = parse_list<float>(p["revert_resolution_list"]);
    kNumKnnPointsToCompare = geti("num_nn_points_within", 3);
    kScanKnnAndMapKnnAvgDiffThreshold = getf("dist_nn_points_within", 0.1f);
    if (has("ExtrinsicLiDARtoPoseBase")) kVecExtrinsicLiDARtoPoseBase = parse_list<double>(p["ExtrinsicLiDARtoPoseBase"]);
    // the reference maps an EMPTY vector here (undefined behaviour, RosParamServer.cpp:29); we require 16 values or use I
    if (kVecExtrinsicLiDARtoPoseBase.size() != 16)
        kVecExtrinsicLiDARtoPoseBase = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    kSE3MatExtrinsicLiDARtoPoseBase = kVecExtrinsicLiDARtoPoseBase;
    kSE3MatExtrinsicPoseBasetoLiDAR.assign(16, 0.0);
    if (!ltremovert::inverse4x4(kSE3MatExtrinsicLiDARtoPoseBase.data(), kSE3MatExtrinsicPoseBasetoLiDAR.data()))
        throw std::runtime_error("ExtrinsicLiDARtoPoseBase is singular");
    kDownsampleVoxelSize = getf("downsample_voxel_size", 0.05f);
    central_sess_scan_dir_ = gets("central_sess_scan_dir", "/use/your/directory/having/*.bin");
    central_sess_pose_path_ = gets("central_sess_pose_path", "/use/your/path/having/pose.txt");
    query_sess_scan_dir_ = gets("query_sess_scan_dir", "/use/your/directory/having/*.bin");
    query_sess_pose_path_ = gets("query_sess_pose_path", "/use/your/path/having/pose.txt");
    start_idx_ = geti("start_idx", 1);
    end_idx_ = geti("end_idx", 100);
    use_keyframe_gap_ = getb("use_keyframe_gap", true);
    use_keyframe_meter_ = getb("use_keyframe_meter", false);
    keyframe_gap_ = geti("keyframe_gap", 10);
    keyframe_gap_meter_ = getf("keyframe_meter", 2.0f);
    repeat_removert_iter_ = geti("repeat_removert_iter", 1);
    kNumOmpCores = geti("num_omp_cores", 4);
    kFlagSaveMapPointcloud = getb("saveMapPCD", false);
    kFlagSaveCleanScans = getb("saveCleanScansPCD", false);
    save_pcd_directory_ = gets("save_pcd_directory", "/");
    gpu_use_self_removert_ = getb("gpu_use_self_removert", false);
    gpu_skip_hd_knn_ = getb("gpu_skip_hd_knn", false);
    gpu_device_ = geti("gpu_device", 0);
    gpu_lanes_ = geti("gpu_lanes", 2);
    if (const char* v = std::getenv("LTM_LANES")) gpu_lanes_ = std::atoi(v);
    gpu_async_io_ = getb("gpu_async_io", true);
    gpu_fetch_chunked_ = getb("gpu_fetch_chunked", true);
    gpu_viz_every_ = geti("gpu_viz_every", 0);
}
