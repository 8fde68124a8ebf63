/*
 * TEST INFRASTRUCTURE -- stand-in for the roscpp / message / tf / image_transport / cv_bridge declarations that the UNMODIFIED
 * reference sources (the .cpp files of /root/reference/ltremovert/src and the headers of include/removert) use, so that they compile here without ROS
 * (oracle/refshim/Makefile builds them into oracle/_ref/).  Nothing of ROS is restated: parameters come from a table (filled from
 * a params_ltmapper.yaml-style file named by $REFSHIM_PARAMS or through refshim::set_param), publishers drop their messages,
 * log macros print only when $REFSHIM_VERBOSE is set.  Original code; no reference or ROS text.
 */
#ifndef REFSHIM_ROS_H
#define REFSHIM_ROS_H

#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

namespace boost {           /* the reference reaches boost only through PCL's smart pointers (Removerter.cpp:938) */
using std::make_shared;
using std::shared_ptr;
}

namespace refshim {

inline bool verbose()
{
    static const bool v = std::getenv("REFSHIM_VERBOSE") != nullptr;
    return v;
}

/* parameter table: full key ("removert/start_idx") -> raw text of the value */
inline std::map<std::string, std::string>& params()
{
    static std::map<std::string, std::string> p;
    return p;
}

inline std::string trim(const std::string& s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

inline std::string strip_comment(const std::string& s)
{
    bool q = false;
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] == '"') q = !q;
        if (s[i] == '#' && !q) return s.substr(0, i);
    }
    return s;
}

/* the subset of YAML that params_ltmapper.yaml uses: one top-level namespace, "key: value" lines, flow lists that may span lines */
inline void load_yaml(const std::string& path)
{
    std::ifstream f(path);
    if (!f) { std::fprintf(stderr, "refshim: cannot read %s\n", path.c_str()); std::exit(2); }
    std::string line, ns, key, acc;
    bool in_list = false;
    while (std::getline(f, line)) {
        std::string body = strip_comment(line);
        if (trim(body).empty()) continue;
        if (in_list) {
            acc += " " + trim(body);
            if (body.find(']') != std::string::npos) { params()[key] = acc; in_list = false; }
            continue;
        }
        const size_t colon = body.find(':');
        if (colon == std::string::npos) continue;
        const bool top = body[0] != ' ' && body[0] != '\t';
        const std::string k = trim(body.substr(0, colon)), v = trim(body.substr(colon + 1));
        if (top && v.empty()) { ns = k + "/"; continue; }
        key = (top ? std::string() : ns) + k;
        if (!v.empty() && v[0] == '[' && v.find(']') == std::string::npos) { acc = v; in_list = true; continue; }
        params()[key] = v;
    }
}

inline void set_param(const std::string& key, const std::string& value) { params()[key] = value; }

inline void ensure_loaded()
{
    static bool done = false;
    if (done) return;
    done = true;
    if (const char* p = std::getenv("REFSHIM_PARAMS")) load_yaml(p);
}

inline std::string unquote(std::string v)
{
    v = trim(v);
    if (v.size() >= 2 && (v.front() == '"' || v.front() == '\'') && v.back() == v.front()) v = v.substr(1, v.size() - 2);
    return v;
}

template <class T> inline void convert(const std::string& raw, T& out) { std::istringstream(unquote(raw)) >> out; }
template <> inline void convert<std::string>(const std::string& raw, std::string& out) { out = unquote(raw); }
template <> inline void convert<bool>(const std::string& raw, bool& out)
{
    const std::string v = unquote(raw);
    out = (v == "true" || v == "True" || v == "1");
}
template <class E> inline void convert_list(const std::string& raw, std::vector<E>& out)
{
    out.clear();
    std::string v = trim(raw);
    const size_t a = v.find('['), b = v.rfind(']');
    if (a == std::string::npos || b == std::string::npos) return;
    std::stringstream ss(v.substr(a + 1, b - a - 1));
    std::string item;
    while (std::getline(ss, item, ',')) {
        if (trim(item).empty()) continue;
        E e{};
        convert(item, e);
        out.push_back(e);
    }
}
template <> inline void convert<std::vector<float>>(const std::string& raw, std::vector<float>& out) { convert_list(raw, out); }
template <> inline void convert<std::vector<double>>(const std::string& raw, std::vector<double>& out) { convert_list(raw, out); }

} // namespace refshim

namespace ros {

struct Time {
    double sec = 0;
    static Time now() { return Time(); }
    double toSec() const { return sec; }
};

struct TransportHints {
    TransportHints& tcpNoDelay() { return *this; }
};

class Publisher {
public:
    template <class M> void publish(const M&) const {}
    uint32_t getNumSubscribers() const { return 0; }
};

class Subscriber {};

class NodeHandle {
public:
    /* roscpp semantics: the stored value if the key exists, the default otherwise */
    template <class T> bool param(const std::string& name, T& out, const T& dflt) const
    {
        refshim::ensure_loaded();
        auto it = refshim::params().find(name);
        if (it == refshim::params().end()) { out = dflt; return false; }
        refshim::convert(it->second, out);
        return true;
    }
    template <class T> bool param(const std::string& name, T& out, const char* dflt) const { return param<T>(name, out, T(dflt)); }
    template <class M> Publisher advertise(const std::string&, uint32_t) { return Publisher(); }
    template <class M, class... A> Subscriber subscribe(A&&...) { return Subscriber(); }
};

inline void init(int&, char**, const std::string&) {}
inline void spin() {}                        /* the reference never leaves ros::spin(); the stand-in returns so the process ends */

} // namespace ros

#define ROS_INFO_STREAM(args) do { if (refshim::verbose()) { std::cerr << args << std::endl; } } while (0)
#define ROS_INFO(...)         do { if (refshim::verbose()) { std::fprintf(stderr, __VA_ARGS__); std::fputc('\n', stderr); } } while (0)
#define ROS_WARN_STREAM(args) ROS_INFO_STREAM(args)

namespace std_msgs {
struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; };
struct Float64MultiArray { std::vector<double> data; };
}

namespace sensor_msgs {
struct PointCloud2 { std_msgs::Header header; std::vector<uint8_t> data; };
typedef boost::shared_ptr<PointCloud2> PointCloud2Ptr;
typedef boost::shared_ptr<PointCloud2 const> PointCloud2ConstPtr;
struct Image { std_msgs::Header header; uint32_t height = 0, width = 0; std::string encoding; std::vector<uint8_t> data; };
typedef boost::shared_ptr<Image> ImagePtr;
struct Imu {};
struct NavSatFix {};
}

namespace nav_msgs { struct Odometry {}; struct Path {}; }
namespace visualization_msgs { struct Marker {}; struct MarkerArray {}; }

namespace image_transport {
class Publisher {
public:
    template <class M> void publish(const M&) const {}
    uint32_t getNumSubscribers() const { return 0; }
};
class ImageTransport {
public:
    explicit ImageTransport(const ros::NodeHandle&) {}
    Publisher advertise(const std::string&, uint32_t) { return Publisher(); }
};
}

#endif
