// host_selftest.cpp -- CPU-only self test of the host plumbing (no GPU, no libltm_hip.so): PCD v0.7 reader/writer
// (ascii, binary, binary_compressed), pose-line parsing, the YAML-subset parameter reader, keyframe selection quirks and
// the VoxelGrid restatement.  Built as lt-mapper_amd/host/host_selftest and run by tests/test_host_cpp.py.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "removert/RosParamServer.h"
#include "removert/utility.h"

using namespace ltremovert;

static int fails = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++fails; } \
    } while (0)

// minimal LZF compressor (literal runs only -- a valid LZF stream) to fabricate a binary_compressed file
static std::string lzf_literals(const std::string& in)
{
    std::string out;
    for (size_t i = 0; i < in.size(); i += 32) {
        const size_t n = std::min<size_t>(32, in.size() - i);
        out.push_back((char)(n - 1));
        out.append(in, i, n);
    }
    return out;
}

// greedy LZF compressor WITH back references (liblzf's stream format, the one pcl::io::savePCDFileBinaryCompressed writes): 3-byte hash chains of depth 1,
// matches of 3..264 bytes at distances 1..8192 -- short and extended length codes, overlapping copies (distance < length) and literal runs all occur
static std::string lzf_greedy(const std::string& in, size_t* n_refs = nullptr, size_t* n_long = nullptr, size_t* n_overlap = nullptr)
{
    const unsigned char* d = reinterpret_cast<const unsigned char*>(in.data());
    const size_t n = in.size();
    std::string out, lit;
    std::vector<long> last(1 << 16, -1);
    auto flush = [&] {
        for (size_t i = 0; i < lit.size(); i += 32) { const size_t m = std::min<size_t>(32, lit.size() - i); out.push_back((char)(m - 1)); out.append(lit, i, m); }
        lit.clear();
    };
    size_t i = 0;
    while (i < n) {
        size_t best = 0, dist = 0;
        if (i + 3 <= n) {
            const unsigned h = ((d[i] * 2654435761u) ^ (d[i + 1] * 40503u) ^ (d[i + 2] * 2246822519u)) & 0xffffu;
            const long c = last[h];
            last[h] = (long)i;
            if (c >= 0 && i - (size_t)c <= 8192) {
                size_t l = 0;
                while (i + l < n && l < 264 && d[(size_t)c + l] == d[i + l]) ++l;      // (reads past i when the match overlaps itself: that is the point)
                if (l >= 3) { best = l; dist = i - (size_t)c; }
            }
        }
        if (!best) { lit.push_back((char)d[i++]); continue; }
        flush();
        const unsigned off = (unsigned)(dist - 1), L = (unsigned)(best - 2);
        if (L < 7) out.push_back((char)((L << 5) | (off >> 8)));
        else { out.push_back((char)((7u << 5) | (off >> 8))); out.push_back((char)(L - 7)); if (n_long) ++*n_long; }
        out.push_back((char)(off & 0xffu));
        if (n_refs) ++*n_refs;
        if (n_overlap && dist < best) ++*n_overlap;
        i += best;
    }
    flush();
    return out;
}

int main(int argc, char** argv)
{
    // host_selftest --voxelgrid <in.pcd> <leaf> <out.pcd>: the loader's per-scan step (loadPCDFile + pcl::VoxelGrid restatement,
    // Session.cpp:272-292) on one file, so that tests can compare it with the oracle's restatement
    if (argc == 5 && std::string(argv[1]) == "--voxelgrid") {
        Cloud in, out;
        std::string err;
        if (!loadPCDFile(argv[2], in, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        voxelGridFilter(std::move(in), std::stof(argv[3]), out);
        if (!savePCDFileBinary(argv[4], out, false, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        return 0;
    }
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    Cloud c;
    for (int i = 0; i < 1000; ++i) c.push_back(PointType{0.1f * i, -0.25f * i, 1.0f / (i + 1), (float)(i % 256)});

    // binary round trip, both layouts
    std::string err;
    CHECK(savePCDFileBinary(dir + "/a.pcd", c, false, &err));
    CHECK(savePCDFileBinary(dir + "/b.pcd", c, true, &err));
    Cloud r;
    CHECK(loadPCDFile(dir + "/a.pcd", r, &err) && r.size() == c.size() && std::memcmp(r.data(), c.data(), c.size() * 16) == 0);
    CHECK(loadPCDFile(dir + "/b.pcd", r, &err) && r.size() == c.size() && std::memcmp(r.data(), c.data(), c.size() * 16) == 0);
    {
        std::ifstream f(dir + "/b.pcd", std::ios::binary);
        std::string head((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        CHECK(head.find("WIDTH 1\nHEIGHT 1000\n") != std::string::npos);
        CHECK(head.find("FIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n") != std::string::npos);
    }
    // the piecewise writer of the chunked output path: header first, points in pieces -- byte for byte the one-shot file
    {
        for (bool octree_layout : {false, true}) {
            std::ofstream f;
            CHECK(openPCDFileBinary(dir + "/pieces.pcd", c.size(), octree_layout, &f, &err));
            const size_t cuts[4] = {0, 1, 637, c.size()};
            for (int k = 0; k < 3; ++k) f.write(reinterpret_cast<const char*>(c.data() + cuts[k]), (std::streamsize)((cuts[k + 1] - cuts[k]) * sizeof(PointType)));
            f.close();
            std::ifstream a(dir + (octree_layout ? "/b.pcd" : "/a.pcd"), std::ios::binary), b(dir + "/pieces.pcd", std::ios::binary);
            const std::string sa((std::istreambuf_iterator<char>(a)), std::istreambuf_iterator<char>()), sb((std::istreambuf_iterator<char>(b)), std::istreambuf_iterator<char>());
            CHECK(!sa.empty() && sa == sb);
        }
        std::ofstream f;
        CHECK(!openPCDFileBinary(dir + "/no/such/dir/x.pcd", 1, false, &f, &err) && !err.empty());
    }
    // AsyncWriter: every task runs, drain() re-throws the first failure and the pool stays usable
    {
        std::atomic<int> ran{0};
        AsyncWriter w(3);
        for (int i = 0; i < 50; ++i) w.submit([&ran, i] { ++ran; if (i == 17) throw std::runtime_error("task 17"); });
        bool threw = false;
        try { w.drain(); } catch (const std::exception& e) { threw = std::string(e.what()) == "task 17"; }
        CHECK(threw && ran == 50);
        w.submit([&ran] { ++ran; });
        bool ok_after = true;
        try { w.drain(); } catch (...) { ok_after = false; }
        CHECK(ran == 51 && ok_after);      // the failure was reported once and cleared
    }
    // ascii with an extra field and a different field order
    {
        std::ofstream f(dir + "/c.pcd");
        f << "# .PCD v0.7\nVERSION 0.7\nFIELDS intensity x y z ring\nSIZE 4 4 4 4 2\nTYPE F F F F U\nCOUNT 1 1 1 1 1\nWIDTH 2\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA ascii\n"
          << "7 1.5 2.5 3.5 11\n9 -1 -2 -3 12\n";
    }
    CHECK(loadPCDFile(dir + "/c.pcd", r, &err) && r.size() == 2 && r[0].x == 1.5f && r[0].intensity == 7.0f && r[1].z == -3.0f && r[1].intensity == 9.0f);
    // binary with an extra field and another field order: the general record decoder (the in-place read is for plain x y z intensity float32 only)
    {
        std::ofstream f(dir + "/c2.pcd", std::ios::binary);
        f << "VERSION 0.7\nFIELDS intensity x y z ring\nSIZE 4 4 4 4 2\nTYPE F F F F U\nCOUNT 1 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA binary\n";
        const float a[4] = {7.0f, 1.5f, 2.5f, 3.5f}, b[4] = {9.0f, -1.0f, -2.0f, -3.0f};
        const uint16_t ra = 11, rb = 12;
        f.write(reinterpret_cast<const char*>(a), 16); f.write(reinterpret_cast<const char*>(&ra), 2);
        f.write(reinterpret_cast<const char*>(b), 16); f.write(reinterpret_cast<const char*>(&rb), 2);
    }
    CHECK(loadPCDFile(dir + "/c2.pcd", r, &err) && r.size() == 2 && r[0].x == 1.5f && r[0].intensity == 7.0f && r[1].z == -3.0f && r[1].intensity == 9.0f);
    // binary_compressed: structure-of-arrays payload, LZF literal stream
    {
        std::string soa;
        for (int fld = 0; fld < 4; ++fld)
            for (const PointType& p : c) { const float v = fld == 0 ? p.x : fld == 1 ? p.y : fld == 2 ? p.z : p.intensity; soa.append(reinterpret_cast<const char*>(&v), 4); }
        const std::string comp = lzf_literals(soa);
        std::ofstream f(dir + "/d.pcd", std::ios::binary);
        f << "VERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 1000\nHEIGHT 1\nPOINTS 1000\nDATA binary_compressed\n";
        const uint32_t cs = (uint32_t)comp.size(), us = (uint32_t)soa.size();
        f.write(reinterpret_cast<const char*>(&cs), 4); f.write(reinterpret_cast<const char*>(&us), 4); f.write(comp.data(), cs);
    }
    CHECK(loadPCDFile(dir + "/d.pcd", r, &err) && r.size() == c.size() && std::memcmp(r.data(), c.data(), c.size() * 16) == 0);
    // binary_compressed with BACK REFERENCES (the literal-only stream above never ran the decoder's match branch): a cloud with
    // constant and periodic columns compresses into short, extended-length and self-overlapping matches
    {
        Cloud rep;
        for (int i = 0; i < 5000; ++i) rep.push_back(PointType{0.1f * (float)(i % 37), i < 2500 ? 1.25f : -0.0f, 7.0f, (float)(i % 4)});
        std::string soa;
        for (int fld = 0; fld < 4; ++fld)
            for (const PointType& p : rep) { const float v = fld == 0 ? p.x : fld == 1 ? p.y : fld == 2 ? p.z : p.intensity; soa.append(reinterpret_cast<const char*>(&v), 4); }
        size_t refs = 0, longs = 0, overlaps = 0;
        const std::string comp = lzf_greedy(soa, &refs, &longs, &overlaps);
        CHECK(refs > 100 && longs > 10 && overlaps > 0 && comp.size() < soa.size() / 4);
        std::string back(soa.size(), '\0');
        CHECK(lzf_decompress(reinterpret_cast<const unsigned char*>(comp.data()), comp.size(), reinterpret_cast<unsigned char*>(&back[0]), back.size()) && back == soa);
        // a truncated stream, a match that reaches before the start of the output and an output that is too small are refused, not read out of bounds
        CHECK(!lzf_decompress(reinterpret_cast<const unsigned char*>(comp.data()), comp.size() - 1, reinterpret_cast<unsigned char*>(&back[0]), back.size()));
        CHECK(!lzf_decompress(reinterpret_cast<const unsigned char*>(comp.data()), comp.size(), reinterpret_cast<unsigned char*>(&back[0]), back.size() - 8));
        const unsigned char bad_ref[3] = {0x20, 0x05, 0x00};      // match of 3 bytes at distance 6 into an empty output
        CHECK(!lzf_decompress(bad_ref, 2, reinterpret_cast<unsigned char*>(&back[0]), 3));
        std::ofstream f(dir + "/e.pcd", std::ios::binary);
        f << "VERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 5000\nHEIGHT 1\nPOINTS 5000\nDATA binary_compressed\n";
        const uint32_t cs = (uint32_t)comp.size(), us = (uint32_t)soa.size();
        f.write(reinterpret_cast<const char*>(&cs), 4); f.write(reinterpret_cast<const char*>(&us), 4); f.write(comp.data(), cs);
        f.close();
        CHECK(loadPCDFile(dir + "/e.pcd", r, &err) && r.size() == rep.size() && std::memcmp(r.data(), rep.data(), rep.size() * 16) == 0);
    }
    CHECK(!loadPCDFile(dir + "/does_not_exist.pcd", r, &err) && !err.empty());

    // pose lines (Session.cpp:102-114)
    auto v = splitPoseLine("1 0 0 1.5 0 1 0 -2 0 0 1 0.25", ' ');
    CHECK(v.size() == 12 && v[3] == 1.5 && v[7] == -2.0 && v[11] == 0.25);
    double m[16] = {0, -1, 0, 3, 1, 0, 0, -4, 0, 0, 1, 5, 0, 0, 0, 1}, inv[16];
    CHECK(inverse4x4(m, inv) && std::fabs(inv[3] - 4.0) < 1e-12 && std::fabs(inv[7] - 3.0) < 1e-12 && std::fabs(inv[11] + 5.0) < 1e-12);
    double sing[16] = {0};
    CHECK(!inverse4x4(sing, inv));
    CHECK(resetRimgSize({50.0f, 360.0f}, 2.5f) == std::make_pair(125, 900));

    // YAML subset: the reference's own parameter file layout (config/params_ltmapper.yaml)
    {
        std::ofstream f(dir + "/p.yaml");
        f << "removert:\n\n  # comment\n  isScanFileKITTIFormat: false\n  saveMapPCD: true \n  save_pcd_directory: \"/tmp/out dir/\" # trailing comment\n"
          << "  sequence_vfov: 50 # including upper\n  ExtrinsicLiDARtoPoseBase: [1.0, 0.0, 0.0, 0.5, \n                             0.0, 1.0, 0.0, 0.0, \n"
          << "                             0.0, 0.0, 1.0, 0.0, \n                             0.0, 0.0, 0.0, 1.0]\n  start_idx: 1100 # change this\n"
          << "  remove_resolution_list: [2.5, 2.0, 1.5] # for Ouster\n  num_nn_points_within: 2\n  dist_nn_points_within: 0.01\nother:\n  start_idx: 7\n";
    }
    RosParamServer::setParamFile(dir + "/p.yaml");
    RosParamServer P;
    CHECK(!P.isScanFileKITTIFormat_ && P.kFlagSaveMapPointcloud && P.save_pcd_directory_ == "/tmp/out dir/");
    CHECK(P.kVFOV == 50.0f && P.kHFOV == 360.0f && P.start_idx_ == 1100 && P.end_idx_ == 100);
    CHECK(P.remove_resolution_list_.size() == 3 && P.remove_resolution_list_[2] == 1.5f);
    CHECK(P.kNumKnnPointsToCompare == 2 && P.kScanKnnAndMapKnnAvgDiffThreshold == 0.01f && P.kDownsampleVoxelSize == 0.05f);
    CHECK(P.kSE3MatExtrinsicLiDARtoPoseBase[3] == 0.5 && std::fabs(P.kSE3MatExtrinsicPoseBasetoLiDAR[3] + 0.5) < 1e-15);
    CHECK(!P.gpu_use_self_removert_ && P.keyframe_gap_ == 10 && P.kNumOmpCores == 4);

    // VoxelGrid restatement: centroid per voxel in linear-index order; overflow early-out returns the input
    Cloud g = {{0.01f, 0.01f, 0.01f, 1}, {0.02f, 0.02f, 0.02f, 3}, {1.01f, 0.0f, 0.0f, 5}}, o;
    voxelGridFilter(g, 0.05f, o);
    CHECK(o.size() == 2 && o[0].intensity == 2.0f && o[1].x == 1.01f);
    Cloud big = {{-1000.f, -1000.f, -100.f, 0}, {1000.f, 1000.f, 100.f, 0}};
    voxelGridFilter(big, 0.05f, o);
    CHECK(o.size() == 2 && o[0].x == -1000.f);

    std::printf(fails ? "host_selftest: %d FAILED\n" : "host_selftest: all checks passed\n", fails);
    return fails ? 1 : 0;
}
