// loader_driver.cpp -- loader_driver <features.txt> <d> <metric> <out.bin> [threads] : parse with the fast loader,
// round-trip through the binary cache, dump rows + classes for tests/test_host_loader.py (no GPU involved).
// loader_driver --videos <file> <d> <metric> <out.bin> : the video-feature file of video.cpp:35-96.
// loader_driver --tokens <file> : parse every whitespace-separated token with parse_float_exact, print the float bits (REJECT for what `istream >> float` refuses).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "fir_loader.h"

int main(int argc, char** argv) {
    if (argc >= 3 && !std::strcmp(argv[1], "--tokens")) {
        std::ifstream in(argv[2]);
        std::stringstream ss;
        ss << in.rdbuf();
        const std::string s = ss.str();
        const char* p = s.data();
        const char* limit = p + s.size();
        for (;;) {
            while (p < limit && (*p == ' ' || *p == '\n')) ++p;
            if (p >= limit) break;
            const char* e = p;
            const float v = fir::parse_float_exact(p, limit, &e);
            if (e == p) {                                   // not a number for `istream >> float`: skip the token
                std::printf("REJECT\n");
                while (p < limit && *p != ' ' && *p != '\n') ++p;
                continue;
            }
            uint32_t bits;
            std::memcpy(&bits, &v, 4);
            std::printf("%u\n", bits);
            p = e;
        }
        return 0;
    }
    if (argc >= 6 && !std::strcmp(argv[1], "--videos")) {       // --videos <file> <d> <metric> <out.bin>: video.cpp:35-96
        fir::PackedVideos v;
        const int64_t persons = fir::load_videos_packed(argv[2], std::atoi(argv[3]), std::atoi(argv[4]), v);
        FILE* fp = std::fopen(argv[5], "wb");
        std::fwrite(v.rows.data(), sizeof(float), v.rows.size(), fp);
        std::fclose(fp);
        std::printf("{\"persons\": %lld, \"total_images\": %lld, \"total_videos\": %lld, \"names\": [", (long long)persons,
                    (long long)v.total_images, (long long)v.total_videos);
        for (size_t i = 0; i < v.person.size(); ++i) std::printf("%s\"%s\"", i ? ", " : "", v.person[i].c_str());
        std::printf("], \"video_first\": [");
        for (size_t i = 0; i < v.video_first.size(); ++i) std::printf("%s%d", i ? ", " : "", v.video_first[i]);
        std::printf("], \"frame_first\": [");
        for (size_t i = 0; i < v.frame_first.size(); ++i) std::printf("%s%lld", i ? ", " : "", (long long)v.frame_first[i]);
        std::printf("]}\n");
        return 0;
    }
    if (argc < 5) { std::fprintf(stderr, "usage: loader_driver <features.txt> <d> <metric> <out.bin> [threads]\n"); return 2; }
    fir::PackedFeatures f, g;
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t n = fir::load_features_packed(argv[1], std::atoi(argv[2]), std::atoi(argv[3]), f, argc > 5 ? std::atoi(argv[5]) : 0);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const std::string cache = std::string(argv[4]) + ".cache";
    if (fir::save_feature_cache(cache, f) != 0 || fir::load_feature_cache(cache, g) != 0) { std::fprintf(stderr, "cache round trip failed\n"); return 1; }
    if (g.n != f.n || g.rows != f.rows || g.class_no != f.class_no || g.class_names != f.class_names || g.index_in_file != f.index_in_file) {
        std::fprintf(stderr, "cache mismatch\n");
        return 1;
    }
    FILE* fp = std::fopen(argv[4], "wb");
    std::fwrite(g.rows.data(), sizeof(float), g.rows.size(), fp);
    std::fwrite(g.class_no.data(), sizeof(int32_t), g.class_no.size(), fp);
    std::fclose(fp);
    std::printf("{\"images\": %lld, \"classes\": %zu, \"seconds\": %.4f}\n", (long long)n, g.class_names.size(), secs);
    return 0;
}
