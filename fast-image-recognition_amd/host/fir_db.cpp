// fir_db.cpp -- host side of the drop-in (see fir_db.h). IO, packing and handle caching only:
// every distance is computed by libfir_amd.so on the GPU.
#include "fir_db.h"

#include <iostream>
#include "fir_loader.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

namespace fir {

namespace {
int g_device = 0;
int g_metric = FIR_DEFAULT_METRIC;
int g_large_batch = 0;

struct CacheKey {
    const void* vec;
    size_t n;
    const float* first;
    const float* last;
    int dim;
    int metric;
    bool operator<(const CacheKey& o) const {
        return std::tie(vec, n, first, last, dim, metric) < std::tie(o.vec, o.n, o.first, o.last, o.dim, o.metric);
    }
};
std::map<CacheKey, fir_gallery*> g_cache;

CacheKey key_of(const std::vector<ImageInfo>& db, int dim) {
    CacheKey k;
    k.vec = &db;
    k.n = db.size();
    k.first = db.empty() ? nullptr : db.front().features.data();
    k.last = db.empty() ? nullptr : db.back().features.data();
    k.dim = dim;
    k.metric = g_metric;
    return k;
}
}  // namespace

void set_large_batch_mfma(int min_queries) { g_large_batch = min_queries; }
void set_device(int device) { g_device = device; }
int device() { return g_device; }
int metric() { return g_metric; }
void set_metric(int m) { g_metric = m; }
void log_error(const char* where) { std::fprintf(stderr, "fir: %s: %s\n", where, fir_last_error()); }

fir_gallery* GalleryCache::get(const std::vector<ImageInfo>& db, int dim) {
    const CacheKey k = key_of(db, dim);
    auto it = g_cache.find(k);
    if (it != g_cache.end()) return it->second;
    // a different split presented through the same vector object replaces the old upload
    for (auto i = g_cache.begin(); i != g_cache.end();) {
        if (i->first.vec == k.vec) { fir_gallery_destroy(i->second); i = g_cache.erase(i); } else ++i;
    }
    std::vector<float> rows((size_t)db.size() * dim);
    std::vector<int32_t> cls(db.size());
    for (size_t j = 0; j < db.size(); ++j) {
        const FeaturesVector& f = db[j].features;
        const size_t have = std::min<size_t>(f.size(), (size_t)dim);
        std::memcpy(&rows[j * dim], f.data(), have * sizeof(float));   // rows shorter than dim are zero padded
        cls[j] = db[j].classNo;
    }
    fir_gallery* g = nullptr;
    if (fir_gallery_create(rows.data(), (int64_t)db.size(), dim, cls.data(), g_metric, g_device, &g) != FIR_OK) {
        log_error("gallery upload");
        return nullptr;
    }
    if (g_large_batch > 0) fir_gallery_set_large_batch_mfma(g, g_large_batch);
    g_cache[k] = g;
    return g;
}

void GalleryCache::invalidate(const std::vector<ImageInfo>& db) {
    for (auto i = g_cache.begin(); i != g_cache.end();) {
        if (i->first.vec == (const void*)&db) { fir_gallery_destroy(i->second); i = g_cache.erase(i); } else ++i;
    }
}

void GalleryCache::clear() {
    for (auto& kv : g_cache) fir_gallery_destroy(kv.second);
    g_cache.clear();
}

std::vector<int> recognize_images_bf(const std::vector<ImageInfo>& dbImages, const std::vector<ImageInfo>& tests,
                                     int max_features, std::vector<float>* best_dist) {
    if (max_features == 0) max_features = FEATURES_COUNT;          // db_features.cpp:320-321
    std::vector<int> out(tests.size(), -1);
    if (best_dist) best_dist->assign(tests.size(), FIR_NOT_FOUND_DIST);
    if (tests.empty() || dbImages.empty()) return out;
    // the gallery is uploaded with as many features as the scan reads
    const int dim = max_features;
    fir_gallery* g = GalleryCache::get(dbImages, dim);
    if (!g) return out;
    std::vector<float> q((size_t)tests.size() * dim, 0.0f);
    for (size_t i = 0; i < tests.size(); ++i) {
        const FeaturesVector& f = tests[i].features;
        std::memcpy(&q[i * dim], f.data(), std::min<size_t>(f.size(), (size_t)dim) * sizeof(float));
    }
    std::vector<int32_t> idx(tests.size());
    std::vector<float> dist(tests.size());
    if (fir_search_top1(g, q.data(), (int32_t)tests.size(), 0, max_features, idx.data(), dist.data()) != FIR_OK) {
        log_error("search_top1");
        return out;
    }
    for (size_t i = 0; i < tests.size(); ++i) out[i] = idx[i];
    if (best_dist) *best_dist = dist;
    return out;
}

}  // namespace fir

float feature_distance(const FeaturesVector& lhs, const FeaturesVector& rhs, int start_pos, int end_pos) {
    float d = 0.0f;
    const int len = (int)std::min(lhs.size(), rhs.size());
    if (fir_feature_distance(lhs.data(), rhs.data(), len, start_pos, end_pos, fir::metric(), fir::device(), &d) != FIR_OK) {
        fir::log_error("feature_distance");
        return std::nanf("");
    }
    return d;
}

int recognize_image_bf(const std::vector<ImageInfo>& dbImages, const ImageInfo& testImageInfo, int max_features) {
    std::vector<ImageInfo> one(1, testImageInfo);
    return fir::recognize_images_bf(dbImages, one, max_features)[0];
}

// ---- feature file: "<path>\n<class>\n<f0> <f1> ... \n" per image (dnn_feature_extractor.py:58-64) ----
// db_features.cpp:44-116 through the fast packed loader (fir_loader.cpp: memory-mapped, threaded, exact float
// parsing; rows bit-identical to the reference's -- tests/test_host_loader.py), then scattered into the
// reference's ImagesDatabase shape.
int loadImages(ImagesDatabase& imagesDb, std::string features_file, std::unordered_map<std::string, int>& person2indexMap,
               bool /*early_stop*/) {
    person2indexMap.clear();
    fir::PackedFeatures packed;
    const int64_t total = fir::load_features_packed(features_file, FEATURES_COUNT, fir::metric(), packed);
    if (total <= 0) return 0;                                        // silently empty, db_features.cpp:49,115
    const size_t first_new = imagesDb.size();                        // the reference appends to whatever the caller passed
    imagesDb.resize(first_new + packed.class_names.size());
    for (size_t c = 0; c < packed.class_names.size(); ++c) person2indexMap.emplace(packed.class_names[c], (int)c);
    for (int64_t r = 0; r < packed.n; ++r) {
        const float* src = &packed.rows[(size_t)r * FEATURES_COUNT];
        imagesDb[first_new + (size_t)packed.class_no[(size_t)r]].emplace_back(src, src + FEATURES_COUNT);
    }
    double avg_count = 0;                                            // db_features.cpp:107-112
    for (size_t i = 0; i < imagesDb.size(); ++i) avg_count += (double)imagesDb[i].size();
    avg_count /= (double)imagesDb.size();
    std::cout << "total size=" << imagesDb.size() << " totalImages=" << total << " avg_count=" << avg_count << std::endl;
    return (int)total;
}

// video.cpp:35-96 through fir_loader's exact parser. Entries are (re)built the way the reference's map ends up:
// a repeated person name re-sizes and overwrites the earlier entry's videos.
void loadVideos(MapOfVideos& dbVideos, const std::string& video_features_file) {
    fir::PackedVideos pv;
    if (fir::load_videos_packed(video_features_file, FEATURES_COUNT, fir::metric(), pv) <= 0) return;
    for (size_t p = 0; p < pv.person.size(); ++p) {
        std::vector<std::vector<FeaturesVector> >& videos = dbVideos[pv.person[p]];
        videos.clear();
        for (int32_t v = pv.video_first[p]; v < pv.video_first[p + 1]; ++v) {
            videos.emplace_back();
            for (int64_t r = pv.frame_first[(size_t)v]; r < pv.frame_first[(size_t)v + 1]; ++r)
                videos.back().emplace_back(&pv.rows[(size_t)r * FEATURES_COUNT], &pv.rows[(size_t)(r + 1) * FEATURES_COUNT]);
        }
    }
    std::cout << "total size=" << dbVideos.size() << " totalVideos=" << pv.total_videos << " totalImages=" << pv.total_images << std::endl;   // :92
}

void getTrainingAndTestImages(const ImagesDatabase& totalImages, std::vector<ImageInfo>& dbImages,
                              std::vector<ImageInfo>& testImages, bool randomize) {
    const int kIndices = 400;                                        // db_features.cpp:119
    int order[kIndices];
    for (int i = 0; i < kIndices; ++i) order[i] = i;
    if (randomize) {
        // what std::random_shuffle(first, last) does in libstdc++ (removed from C++17): rand()-driven swaps
        for (int i = 1; i < kIndices; ++i) {
            const int j = std::rand() % (i + 1);
            if (i != j) std::swap(order[i], order[j]);
        }
    }
    dbImages.clear();
    testImages.clear();
    int base = 0;
    for (size_t c = 0; c < totalImages.size(); ++c) {
        const int count = (int)totalImages[c].size();
#ifndef FIR_SPLIT_BY_FRACTION
        int gallery_size = 30;                                       // Caltech rule, :132-133
#else
        float want = count * FRACTION;                               // :135-142
        int gallery_size = (int)std::ceil(want);
        if (gallery_size == count) gallery_size = count - 1;
        if (gallery_size == 0) gallery_size = 1;
#endif
        int taken = 0;
        for (int i = 0; i < kIndices; ++i) {
            if (order[i] >= count) continue;
            ImageInfo view((int)c, base + order[i], totalImages[c][order[i]]);
            (taken < gallery_size ? dbImages : testImages).push_back(view);
            ++taken;
        }
        base += count;
    }
}
