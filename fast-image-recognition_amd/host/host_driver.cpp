// host_driver.cpp -- exercises the drop-in C++ surface the way the reference's harnesses do
// (testRecognitionMethod ImageTesting.cpp:439-501, testANN ann.cpp:24-81) and prints the
// per-query answers as JSON for tests/test_gpu_host_shim.py to compare with the oracle.
//   host_driver <features.txt> [max_features ...]
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>

#include "compat/ann.h"
#include "compat/db_features.h"

static void print_vec(const char* key, const std::vector<int>& v, bool comma) {
    std::printf("\"%s\": [", key);
    for (size_t i = 0; i < v.size(); ++i) std::printf("%s%d", i ? ", " : "", v[i]);
    std::printf("]%s\n", comma ? "," : "");
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: host_driver <features.txt> [max_features ...]\n"); return 2; }
    ImagesDatabase total;
    std::unordered_map<std::string, int> person2index;
    std::streambuf* cout_buf = std::cout.rdbuf(nullptr);   // the shim prints what the reference prints (std::cout); stdout carries JSON here
    const int n = loadImages(total, argv[1], person2index);
    std::cout.rdbuf(cout_buf);
    std::vector<ImageInfo> dbImages, testImages;
    getTrainingAndTestImages(total, dbImages, testImages, /*randomize=*/false);
    std::printf("{\n\"images\": %d, \"classes\": %zu, \"gallery\": %zu, \"queries\": %zu,\n", n, total.size(), dbImages.size(), testImages.size());

    std::vector<int> truth, dbidx, dbcls;
    for (const ImageInfo& t : testImages) truth.push_back(t.classNo);
    for (const ImageInfo& g : dbImages) { dbidx.push_back(g.indexInDatabase); dbcls.push_back(g.classNo); }
    print_vec("query_class", truth, true);
    print_vec("gallery_index", dbidx, true);
    print_vec("gallery_class", dbcls, true);

    for (int a = 2; a < argc; ++a) {                       // BruteForceClassifier(max_features), one query at a time
        const int maxf = std::atoi(argv[a]);
        BruteForceClassifier bf(maxf);
        bf.train(&dbImages);
        std::vector<int> one, batch = bf.recognize_batch(testImages);
        for (ImageInfo t : testImages) one.push_back(bf.recognize(t));
        char key[64];
        std::snprintf(key, sizeof key, "bf_%d_single", maxf);
        print_vec(key, one, true);
        std::snprintf(key, sizeof key, "bf_%d_batch", maxf);
        print_vec(key, batch, true);
        std::printf("\"bf_%d_name\": \"%s\",\n", maxf, bf.get_name().c_str());
    }
    {   // the TWD classifiers of testRecognition (ImageTesting.cpp:530-535), batched and one at a time
        const int C = (int)total.size();
        ConventionalTWDClassifier post(C, ConventionalTWDClassifier::TWD_Type::Posteriors, 0.24);
        ConventionalTWDClassifier diff(C, ConventionalTWDClassifier::TWD_Type::DistDiff, 0.003);
        ConventionalTWDClassifier ratio(C, ConventionalTWDClassifier::TWD_Type::DistRatio, 0.7);
        ProposedTWDClassifier p32(C, 32, 0.7), p64(C, 64, 0.7);
        ImageClassifier* twd[5] = {&post, &diff, &ratio, &p32, &p64};   // (north_star's name for ImageTesting.cpp:35-49 `Classifier`)
        const char* keys[5] = {"twd_post", "twd_diff", "twd_ratio", "twd_p32", "twd_p64"};
        for (int i = 0; i < 5; ++i) {
            twd[i]->train(&dbImages);
            fir::num_of_unreliable() = 0;
            std::vector<int> b = twd[i]->recognize_batch(testImages);
            const int unrel = fir::num_of_unreliable();
            std::vector<int> one;
            for (size_t k = 0; k < testImages.size() && k < 6; ++k) { ImageInfo t = testImages[k]; one.push_back(twd[i]->recognize(t)); }
            char key[64];
            std::snprintf(key, sizeof key, "%s_batch", keys[i]);
            print_vec(key, b, true);
            std::snprintf(key, sizeof key, "%s_first6", keys[i]);
            print_vec(key, one, true);
            std::printf("\"%s_unreliable\": %d, \"%s_name\": \"%s\",\n", keys[i], unrel, keys[i], twd[i]->get_name().c_str());
        }
    }
    {   // DirectedEnumeration the way testANN drives it (ann.cpp:62-72): one build, several imageCountToCheck
        std::streambuf* old = std::cout.rdbuf(nullptr);     // the constructor prints like the reference's; keep stdout JSON
        std::srand(13);
        DirectedEnumeration dem(dbImages);
        std::cout.rdbuf(old);
        print_vec("dem_pivots", dem.getStartIndices(), true);
        std::printf("\"dem_threshold\": %.9g,\n", dem.getThresholdValue());
        const int counts[3] = {0, 40, 100};
        for (int c = 0; c < 3; ++c) {
            dem.setImageCountToCheck(counts[c]);
            std::vector<int> batch = dem.recognize_batch(testImages), one, calc, found;
            for (ImageInfo t : testImages) {
                one.push_back(dem.recognize(t));
                calc.push_back(dem.getDistanceCalcCount());
                found.push_back(dem.isFoundLessThreshold ? 1 : 0);
            }
            char key[64];
            std::snprintf(key, sizeof key, "dem_%d_batch", counts[c]);
            print_vec(key, batch, true);
            std::snprintf(key, sizeof key, "dem_%d_single", counts[c]);
            print_vec(key, one, true);
            std::snprintf(key, sizeof key, "dem_%d_calc", counts[c]);
            print_vec(key, calc, true);
            std::snprintf(key, sizeof key, "dem_%d_found", counts[c]);
            print_vec(key, found, true);
        }
    }
    BruteForce ann(dbImages);                               // ann.h BruteForce -> gallery rows
    std::vector<int> rows = ann.recognize_batch(testImages);
    print_vec("ann_rows", rows, true);
    std::vector<int> raw;
    for (const ImageInfo& t : testImages) raw.push_back(recognize_image_bf(dbImages, t));
    print_vec("recognize_image_bf", raw, true);
    {   // several GPUs (here: logical shards on device 0): the same answers through fir_sharded_* and its RCCL exchange
        fir::set_devices(std::vector<int>(1, 0), 3);
        BruteForce ann_sh(dbImages);
        print_vec("sharded_ann_rows", ann_sh.recognize_batch(testImages), true);
        BruteForceClassifier bf_sh(256);
        bf_sh.train(&dbImages);
        print_vec("sharded_bf_256_batch", bf_sh.recognize_batch(testImages), true);
        std::vector<int> one;
        for (size_t k = 0; k < testImages.size() && k < 4; ++k) one.push_back(recognize_image_bf(dbImages, testImages[k]));
        print_vec("sharded_first4", one, true);
        fir::set_devices(std::vector<int>());
    }
    if (!testImages.empty() && dbImages.size() > 2) {
        // a gallery row edited IN PLACE (same vector object, same row buffers) between two calls of the free function: the
        // reference reads the rows afresh every time; the upload cache must notice (fir_db.h, GalleryCache)
        const size_t mid = dbImages.size() / 2;
        FeaturesVector& row = const_cast<FeaturesVector&>(dbImages[mid].features);
        const FeaturesVector saved = row;
        const int before = recognize_image_bf(dbImages, testImages[0]);
        fir::set_cache_validation(fir::FIR_CACHE_VALIDATE_ALWAYS);
        row = testImages[0].features;                      // same length: copied into the same buffer
        const int after_always = recognize_image_bf(dbImages, testImages[0]);
        row = saved;
        const int restored = recognize_image_bf(dbImages, testImages[0]);
        fir::set_cache_validation(fir::FIR_CACHE_VALIDATE_THROTTLED);
        row = testImages[0].features;
        struct timespec ts = {0, 40 * 1000 * 1000};
        nanosleep(&ts, nullptr);                           // past the 20 ms window of the default policy
        const int after_default = recognize_image_bf(dbImages, testImages[0]);
        row = saved;
        fir::invalidate(dbImages);
        std::printf("\"edit_row\": %zu, \"edit_before\": %d, \"edit_after_always\": %d, \"edit_restored\": %d, \"edit_after_default\": %d,\n", mid, before,
                    after_always, restored, after_default);
    }
    const float d01 = testImages.empty() ? 0.f : testImages[0].distance(dbImages[0]);
    const float d64 = testImages.empty() ? 0.f : testImages[0].distance(dbImages[0], 0, 64);
    std::printf("\"dist_q0_g0\": %.9g, \"dist_q0_g0_64\": %.9g\n}\n", d01, d64);
    fir::GalleryCache::clear();
    return 0;
}
