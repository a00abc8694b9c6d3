#include "fir_classifiers.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <limits>

float ClassificationMethod::getThreshold(std::vector<float>& otherClassesDists, float falseAcceptRate) {
    const int ind = (int)(otherClassesDists.size() * falseAcceptRate);
    std::nth_element(otherClassesDists.begin(), otherClassesDists.begin() + ind, otherClassesDists.end());
    const float threshold = otherClassesDists[ind];
    std::cout << threshold << " " << *std::min_element(otherClassesDists.begin(), otherClassesDists.end()) << " "
              << *std::max_element(otherClassesDists.begin(), otherClassesDists.end()) << std::endl;
    return threshold;
}

#include <cstring>

namespace fir {
int& num_of_unreliable() {
    static int counter = 0;
    return counter;
}
fir_gallery* twd_gallery(const std::vector<ImageInfo>& dbImages) { return GalleryCache::get(dbImages, 256); }

namespace {
std::vector<float> pack_queries(const std::vector<ImageInfo>& tests, int dim) {
    std::vector<float> q((size_t)tests.size() * dim, 0.0f);
    for (size_t i = 0; i < tests.size(); ++i) {
        const FeaturesVector& f = tests[i].features;
        std::memcpy(&q[i * dim], f.data(), std::min<size_t>(f.size(), (size_t)dim) * sizeof(float));
    }
    return q;
}
}  // namespace
}  // namespace fir

std::vector<int> ConventionalTWDClassifier::recognize_batch(const std::vector<ImageInfo>& tests) {
    std::vector<int> out(tests.size(), -1);
    if (tests.empty() || !pDbImages || pDbImages->empty()) return out;
    fir_gallery* g = fir::twd_gallery(*pDbImages);
    if (!g) return out;
    std::vector<float> q = fir::pack_queries(tests, 256);
    std::vector<int32_t> cls(tests.size()), unrel(tests.size());
    if (fir_twd_conventional(g, q.data(), (int32_t)tests.size(), num_of_classes, (int)type, threshold, reduced_features_count,
                             cls.data(), unrel.data()) != FIR_OK) {
        fir::log_error("twd_conventional");
        return out;
    }
    for (size_t i = 0; i < tests.size(); ++i) { out[i] = cls[i]; fir::num_of_unreliable() += unrel[i]; }
    return out;
}

std::vector<int> ProposedTWDClassifier::recognize_batch(const std::vector<ImageInfo>& tests) {
    std::vector<int> out(tests.size(), -1);
    if (tests.empty() || !pDbImages || pDbImages->empty()) return out;
    fir_gallery* g = fir::twd_gallery(*pDbImages);
    if (!g) return out;
    std::vector<float> q = fir::pack_queries(tests, 256);
    std::vector<int32_t> cls(tests.size()), unrel(tests.size());
    if (fir_twd_proposed(g, q.data(), (int32_t)tests.size(), reduced_features_count, th_, cls.data(), unrel.data(), nullptr) != FIR_OK) {
        fir::log_error("twd_proposed");
        return out;
    }
    for (size_t i = 0; i < tests.size(); ++i) { out[i] = cls[i]; fir::num_of_unreliable() += unrel[i]; }
    return out;
}

// ---- DirectedEnumeration (ann.cpp:270-507, PIVOT build) ----
DirectedEnumeration::DirectedEnumeration(std::vector<ImageInfo>& faceImages, float falseAcceptRate, float threshold_, int imageCountToCheck_)
    : ClassificationMethod("dem", faceImages), isFoundLessThreshold(false), bestDistance(0), threshold(0), gallery(nullptr), dem(nullptr) {
    auto t1 = std::chrono::high_resolution_clock::now();
    setImageCountToCheck(imageCountToCheck_);                          // init(), ann.cpp:363
    const int dbSize = (int)dbImages.size();
    if (threshold_ > 0) threshold = threshold_;                        // ann.cpp:275-277
    if (dbSize == 0) return;
    // ann.cpp:365-376: the first pivot is the head of a random_shuffle of 0..n-1 (same generator: std::rand);
    // the rest of that shuffle is overwritten by the greedy choice below.
    std::vector<int> indices((size_t)dbSize);
    for (int i = 0; i < dbSize; ++i) indices[(size_t)i] = i;
    std::random_shuffle(indices.begin(), indices.end());
    int N = (int)(dbSize * 0.015);
    if (N < 5) N = 5;
    std::cout << N;
    N = std::min(N, dbSize);                                           // the reference reads past `indices` here

    const int dim = FEATURES_COUNT;                                    // ann.h:33-38: always the full range
    std::vector<float> rows((size_t)dbSize * dim, 0.0f);
    std::vector<int32_t> cls((size_t)dbSize);
    for (int j = 0; j < dbSize; ++j) {
        const FeaturesVector& f = dbImages[(size_t)j].features;
        std::memcpy(&rows[(size_t)j * dim], f.data(), std::min<size_t>(f.size(), (size_t)dim) * sizeof(float));
        cls[(size_t)j] = dbImages[(size_t)j].classNo;
    }
    if (fir_gallery_create(rows.data(), dbSize, dim, cls.data(), fir::metric(), fir::device(), &gallery) != FIR_OK ||
        fir_dem_create(gallery, indices[0], N, &dem) != FIR_OK) {
        fir::log_error("DirectedEnumeration");
        return;
    }
    int32_t n_built = 0, n_used = 0;
    fir_dem_info(dem, nullptr, &n_built, &n_used, nullptr);
    std::vector<int32_t> piv((size_t)N);
    std::vector<float> otherClassesDists((size_t)N);
    order0.resize((size_t)dbSize);
    fir_dem_get(dem, piv.data(), otherClassesDists.data(), nullptr, order0.data());
    startIndices.assign(piv.begin(), piv.begin() + n_used);            // ann.cpp:333-334
    otherClassesDists.resize((size_t)n_built);
    if (threshold_ <= 0 && !otherClassesDists.empty()) threshold = getThreshold(otherClassesDists, falseAcceptRate);   // ann.cpp:341-343
    auto t2 = std::chrono::high_resolution_clock::now();
    std::cout << "init took " << std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count() << " milliseconds" << std::endl;
}

DirectedEnumeration::~DirectedEnumeration() {
    if (dem) fir_dem_destroy(dem);
    if (gallery) fir_gallery_destroy(gallery);
}

namespace {
struct LikelihoodsComparator {                                         // ann.cpp:402-414
    const float* likelihoods;
    bool operator()(int lhsIndex, int rhsIndex) const { return likelihoods[lhsIndex] < likelihoods[rhsIndex]; }
};
}  // namespace

// ann.cpp:416-507 for one query whose pivot distances and likelihoods the device already produced.
int DirectedEnumeration::finish_walk(const float* query, const float* pivot_dist, const float* likelihoods) {
    int bestIndex = -1;
    isFoundLessThreshold = false;
    bestDistance = std::numeric_limits<float>::max();
    distanceCalcCount = 0;
    const int dbSize = (int)dbImages.size();
    int start_index = 0;
    for (size_t i = 0; i < startIndices.size(); ++i) {                 // CHECK_FOR_BEST_DIST over the pivots (:427-430)
        const float tmpDist = pivot_dist[i];
        ++distanceCalcCount;
        ++start_index;
        if (tmpDist < bestDistance) {
            bestDistance = tmpDist;
            bestIndex = startIndices[i];
            if (bestDistance < threshold) { isFoundLessThreshold = true; goto end; }
        }
    }
    if (imageCountToCheck > start_index) {
        likelihood_indices = order0;
        LikelihoodsComparator cmp{likelihoods};
        std::partial_sort(likelihood_indices.begin() + start_index, likelihood_indices.begin() + imageCountToCheck, likelihood_indices.end(), cmp);   // :455-456
        const int m = imageCountToCheck - start_index;                 // the loop of :458-462 checks exactly these, in order
        std::vector<float> dist((size_t)m);
        if (fir_rows_distances(gallery, query, 1, likelihood_indices.data() + start_index, m, 0, FEATURES_COUNT, dist.data()) != FIR_OK) {
            fir::log_error("DirectedEnumeration::recognize");
            return -1;
        }
        for (int k = 0; k < m && distanceCalcCount < imageCountToCheck; ++k) {
            const float tmpDist = dist[(size_t)k];
            ++distanceCalcCount;
            if (tmpDist < bestDistance) {
                bestDistance = tmpDist;
                bestIndex = likelihood_indices[(size_t)(start_index + k)];
                if (bestDistance < threshold) { isFoundLessThreshold = true; goto end; }
            }
        }
    }
end:
    avgCheckedPercent += 100. * distanceCalcCount / dbSize;
    return bestIndex;
}

std::vector<int> DirectedEnumeration::recognize_batch(const std::vector<ImageInfo>& tests) {
    std::vector<int> out(tests.size(), -1);
    if (!dem || tests.empty()) return out;
    const int dim = FEATURES_COUNT;
    const size_t n = dbImages.size(), used = startIndices.size();
    const size_t kChunk = 8;                                           // one pass over the table serves 8 queries
    std::vector<float> q(kChunk * dim), pd(kChunk * used), lik(kChunk * n);
    for (size_t i0 = 0; i0 < tests.size(); i0 += kChunk) {
        const size_t nq = std::min(kChunk, tests.size() - i0);
        std::fill(q.begin(), q.end(), 0.0f);
        for (size_t i = 0; i < nq; ++i) {
            const FeaturesVector& f = tests[i0 + i].features;
            std::memcpy(&q[i * dim], f.data(), std::min<size_t>(f.size(), (size_t)dim) * sizeof(float));
        }
        if (fir_dem_likelihoods(dem, q.data(), (int32_t)nq, pd.data(), lik.data()) != FIR_OK) {
            fir::log_error("DirectedEnumeration::recognize");
            return out;
        }
        for (size_t i = 0; i < nq; ++i) out[i0 + i] = finish_walk(&q[i * dim], &pd[i * used], &lik[i * n]);
    }
    return out;
}

int DirectedEnumeration::recognize(ImageInfo& testImage) {
    std::vector<ImageInfo> one(1, testImage);
    return recognize_batch(one)[0];
}
