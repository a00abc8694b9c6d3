#include "fir_classifiers.h"

#include <algorithm>

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
