// fir_classification.cpp -- see fir_classification.h. Dataset IO / split bookkeeping on the host,
// every distance, exp and vote on the GPU (fir_cls_*).
#include "fir_classification.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

namespace fir {
namespace {
ClassificationState g_state;
int g_cls_device = 0;
}  // namespace
ClassificationState& classification_state() { return g_state; }
int classification_device() { return g_cls_device; }
void set_classification_device(int device) { g_cls_device = device; }
}  // namespace fir

using fir::classification_state;

void load_image_dataset(const std::string& features_file, int features_count) {
    fir::ClassificationState& st = classification_state();
    st.dataset.clear(); st.tmp_dataset.clear(); st.indices.clear(); st.training_set.clear(); st.test_set.clear();
    st.num_of_cont_features = (size_t)features_count;
    st.num_of_classes = 0;
    std::cout << features_file << std::endl;                               // classification.cpp:798-799
    std::ifstream in(features_file);
    if (!in) return;
    std::map<std::string, int> class_id;                                   // classification.cpp:803
    std::string path_line, class_line, feat_line;
    while (std::getline(in, path_line) && std::getline(in, class_line) && std::getline(in, feat_line)) {
        class_line.erase(0, class_line.find_first_not_of(" \t\n\r\f\v"));
        if (class_line.find("BACKGROUND_Google") != std::string::npos || class_line.find("257.clutter") != std::string::npos)
            continue;                                                      // :813-817
        auto it = class_id.find(class_line);
        if (it == class_id.end()) it = class_id.emplace(class_line, (int)class_id.size()).first;
        std::vector<FEATURE_TYPE> f((size_t)features_count);
        const char* p = feat_line.c_str();
        FEATURE_TYPE norm = 0;
        double v = 0;
        bool failed = false;
        for (int i = 0; i < features_count; ++i) {
            // `iss >> feature` (:832): nothing is extracted once the line is exhausted (the value read then is the previous
            // one in practice -- the reference's variable is uninitialised); a malformed field stores 0 and fails the stream
            if (!failed) {
                const char* q = p;
                while (*q && std::strchr(" \t\n\r\f\v", *q)) ++q;
                if (!*q) failed = true;
                else {
                    const char* fe = q;
                    if (*fe == '+' || *fe == '-') ++fe;
                    bool digits = false, point = false;
                    for (;; ++fe) {
                        if (*fe >= '0' && *fe <= '9') digits = true;
                        else if (*fe == '.' && !point) point = true;
                        else break;
                    }
                    if (digits && (*fe == 'e' || *fe == 'E')) {
                        const char* x = fe + 1;
                        if (*x == '+' || *x == '-') ++x;
                        const char* xd = x;
                        while (*x >= '0' && *x <= '9') ++x;
                        if (x == xd) digits = false; else fe = x;
                    }
                    if (!digits) { failed = true; v = 0; }
                    else { v = std::strtod(std::string(q, fe).c_str(), nullptr); p = fe; }
                }
            }
            norm += v * v;                                                 // :833
            f[(size_t)i] = v;
        }
        norm = std::sqrt(norm);                                            // :842
        for (double& v : f) v /= norm;                                     // :845-847
        st.dataset.push_back(Feature_vector(f, it->second));
    }
    st.num_of_classes = class_id.size();
    st.indices.assign(st.num_of_classes, std::vector<size_t>());
    for (size_t i = 0; i < st.dataset.size(); ++i) st.indices[(size_t)st.dataset[i].output].push_back(i);
    st.num_of_cont_features_orig = st.num_of_cont_features;               // classification.cpp:995
    std::cout << st.num_of_classes << ' ' << st.num_of_cont_features << ' ' << st.dataset.size() << std::endl;   // :859-860
}

void split_train_test(double fraction, bool shuffle) {
    fir::ClassificationState& st = classification_state();
    const size_t C = st.num_of_classes, D = st.num_of_cont_features_orig;
    st.training_set.assign(C, std::vector<size_t>());
    st.test_set.clear();
    for (size_t c = 0; c < C; ++c) {
        std::vector<size_t>& idx = st.indices[c];
        if (shuffle)                                                       // std::random_shuffle, :952
            for (size_t i = 1; i < idx.size(); ++i) { const size_t j = (size_t)std::rand() % (i + 1); if (i != j) std::swap(idx[i], idx[j]); }
        int end = fraction >= 1 ? (int)fraction : (int)std::ceil(fraction * idx.size());   // :953
        if (end == 0 && !idx.empty()) end = 1;
        else if (end >= (int)idx.size()) end = (int)idx.size();
        st.training_set[c].assign(idx.begin(), idx.begin() + end);
        st.test_set.insert(st.test_set.end(), idx.begin() + end, idx.end());
    }
    st.tmp_dataset = st.dataset;
    st.num_of_cont_features = D;
    st.minValues.assign(D, 0); st.maxValues.assign(D, 0); st.avgValues.assign(D, 0); st.stdValues.assign(D, 0);
    for (size_t f = 0; f < D; ++f) {                                       // :969-989
        double mn = FLT_MAX, mx = -FLT_MAX, sum = 0, sq = 0;
        int count = 0;
        for (size_t c = 0; c < C; ++c)
            for (size_t t : st.training_set[c]) {
                const double v = st.dataset[t].features[f];
                ++count;
                if (v < mn) mn = v;
                if (mx < v) mx = v;
                sum += v;
                sq += v * v;
            }
        st.minValues[f] = mn; st.maxValues[f] = mx;
        st.avgValues[f] = sum / count;
        st.stdValues[f] = std::sqrt((sq - st.avgValues[f] * st.avgValues[f] * count) / (count - 1));
    }
    ++st.split_serial;
}

fir_cls* Classifier::device_model() {
    fir::ClassificationState& st = classification_state();
    if (st.model && st.model_serial == st.split_serial) return st.model;
    if (st.model) { fir_cls_destroy(st.model); st.model = nullptr; }
    const size_t D = st.num_of_cont_features;
    std::vector<double> rows;
    std::vector<int32_t> cls;
    for (size_t c = 0; c < st.num_of_classes; ++c)
        for (size_t t : st.training_set[c]) {                              // the reference's scan order, :121-122
            const std::vector<double>& f = st.tmp_dataset[t].features;
            rows.insert(rows.end(), f.begin(), f.begin() + D);
            cls.push_back((int32_t)c);
        }
    if (fir_cls_create(rows.data(), (int64_t)cls.size(), (int32_t)D, cls.data(), (int32_t)st.num_of_classes, st.avgValues.data(),
                       fir::classification_device(), &st.model) != FIR_OK) {
        std::fprintf(stderr, "fir: training-set upload: %s\n", fir_last_error());
        st.model = nullptr;
    }
    st.model_serial = st.split_serial;
    return st.model;
}

std::vector<int> Classifier::predict_batch(const std::vector<const Feature_vector*>& inputs) {
    std::vector<int> out;
    for (const Feature_vector* fv : inputs) out.push_back(predict(*fv));
    return out;
}

namespace {
std::vector<double> pack(const std::vector<const Feature_vector*>& inputs, size_t D) {
    std::vector<double> q(inputs.size() * D);
    for (size_t i = 0; i < inputs.size(); ++i)
        for (size_t f = 0; f < D; ++f) q[i * D + f] = inputs[i]->features[f];
    return q;
}
template <typename T>
std::string named(const std::string& prefix, T param) {                    // classification.cpp:97-102
    std::ostringstream os;
    os << prefix << ", " << param;
    return os.str();
}
}  // namespace

KNNClassifier::KNNClassifier(int k) : Classifier(named("k-NN", k)), K(k) {}

std::vector<int> KNNClassifier::predict_batch(const std::vector<const Feature_vector*>& inputs) {
    std::vector<int> out(inputs.size(), -1);
    fir_cls* m = device_model();
    if (!m || inputs.empty()) return out;
    std::vector<double> q = pack(inputs, classification_state().num_of_cont_features);
    std::vector<int32_t> best(inputs.size());
    if (fir_cls_knn_predict(m, q.data(), (int32_t)inputs.size(), K, best.data()) != FIR_OK) {
        std::fprintf(stderr, "fir: knn_predict: %s\n", fir_last_error());
        return out;
    }
    out.assign(best.begin(), best.end());
    return out;
}
int KNNClassifier::predict(const Feature_vector& inputFeatures) {
    return predict_batch(std::vector<const Feature_vector*>(1, &inputFeatures))[0];
}

PNNClassifier::PNNClassifier(bool bf, std::string name) : Classifier(name + (bf ? "" : " (seq)")), bruteforce(bf) {}

std::vector<int> PNNClassifier::predict_batch(const std::vector<const Feature_vector*>& inputs) {
    std::vector<int> out(inputs.size(), -1);
    fir_cls* m = device_model();
    if (!m || inputs.empty()) return out;
    std::vector<double> q = pack(inputs, classification_state().num_of_cont_features);
    std::vector<int32_t> best(inputs.size());
    const int rc = bruteforce ? fir_cls_pnn_predict(m, q.data(), (int32_t)inputs.size(), /*reference var*/ 0.0, nullptr, best.data())
                              : fir_cls_pnn_predict_seq(m, q.data(), (int32_t)inputs.size(), 0.0, best.data(), nullptr);   // :297-307
    if (rc != FIR_OK) {
        std::fprintf(stderr, "fir: pnn_predict: %s\n", fir_last_error());
        return out;
    }
    out.assign(best.begin(), best.end());
    return out;
}
int PNNClassifier::predict(const Feature_vector& inputFeatures) {
    return predict_batch(std::vector<const Feature_vector*>(1, &inputFeatures))[0];
}

PNNwithClusteringClassifier::PNNwithClusteringClassifier(int no_clusters)
    : Classifier(named("PNN with clustering", no_clusters)), num_clusters(no_clusters) {}
PNNwithClusteringClassifier::~PNNwithClusteringClassifier() { if (medoid_model) fir_cls_destroy(medoid_model); }

void PNNwithClusteringClassifier::train() {
    fir::ClassificationState& st = classification_state();
    const size_t D = st.num_of_cont_features;
    clustered_training_set.assign(st.num_of_classes, std::vector<size_t>());
    const std::vector<double> zero_mean(D, 0.0);
    for (size_t i = 0; i < st.num_of_classes; ++i) {
        const std::vector<size_t>& members = st.training_set[i];
        const size_t n = members.size();
        if (n <= (size_t)num_clusters) { clustered_training_set[i] = members; continue; }   // :381-385
        // n x n table of mean squared distances between the class's raw rows (:339-343, 360-365), from the GPU
        std::vector<double> rows(n * D);
        for (size_t t = 0; t < n; ++t)
            for (size_t f = 0; f < D; ++f) rows[t * D + f] = st.dataset[members[t]].features[f];
        std::vector<int32_t> one_class(n, 0);
        std::vector<double> table(n * n);
        fir_cls* pair = nullptr;
        if (fir_cls_create(rows.data(), (int64_t)n, (int32_t)D, one_class.data(), 1, zero_mean.data(), fir::classification_device(), &pair) != FIR_OK ||
            fir_cls_distance_sums(pair, rows.data(), (int32_t)n, table.data()) != FIR_OK) {
            std::fprintf(stderr, "fir: clustering distances: %s\n", fir_last_error());
            if (pair) fir_cls_destroy(pair);
            clustered_training_set[i] = members;
            continue;
        }
        fir_cls_destroy(pair);
        for (double& v : table) v /= (double)D;                                           // dist /= num_of_cont_features
        std::vector<long> centroid((size_t)num_clusters), assign(n);
        for (int c = 0; c < num_clusters; ++c) centroid[(size_t)c] = c;
        for (int step = 0; step < 100; ++step) {
            for (size_t t = 0; t < n; ++t) {                                              // nearest medoid, first on ties (:331-350)
                assign[t] = -1;
                double best = DBL_MAX;
                for (int c = 0; c < num_clusters; ++c) {
                    if (centroid[(size_t)c] < 0) continue;
                    const double dist = table[(size_t)centroid[(size_t)c] * n + t];
                    if (dist < best) { best = dist; assign[t] = c; }
                }
            }
            for (int c = 0; c < num_clusters; ++c) {                                      // member with the smallest summed distance (:351-375)
                double best = DBL_MAX;
                centroid[(size_t)c] = -1;
                for (size_t t = 0; t < n; ++t) {
                    if (assign[t] != c) continue;
                    double sum = 0;
                    for (size_t t1 = 0; t1 < n; ++t1)
                        if (assign[t1] == c) sum += table[t * n + t1];
                    if (sum < best) { best = sum; centroid[(size_t)c] = (long)t; }
                }
            }
        }
        for (int c = 0; c < num_clusters; ++c)
            if (centroid[(size_t)c] >= 0) clustered_training_set[i].push_back(members[(size_t)centroid[(size_t)c]]);
    }
    // device model over the medoids; the PNN denominator stays the full training size (:390,393)
    if (medoid_model) { fir_cls_destroy(medoid_model); medoid_model = nullptr; }
    std::vector<double> rows;
    std::vector<int32_t> cls;
    size_t total = 0;
    for (size_t c = 0; c < st.num_of_classes; ++c) {
        total += st.training_set[c].size();
        for (size_t t : clustered_training_set[c]) {
            const std::vector<double>& f = st.tmp_dataset[t].features;
            rows.insert(rows.end(), f.begin(), f.begin() + D);
            cls.push_back((int32_t)c);
        }
    }
    if (fir_cls_create(rows.data(), (int64_t)cls.size(), (int32_t)D, cls.data(), (int32_t)st.num_of_classes, st.avgValues.data(),
                       fir::classification_device(), &medoid_model) != FIR_OK ||
        fir_cls_set_total_training_size(medoid_model, (int64_t)total) != FIR_OK) {
        std::fprintf(stderr, "fir: medoid upload: %s\n", fir_last_error());
        if (medoid_model) { fir_cls_destroy(medoid_model); medoid_model = nullptr; }
    }
}

std::vector<int> PNNwithClusteringClassifier::predict_batch(const std::vector<const Feature_vector*>& inputs) {
    std::vector<int> out(inputs.size(), -1);
    if (!medoid_model || inputs.empty()) return out;
    std::vector<double> q = pack(inputs, classification_state().num_of_cont_features);
    std::vector<int32_t> best(inputs.size());
    if (fir_cls_pnn_predict(medoid_model, q.data(), (int32_t)inputs.size(), 0.0, nullptr, best.data()) != FIR_OK) {
        std::fprintf(stderr, "fir: pnn_predict: %s\n", fir_last_error());
        return out;
    }
    out.assign(best.begin(), best.end());
    return out;
}
int PNNwithClusteringClassifier::predict(const Feature_vector& inputFeatures) {
    return predict_batch(std::vector<const Feature_vector*>(1, &inputFeatures))[0];
}

// ---- FPNNClassifier (classification.cpp:618-791) ----
FPNNClassifier::FPNNClassifier(double scale, bool bf, float output_ratio)
    : PNNClassifier(bf, named("FPNN", scale)), features_scale(scale), exhaustive(bf), ratio(output_ratio) {}
FPNNClassifier::~FPNNClassifier() {
    if (model) fir_fpnn_destroy(model);
}
int FPNNClassifier::harmonics() const {
    int32_t J = 0;
    if (model) fir_fpnn_info(model, &J, nullptr, nullptr);
    return J;
}
void FPNNClassifier::train() {
    fir::ClassificationState& st = classification_state();
    if (model) { fir_fpnn_destroy(model); model = nullptr; }
    const size_t D = st.num_of_cont_features;
    std::vector<double> rows;
    std::vector<int32_t> cls;
    for (size_t c = 0; c < st.num_of_classes; ++c)
        for (size_t t : st.training_set[c]) {                              // :680: tmp_dataset[training_set[i][t]]
            const std::vector<double>& f = st.tmp_dataset[t].features;
            rows.insert(rows.end(), f.begin(), f.begin() + D);
            cls.push_back((int32_t)c);
        }
    if (fir_fpnn_train(rows.data(), (int64_t)cls.size(), (int32_t)D, cls.data(), (int32_t)st.num_of_classes, st.avgValues.data(),
                       st.stdValues.data(), features_scale, fir::classification_device(), &model) != FIR_OK) {
        std::fprintf(stderr, "fir: fpnn_train: %s\n", fir_last_error());
        model = nullptr;
    }
}
std::vector<int> FPNNClassifier::predict_batch(const std::vector<const Feature_vector*>& inputs) {
    std::vector<int> out(inputs.size(), -1);
    if (!model || inputs.empty()) return out;
    std::vector<double> q = pack(inputs, classification_state().num_of_cont_features);
    std::vector<int32_t> best(inputs.size());
    const int rc = exhaustive ? fir_fpnn_predict(model, q.data(), (int32_t)inputs.size(), best.data(), nullptr)
                              : fir_fpnn_predict_seq(model, q.data(), (int32_t)inputs.size(), ratio, best.data(), nullptr);
    if (rc != FIR_OK) {
        std::fprintf(stderr, "fir: fpnn_predict: %s\n", fir_last_error());
        return out;
    }
    out.assign(best.begin(), best.end());
    return out;
}
int FPNNClassifier::predict(const Feature_vector& inputFeatures) {
    return predict_batch(std::vector<const Feature_vector*>(1, &inputFeatures))[0];
}
