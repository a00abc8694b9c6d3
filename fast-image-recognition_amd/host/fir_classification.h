// fir_classification.h -- host-side mirror of the classifier API of qt_cpp/classification.cpp
// (Feature_vector :35-43, the dataset state :53-62, Classifier :82-105, KNNClassifier :108-170,
// PNNClassifier :173-226, load_image_dataset :795-862, split_train_test :942-990), backed by the
// GPU through fir_cls_* (include/fir_amd.h). Like the reference, this header defines a class
// named `Classifier`; it is the classification.cpp one, so do not include it together with
// fir_classifiers.h (the ImageTesting.cpp one) -- the reference keeps them in separate
// translation units too.
#ifndef FIR_CLASSIFICATION_H
#define FIR_CLASSIFICATION_H
#ifdef FIR_CLASSIFIERS_H
#error "fir_classification.h and fir_classifiers.h both define `Classifier` (as the reference's two .cpp files do): use separate translation units"
#endif

#include <string>
#include <vector>

#include "../../include/fir_amd.h"

#ifndef FEATURES_COUNT
#define FEATURES_COUNT 1536
#endif

typedef double FEATURE_TYPE;                                   // classification.cpp:31

class Feature_vector {                                         // classification.cpp:35-43
public:
    Feature_vector(const std::vector<FEATURE_TYPE>& fv, double out) : features(fv), output(out) {}
    std::vector<FEATURE_TYPE> features;
    double output;
};

namespace fir {
// The reference keeps this state in file-scope globals (classification.cpp:53-62); it is one
// explicit object here with the same member names.
struct ClassificationState {
    size_t num_of_classes = 0, num_of_cont_features = 0, num_of_cont_features_orig = 0;
    std::vector<Feature_vector> dataset, tmp_dataset;
    std::vector<std::vector<size_t> > indices;
    std::vector<std::vector<size_t> > training_set;
    std::vector<size_t> test_set;
    std::vector<FEATURE_TYPE> minValues, maxValues, avgValues, stdValues;
    fir_cls* model = nullptr;          // device copy of the current training set (built by train())
    unsigned long long split_serial = 0, model_serial = ~0ull;
};
ClassificationState& classification_state();
int classification_device();
void set_classification_device(int device);
}  // namespace fir

// classification.cpp:795-862 (reads `features_file`; the reference hard-wires FEATURES_FILE_NAME).
void load_image_dataset(const std::string& features_file, int features_count = FEATURES_COUNT);
#ifdef DB_H
// the reference's call form (classification.cpp:795,992): reads FEATURES_FILE_NAME of its db.h, when that header came first
inline void load_image_dataset() { load_image_dataset(FEATURES_FILE_NAME, FEATURES_COUNT); }
#endif
// classification.cpp:942-990. `shuffle` = false keeps file order inside each class (the reference always shuffles).
void split_train_test(double fraction, bool shuffle = true);

class Classifier {                                             // classification.cpp:82-105
public:
    Classifier(std::string name) : method_name(name) {}
    virtual ~Classifier() {}
    virtual void train() = 0;
    virtual int predict(const Feature_vector& inputFeatures) = 0;
    // batched extension: one training-set pass per 4 queries
    virtual std::vector<int> predict_batch(const std::vector<const Feature_vector*>& inputs);
    std::string get_name() { return method_name; }

protected:
    std::string method_name;
    // uploads the current training_set (class-major) + avgValues if the split changed since the last upload
    static fir_cls* device_model();
};

class KNNClassifier : public Classifier {                      // classification.cpp:108-170
public:
    KNNClassifier(int k);
    void train() override { device_model(); }
    int predict(const Feature_vector& inputFeatures) override;
    std::vector<int> predict_batch(const std::vector<const Feature_vector*>& inputs) override;

private:
    int K;
};

class PNNClassifier : public Classifier {                      // classification.cpp:173-226 (brute-force form)
public:
    PNNClassifier(bool bf = true, std::string name = "PNN");
    void train() override { device_model(); }
    int predict(const Feature_vector& inputFeatures) override;
    std::vector<int> predict_batch(const std::vector<const Feature_vector*>& inputs) override;

private:
    bool bruteforce;
};

// classification.cpp:311-428: k-medoids (100 rounds) inside every class, then the PNN over the medoids only.
// The pairwise within-class distances come from the GPU (fir_cls_distance_sums with a zero mean vector);
// the assign / re-pick bookkeeping of the 100 rounds runs on the host over that n x n table.
class PNNwithClusteringClassifier : public Classifier {
public:
    PNNwithClusteringClassifier(int no_clusters);
    ~PNNwithClusteringClassifier();
    void train() override;
    int predict(const Feature_vector& inputFeatures) override;
    std::vector<int> predict_batch(const std::vector<const Feature_vector*>& inputs) override;
    const std::vector<std::vector<size_t> >& clusters() const { return clustered_training_set; }

private:
    const int num_clusters;
    std::vector<std::vector<size_t> > clustered_training_set;
    fir_cls* medoid_model = nullptr;
};

// classification.cpp:618-791: orthogonal-series (trigonometric) PNN. train() builds the D x C x (2J+1) coefficient
// model on the GPU (fir_fpnn_train); predict() is predict_bf or, with bf = false, predict_sequentional with the
// class-pruning threshold fastlog(output_ratio) per feature seen.
class FPNNClassifier : public PNNClassifier {
public:
    FPNNClassifier(double scale = 1.0, bool bf = true, float output_ratio = 0.9f);
    ~FPNNClassifier();
    void train() override;
    int predict(const Feature_vector& inputFeatures) override;
    std::vector<int> predict_batch(const std::vector<const Feature_vector*>& inputs) override;
    int harmonics() const;                                     // J (classification.cpp:669-676)

private:
    double features_scale;
    bool exhaustive;
    float ratio;
    fir_fpnn* model = nullptr;
};

#endif  // FIR_CLASSIFICATION_H
