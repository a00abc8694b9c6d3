// fir_classifiers.h -- the reference's two classifier interfaces on top of the GPU matcher:
//   Classifier / BruteForceClassifier              qt_cpp/ImageTesting.cpp:35-71   (-> class id)
//   ClassificationMethod / BruteForce              qt_cpp/ann.h:9-47, ann.cpp:84-126 (-> gallery row)
// Same constructors, virtuals, names and -1 conventions, so the reference's drivers
// (testRecognitionMethod ImageTesting.cpp:439, testSetRecognition ann.cpp:94) compile against
// them unchanged. Added: batched recognize_batch() -- one gallery pass per 8 test images.
#ifndef FIR_CLASSIFIERS_H
#define FIR_CLASSIFIERS_H

#include <chrono>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "fir_db.h"

// ---- ImageTesting.cpp:35-49 ----
class Classifier {
public:
    Classifier(std::string n) : name(n), pDbImages(nullptr) {}
    virtual ~Classifier() {}
    // The reference only stores the pointer (ImageTesting.cpp:40); the upload happens lazily on
    // the first recognize and is redone when a new split arrives through the same vector.
    virtual void train(std::vector<ImageInfo>* pDb) { pDbImages = pDb; if (pDb) fir::invalidate(*pDb); }
    virtual int recognize(ImageInfo& testImageInfo) = 0;
    // Batched extension: class id (or -1) for every test image.
    virtual std::vector<int> recognize_batch(const std::vector<ImageInfo>& tests) {
        std::vector<int> out;
        out.reserve(tests.size());
        for (ImageInfo t : tests) out.push_back(recognize(t));
        return out;
    }
    std::string get_name() { return name; }

private:
    std::string name;

protected:
    std::vector<ImageInfo>* pDbImages;
    static std::string build_name(std::string prefix, int param) {   // ImageTesting.cpp:51-55
        std::ostringstream os;
        os << prefix << ", " << param;
        return os.str();
    }
};

// ---- ImageTesting.cpp:58-71 ----
class BruteForceClassifier : public Classifier {
public:
    BruteForceClassifier(int max_feats = FEATURES_COUNT) : Classifier(Classifier::build_name("BF", max_feats)), max_features(max_feats) {}
    int recognize(ImageInfo& testImageInfo) override {
        const int best = recognize_image_bf(*pDbImages, testImageInfo, max_features);
        return best == -1 ? -1 : (*pDbImages)[best].classNo;
    }
    std::vector<int> recognize_batch(const std::vector<ImageInfo>& tests) override {
        std::vector<int> rows = fir::recognize_images_bf(*pDbImages, tests, max_features);
        for (int& r : rows) r = (r == -1) ? -1 : (*pDbImages)[r].classNo;
        return rows;
    }

private:
    int max_features;
};

// ---- ImageTesting.cpp:33: how many queries needed the second stage / a second chunk ----
namespace fir {
int& num_of_unreliable();
fir_gallery* twd_gallery(const std::vector<ImageInfo>& dbImages);   // the 256-feature upload the TWD classifiers scan
}  // namespace fir

// ---- ImageTesting.cpp:74-186 ----
class ConventionalTWDClassifier : public Classifier {
public:
    enum class TWD_Type { Posteriors, DistDiff, DistRatio };
    ConventionalTWDClassifier(int cls_num, TWD_Type t, double th, int feat_count = 64)
        : Classifier(build_name(t, th)), num_of_classes(cls_num), reduced_features_count(feat_count), threshold(th), type(t) {}
    int recognize(ImageInfo& testImageInfo) override {
        return recognize_batch(std::vector<ImageInfo>(1, testImageInfo))[0];
    }
    std::vector<int> recognize_batch(const std::vector<ImageInfo>& tests) override;

private:
    int num_of_classes;
    int reduced_features_count;
    double threshold;
    TWD_Type type;
    static std::string build_name(TWD_Type type, double threshold) {   // ImageTesting.cpp:91-106
        const char* prefix = type == TWD_Type::Posteriors ? "TWD posteriors" : type == TWD_Type::DistDiff ? "TWD diff" : "TWD ratio";
        std::ostringstream os;
        os << prefix << ", " << threshold;
        return os.str();
    }
};

// ---- ImageTesting.cpp:188-288 ----
class ProposedTWDClassifier : public Classifier {
public:
    ProposedTWDClassifier(int cls_num, int feat_count, double th)
        : Classifier(build_name(feat_count, th)), num_of_classes(cls_num), reduced_features_count(feat_count), th_(th) {}
    int recognize(ImageInfo& testImageInfo) override {
        return recognize_batch(std::vector<ImageInfo>(1, testImageInfo))[0];
    }
    std::vector<int> recognize_batch(const std::vector<ImageInfo>& tests) override;

private:
    int num_of_classes, reduced_features_count;
    double th_;
    static std::string build_name(int feat_count, double threshold) {  // ImageTesting.cpp:199-203
        std::ostringstream os;
        os << "Proposed TWD, " << feat_count << ", " << threshold;
        return os.str();
    }
};

// ---- ann.h:9-39 ----
class ClassificationMethod {
public:
    ClassificationMethod(std::string name, std::vector<ImageInfo>& db) : method_name(name), dbImages(db), distanceCalcCount(0), avgCheckedPercent(0) {
        imageCountToCheck = (int)dbImages.size();
    }
    virtual ~ClassificationMethod() {}
    virtual int recognize(ImageInfo& testImageInfo) = 0;
    virtual std::vector<int> recognize_batch(const std::vector<ImageInfo>& tests) {
        std::vector<int> out;
        out.reserve(tests.size());
        for (ImageInfo t : tests) out.push_back(recognize(t));
        return out;
    }
    // ann.cpp:94-109: error rate and ms per query over a test set (batched underneath).
    void testSetRecognition(std::vector<ImageInfo>& testImages) {
        int errorsCount = 0;
        avgCheckedPercent = 0;
        auto t1 = std::chrono::high_resolution_clock::now();
        std::vector<int> best = recognize_batch(testImages);
        for (size_t i = 0; i < testImages.size(); ++i)
            if (best[i] == -1 || testImages[i].classNo != dbImages[best[i]].classNo) ++errorsCount;
        auto t2 = std::chrono::high_resolution_clock::now();
        double total_time = (double)std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count();
        double errorRate = 100. * errorsCount / testImages.size();
        std::cout << method_name.c_str() << " error=" << errorRate << "% total_time (ms)" << total_time / testImages.size()
                  << " checkedPercent=" << (avgCheckedPercent > 0 ? avgCheckedPercent / testImages.size() : -1) << std::endl;
    }
    virtual void setImageCountToCheck(int count) {                    // ann.h:20-22
        imageCountToCheck = (count > 0 && count < (int)dbImages.size()) ? count : (int)dbImages.size();
    }
    // ann.cpp:84-93: the distance at rank (int)(n * rate).
    static float getThreshold(std::vector<float>& otherClassesDists, float falseAcceptRate);

protected:
    std::string method_name;
    std::vector<ImageInfo>& dbImages;
    int distanceCalcCount;
    float avgCheckedPercent;
    int imageCountToCheck;
    float distance(ImageInfo& testImage, int modelInd, bool updateCounters = true) {   // ann.h:33-38
        if (updateCounters) ++distanceCalcCount;
        return testImage.distance(dbImages[modelInd]);
    }
};

// ---- ann.h:42-47, ann.cpp:113-126: exhaustive nearest neighbour over all FEATURES_COUNT features ----
class BruteForce : public ClassificationMethod {
public:
    BruteForce(std::vector<ImageInfo>& db) : ClassificationMethod("BF", db) {}
    int recognize(ImageInfo& testImage) override {
        distanceCalcCount = (int)dbImages.size();                    // every row is evaluated once (ann.cpp:114,118)
        return recognize_image_bf(dbImages, testImage, FEATURES_COUNT);
    }
    std::vector<int> recognize_batch(const std::vector<ImageInfo>& tests) override {
        distanceCalcCount = (int)dbImages.size();
        return fir::recognize_images_bf(dbImages, tests, FEATURES_COUNT);
    }
};

// ---- ann.h:64-100, ann.cpp:270-507 (PIVOT build): maximum-likelihood directed enumeration ----
// The constructor's pivot table (n_pivots gallery scans), the per-query likelihood update over the table and the
// candidate distance checks run on the GPU (fir_dem_*, fir_rows_distances); the walk itself -- the index
// bookkeeping, std::partial_sort, the early exit below `threshold` -- is the reference's sequential logic on the
// host, so row, bestDistance, isFoundLessThreshold and the checked-percent counter come out the same.
class DirectedEnumeration : public ClassificationMethod {
public:
    DirectedEnumeration(std::vector<ImageInfo>& faceImages, float falseAcceptRate = 0.01f, float threshold = 0, int imageCountToCheck = 0);
    ~DirectedEnumeration();
    DirectedEnumeration(const DirectedEnumeration&) = delete;
    DirectedEnumeration& operator=(const DirectedEnumeration&) = delete;

    int recognize(ImageInfo& testImage) override;
    std::vector<int> recognize_batch(const std::vector<ImageInfo>& tests) override;

    bool isFoundLessThreshold;
    float bestDistance;

    // added: what the constructor built (tests, diagnostics)
    float getThresholdValue() const { return threshold; }
    const std::vector<int>& getStartIndices() const { return startIndices; }
    int getDistanceCalcCount() const { return distanceCalcCount; }

private:
    int finish_walk(const float* query, const float* pivot_dist, const float* likelihoods);
    float threshold;
    std::vector<int> startIndices;            // <= 32 kept (ann.cpp:333-334)
    std::vector<int> order0;                  // likelihood_indices after the pivot loop (ann.cpp:427-432)
    std::vector<int> likelihood_indices;
    fir_gallery* gallery;                     // owned: all FEATURES_COUNT features of dbImages
    fir_dem* dem;
};

// The names BASELINE.json's north_star uses for the two abstract interfaces: the ImageTesting-side `Classifier`
// (ImageTesting.cpp:35-49: train(vector<ImageInfo>*), recognize(ImageInfo&) -> class) and the ann-side
// `ClassificationMethod` (ann.h:9-39: recognize(ImageInfo&) -> row index, testSetRecognition, getThreshold).
using ImageClassifier = Classifier;
using ImageRecognizer = ClassificationMethod;

#endif  // FIR_CLASSIFIERS_H
