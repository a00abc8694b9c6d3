// cls_driver.cpp -- drives the classification.cpp-side API the way testClassification1 does
// (classification.cpp:991-1060: load, split, train, predict every test item) and prints JSON.
//   cls_driver <features.txt> <features_count> <fraction>
//   cls_driver --dump <features.txt> <features_count> <out.bin>   (load_image_dataset only; runs without a GPU)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "fir_classification.h"

static void print_vec(const char* key, const std::vector<int>& v, bool comma) {
    std::printf("\"%s\": [", key);
    for (size_t i = 0; i < v.size(); ++i) std::printf("%s%d", i ? ", " : "", v[i]);
    std::printf("]%s\n", comma ? "," : "");
}

int main(int argc, char** argv) {
    std::cout.rdbuf(nullptr);   // stdout carries this driver's JSON (printf); the loader's reference-style cout lines are dropped
    if (argc >= 5 && !std::strcmp(argv[1], "--dump")) {       // --dump <features.txt> <features_count> <out.bin>: loader only, no GPU
        load_image_dataset(argv[2], std::atoi(argv[3]));
        fir::ClassificationState& st = fir::classification_state();
        FILE* fp = std::fopen(argv[4], "wb");
        for (const Feature_vector& fv : st.dataset) std::fwrite(fv.features.data(), sizeof(double), fv.features.size(), fp);
        std::fclose(fp);
        std::printf("{\"rows\": %zu, \"classes\": %zu, \"labels\": [", st.dataset.size(), st.num_of_classes);
        for (size_t i = 0; i < st.dataset.size(); ++i) std::printf("%s%d", i ? ", " : "", (int)st.dataset[i].output);
        std::printf("]}\n");
        return 0;
    }
    if (argc < 4) { std::fprintf(stderr, "usage: cls_driver <features.txt> <features_count> <fraction>\n"); return 2; }
    load_image_dataset(argv[1], std::atoi(argv[2]));
    split_train_test(std::atof(argv[3]), /*shuffle=*/false);
    fir::ClassificationState& st = fir::classification_state();
    std::vector<int> truth, test_rows, train_rows;
    std::vector<const Feature_vector*> inputs;
    for (size_t j : st.test_set) { truth.push_back((int)st.dataset[j].output); test_rows.push_back((int)j); inputs.push_back(&st.tmp_dataset[j]); }
    for (auto& c : st.training_set) for (size_t t : c) train_rows.push_back((int)t);
    std::printf("{\n\"classes\": %zu, \"features\": %zu, \"rows\": %zu,\n", st.num_of_classes, st.num_of_cont_features, st.dataset.size());
    print_vec("truth", truth, true);
    print_vec("test_rows", test_rows, true);
    print_vec("train_rows", train_rows, true);
    KNNClassifier knn1(1), knn3(3);
    PNNClassifier pnn(true), pnn_seq(false);
    PNNwithClusteringClassifier clust(5);
    FPNNClassifier fpnn(1.0, true), fpnn033(0.33, true), fpnn_seq(1.0, false), fpnn033_seq(0.33, false, 0.99f);   // :1002-1007
    Classifier* all[9] = {&knn1, &knn3, &pnn, &pnn_seq, &clust, &fpnn, &fpnn033, &fpnn_seq, &fpnn033_seq};
    const char* keys[9] = {"knn1", "knn3", "pnn", "pnn_seq", "pnn_clust5", "fpnn", "fpnn033", "fpnn_seq", "fpnn033_seq"};
    for (int i = 0; i < 9; ++i) {
        all[i]->train();
        std::vector<int> one;
        for (const Feature_vector* fv : inputs) one.push_back(all[i]->predict(*fv));
        std::vector<int> batch = all[i]->predict_batch(inputs);
        char key[32];
        std::snprintf(key, sizeof key, "%s_single", keys[i]);
        print_vec(key, one, true);
        std::snprintf(key, sizeof key, "%s_batch", keys[i]);
        print_vec(key, batch, true);
        std::printf("\"%s_name\": \"%s\",\n", keys[i], all[i]->get_name().c_str());
    }
    std::vector<int> medoids;
    for (const auto& c : clust.clusters()) for (size_t t : c) medoids.push_back((int)t);
    print_vec("medoid_rows", medoids, true);
    std::printf("\"fpnn_J\": %d,\n", fpnn.harmonics());
    std::printf("\"avg0\": %.17g\n}\n", st.avgValues.empty() ? 0.0 : st.avgValues[0]);
    if (st.model) fir_cls_destroy(st.model);
    return 0;
}
