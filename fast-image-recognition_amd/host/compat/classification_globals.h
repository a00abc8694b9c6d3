// Source-compatibility shim for code written against the file-scope state of qt_cpp/classification.cpp:53-62 (the
// harnesses testClassification1 / testClassification read and write `dataset`, `test_set`, `indices`, ... directly):
// the same names, bound to the one state object the classifiers of fir_classification.h use. Include it in the
// translation unit that holds such code, after "fir_classification.h".
#ifndef FIR_CLASSIFICATION_GLOBALS_H
#define FIR_CLASSIFICATION_GLOBALS_H

#include "../fir_classification.h"

namespace {
size_t& num_of_classes = fir::classification_state().num_of_classes;
size_t& num_of_cont_features = fir::classification_state().num_of_cont_features;
size_t& num_of_cont_features_orig = fir::classification_state().num_of_cont_features_orig;
std::vector<Feature_vector>& dataset = fir::classification_state().dataset;
std::vector<Feature_vector>& tmp_dataset = fir::classification_state().tmp_dataset;
std::vector<std::vector<size_t> >& indices = fir::classification_state().indices;
std::vector<std::vector<size_t> >& training_set = fir::classification_state().training_set;
std::vector<size_t>& test_set = fir::classification_state().test_set;
std::vector<FEATURE_TYPE>& minValues = fir::classification_state().minValues;
std::vector<FEATURE_TYPE>& maxValues = fir::classification_state().maxValues;
std::vector<FEATURE_TYPE>& avgValues = fir::classification_state().avgValues;
std::vector<FEATURE_TYPE>& stdValues = fir::classification_state().stdValues;
}  // namespace

#endif  // FIR_CLASSIFICATION_GLOBALS_H
