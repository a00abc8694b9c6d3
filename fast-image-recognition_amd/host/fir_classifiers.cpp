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
