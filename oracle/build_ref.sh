#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY. Builds the REAL reference hot path into oracle/_ref/*.so so that the
# tests (this container) and bench.py's cpu_baseline leg can call the reference's own code.
#
# The reference's .cpp files do not compile whole in this image: each one reaches
# `#include <opencv2/...>` (db_features.cpp:164, ImageTesting.cpp:290, ann.h:49,
# classification.cpp:11) or Qt (main.cpp:1-3), and neither library is installed. No stand-in
# header is written for them. Instead the hot-path LINE RANGES that need neither library
#  are streamed, unmodified, from where they lie under /root/reference
# straight into g++'s stdin, followed by this repo's own extern "C" marshalling code
# (ref_wrap_*.inc). Nothing from the reference is written into the repository; the only
# outputs are shared objects under oracle/_ref/ (git-ignored).
#
# Three builds of the match path, one per arm of the compile-time metric switch
# (db_features.h:12, db_features.cpp:25-39):
#   libref_l2.so    as shipped                      (USE_L2_DISTANCE defined)
#   libref_chi2.so  `#undef USE_L2_DISTANCE` after the header -> the active `#if 1` chi-square arm
#   libref_kl.so    same, and the `#if 1` at db_features.cpp:30 read as `#if 0` -> the reference's
#                   own (otherwise dead) KL / Jensen-Shannon arm, lines 33-36
# plus libref_cls.so from classification.cpp (double-precision kNN / PNN).
# The match TU also carries ann.h:1-47,61-100 and ann.cpp:2-22,84-126,268-507 (ClassificationMethod, BruteForce,
# DirectedEnumeration) and video.cpp:21-155 (loadVideos); the class bodies are read with `private`/`protected` defined to `public` so that the
# marshalling code can copy the pivot table and counters out -- the reference's lines themselves are untouched.
#
# Flags follow the reference's qmake project (recognition_testing.pro: c++11, release -O2).
set -euo pipefail

REF=${REF_ROOT:-/root/reference}/qt_cpp
HERE=$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)
OUT=$HERE/_ref
if [ ! -d "$REF" ]; then
    echo "build_ref.sh: $REF not present (GPU box?) - keeping prebuilt oracle/_ref" >&2
    exit 0
fi
mkdir -p "$OUT"
CXX=${CXX:-g++}
CXXFLAGS="-std=c++11 -O2 -fPIC -shared -w -I$REF -I$HERE -include cmath -include cfloat -include cstdint -include iostream -include algorithm"

match_tu() {   # $1 = metric id (0 l2, 1 chi2, 2 kl)
    echo '#include "db_features.h"'
    if [ "$1" != 0 ]; then echo '#undef USE_L2_DISTANCE'; fi
    echo "#define REF_METRIC_ID $1"
    if [ "$1" = 2 ]; then
        sed -n '1,162p;319,335p' "$REF/db_features.cpp" | sed '30s/#if 1/#if 0/'
    else
        sed -n '1,162p;319,335p' "$REF/db_features.cpp"
    fi
    sed -n '1,288p' "$REF/ImageTesting.cpp"
    echo '#define private public     /* the wrapper reads DirectedEnumeration::P_matrix / startIndices / threshold (ann.h:93-96) */'
    echo '#define protected public   /* ... and ClassificationMethod::distanceCalcCount (ann.h:30) */'
    sed -n '1,47p;61,100p' "$REF/ann.h"
    echo '#undef private'
    echo '#undef protected'
    echo '#endif'
    sed -n '2,22p;84,126p;268,507p' "$REF/ann.cpp"
    echo '#include <fstream>'
    echo '#include <sstream>'
    echo '#include <map>'
    sed -n '21,155p' "$REF/video.cpp"         # MapOfVideos, the feature file names, loadVideos (:98-154 is an '#if 0' block)
    echo '#include "ref_wrap_match.inc"'
}

cls_tu() {
    sed -n '1,10p;19,428p' "$REF/classification.cpp"
    echo '#define private public     /* the wrapper reads FPNNClassifier::a / J (classification.cpp:630-632) */'
    sed -n '617,791p' "$REF/classification.cpp"
    echo '#undef private'
    sed -n '792,862p;942,990p' "$REF/classification.cpp"
    echo '#include "ref_wrap_cls.inc"'
}

# The reference's own experiment harness (testRecognitionMethod + testRecognition, ImageTesting.cpp:439-501,503-548, minus the
# three OpenCV classifiers of :536-538), twice:
#   harness_reference  with the reference's own classes (ImageTesting.cpp:35-288) and db_features.cpp -- pure CPU
#   harness_dropin     the same harness lines against THIS repo's host shim: db.h stays the reference's (configuration only),
#                      "db_features.h" resolves to host/compat, line 33 (the file-static counter) becomes the shim's counter,
#                      lines 35-288 become one #include -- exactly the edit INTEGRATION.md describes
harness_tu() {   # $1 = reference | dropin
    echo "#include \"$REF/db.h\""
    sed -n '1,18p;21,32p' "$REF/ImageTesting.cpp"            # the includes, `using namespace std`, print_endl (not :19-20, the two project headers)
    if [ "$1" = reference ]; then
        echo "#include \"$REF/db_features.h\""
        sed -n '1,162p;319,335p' "$REF/db_features.cpp" | grep -v '^#include "db'
        sed -n '33,288p' "$REF/ImageTesting.cpp"
    else
        echo '#include "compat/db_features.h"'
        echo '#include "fir_classifiers.h"                          /* was: the classes of ImageTesting.cpp:35-288 */'
        echo '#define num_of_unreliable fir::num_of_unreliable()   /* was: static int num_of_unreliable (ImageTesting.cpp:33) */'
    fi
    sed -n '439,501p;503,535p;539,548p' "$REF/ImageTesting.cpp"
    echo 'int main() { testRecognition(); return 0; }'
}
# testANN (ann.cpp:24-81, minus the FLANN method of :56): BruteForce and DirectedEnumeration over imageCountToCheck ratios.
ann_harness_tu() {   # $1 = reference | dropin
    echo "#include \"$REF/db.h\""
    if [ "$1" = reference ]; then
        echo "#include \"$REF/db_features.h\""
        sed -n '1,162p;319,335p' "$REF/db_features.cpp" | grep -v '^#include "db'
        sed -n '1,47p;61,100p' "$REF/ann.h"
        echo '#endif'
        sed -n '3,22p' "$REF/ann.cpp"
        sed -n '24,55p;57,81p;84,126p;268,507p' "$REF/ann.cpp"
    else
        echo '#include "compat/db_features.h"'
        echo '#include "compat/ann.h"                               /* was: ann.h and the classes of ann.cpp:84-126,268-507 */'
        sed -n '3,22p' "$REF/ann.cpp"
        sed -n '24,55p;57,81p' "$REF/ann.cpp"
    fi
    echo 'int main() { testANN(); return 0; }'
}
# testClassification1 (classification.cpp:991-1089, minus the three OpenCV classifiers of :1009-1011 and the OpenCV PCA
# projection of :1032-1034): kNN-1/3, PNN, PNN with clustering, FPNN x2, sequential PNN, sequential FPNN x2 over six
# training fractions, two random splits each.
cls_harness_tu() {   # $1 = reference | dropin
    if [ "$1" = reference ]; then
        sed -n '1,10p;19,428p;617,862p;942,990p' "$REF/classification.cpp"
    else
        echo "#include \"$REF/db.h\""
        sed -n '1,9p;20,28p' "$REF/classification.cpp"      # the std includes, `using namespace std`, print_endl
        echo '#include "fir_classification.h"                  /* was: the classes and loaders of classification.cpp:31-990 */'
        echo '#include "compat/classification_globals.h"       /* was: the file-scope state of classification.cpp:53-62 */'
        sed -n '33p' "$REF/classification.cpp"                # NO_PCA_FEATURES
    fi
    sed -n '991,1008p;1012,1031p;1035,1089p' "$REF/classification.cpp"
    echo 'int main() { testClassification1(); return 0; }'
}
HOSTDIR=$HERE/../fast-image-recognition_amd/host
cls_harness_tu reference | $CXX -std=c++11 -O2 -w -I"$REF" -include cmath -include cfloat -include cstdint -x c++ - -o "$OUT/harness_cls_reference"
if [ -f "$HOSTDIR/../libfir_host.so" ]; then
    cls_harness_tu dropin | $CXX -std=c++11 -O2 -w -I"$HOSTDIR" -include cmath -include cfloat -include cstdint -x c++ - -o "$OUT/harness_cls_dropin" \
        -L"$HOSTDIR/.." -lfir_host -lfir_amd -Wl,-rpath,'$ORIGIN/../../fast-image-recognition_amd'
fi
ann_harness_tu reference | $CXX -std=c++11 -O2 -w -I"$REF" -include cmath -include cfloat -include unordered_map -include iostream -include algorithm -x c++ - -o "$OUT/harness_ann_reference"
if [ -f "$HOSTDIR/../libfir_host.so" ]; then
    ann_harness_tu dropin | $CXX -std=c++11 -O2 -w -I"$HOSTDIR" -include cmath -include cfloat -include unordered_map -x c++ - -o "$OUT/harness_ann_dropin" \
        -L"$HOSTDIR/.." -lfir_host -lfir_amd -Wl,-rpath,'$ORIGIN/../../fast-image-recognition_amd'
fi
harness_tu reference | $CXX -std=c++11 -O2 -w -I"$REF" -include cmath -x c++ - -o "$OUT/harness_reference"
if [ -f "$HOSTDIR/../libfir_host.so" ]; then
    harness_tu dropin | $CXX -std=c++11 -O2 -w -I"$HOSTDIR" -include cmath -x c++ - -o "$OUT/harness_dropin" \
        -L"$HOSTDIR/.." -lfir_host -lfir_amd -Wl,-rpath,'$ORIGIN/../../fast-image-recognition_amd'
fi

match_tu 0 | $CXX $CXXFLAGS -x c++ - -o "$OUT/libref_l2.so"
match_tu 1 | $CXX $CXXFLAGS -x c++ - -o "$OUT/libref_chi2.so"
match_tu 2 | $CXX $CXXFLAGS -x c++ - -o "$OUT/libref_kl.so"
cls_tu     | $CXX $CXXFLAGS -x c++ - -o "$OUT/libref_cls.so"
echo "built: $(ls "$OUT"/*.so | tr '\n' ' ')"
