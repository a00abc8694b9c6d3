// Source-compatibility shim: the reference includes "ann.h" (qt_cpp/ann.h).
#include "../fir_classifiers.h"
