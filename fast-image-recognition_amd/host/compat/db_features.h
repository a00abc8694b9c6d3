// Source-compatibility shim: the reference includes "db_features.h" (qt_cpp/db_features.h).
#include "../fir_db.h"
