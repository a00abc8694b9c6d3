// Source-compatibility shim: the reference includes "db.h" (qt_cpp/db.h).
#include "../fir_db.h"
