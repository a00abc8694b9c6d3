"""MI355X-native gallery matcher: Python plumbing over the C ABI (include/fir_amd.h).

The product is the HIP library ``libfir_amd.so`` and the C++ host shim in ``host/``; this
module only loads the library with ctypes so that tests and bench.py can drive it. It computes
nothing itself and has no fallback: if the library or a gfx950 device is missing, calls raise.

The directory name is not an importable identifier; load it with
``__graft_entry__.load_package()`` (importlib by path) as ``fast_image_recognition_amd``.
"""
from .capi import (  # noqa: F401
    ClsModel,
    FirError,
    Fpnn,
    Gallery,
    Dem,
    GemmSearch,
    METRIC_CHI2,
    METRIC_KL,
    METRIC_L2,
    SHADOW_ALL,
    SHADOW_FP16,
    SHADOW_NONE,
    ShardedClsModel,
    ShardedGallery,
    comm_unique_id,
    device_count,
    device_info,
    device_peak_hbm_gbs,
    feature_distance,
    key_pack,
    keys_unpack,
    lib,
    lib_path,
)
