// fir_loader.h -- fast reader of the reference's feature files and a binary cache.
//
// Format (producer qt_cpp/dnn_feature_extractor.py:58-64): per image three lines -- path, class name,
// "%f " x D. Consumer semantics reproduced exactly (qt_cpp/db_features.cpp:44-116): classes
// numbered by first appearance, BACKGROUND_Google / 257.clutter skipped, |x| < 1e-4 -> 0, then the
// row divided by its L2 norm (L2 metric) or its sum (chi2 / KL), all in float, in file order --
// the packed rows are bit-identical to what loadImages leaves in ImagesDatabase.
//
// What is fast: the file is memory-mapped, records are indexed once and parsed by a pool of
// threads; a token of the usual fixed-point form is converted as (double)digits / 10^k -> float
// with a guard that detects the (rare) double-rounding cases and hands only those to strtof.
#ifndef FIR_LOADER_H
#define FIR_LOADER_H

#include <cstdint>
#include <string>
#include <vector>

namespace fir {

struct PackedFeatures {
    int64_t n = 0;                         // images
    int d = 0;                             // features per image
    int metric = 0;                        // FIR_METRIC_* the rows were normalised for
    std::vector<float> rows;               // [n][d], class-major (ImagesDatabase order)
    std::vector<int32_t> class_no;         // [n]
    std::vector<int32_t> index_in_file;    // [n] record number in the file (after skipping)
    std::vector<std::string> class_names;  // by class id
};

// Returns the number of images (0 when the file cannot be opened, like db_features.cpp:49,115).
// threads <= 0: one per hardware thread.
int64_t load_features_packed(const std::string& features_file, int d, int metric, PackedFeatures& out, int threads = 0);

// Binary cache of a PackedFeatures (little endian, versioned). Return 0 on success, -1 on any error.
int save_feature_cache(const std::string& cache_file, const PackedFeatures& f);
int load_feature_cache(const std::string& cache_file, PackedFeatures& out);

// The video-feature file of qt_cpp/video.cpp:35-96 (loadVideos): per person a name line, the number of videos, per
// video the number of frames and per frame a file-name line and a "%f " x D line. Same clip (|x| < 1e-4 -> 0) as the
// image files; the row is divided by its L2 norm (L2 metric) or -- as the reference does here, unlike loadImages --
// by its sum of squares (other metrics), video.cpp:74-83. Persons come back sorted by name (the reference's std::map).
struct PackedVideos {
    int d = 0;
    std::vector<std::string> person;            // sorted, unique
    std::vector<int32_t> video_first;           // [persons + 1]: videos of person p are video_first[p] .. video_first[p+1]
    std::vector<int64_t> frame_first;           // [videos + 1]: rows of video v are frame_first[v] .. frame_first[v+1]
    std::vector<float> rows;                    // [frames][d]
    int64_t total_images = 0, total_videos = 0; // the counters loadVideos prints (video.cpp:92)
};
// Returns the number of persons (0 when the file cannot be opened, video.cpp:38).
int64_t load_videos_packed(const std::string& video_features_file, int d, int metric, PackedVideos& out);

// One token -> float exactly as `istream >> float` converts it (strtof on the characters num_get accepts: no nan / inf /
// hex); *end is set past the token (== p when nothing could be extracted).
float parse_float_exact(const char* p, const char* limit, const char** end);

}  // namespace fir

#endif  // FIR_LOADER_H
