#include "fir_loader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>

namespace fir {

namespace {
const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// The general case: the characters `istream >> float` (libstdc++ num_get) would accept -- sign, digits, one decimal
// point, one exponent with optional sign -- converted by strtof. No "nan" / "inf" / hex forms: the stream rejects them.
float slow_path(const char* p, const char* limit, const char** end) {
    char buf[128];
    const char* q = p;
    while (q < limit && (*q == ' ' || *q == '\t')) ++q;
    size_t len = 0;
    bool digits = false, point = false;
    if (q < limit && (*q == '+' || *q == '-')) buf[len++] = *q;
    while (q + len < limit && len < sizeof(buf) - 2) {
        const char c = q[len];
        if (c >= '0' && c <= '9') digits = true;
        else if (c == '.' && !point) point = true;
        else break;
        buf[len++] = c;
    }
    if (!digits) { *end = p; return 0.0f; }
    if (q + len < limit && (q[len] == 'e' || q[len] == 'E')) {
        size_t l2 = len;
        buf[l2++] = q[len];
        if (q + l2 < limit && (q[l2] == '+' || q[l2] == '-')) { buf[l2] = q[l2]; ++l2; }
        bool ed = false;
        while (q + l2 < limit && l2 < sizeof(buf) - 1 && q[l2] >= '0' && q[l2] <= '9') { buf[l2] = q[l2]; ++l2; ed = true; }
        if (!ed) { *end = p; return 0.0f; }                            // "1e": the stream fails the whole field
        len = l2;
    }
    buf[len] = 0;
    char* e = nullptr;
    const float v = std::strtof(buf, &e);
    *end = (e == buf) ? p : q + (e - buf);
    return v;
}
}  // namespace

float parse_float_exact(const char* p, const char* limit, const char** end) {
    const char* s = p;
    while (s < limit && (*s == ' ' || *s == '\t')) ++s;
    const char* tok = s;
    bool neg = false;
    if (s < limit && (*s == '-' || *s == '+')) { neg = *s == '-'; ++s; }
    uint64_t mant = 0;
    int digits = 0, frac = 0;
    bool any = false;
    while (s < limit && *s >= '0' && *s <= '9') { if (digits < 19) { mant = mant * 10 + (uint64_t)(*s - '0'); if (mant) ++digits; } else return slow_path(p, limit, end); ++s; any = true; }
    if (s < limit && *s == '.') {
        ++s;
        while (s < limit && *s >= '0' && *s <= '9') {
            if (digits >= 19) return slow_path(p, limit, end);
            mant = mant * 10 + (uint64_t)(*s - '0');
            if (mant) ++digits;
            ++frac;
            ++s;
            any = true;
        }
    }
    if (!any) { (void)tok; *end = p; return 0.0f; }                    // nan / inf / garbage: the stream extracts nothing
    if (s < limit && (*s == 'e' || *s == 'E')) return slow_path(p, limit, end);   // exponent form
    if (digits > 15 || frac > 22) return slow_path(p, limit, end);     // (double)mant must be exact, 10^frac too
    *end = s;
    if (mant == 0) return neg ? -0.0f : 0.0f;
    const double v = (double)mant / kPow10[frac];                      // one correctly rounded double operation
    // (float)v equals the correctly rounded float of the decimal unless the decimal sits within one double ulp of
    // a float rounding boundary: the 29 bits below float precision are then 0x0FFFFFFF, 0x10000000 or 0x10000001.
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    const uint32_t low = (uint32_t)(bits & 0x1FFFFFFFull);
    const int exp2 = (int)((bits >> 52) & 0x7FF) - 1023;
    if (low == 0x0FFFFFFFu || low == 0x10000000u || low == 0x10000001u || exp2 < -126 || exp2 > 126) return slow_path(p, limit, end);
    const float f = (float)v;
    return neg ? -f : f;
}

namespace {

struct Record {
    const char* feat;      // start of the feature line
    const char* feat_end;  // its end (exclusive)
    int32_t cls;
};

void parse_range(const std::vector<Record>& recs, size_t lo, size_t hi, const std::vector<int64_t>& dest, int d, bool l2, float* rows) {
    for (size_t r = lo; r < hi; ++r) {
        float* f = rows + dest[r] * d;
        const char* p = recs[r].feat;
        const char* limit = recs[r].feat_end;
        float norm = 0.0f;
        float v = 0.0f;                                                // `dfeature` lives across the loop (db_features.cpp:81)
        bool failed = false;
        for (int i = 0; i < d; ++i) {
            if (!failed) {
                // `iss >> dfeature`: at the end of the line nothing is extracted and dfeature keeps its (clipped) value;
                // a malformed token stores 0 (C++11); either way the stream stays failed for the rest of the row
                const char* q = p;
                while (q < limit && std::strchr(" \t\n\r\f\v", *q)) ++q;
                if (q >= limit) failed = true;
                else {
                    const char* e = q;
                    const float parsed = parse_float_exact(q, limit, &e);
                    if (e == q) { failed = true; v = 0.0f; } else { v = parsed; p = e; }
                }
            }
            if (std::fabs(v) < 0.0001) v = 0.0f;                       // db_features.cpp:85-86
            f[i] = v;
            norm += l2 ? v * v : v;                                    // :88 / :91
        }
        if (l2) norm = std::sqrt(norm);                                // :95
        for (int i = 0; i < d; ++i) f[i] /= norm;                      // :98-99
    }
}

}  // namespace

int64_t load_features_packed(const std::string& features_file, int d, int metric, PackedFeatures& out, int threads) {
    out = PackedFeatures();
    out.d = d;
    out.metric = metric;
    const int fd = ::open(features_file.c_str(), O_RDONLY);
    if (fd < 0) return 0;
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size == 0) { ::close(fd); return 0; }
    const size_t size = (size_t)st.st_size;
    const char* base = (const char*)::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (base == MAP_FAILED) return 0;
    ::madvise((void*)base, size, MADV_SEQUENTIAL);
    const char* limit = base + size;

    // index the records: three lines each; a record is complete only if all three lines could be read
    std::vector<Record> recs;
    std::map<std::string, int32_t> class_id;
    const char* p = base;
    auto next_line = [&](const char*& b, const char*& e) -> bool {
        if (p >= limit) return false;
        b = p;
        const char* nl = (const char*)std::memchr(p, '\n', (size_t)(limit - p));
        e = nl ? nl : limit;
        p = nl ? nl + 1 : limit;
        return true;
    };
    for (;;) {
        const char *b1, *e1, *b2, *e2, *b3, *e3;
        if (!next_line(b1, e1) || !next_line(b2, e2) || !next_line(b3, e3)) break;
        while (b2 < e2 && std::strchr(" \t\n\r\f\v", *b2)) ++b2;       // leading blanks of the class line (:57)
        std::string name(b2, e2);
        if (name.find("BACKGROUND_Google") != std::string::npos || name.find("257.clutter") != std::string::npos) continue;   // :60-64
        auto it = class_id.find(name);
        if (it == class_id.end()) {
            it = class_id.emplace(name, (int32_t)out.class_names.size()).first;
            out.class_names.push_back(name);
        }
        Record r;
        r.feat = b3;
        r.feat_end = e3;
        r.cls = it->second;
        recs.push_back(r);
    }
    const size_t n = recs.size();
    out.n = (int64_t)n;
    // class-major destination rows (ImagesDatabase order: class by first appearance, then file order)
    std::vector<int64_t> start(out.class_names.size() + 1, 0), dest(n);
    for (const Record& r : recs) start[(size_t)r.cls + 1]++;
    for (size_t c = 0; c < out.class_names.size(); ++c) start[c + 1] += start[c];
    out.class_no.resize(n);
    out.index_in_file.resize(n);
    for (size_t r = 0; r < n; ++r) {
        dest[r] = start[(size_t)recs[r].cls]++;
        out.class_no[(size_t)dest[r]] = recs[r].cls;
        out.index_in_file[(size_t)dest[r]] = (int32_t)r;
    }
    out.rows.assign(n * (size_t)d, 0.0f);
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min<int>(nt, (int)std::max<size_t>(n / 8, 1)));
    const bool l2 = metric == 0;
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) {
        const size_t lo = n * (size_t)t / (size_t)nt, hi = n * (size_t)(t + 1) / (size_t)nt;
        pool.emplace_back(parse_range, std::cref(recs), lo, hi, std::cref(dest), d, l2, out.rows.data());
    }
    for (std::thread& th : pool) th.join();
    ::munmap((void*)base, size);
    return out.n;
}

namespace {
const char kMagic[8] = {'F', 'I', 'R', 'F', 'E', 'A', 'T', '1'};
}

int save_feature_cache(const std::string& cache_file, const PackedFeatures& f) {
    FILE* fp = std::fopen(cache_file.c_str(), "wb");
    if (!fp) return -1;
    const int64_t hdr[4] = {f.n, f.d, f.metric, (int64_t)f.class_names.size()};
    bool ok = std::fwrite(kMagic, 1, 8, fp) == 8 && std::fwrite(hdr, sizeof(int64_t), 4, fp) == 4;
    ok = ok && std::fwrite(f.rows.data(), sizeof(float), f.rows.size(), fp) == f.rows.size();
    ok = ok && std::fwrite(f.class_no.data(), sizeof(int32_t), f.class_no.size(), fp) == f.class_no.size();
    ok = ok && std::fwrite(f.index_in_file.data(), sizeof(int32_t), f.index_in_file.size(), fp) == f.index_in_file.size();
    for (const std::string& s : f.class_names) {
        const int32_t len = (int32_t)s.size();
        ok = ok && std::fwrite(&len, 4, 1, fp) == 1 && std::fwrite(s.data(), 1, s.size(), fp) == s.size();
    }
    return (std::fclose(fp) == 0 && ok) ? 0 : -1;
}

int load_feature_cache(const std::string& cache_file, PackedFeatures& out) {
    out = PackedFeatures();
    FILE* fp = std::fopen(cache_file.c_str(), "rb");
    if (!fp) return -1;
    char magic[8];
    int64_t hdr[4];
    bool ok = std::fread(magic, 1, 8, fp) == 8 && std::memcmp(magic, kMagic, 8) == 0 && std::fread(hdr, sizeof(int64_t), 4, fp) == 4;
    ok = ok && hdr[0] >= 0 && hdr[1] > 0 && hdr[3] >= 0;
    // a truncated or damaged cache must not size an allocation: the fixed part of the file has to be there in full
    // (rows, class ids, file indices; the class names follow, at least their length words)
    if (ok) {
        struct stat st;
        ok = ::fstat(::fileno(fp), &st) == 0 && hdr[0] < ((int64_t)1 << 31) && hdr[1] < (1 << 24) && hdr[3] <= hdr[0] + 1 && hdr[2] >= 0 && hdr[2] <= 2;
        if (ok) {
            const unsigned __int128 fixed = (unsigned __int128)8 + 32 + (unsigned __int128)hdr[0] * (unsigned __int128)hdr[1] * 4 + (unsigned __int128)hdr[0] * 8 +
                                            (unsigned __int128)hdr[3] * 4;
            ok = fixed <= (unsigned __int128)st.st_size;
        }
    }
    if (ok) {
        out.n = hdr[0]; out.d = (int)hdr[1]; out.metric = (int)hdr[2];
        out.rows.resize((size_t)out.n * out.d);
        out.class_no.resize((size_t)out.n);
        out.index_in_file.resize((size_t)out.n);
        ok = std::fread(out.rows.data(), sizeof(float), out.rows.size(), fp) == out.rows.size();
        ok = ok && std::fread(out.class_no.data(), sizeof(int32_t), out.class_no.size(), fp) == out.class_no.size();
        ok = ok && std::fread(out.index_in_file.data(), sizeof(int32_t), out.index_in_file.size(), fp) == out.index_in_file.size();
        for (int64_t c = 0; ok && c < hdr[3]; ++c) {
            int32_t len = 0;
            ok = std::fread(&len, 4, 1, fp) == 1 && len >= 0 && len < (1 << 20);
            std::string s((size_t)(ok ? len : 0), '\0');
            ok = ok && std::fread(&s[0], 1, s.size(), fp) == s.size();
            out.class_names.push_back(s);
        }
        // class ids index class_names (fir_gallery_create and the classifiers trust them)
        for (int64_t i = 0; ok && i < out.n; ++i) ok = out.class_no[(size_t)i] >= 0 && out.class_no[(size_t)i] < hdr[3];
        ok = ok && std::fgetc(fp) == EOF;                                   // nothing may follow the last class name
    }
    std::fclose(fp);
    if (!ok) out = PackedFeatures();
    return ok ? 0 : -1;
}

// ---- video.cpp:35-96 ----
namespace {
struct Cursor {                                  // the reference's ifstream, over the mapped file
    const char* p;
    const char* limit;
    bool fail = false;
    bool getline(const char*& b, const char*& e) {                     // std::getline: fails only when nothing is left
        if (fail || p >= limit) { fail = true; return false; }
        b = p;
        const char* nl = (const char*)std::memchr(p, '\n', (size_t)(limit - p));
        e = nl ? nl : limit;
        p = nl ? nl + 1 : limit;
        return true;
    }
    bool read_int(int& v) {                                            // ifs >> int
        if (fail) return false;
        while (p < limit && std::strchr(" \t\n\r\f\v", *p)) ++p;
        const char* s = p;
        bool neg = false;
        if (s < limit && (*s == '-' || *s == '+')) { neg = *s == '-'; ++s; }
        long long acc = 0;
        const char* digits = s;
        while (s < limit && *s >= '0' && *s <= '9' && acc < (1ll << 40)) acc = acc * 10 + (*s++ - '0');
        if (s == digits) { fail = true; v = 0; return false; }         // num_get failure stores 0
        p = s;
        v = (int)(neg ? -acc : acc);
        return true;
    }
};
}  // namespace

int64_t load_videos_packed(const std::string& video_features_file, int d, int metric, PackedVideos& out) {
    out = PackedVideos();
    out.d = d;
    const int fd = ::open(video_features_file.c_str(), O_RDONLY);
    if (fd < 0) return 0;
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size == 0) { ::close(fd); return 0; }
    const size_t size = (size_t)st.st_size;
    const char* base = (const char*)::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (base == MAP_FAILED) return 0;
    Cursor in{base, base + size};
    const bool l2 = metric == 0;
    // person -> videos -> frames (rows), exactly the reference's MapOfVideos while reading
    std::map<std::string, std::vector<std::vector<std::vector<float> > > > db;
    while (!in.fail) {
        const char *b, *e;
        if (!in.getline(b, e)) break;                                  // :42-43
        while (b < e && std::strchr(" \t\n\r\f\v", *b)) ++b;           // :44
        const std::string personName(b, e);
        int videos_count = 0;
        in.read_int(videos_count);                                     // :46
        if (videos_count < 0) videos_count = 0;
        std::vector<std::vector<std::vector<float> > >& person_videos = db[personName];   // :49-50
        person_videos.resize((size_t)videos_count);                    // :51 (a repeated name re-sizes the earlier entry)
        for (int i = 0; i < videos_count; ++i) {
            int frames_count = 0;
            in.read_int(frames_count);                                 // :55
            if (frames_count < 0) frames_count = 0;
            person_videos[(size_t)i].assign((size_t)frames_count, std::vector<float>());   // :57
            if (!in.getline(b, e)) break;                              // rest of the count line, :59-60
            for (int j = 0; j < frames_count; ++j) {
                const char *fb, *fe;
                if (!in.getline(b, e)) break;                          // file name, :63
                if (!in.getline(fb, fe)) break;                        // features, :65
                std::vector<float>& f = person_videos[(size_t)i][(size_t)j];
                f.assign((size_t)d, 0.0f);
                float dfeature = 0.0f, sum = 0.0f;                     // dfeature is uninitialised in the reference (:70)
                bool stream_failed = false;
                const char* p = fb;
                for (int k = 0; k < d; ++k) {
                    if (!stream_failed) {
                        const char* q = p;
                        while (q < fe && std::strchr(" \t\n\r\f\v", *q)) ++q;
                        if (q >= fe) stream_failed = true;             // end of line: the extraction leaves dfeature as it was
                        else {
                            const char* endp = q;
                            const float v = parse_float_exact(q, fe, &endp);
                            if (endp == q) { stream_failed = true; dfeature = 0.0f; }   // malformed token: 0 is stored
                            else { dfeature = v; p = endp; }
                        }
                    }
                    if (std::fabs(dfeature) < 0.0001) dfeature = 0;    // :74-75
                    f[(size_t)k] = dfeature;
                    sum += dfeature * dfeature;                        // :78
                }
                if (l2) sum = std::sqrt(sum);                          // :81-83
                for (int k = 0; k < d; ++k) f[(size_t)k] /= sum;       // :84-85
                ++out.total_images;
            }
        }
        out.total_videos += videos_count;                              // :91
    }
    ::munmap((void*)base, size);
    out.video_first.push_back(0);
    out.frame_first.push_back(0);
    for (auto& kv : db) {
        out.person.push_back(kv.first);
        for (auto& video : kv.second) {
            for (auto& frame : video) {
                const size_t at = out.rows.size();
                out.rows.resize(at + (size_t)d, 0.0f);
                if (frame.size() == (size_t)d) std::memcpy(&out.rows[at], frame.data(), sizeof(float) * (size_t)d);
            }
            out.frame_first.push_back(out.frame_first.back() + (int64_t)video.size());
        }
        out.video_first.push_back((int32_t)(out.frame_first.size() - 1));
    }
    return (int64_t)out.person.size();
}

}  // namespace fir
