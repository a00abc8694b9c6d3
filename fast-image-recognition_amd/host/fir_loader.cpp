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

float slow_path(const char* p, const char* limit, const char** end) {
    // strtof needs a terminated string: copy the token
    char buf[128];
    const char* q = p;
    while (q < limit && (*q == ' ' || *q == '\t')) ++q;
    size_t len = 0;
    while (q + len < limit && len < sizeof(buf) - 1 && q[len] != ' ' && q[len] != '\t' && q[len] != '\n' && q[len] != '\r') ++len;
    std::memcpy(buf, q, len);
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
    if (!any) { (void)tok; return slow_path(p, limit, end); }          // nan / inf / garbage: let strtof decide
    if (s < limit && (*s == 'e' || *s == 'E' || *s == 'x' || *s == 'X' || *s == 'n' || *s == 'N' || *s == 'i' || *s == 'I'))
        return slow_path(p, limit, end);                               // exponent / hex forms
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
        bool failed = false;
        for (int i = 0; i < d; ++i) {
            float v = 0.0f;
            if (!failed) {
                const char* e = p;
                v = parse_float_exact(p, limit, &e);
                if (e == p) { failed = true; v = 0.0f; } else p = e;   // a failed extraction stores 0 and fails the stream (C++11)
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
    }
    std::fclose(fp);
    if (!ok) out = PackedFeatures();
    return ok ? 0 : -1;
}

}  // namespace fir
