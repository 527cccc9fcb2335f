// tle_host.cpp -- see tle_host.h
#include "tle_host.h"

#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <system_error>
#include <exception>
#include <mutex>
#include <thread>

namespace azh {

namespace {

std::string_view strip(std::string_view s, const char *set)
{
    size_t a = s.find_first_not_of(set);
    if (a == std::string_view::npos) return {};
    size_t b = s.find_last_not_of(set);
    return s.substr(a, b - a + 1);
}

inline std::string_view cols(std::string_view line, size_t from, size_t to)
{
    return strip(line.substr(from, to - from), " ");
}

bool to_double_slow(std::string_view f, double &v)
{
    if (f.empty() || f.size() >= 40) return false;
    char buf[40];
    memcpy(buf, f.data(), f.size());
    buf[f.size()] = 0;
    char *end = nullptr;
    v = strtod(buf, &end);
    return end == buf + f.size();
}

// Fixed-point decimal fields ([sign] digits [. digits], at most 15 significant digits -- every numeric TLE column):
// the digits form an integer below 2^53 and the scale 10^k (k <= 15) is a double too, so ONE correctly rounded
// division gives the correctly rounded value, bit for bit what strtod returns (Clinger's fast path).  Anything else --
// exponents, more digits, stray characters -- goes to strtod.  (Catalog-scale ingest, SURVEY.md 8-f4: strtod and its
// NUL-terminated copy were three quarters of the per-record time.)
bool to_double(std::string_view f, double &v)
{
    static const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    size_t i = 0;
    const size_t n = f.size();
    bool neg = false;
    if (i < n && (f[i] == '+' || f[i] == '-')) neg = f[i++] == '-';
    uint64_t m = 0;
    int digits = 0, frac = 0;
    bool seen_point = false, any = false;
    for (; i < n; ++i) {
        const char c = f[i];
        if (c >= '0' && c <= '9') {
            any = true;
            if (m != 0 || c != '0') ++digits; // leading zeros carry no significance
            if (digits > 15) return to_double_slow(f, v);
            m = m * 10u + (uint64_t)(c - '0');
            if (seen_point) ++frac;
        } else if (c == '.' && !seen_point) {
            seen_point = true;
        } else {
            return to_double_slow(f, v);
        }
    }
    if (!any || frac > 15) return to_double_slow(f, v);
    const double x = (double)m / p10[frac];
    v = neg ? -x : x;
    return true;
}

bool to_long_slow(std::string_view f, long &v)
{
    if (f.empty() || f.size() >= 24) return false;
    char buf[24];
    memcpy(buf, f.data(), f.size());
    buf[f.size()] = 0;
    char *end = nullptr;
    v = strtol(buf, &end, 10);
    return end == buf + f.size();
}
bool to_long(std::string_view f, long &v)
{
    size_t i = 0;
    const size_t n = f.size();
    if (n == 0 || n > 10) return to_long_slow(f, v);
    bool neg = false;
    if (f[0] == '+' || f[0] == '-') neg = f[i++] == '-';
    if (i == n) return false;
    long m = 0;
    for (; i < n; ++i) {
        if (f[i] < '0' || f[i] > '9') return to_long_slow(f, v);
        m = m * 10 + (f[i] - '0');
    }
    v = neg ? -m : m;
    return true;
}

// next line with at least 69 characters once blanks/tabs are trimmed
bool next_record_line(std::string_view text, size_t &pos, std::string_view &line)
{
    while (pos < text.size()) {
        size_t e = text.find_first_of("\n\r", pos);
        if (e == std::string_view::npos) e = text.size();
        std::string_view raw = strip(text.substr(pos, e - pos), " \t");
        pos = (e < text.size()) ? e + 1 : e;
        if (raw.size() >= 69) {
            line = raw;
            return true;
        }
    }
    return false;
}

} // namespace

double year_doy_to_jd(int full_year, double doy)
{
    // Julian day number of 1 January (Gregorian), then midnight-based day-of-year offset
    const double y = double(full_year) + 4800.0 - 1.0; // January: year shifted by one, month index 10
    const double m = 10.0;
    const double jdn = 1.0 + std::floor((153.0 * m + 2.0) / 5.0) + 365.0 * y + std::floor(y / 4.0) -
                       std::floor(y / 100.0) + std::floor(y / 400.0) - 32045.0;
    return jdn + doy - 1.5;
}

int parse_lines(std::string_view l1, std::string_view l2, TleRecord &t)
{
    if (l1.size() < 69 || l2.size() < 69) return -1;
    t = TleRecord{};
    long iv;
    double dv;

    // NORAD id, alpha-5 aware: leading letter A..Z stands for 10..35
    std::string_view id = cols(l1, 2, 7);
    if (id.empty()) return -999;
    if (id[0] >= 'A' && id[0] <= 'Z') {
        if (!to_long(id.substr(1), iv)) return -999;
        t.satnum = uint32_t(id[0] - 'A' + 10) * 10000u + uint32_t(iv);
    } else {
        if (!to_long(id, iv)) return -999;
        t.satnum = uint32_t(iv);
    }
    t.classification = l1[7];

    // B*: five-digit mantissa with implied leading "0." and a signed decimal exponent
    if (!to_double(cols(l1, 53, 59), dv) || !to_long(cols(l1, 59, 61), iv)) return -999;
    t.bstar = (dv * 1e-5) * std::pow(10.0, double(iv));

    if (!to_long(cols(l1, 18, 20), iv)) return -999;
    t.epoch_year = int(iv);
    if (!to_double(cols(l1, 20, 32), t.epoch_day)) return -999;
    t.epoch_jd = year_doy_to_jd(t.epoch_year < 57 ? 2000 + t.epoch_year : 1900 + t.epoch_year, t.epoch_day);

    if (!to_double(cols(l1, 33, 43), t.ndot)) return -999;
    if (!to_long(cols(l1, 64, 68), iv)) return -999;
    t.elnum = uint32_t(iv);

    if (!to_double(cols(l2, 8, 16), t.incl_deg)) return -999;
    if (!to_double(cols(l2, 17, 25), t.raan_deg)) return -999;
    if (!to_double(cols(l2, 26, 33), dv)) return -999;
    t.ecc = dv / 1e7;
    if (!to_double(cols(l2, 34, 42), t.argp_deg)) return -999;
    if (!to_double(cols(l2, 43, 51), t.ma_deg)) return -999;
    if (!to_double(cols(l2, 52, 63), t.mm_revday)) return -999;
    if (!to_long(cols(l2, 63, 68), iv)) return -999;
    t.revnum = uint32_t(iv);
    return 0;
}

int parse_first(std::string_view text, TleRecord &out)
{
    size_t pos = 0;
    std::string_view a, b;
    if (!next_record_line(text, pos, a)) return -1;
    if (!next_record_line(text, pos, b)) return -1;
    return parse_lines(a, b, out);
}

namespace {
void parse_range(std::string_view text, std::vector<TleRecord> &out)
{
    size_t pos = 0;
    std::string_view line, pending;
    bool have = false;
    while (next_record_line(text, pos, line)) {
        if (line[0] == '1') {
            pending = line;
            have = true;
        } else if (line[0] == '2') {
            if (have) {
                have = false;
                TleRecord r;
                if (parse_lines(pending, line, r) == 0) out.push_back(r);
            }
        } else {
            have = false;
        }
    }
}
std::atomic<unsigned> g_parse_threads{0}; // 0 = automatic

// tasks 0 .. n-1, task 0 on the calling thread; a task whose thread cannot be created runs on the calling thread as well.
// Exception-safe: a task that throws (bad_alloc on a 140-MB catalog) is caught inside its thread, every thread is joined, and
// the FIRST exception is rethrown on the calling thread -- where the extern "C" entry points turn it into an error code.
void run_tasks(unsigned n, const std::function<void(unsigned)> &task)
{
    std::vector<std::thread> th;
    std::vector<unsigned> here;
    std::exception_ptr first;
    std::mutex mu;
    auto guarded = [&](unsigned k) {
        try {
            task(k);
        } catch (...) {
            std::lock_guard<std::mutex> lock(mu);
            if (!first) first = std::current_exception();
        }
    };
    try {
        th.reserve(n);
        for (unsigned k = 1; k < n; ++k) {
            try {
                th.emplace_back(guarded, k);
            } catch (const std::system_error &) {
                here.push_back(k);
            }
        }
    } catch (...) {
        std::lock_guard<std::mutex> lock(mu);
        if (!first) first = std::current_exception();
    }
    if (n) guarded(0);
    for (unsigned k : here) guarded(k);
    for (auto &t : th) t.join();
    if (first) std::rethrow_exception(first);
}
} // namespace

void set_parse_threads(unsigned n) { g_parse_threads.store(n, std::memory_order_relaxed); }

// Catalog-scale text (config 5: 10^6 TLEs = 140 MB) is cut at record boundaries and parsed by several threads.  A cut is
// moved forward to the start of the next record line that begins with '1': whatever precedes it, the serial reader's
// state after such a line is (pending = that line), so every piece reproduces the serial result and the pieces
// concatenate in order.
void parse_all(std::string_view text, std::vector<TleRecord> &out)
{
    const unsigned want = parse_threads_for(text.size()); // (pieces of at least 256 KiB, ~1,800 records: below that a thread costs more than it parses)
    if (want <= 1) {
        parse_range(text, out);
        return;
    }
    std::vector<size_t> cut(want + 1, text.size());
    cut[0] = 0;
    for (unsigned k = 1; k < want; ++k) {
        size_t pos = std::max(cut[k - 1], text.size() / want * k);
        // start of the next line
        size_t e = text.find_first_of("\n\r", pos);
        pos = e == std::string_view::npos ? text.size() : e + 1;
        // ... that is a line 1 (>= 69 significant characters, first one '1')
        while (pos < text.size()) {
            size_t le = text.find_first_of("\n\r", pos);
            if (le == std::string_view::npos) le = text.size();
            std::string_view raw = strip(text.substr(pos, le - pos), " \t");
            if (raw.size() >= 69 && raw[0] == '1') break;
            pos = le < text.size() ? le + 1 : le;
        }
        cut[k] = pos;
    }
    std::vector<std::vector<TleRecord>> piece(want);
    run_tasks(want, [&](unsigned k) { parse_range(text.substr(cut[k], cut[k + 1] - cut[k]), piece[k]); });
    std::vector<size_t> at(want + 1, out.size());
    for (unsigned k = 0; k < want; ++k) at[k + 1] = at[k] + piece[k].size();
    out.resize(at[want]);
    parallel_ranges(want, want, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k)
            if (!piece[k].empty()) memcpy(static_cast<void *>(out.data() + at[k]), piece[k].data(), piece[k].size() * sizeof(TleRecord));
    });
}

unsigned parse_threads_for(size_t bytes)
{
    // automatic: the host's cores (at most 16), pieces of at least 256 KiB; an explicit count is honoured down to 4-KiB pieces
    if (const unsigned forced = g_parse_threads.load(std::memory_order_relaxed))
        return (unsigned)std::min<size_t>(forced, std::max<size_t>(1, bytes >> 12));
    const unsigned want = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    return (unsigned)std::min<size_t>(want, std::max<size_t>(1, bytes >> 18));
}

void parallel_ranges(size_t n, unsigned threads, const std::function<void(size_t, size_t)> &fn)
{
    threads = (unsigned)std::min<size_t>(std::max(1u, threads), std::max<size_t>(1, n));
    if (threads <= 1) {
        fn(0, n);
        return;
    }
    run_tasks(threads, [&](unsigned k) { fn(n * k / threads, n * (k + 1) / threads); });
}

// ---------------------------------------------------------------------------------------------
// OMM JSON.  A deliberately small reader: objects whose values are strings, numbers, true/false/null
// (anything nested is skipped), which is all an OMM record holds.
namespace {

struct JsonCursor {
    std::string_view t;
    size_t i = 0;
    void ws() { while (i < t.size() && (t[i] == ' ' || t[i] == '\t' || t[i] == '\n' || t[i] == '\r')) ++i; }
    bool eat(char c) { ws(); if (i < t.size() && t[i] == c) { ++i; return true; } return false; }
    bool peek(char c) { ws(); return i < t.size() && t[i] == c; }
    // string token -> raw contents (escapes are kept verbatim: no OMM field needs them decoded)
    bool string(std::string_view &out)
    {
        ws();
        if (i >= t.size() || t[i] != '"') return false;
        size_t a = ++i;
        while (i < t.size() && t[i] != '"') i += (t[i] == '\\' && i + 1 < t.size()) ? 2 : 1;
        if (i >= t.size()) return false;
        out = t.substr(a, i - a);
        ++i;
        return true;
    }
    // any value; strings and bare tokens (numbers, true, false, null) are returned as text
    bool value(std::string_view &out, bool &is_string)
    {
        ws();
        if (i >= t.size()) return false;
        if (t[i] == '"') { is_string = true; return string(out); }
        is_string = false;
        if (t[i] == '{' || t[i] == '[') { // nested container: skip it
            int depth = 0;
            size_t a = i;
            do {
                if (t[i] == '"') { std::string_view dummy; if (!string(dummy)) return false; continue; }
                if (t[i] == '{' || t[i] == '[') ++depth;
                if (t[i] == '}' || t[i] == ']') --depth;
                ++i;
            } while (i < t.size() && depth > 0);
            out = t.substr(a, i - a);
            return depth == 0;
        }
        size_t a = i;
        while (i < t.size() && t[i] != ',' && t[i] != '}' && t[i] != ']' && t[i] != ' ' && t[i] != '\n' && t[i] != '\r' && t[i] != '\t') ++i;
        out = t.substr(a, i - a);
        return !out.empty();
    }
};

int days_before_month(int year, int month)
{
    static const int cum[12] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334};
    const bool leap = (year % 4 == 0 && year % 100 != 0) || year % 400 == 0;
    return cum[month - 1] + ((leap && month > 2) ? 1 : 0);
}

// "YYYY-MM-DDThh:mm:ss[.ffffff][Z]" -> two-digit year, fractional day of year, JD (Tle.zig L198-238)
int parse_iso_epoch(std::string_view e, TleRecord &r)
{
    if (e.size() < 19) return -1;
    long year, month, day, hour, minute;
    double sec;
    if (!to_long(e.substr(0, 4), year) || !to_long(e.substr(5, 2), month) || !to_long(e.substr(8, 2), day) ||
        !to_long(e.substr(11, 2), hour) || !to_long(e.substr(14, 2), minute))
        return -1;
    std::string_view s = e.substr(17);
    if (!s.empty() && s.back() == 'Z') s.remove_suffix(1);
    if (!to_double(s, sec) || month < 1 || month > 12) return -1;
    const double doy = double(days_before_month(int(year), int(month)) + day) +
                       (double(hour) + (double(minute) + sec / 60.0) / 60.0) / 24.0;
    r.epoch_year = int(year % 100);
    r.epoch_day = doy;
    r.epoch_jd = year_doy_to_jd(int(year), doy);
    return 0;
}

int parse_omm_object(JsonCursor &c, TleRecord &r)
{
    if (!c.eat('{')) return -999;
    r = TleRecord{};
    enum { K_EPOCH = 1, K_MM = 2, K_ECC = 4, K_INC = 8, K_RAAN = 16, K_ARGP = 32, K_MA = 64, K_ID = 128, K_BSTAR = 256 };
    unsigned seen = 0;
    int rc = 0;
    if (!c.peek('}')) {
        do {
            std::string_view key, val;
            bool is_str = false;
            if (!c.string(key) || !c.eat(':') || !c.value(val, is_str)) return -999;
            const bool null = !is_str && val == "null";
            double d = 0.0;
            // numeric fields also arrive as JSON strings (Space-Track style: "MEAN_MOTION":"15.5"); the reference's
            // std.json accepts string tokens for f64 / u32 fields (src/Tle.zig parseOmm), so does this reader
            const bool num = !null && !val.empty() && to_double(val, d);
            auto want = [&](double &dst, unsigned bit) { if (!num) rc = -999; else { dst = d; seen |= bit; } };
            if (key == "EPOCH") { if (!is_str) rc = -999; else { int e = parse_iso_epoch(val, r); if (e) rc = e; seen |= K_EPOCH; } }
            else if (key == "MEAN_MOTION") want(r.mm_revday, K_MM);
            else if (key == "ECCENTRICITY") want(r.ecc, K_ECC);
            else if (key == "INCLINATION") want(r.incl_deg, K_INC);
            else if (key == "RA_OF_ASC_NODE") want(r.raan_deg, K_RAAN);
            else if (key == "ARG_OF_PERICENTER") want(r.argp_deg, K_ARGP);
            else if (key == "MEAN_ANOMALY") want(r.ma_deg, K_MA);
            else if (key == "BSTAR") want(r.bstar, K_BSTAR);
            else if (key == "NORAD_CAT_ID") { if (!num) rc = -999; else { r.satnum = uint32_t(d); seen |= K_ID; } }
            else if (key == "MEAN_MOTION_DOT") { if (num) r.ndot = d; }
            else if (key == "ELEMENT_SET_NO") { if (num) r.elnum = uint32_t(d); }
            else if (key == "REV_AT_EPOCH") { if (num) r.revnum = uint32_t(d); }
            else if (key == "CLASSIFICATION_TYPE") { if (is_str && !val.empty()) r.classification = val[0]; }
        } while (c.eat(','));
    }
    if (!c.eat('}')) return -999;
    if (rc != 0) return rc;
    const unsigned all = K_EPOCH | K_MM | K_ECC | K_INC | K_RAAN | K_ARGP | K_MA | K_ID | K_BSTAR;
    return (seen & all) == all ? 0 : -999;
}

} // namespace

int parse_omm_json(std::string_view text, std::vector<TleRecord> &out)
{
    JsonCursor c{text};
    TleRecord r;
    if (c.peek('{')) {
        int rc = parse_omm_object(c, r);
        if (rc == 0) out.push_back(r);
        return rc;
    }
    if (!c.eat('[')) return -999;
    if (c.eat(']')) return 0;
    do {
        int rc = parse_omm_object(c, r);
        if (rc != 0) return rc;
        out.push_back(r);
    } while (c.eat(','));
    return c.eat(']') ? 0 : -999;
}

} // namespace azh
