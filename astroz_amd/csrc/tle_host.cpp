// tle_host.cpp -- see tle_host.h
#include "tle_host.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

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

bool to_double(std::string_view f, double &v)
{
    if (f.empty() || f.size() >= 40) return false;
    char buf[40];
    memcpy(buf, f.data(), f.size());
    buf[f.size()] = 0;
    char *end = nullptr;
    v = strtod(buf, &end);
    return end == buf + f.size();
}

bool to_long(std::string_view f, long &v)
{
    if (f.empty() || f.size() >= 24) return false;
    char buf[24];
    memcpy(buf, f.data(), f.size());
    buf[f.size()] = 0;
    char *end = nullptr;
    v = strtol(buf, &end, 10);
    return end == buf + f.size();
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

void parse_all(std::string_view text, std::vector<TleRecord> &out)
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

} // namespace azh
