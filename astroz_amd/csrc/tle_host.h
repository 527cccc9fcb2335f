// tle_host.h -- host-side TLE text ingest for libastroz_hip.so (text -> raw element columns).
// Follows the reference's fixed-column reader (src/Tle.zig L49-101), its multi-TLE iterator
// (L103-132) and epoch conversion (L298-304, src/Datetime.zig L222-231).  No checksum
// verification, exactly like the reference.  Only text handling lives on the host; every derived
// quantity is computed by the init kernel.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <string_view>
#include <vector>

namespace azh {

struct TleRecord {
    uint32_t satnum = 0;
    char classification = 'U';
    int epoch_year = 0;   // two digits, as in the TLE
    double epoch_day = 0; // fractional day of year
    double epoch_jd = 0;
    double ndot = 0;      // TLE units
    double bstar = 0;
    double incl_deg = 0, raan_deg = 0, ecc = 0, argp_deg = 0, ma_deg = 0, mm_revday = 0;
    uint32_t elnum = 0, revnum = 0;
};

// 0 on success, -1 (bad length) or -999 (unparsable field)
int parse_lines(std::string_view line1, std::string_view line2, TleRecord &out);
// first two lines with >= 69 significant characters (Tle.parse, L32-47)
int parse_first(std::string_view text, TleRecord &out);
// every '1 ...' line followed by a '2 ...' line (MultiIterator); unparsable pairs are skipped
void parse_all(std::string_view text, std::vector<TleRecord> &out);
// threads parse_all may use on text beyond a few hundred KB (0 = automatic: the host's cores, at most 16; 1 = serial)
void set_parse_threads(unsigned n);
// threads worth using on `bytes` of text under that setting, and a helper: fn(begin, end) over [0, n) cut into `threads` ranges
unsigned parse_threads_for(size_t bytes);
void parallel_ranges(size_t n, unsigned threads, const std::function<void(size_t, size_t)> &fn);

double year_doy_to_jd(int full_year, double doy);

// OMM (CCSDS Orbit Mean-elements Message) in its JSON form, one object or an array of objects, as
// CelesTrak serves it: Tle.parseOmm / parseOmmArray (src/Tle.zig L134-238).  Elements keep their full
// JSON precision (no detour through 69-column text); unknown keys are ignored, the optional keys default
// as in the reference's OmmRecord (L164-182).  Returns 0, -1 (EPOCH shorter than 19 characters /
// malformed record) or -999 (not JSON of that shape, missing mandatory key).
int parse_omm_json(std::string_view text, std::vector<TleRecord> &out);

} // namespace azh
