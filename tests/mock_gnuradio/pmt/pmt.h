// TEST-ONLY stand-in for GNU Radio's pmt: just what a block needs to name a port and publish a blob.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
namespace pmt {
struct pmt_base { std::string symbol; std::vector<uint8_t> blob; bool is_blob = false; };
typedef std::shared_ptr<pmt_base> pmt_t;
inline pmt_t mp(const char *s) { auto p = std::make_shared<pmt_base>(); p->symbol = s; return p; }
inline pmt_t make_blob(const void *data, size_t len)
{
    auto p = std::make_shared<pmt_base>();
    p->is_blob = true;
    p->blob.assign(static_cast<const uint8_t *>(data), static_cast<const uint8_t *>(data) + len);
    return p;
}
inline const void *blob_data(const pmt_t &p) { return p->blob.data(); }
inline size_t blob_length(const pmt_t &p) { return p->blob.size(); }
inline std::string symbol_to_string(const pmt_t &p) { return p->symbol; }
} // namespace pmt
