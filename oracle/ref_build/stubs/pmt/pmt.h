/* Stand-in for <pmt/pmt.h>: just enough PMT for decoder_impl.cc - interned port names and u8 blobs. */
#ifndef REFSTUB_PMT_H
#define REFSTUB_PMT_H
#include <cstdint>
#include <cstddef>
#include <memory>
#include <string>
#include <vector>
namespace pmt {
struct pmt_base {
    std::string sym;
    std::vector<uint8_t> blob;
    bool is_blob = false;
};
typedef std::shared_ptr<pmt_base> pmt_t;
static inline pmt_t mp(const std::string& s)
{
    pmt_t p = std::make_shared<pmt_base>();
    p->sym = s;
    return p;
}
static inline pmt_t intern(const std::string& s) { return mp(s); }
static inline pmt_t make_blob(const void* buf, size_t len)
{
    pmt_t p = std::make_shared<pmt_base>();
    p->is_blob = true;
    p->blob.assign((const uint8_t*)buf, (const uint8_t*)buf + len);
    return p;
}
static inline const void* blob_data(pmt_t p) { return p->blob.data(); }
static inline size_t blob_length(pmt_t p) { return p->blob.size(); }
static inline std::string symbol_to_string(pmt_t p) { return p->sym; }
} // namespace pmt
#endif
