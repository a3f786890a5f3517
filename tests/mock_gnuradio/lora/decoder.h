// TEST-ONLY stand-in for gr-lora's public block header (include/lora/decoder.h:38-715 of the reference): the same class
// shape and make() signature (:705), so that shim/gnuradio/decoder_impl.{h,cc} compile unchanged against it.
#pragma once
#include <cstdint>
#include <memory>

#include <gnuradio/sync_block.h>

namespace gr {
namespace lora {
class decoder : virtual public gr::sync_block {
public:
    typedef std::shared_ptr<decoder> sptr;
    static sptr make(float samp_rate, uint32_t bandwidth, uint8_t sf, bool implicit, uint8_t cr, bool crc, bool reduced_rate,
                     bool disable_drift_correction);
    virtual void set_sf(uint8_t sf) = 0;
    virtual void set_samp_rate(float samp_rate) = 0;
};
} // namespace lora
} // namespace gr
