/* decoder_impl.h -- GNU Radio side of the drop-in: replaces lib/decoder_impl.h of gr-lora.
 * The public block header include/lora/decoder.h (decoder::make with its 8 arguments, :705) stays the reference's own.
 * Everything the block did in work() (lib/decoder_impl.cc:740-903) happens behind the C ABI of lora_hip.h.             */
#ifndef INCLUDED_LORA_DECODER_IMPL_H
#define INCLUDED_LORA_DECODER_IMPL_H

#include <lora/decoder.h>
#include <lora_hip.h>

namespace gr {
namespace lora {

class decoder_impl : public decoder {
    lora_hip_decoder_t *d_h = nullptr;
    uint32_t d_sps = 0;
    void publish_frames();

public:
    decoder_impl(float samp_rate, uint32_t bandwidth, uint8_t sf, bool implicit, uint8_t cr, bool crc, bool reduced_rate,
                 bool disable_drift_correction);
    ~decoder_impl() override;
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) override;
    bool stop() override;
    void set_sf(uint8_t sf) override { lora_hip_set_sf(d_h, sf); }                 /* warn-only, as :905-909 */
    void set_samp_rate(float samp_rate) override { lora_hip_set_samp_rate(d_h, samp_rate); } /* :911-915 */
};

} // namespace lora
} // namespace gr
#endif
