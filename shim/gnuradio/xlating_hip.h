/* xlating_hip.h -- the block gr-lora's channelizer hier block wraps (lib/channelizer_impl.cc:46-48 builds a
 * gr::filter::freq_xlating_fir_filter_ccf there), on the MI355X: same decimation, same firdes::low_pass taps, same
 * translation frequency, apply_cfo as channelizer_impl::apply_cfo (:68-71).  channelizer_impl keeps its hier_block2 shell
 * and connects self() -> xlating_hip -> self() exactly as it connects the GNU Radio filter (:52-53).                  */
#ifndef INCLUDED_LORA_XLATING_HIP_H
#define INCLUDED_LORA_XLATING_HIP_H

#include <gnuradio/sync_decimator.h>
#include <lora_hip_channelizer.h>

#include <memory>
#include <vector>

namespace gr {
namespace lora {

class xlating_hip : public gr::sync_decimator {
    lora_hip_channelizer_t *d_h = nullptr;

public:
    typedef std::shared_ptr<xlating_hip> sptr;
    static sptr make(float samp_rate, float center_freq, const std::vector<float> &channel_list, uint32_t bandwidth, uint32_t decimation);
    xlating_hip(float samp_rate, float center_freq, const std::vector<float> &channel_list, uint32_t bandwidth, uint32_t decimation);
    ~xlating_hip() override;
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) override;
    void apply_cfo(float cfo) { lora_hip_channelizer_apply_cfo(d_h, cfo); }
    std::vector<float> taps() const;
};

} // namespace lora
} // namespace gr
#endif
