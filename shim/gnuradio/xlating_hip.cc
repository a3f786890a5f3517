#include "xlating_hip.h"

#include <gnuradio/io_signature.h>

#include <cstdlib>
#include <iostream>

namespace gr {
namespace lora {

xlating_hip::sptr xlating_hip::make(float samp_rate, float center_freq, const std::vector<float> &channel_list, uint32_t bandwidth,
                                    uint32_t decimation)
{
    return gnuradio::get_initial_sptr(new xlating_hip(samp_rate, center_freq, channel_list, bandwidth, decimation));
}

xlating_hip::xlating_hip(float samp_rate, float center_freq, const std::vector<float> &channel_list, uint32_t bandwidth, uint32_t decimation)
    : gr::sync_decimator("xlating_hip", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(gr_complex)),
                         decimation)
{
    lora_hip_channelizer_config_t c{};
    c.struct_size = sizeof c;
    c.samp_rate = samp_rate; c.center_freq = center_freq;
    c.channel_list = channel_list.data();
    c.n_channels = 1; /* the reference translates channel_list[0] only (channelizer_impl.cc:45) */
    c.bandwidth = bandwidth; c.decimation = decimation; c.device = 0;
    if (channel_list.empty() || lora_hip_channelizer_create(&c, &d_h) != LORA_HIP_OK) {
        std::cerr << "[LoRa Channelizer] ERROR : " << lora_hip_channelizer_last_error(nullptr) << std::endl;
        exit(1);
    }
}

xlating_hip::~xlating_hip() { lora_hip_channelizer_destroy(d_h); }

int xlating_hip::work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
{
    size_t n_out = 0; /* no set_history: the library keeps the filter's delay line across calls itself */
    if (lora_hip_channelizer_work(d_h, static_cast<const float *>(input_items[0]), (size_t)noutput_items * decimation(),
                                  static_cast<float *>(output_items[0]), (size_t)noutput_items, &n_out) != LORA_HIP_OK) {
        std::cerr << "[LoRa Channelizer] ERROR : " << lora_hip_channelizer_last_error(d_h) << std::endl;
        exit(1);
    }
    return (int)n_out;
}

std::vector<float> xlating_hip::taps() const
{
    size_t n = 0;
    lora_hip_channelizer_taps(d_h, nullptr, 0, &n);
    std::vector<float> t(n);
    lora_hip_channelizer_taps(d_h, t.data(), t.size(), &n);
    return t;
}

} // namespace lora
} // namespace gr
