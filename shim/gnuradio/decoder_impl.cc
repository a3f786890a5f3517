/* decoder_impl.cc -- replaces lib/decoder_impl.cc of gr-lora: the block keeps its name, ports and make() signature;
 * the receive chain runs on the MI355X behind lora_hip.h.  Build: add this repository's include/ to the include path
 * and link liblora_hip.so instead of liquid (lib/CMakeLists.txt:42).                                                  */
#include "decoder_impl.h"

#include <gnuradio/io_signature.h>

#include <cstdlib>
#include <iostream>
#include <string>

namespace gr {
namespace lora {

decoder::sptr decoder::make(float samp_rate, uint32_t bandwidth, uint8_t sf, bool implicit, uint8_t cr, bool crc, bool reduced_rate,
                            bool disable_drift_correction)
{
    return gnuradio::get_initial_sptr(new decoder_impl(samp_rate, bandwidth, sf, implicit, cr, crc, reduced_rate, disable_drift_correction));
}

decoder_impl::decoder_impl(float samp_rate, uint32_t bandwidth, uint8_t sf, bool implicit, uint8_t cr, bool crc, bool reduced_rate,
                           bool disable_drift_correction)
    : gr::sync_block("decoder", gr::io_signature::make(1, -1, sizeof(gr_complex)), gr::io_signature::make(0, 0, 0))
{
    lora_hip_config_t c{};
    c.struct_size = sizeof c;
    c.samp_rate = samp_rate; c.bandwidth = bandwidth; c.sf = sf; c.implicit = implicit; c.cr = cr; c.crc = crc;
    c.reduced_rate = reduced_rate; c.disable_drift_correction = disable_drift_correction;
    /* make() has no argument for these (its signature is the reference's): process-wide settings from the environment */
    c.device = 0;
    if (const char *e = std::getenv("LORA_HIP_DEVICE")) c.device = std::atoi(e);
    c.demod = LORA_HIP_DEMOD_FFT_COMPAT; /* LORA_HIP_DEMOD_GRAD selects the upstream default estimator */
    if (const char *e = std::getenv("LORA_HIP_DEMOD")) {
        const std::string v(e);
        if (v == "grad") c.demod = LORA_HIP_DEMOD_GRAD;
        else if (v == "fft") c.demod = LORA_HIP_DEMOD_FFT;
        else if (v != "fft_compat") {
            std::cerr << "[LoRa Decoder] ERROR : LORA_HIP_DEMOD must be one of grad, fft, fft_compat" << std::endl;
            exit(1);
        }
    }
    const lora_hip_status s = lora_hip_create(&c, &d_h);
    if (s != LORA_HIP_OK) { /* the reference prints and exit(1)s on a bad configuration (decoder_impl.cc:57-61) */
        std::cerr << "[LoRa Decoder] ERROR : " << lora_hip_strerror(s) << ": " << lora_hip_last_error(nullptr) << std::endl;
        exit(1);
    }
    /* when a frame reaches the `frames` port: upstream inside the work() call that completes the packet (:870-881); here within
     * this many milliseconds (+ one call period) of its last sample - the library decodes in passes (default 50 ms) */
    if (const char *e = std::getenv("LORA_HIP_LATENCY_MS")) lora_hip_set_stream_latency(d_h, (float)std::atof(e));
    uint32_t bins = 0, decim = 0;
    lora_hip_get_geometry(d_h, &d_sps, &bins, &decim);
    std::cout << "Bins per symbol: \t" << bins << std::endl;    /* the constructor's banner, :94-96 */
    std::cout << "Samples per symbol: \t" << d_sps << std::endl;
    std::cout << "Decimation: \t\t" << decim << std::endl;
    set_output_multiple(2 * (int)d_sps);                        /* :91 */
    message_port_register_out(pmt::mp("frames"));               /* :120 */
    message_port_register_out(pmt::mp("control"));              /* :121 */
}

decoder_impl::~decoder_impl() { lora_hip_destroy(d_h); }

int decoder_impl::work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
{
    size_t consumed = 0;
    if (lora_hip_work(d_h, static_cast<const float *>(input_items[0]), (size_t)noutput_items, &consumed) != LORA_HIP_OK) {
        std::cerr << "[LoRa Decoder] ERROR : " << lora_hip_last_error(d_h) << std::endl;
        exit(1);
    }
    publish_frames();
    consume_each(static_cast<int>(consumed)); /* the reference also consumes by hand and returns 0 (:902) */
    return 0;
}

bool decoder_impl::stop()
{
    lora_hip_flush(d_h); /* the library batches: what is still buffered is decoded now */
    publish_frames();
    return true;
}

void decoder_impl::publish_frames()
{
    uint8_t buf[320];
    size_t len = 0;
    while (lora_hip_frames_available(d_h) > 0 && lora_hip_poll_frame(d_h, buf, sizeof buf, &len, nullptr) == LORA_HIP_OK && len > 0)
        message_port_pub(pmt::mp("frames"), pmt::make_blob(buf, len)); /* msg_lora_frame, :607-608 */
}

} // namespace lora
} // namespace gr
