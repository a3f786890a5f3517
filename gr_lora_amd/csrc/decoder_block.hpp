// decoder_block.hpp -- C++ host-side mirror of gr::lora::decoder on top of the C ABI.
//
// Same factory signature and public methods as the reference block
// (include/lora/decoder.h:693-709), minus GNU Radio: work() takes the input
// buffer the scheduler would hand to decoder_impl::work (lib/decoder_impl.cc:740)
// and frames are delivered to subscribers of the "frames" port (:120, :607-608).
// A real gr::sync_block shim is this class with the two GNU Radio types swapped
// in -- see INTEGRATION.md.
#pragma once
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lora_hip.h"

namespace lora_hip {

class decoder {
public:
    typedef std::shared_ptr<decoder> sptr;
    using frame_handler = std::function<void(const std::vector<uint8_t> &)>;

    // decoder::make(samp_rate, bandwidth, sf, implicit, cr, crc, reduced_rate, disable_drift_correction)
    static sptr make(float samp_rate, uint32_t bandwidth, uint8_t sf, bool implicit, uint8_t cr, bool crc,
                     bool reduced_rate, bool disable_drift_correction, int device = 0,
                     int demod = LORA_HIP_DEMOD_FFT_COMPAT)
    {
        return sptr(new decoder(samp_rate, bandwidth, sf, implicit, cr, crc, reduced_rate, disable_drift_correction, device, demod));
    }

    ~decoder() { lora_hip_destroy(d_h); }

    // set_output_multiple(2 * samples_per_symbol) (decoder_impl.cc:91)
    uint32_t output_multiple() const { return 2u * d_sps; }

    // One scheduler call: consumes all noutput_items items of input 0, returns 0 like the reference (:902).
    int work(int noutput_items, const std::complex<float> *input)
    {
        size_t consumed = 0;
        check(lora_hip_work(d_h, reinterpret_cast<const float *>(input), (size_t)noutput_items, &consumed));
        publish();
        return 0;
    }

    // gr::block::stop(): decode what is still buffered
    bool stop()
    {
        check(lora_hip_flush(d_h));
        publish();
        return true;
    }

    void set_sf(uint8_t sf) { lora_hip_set_sf(d_h, sf); }                        // warn only (:905-909)
    void set_samp_rate(float samp_rate) { lora_hip_set_samp_rate(d_h, samp_rate); } // warn only (:911-915)

    // message_port "frames": each subscriber receives the blob the reference publishes (:588-609)
    void subscribe_frames(frame_handler fn) { d_subs.push_back(std::move(fn)); }

private:
    decoder(float samp_rate, uint32_t bandwidth, uint8_t sf, bool implicit, uint8_t cr, bool crc, bool reduced_rate,
            bool disable_drift_correction, int device, int demod)
    {
        lora_hip_config_t c{};
        c.struct_size = sizeof c;
        c.samp_rate = samp_rate; c.bandwidth = bandwidth; c.sf = sf; c.implicit = implicit; c.cr = cr; c.crc = crc;
        c.reduced_rate = reduced_rate; c.disable_drift_correction = disable_drift_correction;
        c.device = device; c.demod = demod;
        const lora_hip_status s = lora_hip_create(&c, &d_h);
        if (s != LORA_HIP_OK) { // the reference exit(1)s on a bad configuration (:57-61)
            std::fprintf(stderr, "[LoRa Decoder] ERROR : %s (%s)\n", lora_hip_strerror(s), lora_hip_last_error(nullptr));
            std::exit(1);
        }
        lora_hip_get_geometry(d_h, &d_sps, nullptr, nullptr);
    }

    void check(lora_hip_status s)
    {
        if (s != LORA_HIP_OK) {
            std::fprintf(stderr, "[LoRa Decoder] ERROR : %s (%s)\n", lora_hip_strerror(s), lora_hip_last_error(d_h));
            std::exit(1);
        }
    }

    void publish()
    {
        uint8_t buf[320];
        size_t len = 0;
        while (lora_hip_frames_available(d_h) > 0) {
            check(lora_hip_poll_frame(d_h, buf, sizeof buf, &len, nullptr));
            if (len == 0) break;
            const std::vector<uint8_t> blob(buf, buf + len);
            for (auto &fn : d_subs) fn(blob);
        }
    }

    lora_hip_decoder_t *d_h = nullptr;
    uint32_t d_sps = 0;
    std::vector<frame_handler> d_subs;
};

} // namespace lora_hip
