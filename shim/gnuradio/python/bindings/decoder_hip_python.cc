/* decoder_hip_python.cc -- Python binding of the drop-in decoder block.
 *
 * Takes the place of python/bindings/decoder_python.cc in a gr-lora tree built against liblora_hip.so: it supplies the
 * `bind_decoder(py::module &)` that python/bindings/python_bindings.cc:45 calls from PYBIND11_MODULE(lora_python, m),
 * so `lora.decoder(samp_rate, bandwidth, sf, implicit, cr, crc, reduced_rate, disable_drift_correction)` keeps the
 * argument names, order and defaults-free signature of the reference binding (decoder_python.cc:36-66), and
 * python/lora_receiver.py:55 constructs it unchanged.  The class registered here is the PUBLIC block type
 * gr::lora::decoder (include/lora/decoder.h:705); the object behind it is shim/gnuradio/decoder_impl.cc, i.e. the HIP
 * decoder behind the C ABI of include/lora_hip.h.
 *
 * The docstrings live here (the reference generates decoder_pydoc.h from Doxygen at build time; nothing is generated
 * for this file).  The base classes must already be registered when bind_decoder runs: python_bindings.cc:54 imports
 * gnuradio.gr first; tests/test_gnuradio_pybind.py registers the stand-ins of tests/mock_gnuradio/.               */
#include <pybind11/pybind11.h>

#include <lora/decoder.h>

namespace py = pybind11;

namespace {
const char *const kDocClass =
    "LoRa PHY decoder block: complex baseband in, decoded frames (loratap header + PHY header + payload) as PMT blobs "
    "on the message port \"frames\".  Decoding runs on an AMD Instinct GPU through liblora_hip.so.";
const char *const kDocMake =
    "decoder(samp_rate, bandwidth, sf, implicit, cr, crc, reduced_rate, disable_drift_correction)\n\n"
    "samp_rate: input rate in samples/s (a multiple of bandwidth); bandwidth: LoRa bandwidth in Hz; sf: spreading factor "
    "6..12; implicit: no PHY header on air, use cr / crc below; cr: coding rate 1..4 (4/5..4/8); crc: payload carries a "
    "CRC; reduced_rate: low data rate optimisation for every symbol; disable_drift_correction: keep the symbol clock "
    "fixed after the SFD.  Unsupported combinations end the process with the reference's message and exit(1).";
const char *const kDocSetSf = "Kept for interface compatibility: warns that the spreading factor cannot be changed at run time.";
const char *const kDocSetRate = "Kept for interface compatibility: warns that the sample rate cannot be changed at run time.";
} // namespace

void bind_decoder(py::module &m)
{
    using gr::lora::decoder;
    py::class_<decoder, gr::sync_block, gr::block, gr::basic_block, std::shared_ptr<decoder>> cls(m, "decoder", kDocClass);
    cls.def(py::init(&decoder::make), py::arg("samp_rate"), py::arg("bandwidth"), py::arg("sf"), py::arg("implicit"), py::arg("cr"),
            py::arg("crc"), py::arg("reduced_rate"), py::arg("disable_drift_correction"), kDocMake);
    cls.def("set_sf", &decoder::set_sf, py::arg("sf"), kDocSetSf);
    cls.def("set_samp_rate", &decoder::set_samp_rate, py::arg("samp_rate"), kDocSetRate);
}
