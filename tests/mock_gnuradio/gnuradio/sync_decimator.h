// TEST-ONLY stand-in for gnuradio/sync_decimator.h (noutput_items outputs from noutput_items * decimation inputs).
#pragma once
#include <gnuradio/sync_block.h>
namespace gr {
class sync_decimator : public sync_block {
    unsigned d_decimation = 1;
public:
    unsigned decimation() const { return d_decimation; }
protected:
    sync_decimator() {}
    sync_decimator(const std::string &name, io_signature::sptr in, io_signature::sptr out, unsigned decimation)
        : sync_block(name, in, out), d_decimation(decimation) {}
};
} // namespace gr
