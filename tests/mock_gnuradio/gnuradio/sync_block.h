// TEST-ONLY stand-in for gnuradio/sync_block.h: the part of the block interface gr-lora's decoder uses
// (lib/decoder_impl.cc:49-52, :91, :120-121, :607-608, :902), recording what the block does so that a test can look at it.
#pragma once
#include <complex>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>

typedef std::complex<float> gr_complex;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace gr {
class basic_block { public: virtual ~basic_block() {} };   // (the binding names the three base types, decoder_python.cc:36)
class block : public basic_block {};
class sync_block : public block {
public:
    // what the mock runtime records
    std::string mock_name;
    io_signature::sptr mock_in, mock_out;
    int mock_output_multiple = 1;
    long long mock_consumed = 0;
    std::vector<std::string> mock_ports;
    std::vector<std::pair<std::string, pmt::pmt_t>> mock_published;

    virtual ~sync_block() {}
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
    virtual bool stop() { return true; }

protected:
    sync_block() {}
    sync_block(const std::string &name, io_signature::sptr in, io_signature::sptr out) : mock_name(name), mock_in(in), mock_out(out) {}
    void set_output_multiple(int m) { mock_output_multiple = m; }
    void message_port_register_out(pmt::pmt_t port) { mock_ports.push_back(pmt::symbol_to_string(port)); }
    void message_port_pub(pmt::pmt_t port, pmt::pmt_t msg) { mock_published.emplace_back(pmt::symbol_to_string(port), msg); }
    void consume_each(int n) { mock_consumed += n; }
};
} // namespace gr

namespace gnuradio {
template <class T>
std::shared_ptr<T> get_initial_sptr(T *p) { return std::shared_ptr<T>(p); }
} // namespace gnuradio
