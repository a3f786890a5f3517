/* Stand-in for <gnuradio/sync_block.h> (+ basic_block / block): the slice of the GNU Radio block contract
 * that lib/decoder_impl.cc touches - constructor (name, in, out), set_output_multiple, consume_each,
 * message_port_register_out / message_port_pub, work() - recording what the block does so that a driver
 * standing in for the scheduler can read it back.  TEST INFRASTRUCTURE (oracle/ref_build). */
#ifndef REFSTUB_GNURADIO_SYNC_BLOCK_H
#define REFSTUB_GNURADIO_SYNC_BLOCK_H
#include <gnuradio/gr_complex.h>
#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;
typedef std::vector<int> gr_vector_int;

namespace gr {
class sync_block
{
public:
    std::string stub_name;
    io_signature::sptr stub_in, stub_out;
    int stub_output_multiple = 1;
    long long stub_consumed = -1; /* consume_each() of the current work() call; -1 = not called */
    std::vector<std::string> stub_ports;
    std::vector<std::pair<std::string, pmt::pmt_t>> stub_published;

    sync_block() {}
    sync_block(const std::string& name, io_signature::sptr in, io_signature::sptr out)
        : stub_name(name), stub_in(in), stub_out(out)
    {
    }
    virtual ~sync_block() {}

    void set_output_multiple(int m) { stub_output_multiple = m; }
    int output_multiple() const { return stub_output_multiple; }
    void consume_each(int n) { stub_consumed = n; }
    void message_port_register_out(pmt::pmt_t id) { stub_ports.push_back(id->sym); }
    void message_port_pub(pmt::pmt_t id, pmt::pmt_t msg) { stub_published.emplace_back(id->sym, msg); }
    std::string name() const { return stub_name; }

    virtual int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) = 0;
};
} // namespace gr

namespace gnuradio {
template <class T>
std::shared_ptr<T> get_initial_sptr(T* p)
{
    return std::shared_ptr<T>(p);
}
} // namespace gnuradio
#endif
