/* Stand-in for <gnuradio/io_signature.h>: records the stream counts / item size the block declares. */
#ifndef REFSTUB_GNURADIO_IO_SIGNATURE_H
#define REFSTUB_GNURADIO_IO_SIGNATURE_H
#include <memory>
#include <cstddef>
#include <iostream>
#include <vector>
#include <string>
namespace gr {
class io_signature
{
public:
    typedef std::shared_ptr<io_signature> sptr;
    int d_min, d_max, d_item;
    io_signature(int mn, int mx, int item) : d_min(mn), d_max(mx), d_item(item) {}
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item)
    {
        return std::make_shared<io_signature>(min_streams, max_streams, sizeof_stream_item);
    }
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    int sizeof_stream_item(int) const { return d_item; }
};
} // namespace gr
#endif
