// TEST-ONLY stand-in for gnuradio/io_signature.h.
#pragma once
#include <memory>
namespace gr {
class io_signature {
public:
    typedef std::shared_ptr<io_signature> sptr;
    int min_streams, max_streams, item_size;
    static sptr make(int min_streams, int max_streams, int item_size)
    {
        auto p = std::make_shared<io_signature>();
        p->min_streams = min_streams; p->max_streams = max_streams; p->item_size = item_size;
        return p;
    }
};
} // namespace gr
