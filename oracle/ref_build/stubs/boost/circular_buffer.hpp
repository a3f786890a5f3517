/* Stand-in for <boost/circular_buffer.hpp>: fixed-capacity ring with the members decoder_impl uses
 * (capacity constructor, push_back overwriting the oldest element when full, size, operator[] from the oldest). */
#ifndef REFSTUB_BOOST_CIRCULAR_BUFFER_HPP
#define REFSTUB_BOOST_CIRCULAR_BUFFER_HPP
#include <cstddef>
#include <vector>
namespace boost {
template <class T>
class circular_buffer
{
    std::vector<T> d_buf;
    size_t d_cap, d_head = 0, d_size = 0;

public:
    explicit circular_buffer(size_t capacity = 0) : d_buf(capacity), d_cap(capacity) {}
    void push_back(const T& v)
    {
        if (d_cap == 0)
            return;
        if (d_size < d_cap) {
            d_buf[(d_head + d_size) % d_cap] = v;
            d_size++;
        } else {
            d_buf[d_head] = v;
            d_head = (d_head + 1) % d_cap;
        }
    }
    size_t size() const { return d_size; }
    size_t capacity() const { return d_cap; }
    bool empty() const { return d_size == 0; }
    void clear() { d_head = d_size = 0; }
    T& operator[](size_t i) { return d_buf[(d_head + i) % d_cap]; }
    const T& operator[](size_t i) const { return d_buf[(d_head + i) % d_cap]; }
    T& front() { return (*this)[0]; }
    T& back() { return (*this)[d_size - 1]; }
};
} // namespace boost
#endif
