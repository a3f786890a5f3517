/* Stand-in for <gnuradio/gr_complex.h>: gr_complex is std::complex<float> in GNU Radio as well. */
#ifndef REFSTUB_GNURADIO_GR_COMPLEX_H
#define REFSTUB_GNURADIO_GR_COMPLEX_H
#include <complex>
#include <string>
#include <vector>
typedef std::complex<float> gr_complex;
typedef std::complex<double> gr_complexd;
#endif
