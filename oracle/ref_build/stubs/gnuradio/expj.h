/* Stand-in for <gnuradio/expj.h>.  GNU Radio 3.9 defines gr_expj(float phase) as sincosf(phase) packed
 * into a gr_complex(cos, sin) (gnuradio-runtime/include/gnuradio/expj.h); glibc's sincosf is used here. */
#ifndef REFSTUB_GNURADIO_EXPJ_H
#define REFSTUB_GNURADIO_EXPJ_H
#include <gnuradio/gr_complex.h>
#include <math.h>
static inline gr_complex gr_expj(float phase)
{
    float t_imag, t_real;
    ::sincosf(phase, &t_imag, &t_real);
    return gr_complex(t_real, t_imag);
}
#endif
