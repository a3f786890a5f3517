/* Stand-in for <gnuradio/attributes.h> (TEST INFRASTRUCTURE: lets oracle/ref_build compile the reference's
 * lib/decoder_impl.cc unmodified without GNU Radio).  Only the two visibility macros include/lora/api.h uses. */
#ifndef REFSTUB_GNURADIO_ATTRIBUTES_H
#define REFSTUB_GNURADIO_ATTRIBUTES_H
#define __GR_ATTR_EXPORT __attribute__((visibility("default")))
#define __GR_ATTR_IMPORT __attribute__((visibility("default")))
#endif
