/* include/vpx/vpx_integer.h -- fixed-width integers for the vpx API (reference: vpx/vpx_integer.h). */
#ifndef VPX_INTEGER_H
#define VPX_INTEGER_H
#include <stddef.h>
#include <stdint.h>
#include <inttypes.h>
#endif
