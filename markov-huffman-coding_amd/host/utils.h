// utils.h — forwarding header: the reference's callers include "utils.h" (src/main.cpp:12) for `null`,
// `eprintf`, charv, check_access, read_buffer and write_buffer (src/utils.h:7-21).  Here all of them come
// with the coding.h face; this file only adds the `null` macro the reference's main() uses.
#ifndef MHC_HOST_UTILS_H
#define MHC_HOST_UTILS_H

#include "coding.h"

#ifndef null
#define null 0   // src/utils.h:7
#endif

#endif
