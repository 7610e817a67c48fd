// thinGuoHall.h - drop-in for PRLib's header of the same name (src/thinning/thinGuoHall.h): declares prl::thinGuoHall with the
// reference's signature, defaults and CV_EXPORTS linkage.  A caller that includes "thinGuoHall.h" (as
// samples/binarizations/binarizeSauvola_sample.cpp:25 does) builds against this repository with only its include path
// changed to include/prl; the declarations themselves live in prl.h.
#ifndef PRLIB_HIP_DROPIN_thinGuoHall_h
#define PRLIB_HIP_DROPIN_thinGuoHall_h
#include "prl.h"
#endif  // PRLIB_HIP_DROPIN_thinGuoHall_h
