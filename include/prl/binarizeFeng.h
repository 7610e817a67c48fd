// binarizeFeng.h - drop-in for PRLib's header of the same name (src/binarizations/binarizeFeng.h:46-53): declares prl::binarizeFeng with the
// reference's signature, defaults and CV_EXPORTS linkage.  A caller that includes "binarizeFeng.h" (as
// samples/binarizations/binarizeSauvola_sample.cpp:25 does) builds against this repository with only its include path
// changed to include/prl; the declarations themselves live in prl.h.
#ifndef PRLIB_HIP_DROPIN_binarizeFeng_h
#define PRLIB_HIP_DROPIN_binarizeFeng_h
#include "prl.h"
#endif  // PRLIB_HIP_DROPIN_binarizeFeng_h
