// deskew.h - drop-in for PRLib's header of the same name (src/deskew/deskew.h:42,52,62): declares prl::deskew,
// prl::findOrientation and prl::findAngle with the reference's signatures and CV_EXPORTS linkage.  A caller that includes "deskew.h" (as
// samples/binarizations/binarizeSauvola_sample.cpp:25 does) builds against this repository with only its include path
// changed to include/prl; the declarations themselves live in prl.h.
#ifndef PRLIB_HIP_DROPIN_deskew_h
#define PRLIB_HIP_DROPIN_deskew_h
#include "prl.h"
#endif  // PRLIB_HIP_DROPIN_deskew_h
