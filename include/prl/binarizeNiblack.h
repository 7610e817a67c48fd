// binarizeNiblack.h - drop-in for PRLib's header of the same name (src/binarizations/binarizeNiblack.h:43-47): declares prl::binarizeNiblack with the
// reference's signature, defaults and CV_EXPORTS linkage.  A caller that includes "binarizeNiblack.h" (as
// samples/binarizations/binarizeSauvola_sample.cpp:25 does) builds against this repository with only its include path
// changed to include/prl; the declarations themselves live in prl.h.
#ifndef PRLIB_HIP_DROPIN_binarizeNiblack_h
#define PRLIB_HIP_DROPIN_binarizeNiblack_h
#include "prl.h"
#endif  // PRLIB_HIP_DROPIN_binarizeNiblack_h
