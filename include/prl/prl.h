// prl.h (include/prl) - the C++ host layer's declarations for callers that use -I include/prl: one set of declarations,
// kept beside its implementation (prlib_amd/csrc/prl/prl.h, prl_host.cpp).
#pragma once
#include "../../prlib_amd/csrc/prl/prl.h"
