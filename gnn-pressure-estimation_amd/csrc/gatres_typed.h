// The typed per-op launchers (storage type argument) are part of the C-ABI: see include/gatres.h, "Typed per-op kernels".
#pragma once
#include "gatres_common.h"
