// host_types.h -- which declarations the Ceres-free cost holders (and the adapter sources) are compiled against.
#pragma once
#ifdef PVIO_HOST_USE_REFERENCE_TYPES // inside the PVIO tree: the reference's own headers, unedited
#include <pvio/common.h>
#include <pvio/estimation/factor.h>
#include <pvio/estimation/state.h>
#include <pvio/map/frame.h>
#else // standalone: the look-alike declarations of tests/host/standin/pvio_min.h
#include "pvio_min.h"
#endif
