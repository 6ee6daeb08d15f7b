// host_seam.h -- the declarations every adapter source under pvio_amd/host/ is written against.
//
// Inside the PVIO tree (-DPVIO_HOST_USE_REFERENCE_TYPES; also `make -C tests/host refcheck`): the reference's own headers,
// unedited, plus the three Ceres-free cost holders of pvio_amd/host/dropin/ that take the place of
// pvio/src/pvio/estimation/ceres/{marginalization,preintegration,reprojection}_error_cost.h (-I .../dropin comes first).
// Standalone (tests only): tests/host/standin/pvio_min.h -- look-alike declarations with the same names and signatures, on the
// include path of tests/host/Makefile; nothing of it is part of what a PVIO maintainer links -- and the same three holders.
#pragma once
#ifdef PVIO_HOST_USE_REFERENCE_TYPES
#include <pvio/common.h>
#include <pvio/core/plane_extractor.h>
#include <pvio/estimation/bundle_adjustor.h>
#include <pvio/estimation/factor.h>
#include <pvio/estimation/pnp.h>
#include <pvio/estimation/state.h>
#include <pvio/map/frame.h>
#include <pvio/map/map.h>
#include <pvio/map/plane.h>
#include <pvio/map/track.h>
#include <pvio/forensics.h>
#include <pvio/utility/unique_timer.h>
#else
#include "pvio_min.h"
#endif
#include "dropin/pvio/estimation/ceres/marginalization_error_cost.h"
#include "dropin/pvio/estimation/ceres/preintegration_error_cost.h"
#include "dropin/pvio/estimation/ceres/reprojection_error_cost.h"
