// seq_bootstrap.h -- TEST INFRASTRUCTURE: the supplied body poses gt_initializer.cpp bootstraps the first window from (set by seq_capi.cpp).
#pragma once
#include <pvio/pvio.h>

#include <vector>

namespace pvio {
struct SeqTimedPose {
    double t;
    quaternion q;
    vector<3> p;
};
std::vector<SeqTimedPose> &seq_bootstrap();
} // namespace pvio
