// klt.hip -- placeholder until the KLT kernels land (see DESIGN.md roadmap); every entry point fails loudly.
#include "klt.h"

#include "../../include/pvio_hip.h"

namespace pvklt {
struct Image {
    int w, h;
};
Klt::Klt(int device) : device_(device) {}
Klt::~Klt() {}
int Klt::create_image(const uint8_t *, int, int, int, bool, Image **) {
    err_ = "KLT not built yet";
    return PVIO_ERR_UNSUPPORTED;
}
void Klt::release_image(Image *) {}
int Klt::download_level(const Image *, int, uint8_t *, int16_t *, int32_t *, int32_t *) { return PVIO_ERR_UNSUPPORTED; }
int Klt::track(const Image *, const Image *, int, const float *, float *, uint8_t *) { return PVIO_ERR_UNSUPPORTED; }
} // namespace pvklt
