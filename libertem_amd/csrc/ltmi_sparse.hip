// CSR mask stacks (placeholder until the SELL kernel lands)
#include "ltmi_common.h"
struct ltmi_masks;
namespace ltmi {
int csr_destroy(ltmi_masks *) { return LTMI_OK; }
int csr_apply(ltmi_masks *, const void *, int, int64_t, int64_t, void *, int64_t, int,
              hipStream_t) {
    LTMI_FAIL(LTMI_E_INVALID, "sparse path not built yet");
}
}  // namespace ltmi
extern "C" int ltmi_masks_create_csr(int, const int64_t *, const int64_t *, const void *, int,
                                     int64_t, int64_t, ltmi_masks **) {
    LTMI_FAIL(LTMI_E_INVALID, "sparse path not built yet");
}
