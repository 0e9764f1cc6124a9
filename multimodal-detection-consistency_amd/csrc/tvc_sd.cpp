// Latent-diffusion reference generator behind the C-ABI (filled in below).
#include "handle.hpp"

struct SdState { int unused = 0; };

void tvc_sd_free(tvc_handle* h) {
    if (h && h->sd) { delete h->sd; h->sd = nullptr; }
}
