// The compiled FSPEN kernel (models/fspen/model.py of the reference; configs/others/fspen.yaml: hop 256)
#include "fspen_kernels.hip.h"

extern "C" const fe::FImpl* fe_fimpl_h256() {
    static const fe::FImpl impl = fe::make_fimpl<fe::FShape<256>>();
    return &impl;
}
