// The compiled LiSenNet kernel (models/lisennet/model.py of the reference; configs/others/lisennet.yaml: hop 256)
#include "lisennet_kernels.hip.h"

extern "C" const fe::LImpl* fe_limpl_h256() {
    static const fe::LImpl impl = fe::make_limpl<fe::LShape<256>>();
    return &impl;
}
