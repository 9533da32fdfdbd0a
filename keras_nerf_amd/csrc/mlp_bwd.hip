#include "kernels.h"
namespace knerf { hipError_t launch_mlp_bwd(const BwdArgs&, hipStream_t) { return hipErrorNotSupported; } }
