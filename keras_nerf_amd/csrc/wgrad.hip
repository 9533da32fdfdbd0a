#include "kernels.h"
namespace knerf { hipError_t launch_wgrad(const WgradArgs&, hipStream_t) { return hipErrorNotSupported; } }
